# What distinguishes the problems whose full step is rejected (config 2 tiled from t = 0: 1 of 256, 6 of 4096)?  Per problem: accepted step length and the
# quantities the QP step leaves behind (step norms, Armijo descent, baseline violation) - the material of a predictor for a speculative alpha = 1/2 trial.
import numpy as np, bipedal_control_amd as bp
from bipedal_control_amd import scenarios as sc
np.set_printoptions(linewidth=220, suppress=True, precision=4)
for robot, gait, B, gs in (("h1", "trot", 4096, 0.0), ("h1", "trot", 4096, -1.225), ("g1", "standing_trot", 1024, 0.0), ("h1", "flying_trot", 1024, 0.0)):
    itf = sc.interface(robot)
    try:
        prob = sc.trot_problem(itf, batch=B, n_intervals=100, gait=gait, gait_start=gs)
    except Exception as e:
        print(robot, gait, "skipped:", e); continue
    mpc = bp.BatchedSqpMpc(itf, B, 140)
    t, x, u, K, st = mpc.run(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
    a = np.array([s.step_size for s in st])
    q = {k: np.array([getattr(s, k) for s in st]) for k in ("dx_norm", "du_norm", "armijo_descent", "merit_before", "dynamics_sse_before", "equality_sse_before", "merit_after", "dynamics_sse_after", "equality_sse_after")}
    bt = a < 1.0
    print(robot, gait, "batch", B, "gait_start", gs, "back-trackers", int(bt.sum()), "step sizes", dict(zip(*np.unique(a, return_counts=True))))
    for k, v in q.items():
        o = np.sort(v[~bt])
        print("  %-22s accepted: median %.4g  p99 %.4g  max %.4g | back-trackers: %s" % (k, np.median(o), o[int(0.99 * (len(o) - 1))], o[-1], np.sort(v[bt])[:12]))
    for k in ("du_norm", "dx_norm"):
        thr = q[k][bt].min() if bt.any() else np.inf
        print("  threshold %s >= %.4g would flag %d problems (%d real)" % (k, thr, int((q[k] >= thr).sum()), int(bt.sum())))
