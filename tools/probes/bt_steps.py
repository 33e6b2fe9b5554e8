import numpy as np, bipedal_control_amd as bp
from bipedal_control_amd import scenarios as sc
itf = sc.h1_interface()
for gs in (0.0, -1.225):
    for B in (256, 4096):
        prob = sc.trot_problem(itf, batch=B, n_intervals=100, gait_start=gs)
        mpc = bp.BatchedSqpMpc(itf, B, 116)
        t, x, u, K, st = mpc.run(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
        a = np.array([s.step_size for s in st]); status = np.array([s.status for s in st])
        vals, cnt = np.unique(a, return_counts=True)
        print("gait_start", gs, "batch", B, "step sizes", dict(zip(vals.tolist(), cnt.tolist())), "status", dict(zip(*np.unique(status, return_counts=True))))
