# Soak test: the same batch solved many times on one handle must give the same bits every time (races between waves, stale
# LDS or HBM scratch would show up as run-to-run differences).  usage: python tools/soak_determinism.py [steps] [batch]
import sys, numpy as np, bipedal_control_amd as bp
from bipedal_control_amd import scenarios as sc
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
itf = sc.h1_interface()
for gait, ni in (("trot", 100), ("flying_trot", 60), ("stance", 40)):
    prob = sc.trot_problem(itf, batch=batch, n_intervals=ni, gait=gait)
    mpc = bp.BatchedSqpMpc(itf, batch, ni + 24, sqp_iterations=2, return_gains=True)
    mpc.setup(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
    ref = None
    bad = 0
    for i in range(steps):
        mpc.reset(); mpc.enqueue()
        if i % 10 == 0 or i == steps - 1:
            t, x, u, K, st = mpc.fetch(gains=True)
            cur = (x.copy(), u.copy(), K.copy(), [s.step_size for s in st])
            if ref is None:
                ref = cur
            elif not (np.array_equal(cur[0], ref[0]) and np.array_equal(cur[1], ref[1]) and np.array_equal(cur[2], ref[2]) and cur[3] == ref[3]):
                bad += 1
    print(gait, "steps", steps, "batch", batch, "mismatching fetches", bad, "status", sorted(set(s.status for s in st)), "finite", bool(np.isfinite(x).all()))
    assert bad == 0
print("soak ok")
