# as sol_hash.py with the trot template tiled from t = 0 (problems that back-track) and more than one SQP iteration
import sys, hashlib, numpy as np
import bipedal_control_amd as bp
from bipedal_control_amd import scenarios as sc
tag = sys.argv[1] if len(sys.argv) > 1 else ""
for robot, batch, ni, its in (("h1", 256, 100, 1), ("h1", 256, 100, 3), ("h1", 4096, 100, 1), ("g1", 1024, 60, 2)):
    itf = sc.interface(robot)
    prob = sc.trot_problem(itf, batch=batch, n_intervals=ni, gait_start=0.0)
    mpc = bp.BatchedSqpMpc(itf, max_batch=batch, max_nodes=ni + 16, sqp_iterations=its)
    t, x, u, K, st = mpc.run(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
    steps = np.array([s.step_size for s in st]); iters = np.array([s.iterations for s in st]); merit = np.array([s.merit_after for s in st])
    print(tag, robot, batch, ni, its, hashlib.sha1(x.tobytes() + u.tobytes() + steps.tobytes() + iters.tobytes() + merit.tobytes()).hexdigest()[:16], "back-tracked", int((steps < 1).sum()), flush=True)
