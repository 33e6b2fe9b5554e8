# hashes of complete solves on a few shapes (bitwise comparison of libraries / switches): python tools/probes/sol_hash.py [tag]
import sys, hashlib, numpy as np
import bipedal_control_amd as bp
from bipedal_control_amd import scenarios as sc
tag = sys.argv[1] if len(sys.argv) > 1 else ""
for robot, batch, ni in (("h1", 256, 100), ("h1", 512, 100), ("h1", 2048, 40), ("g1", 256, 60), ("g1", 1024, 40)):
    itf = sc.interface(robot)
    for gait in ("trot", "stance"):
        try: prob = sc.trot_problem(itf, batch=batch, n_intervals=ni, gait=gait)
        except TypeError:
            if gait != "trot": continue
            prob = sc.trot_problem(itf, batch=batch, n_intervals=ni)
        mpc = bp.BatchedSqpMpc(itf, max_batch=batch, max_nodes=ni + 16)
        t, x, u, K, st = mpc.run(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
        print(tag, robot, batch, ni, gait, hashlib.sha1(x.tobytes() + u.tobytes() + (K.tobytes() if K is not None else b"")).hexdigest()[:16], flush=True)
