# repeated solves of one batch must give the same bits: python tools/probes/det_check.py [batch] [intervals] [robot]
import sys
import numpy as np
import bipedal_control_amd as bp
from bipedal_control_amd import scenarios as sc
B = int(sys.argv[1]) if len(sys.argv) > 1 else 25
NI = int(sys.argv[2]) if len(sys.argv) > 2 else 30
itf = sc.interface(sys.argv[3] if len(sys.argv) > 3 else "h1")
prob = sc.trot_problem(itf, batch=B, n_intervals=NI)
mpc = bp.BatchedSqpMpc(itf, max_batch=B, max_nodes=NI + 12, materialize_lq=True)
lay = mpc.setup(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
ref = None
bad = 0
for it in range(30):
    mpc.stage("linearize"); mpc.synchronize()
    cur = {k: mpc.read(k).copy() for k in ("A", "B", "b", "C", "D", "e", "q", "r", "Q", "R", "perf", "qrd")}
    if ref is None:
        ref = cur
    else:
        for k in cur:
            if not np.array_equal(cur[k], ref[k]):
                idx = np.flatnonzero(cur[k] != ref[k])
                print("iteration %d: %s differs in %d entries, first at %d (%.17g vs %.17g)" % (it, k, idx.size, idx[0], cur[k].flat[idx[0]], ref[k].flat[idx[0]]))
                bad += 1
print("non-deterministic arrays:", bad)
