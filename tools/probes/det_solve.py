# repeated complete solves of one batch must give the same bits: python tools/probes/det_solve.py [batch] [intervals]
import sys
import numpy as np
import bipedal_control_amd as bp
from bipedal_control_amd import scenarios as sc
B = int(sys.argv[1]) if len(sys.argv) > 1 else 25
NI = int(sys.argv[2]) if len(sys.argv) > 2 else 30
itf = sc.interface("h1")
prob = sc.trot_problem(itf, batch=B, n_intervals=NI)
mpc = bp.BatchedSqpMpc(itf, max_batch=B, max_nodes=NI + 12, materialize_lq=True)
ref = None
bad = 0
for it in range(40):
    t, x, u, _, st = mpc.run(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
    if ref is None:
        ref = (x.copy(), u.copy())
    elif not (np.array_equal(x, ref[0]) and np.array_equal(u, ref[1])):
        bad += 1
        d = np.argwhere(x != ref[0])
        print("solve %d differs: %d x entries, first %s; max |dx| %.3e" % (it, len(d), d[0] if len(d) else None, np.abs(x - ref[0]).max()))
print("differing solves:", bad)
