# Dumps the LQ model the lineariser writes (materialised and fused mode) for mixed-gait batches of two robots into an npz:
#   python tools/probes/lq_dump.py out.npz        then compare two dumps (two libraries) with  python tools/probes/lq_dump.py a.npz b.npz
import sys
import numpy as np

if len(sys.argv) == 3:
    a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
    worst = 0.0
    for k in a.files:
        x, y = a[k], b[k]
        if x.dtype.kind == 'f':
            same = np.array_equal(x, y)
            d = np.abs(x - y).max() if x.size else 0.0
            scale = max(1.0, np.abs(x).max()) if x.size else 1.0
            nz = int((x != y).sum())
            worst = max(worst, d / scale)
            print("%-22s %-9s max|d| %.3e (rel to max %.1e)  differing %d / %d" % (k, "BITWISE" if same else "differs", d, d / scale, nz, x.size))
        else:
            print("%-22s %s" % (k, "equal" if np.array_equal(x, y) else "DIFFERS"))
    print("worst relative difference %.3e" % worst)
    sys.exit(0)

import bipedal_control_amd as bp
from bipedal_control_amd import scenarios as sc

out = {}
for robot in ("h1", "g1", "hunter"):
    itf = sc.interface(robot)
    for gait in ("trot", "standing_trot", "flying_trot"):
        prob = sc.trot_problem(itf, batch=6, n_intervals=40, gait=gait)
        for mat in (1, 0):
            mpc = bp.BatchedSqpMpc(itf, max_batch=6, max_nodes=64, materialize_lq=bool(mat))
            lay = mpc.setup(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
            mpc.enqueue(); mpc.synchronize()          # one accepted step: away from the cold start
            mpc.stage("linearize"); mpc.synchronize()
            names = ("A", "B", "b", "Q", "R", "q", "r", "c", "C", "D", "e", "nc", "perf", "qrd") if mat else ("b", "q", "r", "e", "nc", "perf", "qrd")
            n = lay["n_nodes_max"]
            for k in names:
                v = mpc.read(k)
                v = v.reshape(6, 64, -1)[:, :n]
                if k in ("C", "D") :      # rows beyond nc are padding in both modes
                    pass
                out["%s/%s/%s/%s" % (robot, gait, "mat" if mat else "fused", k)] = v
np.savez(sys.argv[1], **out)
print("wrote", sys.argv[1], len(out), "arrays")
