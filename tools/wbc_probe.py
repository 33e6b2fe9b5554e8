# Throughput of the batched whole-body controller (bpmpc_wbc_update incl. the host round trip of its inputs and outputs).
# usage (GPU box, repository root): PYTHONPATH=. python tools/wbc_probe.py [robot] [batch]
import sys
import time

import numpy as np

import bipedal_control_amd as bp
from bipedal_control_amd import scenarios as sc

robot = sys.argv[1] if len(sys.argv) > 1 else "h1"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
itf = sc.interface(robot)
nj = itf.actuatedDofNum
nv = 6 + nj
rng = np.random.default_rng(0)
x = np.tile(itf.getInitialState(), (B, 1)); x[:, 6:] += 0.02 * rng.standard_normal((B, nv))
u = np.zeros((B, itf.inputDim)); u[:, 2:12:3] = itf.robotMass() * 9.81 / 4
q = x[:, 6:] + 0.02 * rng.standard_normal((B, nv))
rbd = np.concatenate([q[:, 3:6], q[:, 0:3], q[:, 6:], np.zeros((B, nv))], axis=1)       # at rest: every stance constraint is consistent
modes = rng.choice([1, 2, 3], size=B).astype(np.int32)
wbc = bp.WeightedWbc(itf, max_batch=B)
sol, status = wbc.update(x, u, rbd, modes)
t0 = time.perf_counter()
n = 20
for _ in range(n):
    sol, status = wbc.update(x, u, rbd, modes)
dt = (time.perf_counter() - t0) / n
print("%s: %d robots, %.3f ms per batched update (%.0f QPs/s), unsolved %d" % (robot, B, 1e3 * dt, B / dt, int(status.sum())))
