# Needs libbpmpc.so built with BPMPC_EXTRA_FLAGS=-DBPMPC_LIN_TIMELINE: start / model-staged / end wall times of every wave of k_linearize_fast
# (H1 trot, batch 256): how long a workgroup holds its slot, how much of that is the model staging, how many are resident over time.
import sys
import numpy as np
import bipedal_control_amd as bp
from bipedal_control_amd import scenarios

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
itf = scenarios.h1_interface()
prob = scenarios.trot_problem(itf, batch=batch, n_intervals=100)
mpc = bp.BatchedSqpMpc(itf, batch, 116, materialize_lq=True)
lay = mpc.setup(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
for _ in range(3):
    mpc.stage("linearize"); mpc.synchronize()
n_wg = min(2048, (batch * lay["n_nodes_max"] + 15) // 16)
r = mpc.read("rprof")[:16 * n_wg].reshape(n_wg, 4, 4)
t0, t1, t2 = r[:, :, 0].min(axis=1), r[:, :, 1].max(axis=1), r[:, :, 2].max(axis=1)
base = t0.min()
t0, t1, t2 = (t0 - base) * 0.01, (t1 - base) * 0.01, (t2 - base) * 0.01       # microseconds
hw = r[:, 0, 3].astype(np.int64)
xcc, cu, se = (hw >> 32) & 15, (hw >> 8) & 15, (hw >> 13) & 7
print("workgroups %d, kernel span %.1f us (first start to last end)" % (n_wg, t2.max()))
print("slot time per workgroup: mean %.1f us (min %.1f, max %.1f); model staging %.1f us; body %.1f us" % ((t2 - t0).mean(), (t2 - t0).min(), (t2 - t0).max(), (t1 - t0).mean(), (t2 - t1).mean()))
wave_end = r[:, :, 2] * 0.01 - base * 0.01
print("spread of a workgroup's four wave ends: mean %.1f us" % (wave_end.max(axis=1) - wave_end.min(axis=1)).mean())
order = np.argsort(t0)
print("start times (us) of workgroups 0, 256, 512, 768, 1024, 1280, 1536:", [round(float(np.sort(t0)[i]), 1) for i in range(0, n_wg, 256)])
for t in np.linspace(0, t2.max(), 15):
    print("  t = %6.1f us: resident workgroups %4d" % (t, int(((t0 <= t) & (t2 > t)).sum())))
print("distinct (xcc, se, cu):", len(set(zip(xcc.tolist(), se.tolist(), cu.tolist()))))
