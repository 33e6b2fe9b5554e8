# Needs libbpmpc.so built with BPMPC_EXTRA_FLAGS=-DBPMPC_LINFAST_PROFILE [-DBPMPC_LIN_PROF_PROBLEM=<b>]; cycles per phase of k_linearize_fast
# (nodes 0..63 of problem 0 - an empty chip - or of problem b - steady state), fused and materialised variant.
import numpy as np
import bipedal_control_amd as bp
from bipedal_control_amd import scenarios
itf = scenarios.h1_interface()
prob = scenarios.trot_problem(itf, batch=256, n_intervals=100)
print("cycles: stage inputs, eval 1, contact rows, eval 2, RK2 A/B/b, cost+perf | wave cycles, wall (10 ns ticks)")
np.set_printoptions(linewidth=200, suppress=True)
for mat in (False, True):
    mpc = bp.BatchedSqpMpc(itf, 256, 116, materialize_lq=mat)
    mpc.setup(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
    mpc.stage("linearize"); mpc.synchronize(); mpc.stage("linearize"); mpc.synchronize()
    r = mpc.read("rprof").reshape(-1, 8)[:64]
    r = r[r.sum(axis=1) > 0]
    print("materialised" if mat else "fused       ", r.mean(axis=0).round(0), "phases sum", r[:, :6].mean(axis=0).sum().round(0))
