# Needs libbpmpc.so built with BPMPC_EXTRA_FLAGS=-DBPMPC_LINFAST_PROFILE; cycles per phase of k_linearize_fast (problem 0, nodes 0..63).
import bipedal_control_amd as bp
from bipedal_control_amd import scenarios
itf=scenarios.h1_interface()
prob=scenarios.trot_problem(itf,batch=256,n_intervals=100)
mpc=bp.BatchedSqpMpc(itf,256,116)
mpc.setup(prob["t0"],prob["x0"],prob["schedule"],prob["targets"],horizon=prob["horizon"])
mpc.stage("linearize"); mpc.synchronize(); mpc.stage("linearize"); mpc.synchronize()
r=mpc.read("rprof").reshape(-1,8)[:64]
r=r[r.sum(axis=1)>0]
print("cycles: stage inputs, eval 1, contact rows, eval 2, RK2 A/B/b, cost+perf")
print(r.mean(axis=0).round(0), r.mean(axis=0).sum())
