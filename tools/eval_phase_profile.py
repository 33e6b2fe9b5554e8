import numpy as np, bipedal_control_amd as bp
from bipedal_control_amd import scenarios
itf=scenarios.h1_interface()
prob=scenarios.trot_problem(itf,batch=256,n_intervals=100)
mpc=bp.BatchedSqpMpc(itf,256,116)
mpc.setup(prob["t0"],prob["x0"],prob["schedule"],prob["targets"],horizon=prob["horizon"])
mpc.stage("linearize"); mpc.synchronize(); mpc.read("evprof")
mpc.stage("linearize"); mpc.synchronize()
r=mpc.read("evprof")
print("eval phases (both evals of node 40 summed):", r[:13].round(0), r[:13].sum())
