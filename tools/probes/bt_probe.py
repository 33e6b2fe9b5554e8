# Diagnostic: which scenarios make the filter line search back-track more than once?
import numpy as np, bipedal_control_amd as bp
from bipedal_control_amd import scenarios as sc
itf = sc.h1_interface()
for gait in ("trot", "flying_trot", "standing_trot"):
    for its in (1, 2, 3, 5):
        prob = sc.trot_problem(itf, batch=8, n_intervals=40, gait=gait, cmd_vel=(0.5, 0.0, 0.0, 0.3))
        mpc = bp.BatchedSqpMpc(itf, max_batch=8, max_nodes=64, sqp_iterations=its)
        t, x, u, _, st = mpc.run(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
        print(gait, its, [s.step_size for s in st], [s.iterations for s in st], [s.status for s in st])
# warm start far from the measured state
prob = sc.trot_problem(itf, batch=8, n_intervals=40)
mpc = bp.BatchedSqpMpc(itf, max_batch=8, max_nodes=64)
t, x, u, _, st = mpc.run(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
for d in (0.05, 0.2, 0.5):
    x0b = prob["x0"] + d * np.sin(np.arange(prob["x0"].size).reshape(prob["x0"].shape))
    t2, x2, u2, _, st2 = mpc.run(prob["t0"], x0b, prob["schedule"], prob["targets"], horizon=prob["horizon"], warm_x=x, warm_u=u)
    print("warm", d, [s.step_size for s in st2], [s.status for s in st2])
