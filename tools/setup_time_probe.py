import time, numpy as np, bipedal_control_amd as bp
from bipedal_control_amd import scenarios as sc
itf = sc.h1_interface()
gaits = ["stance", "trot", "standing_trot", "flying_trot"]
cmds = [(vx, wz) for vx in np.linspace(-0.5, 0.5, 16) for wz in np.linspace(-0.3, 0.3, 4)]
prob = sc.gait_sweep_problem(itf, gaits, cmds, n_intervals=150)
nb = len(prob["schedule"])
mpc = bp.BatchedSqpMpc(itf, max_batch=nb, max_nodes=200)
for rep in range(2):
    t = time.perf_counter(); lay = mpc.setup(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"]); mpc.synchronize()
    print("batch", nb, "per-problem schedules: setup %.1f ms" % (1e3 * (time.perf_counter() - t)), lay)
t = time.perf_counter(); mpc.enqueue(); mpc.synchronize(); print("solve %.2f ms" % (1e3 * (time.perf_counter() - t)))
