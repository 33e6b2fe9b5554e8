# Host pre-pass and solve time of a gait-library sweep (BASELINE.json configs[4] in miniature), shared vs distinct grids.
import time, numpy as np, bipedal_control_amd as bp
from bipedal_control_amd import scenarios as sc
itf = sc.h1_interface()
gaits = ["stance", "trot", "standing_trot", "flying_trot"]
cmds = [(vx, wz) for vx in np.linspace(-0.5, 0.5, 16) for wz in np.linspace(-0.3, 0.3, 4)]
prob = sc.gait_sweep_problem(itf, gaits, cmds, n_intervals=150)
nb = len(prob["schedule"])
mpc = bp.BatchedSqpMpc(itf, max_batch=nb, max_nodes=200, profile=True)
for label, t0 in (("shared t0", prob["t0"]), ("distinct t0 (one grid per problem)", prob["t0"] + 1e-7 * np.arange(nb))):
    for rep in range(2):
        t = time.perf_counter(); lay = mpc.setup(t0, prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"]); mpc.synchronize()
        ts = 1e3 * (time.perf_counter() - t)
        t = time.perf_counter(); mpc.enqueue(); mpc.synchronize(); tr = 1e3 * (time.perf_counter() - t)
    print("%s: grids %d, setup %.2f ms, solve %.2f ms" % (label, lay["n_grids"], ts, tr), {k: round(mpc.kernel_time(k)[0], 3) for k in ("linearize", "project_lu", "project", "riccati", "linesearch")})
