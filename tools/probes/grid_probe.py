# Diagnostic: first solve after repeated setups with one grid per problem.
import time, numpy as np, bipedal_control_amd as bp
from bipedal_control_amd import scenarios as sc
itf = sc.h1_interface()
gaits = ["stance", "trot", "standing_trot", "flying_trot"]
cmds = [(vx, wz) for vx in np.linspace(-0.5, 0.5, 16) for wz in np.linspace(-0.3, 0.3, 4)]
prob = sc.gait_sweep_problem(itf, gaits, cmds, n_intervals=150)
nb = len(prob["schedule"])
mpc = bp.BatchedSqpMpc(itf, max_batch=nb, max_nodes=200)
for do_fetch in (True, False, True, False):
  for label, t0 in (("shared", prob["t0"]), ("distinct", prob["t0"] + 1e-7 * np.arange(nb))):
    for rep in range(3):
        lay = mpc.setup(t0, prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"]); mpc.synchronize()
        t = time.perf_counter(); mpc.enqueue(); te = 1e3 * (time.perf_counter() - t); mpc.synchronize(); tr = 1e3 * (time.perf_counter() - t)
        if do_fetch: st = mpc.fetch()[-1]
        print("fetch", do_fetch, label, rep, lay["n_grids"], "enqueue %.2f total %.2f ms" % (te, tr))
