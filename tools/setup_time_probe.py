# Pre-pass time of a gait-library sweep (BASELINE.json configs[4] in miniature): host path (bpmpc_solver_setup fed by Python-side
# GaitSchedule / cmdVelToTargetTrajectories objects) against the device path (bpmpc_solver_setup_commands).
import time, numpy as np, bipedal_control_amd as bp
from bipedal_control_amd import scenarios as sc
itf = sc.h1_interface()
gaits = ["stance", "trot", "standing_trot", "flying_trot"]
tm = [bp.loadModeSequenceTemplate(sc.H1["gait"], g) for g in gaits[1:]]
cmds = [(vx, wz) for vx in np.linspace(-0.5, 0.5, 16) for wz in np.linspace(-0.3, 0.3, 4)]
t = time.perf_counter(); prob = sc.gait_sweep_problem(itf, gaits, cmds, n_intervals=150); t_objs = 1e3 * (time.perf_counter() - t)
nb = len(prob["schedule"])
gop = np.repeat(np.arange(len(gaits)) - 1, len(cmds)).astype(np.int32)
cmd4 = np.array([(vx, 0.0, 0.0, wz) for _ in gaits for (vx, wz) in cmds])
mpc = bp.BatchedSqpMpc(itf, max_batch=nb, max_nodes=200)
print("building %d schedule / target objects in Python: %.2f ms" % (nb, t_objs))
for label, t0 in (("shared t0", prob["t0"]), ("distinct t0 (one grid per problem)", prob["t0"] + 1e-7 * np.arange(nb))):
    for rep in range(3):
        t = time.perf_counter(); lay = mpc.setup(t0, prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"]); th = 1e3 * (time.perf_counter() - t)
        t = time.perf_counter(); mpc.enqueue(); mpc.synchronize(); tr = 1e3 * (time.perf_counter() - t)
        t = time.perf_counter(); lay2 = mpc.setup_commands(t0, prob["x0"], tm, gop, sc.GAIT_START, cmd4, horizon=prob["horizon"]); td = 1e3 * (time.perf_counter() - t)
        print("   rep %d: host setup %.2f ms, device setup %.3f ms, solve %.2f ms" % (rep, th, td, tr))
    print("%s: grids %d / %d, host setup %.2f ms, device setup %.3f ms, solve %.2f ms" % (label, lay["n_grids"], lay2["n_grids"], th, td, tr))
