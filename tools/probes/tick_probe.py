# where a batch-1 MPC tick goes: setup_commands (host part / device part), solve, fetch
import time, numpy as np, bipedal_control_amd as bp
from bipedal_control_amd import scenarios as sc
itf = sc.h1_interface(); NI = 67
prob = sc.trot_problem(itf, batch=1, n_intervals=NI)
tm = [bp.loadModeSequenceTemplate(sc.H1["gait"], "trot")]
x0 = prob["x0"].copy()
for gains in (True, False):
    mpc = bp.BatchedSqpMpc(itf, max_batch=1, max_nodes=sc.max_nodes_for(NI, prob["horizon"]), return_gains=gains)
    mpc.setup_commands(0.0, x0, tm, 0, sc.GAIT_START, (0.3, 0, 0, 0), horizon=prob["horizon"]); mpc.enqueue(); mpc.fetch(gains=gains)
    rec = []
    for k in range(1, 201):
        t0 = time.perf_counter()
        mpc.setup_commands(0.02 * k, x0, tm, 0, sc.GAIT_START, (0.3, 0, 0, 0), horizon=prob["horizon"], from_previous=True)
        t1 = time.perf_counter(); mpc.synchronize(); t2 = time.perf_counter()
        mpc.enqueue(); t3 = time.perf_counter(); mpc.synchronize(); t4 = time.perf_counter()
        out = mpc.fetch(gains=gains); t5 = time.perf_counter()
        rec.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4))
    r = 1e3 * np.median(np.array(rec[20:]), axis=0)
    print("gains=%s  setup_commands call %.3f ms, its device work until idle %.3f, enqueue call %.3f, solve until idle %.3f, fetch %.3f   (sum %.3f)" % ((gains,) + tuple(r) + (r.sum(),)))

# the generic tick of the OCS2 adaptor: mode schedule and targets from the host (setup_from_previous), solve, fetch
mpc = bp.BatchedSqpMpc(itf, max_batch=1, max_nodes=sc.max_nodes_for(NI, prob["horizon"]), return_gains=True)
mpc.run(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"], gains=True)
rec = []
for k in range(1, 201):
    t0 = time.perf_counter()
    mpc.setup_from_previous(prob["t0"] + 0.02 * k, prob["x0"], prob["schedule"], prob["targets"], prob["horizon"])
    t1 = time.perf_counter(); mpc.enqueue(); out = mpc.fetch(gains=True); t2 = time.perf_counter()
    rec.append((t1 - t0, t2 - t1))
r = 1e3 * np.median(np.array(rec[20:]), axis=0)
print("generic tick: setup_from_previous call %.3f ms, solve + fetch %.3f   (sum %.3f)" % (r[0], r[1], r.sum()))
