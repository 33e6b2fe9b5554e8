# Batch = 1 latency of one solve (the reference's budget is 20 ms per solve at 50 Hz).
import time, numpy as np, bipedal_control_amd as bp
from bipedal_control_amd import scenarios as sc
itf = sc.h1_interface()
for NI in (67, 100):
    prob = sc.trot_problem(itf, batch=1, n_intervals=NI)
    mpc = bp.BatchedSqpMpc(itf, max_batch=1, max_nodes=sc.max_nodes_for(NI, prob["horizon"]), profile=False)
    mpc.setup(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
    for _ in range(20): mpc.reset(); mpc.enqueue()
    mpc.synchronize()
    lat = []
    for _ in range(200):
        mpc.reset(); t = time.perf_counter(); mpc.enqueue(); mpc.synchronize(); lat.append(time.perf_counter() - t)
    t = time.perf_counter()
    for _ in range(200): mpc.reset(); mpc.enqueue()
    mpc.synchronize(); thr = (time.perf_counter() - t) / 200
    # a whole MPC tick: new measurement + shifted warm start + solve + policy export
    x0 = prob["x0"].copy(); tick = []
    tm = [bp.loadModeSequenceTemplate(sc.H1["gait"], "trot")]
    mpc2 = bp.BatchedSqpMpc(itf, max_batch=1, max_nodes=sc.max_nodes_for(NI, prob["horizon"]), return_gains=True)
    mpc2.setup_commands(0.0, x0, tm, 0, sc.GAIT_START, (0.3, 0, 0, 0), horizon=prob["horizon"]); mpc2.enqueue(); mpc2.fetch(gains=True)
    for k in range(1, 101):
        t = time.perf_counter()
        mpc2.setup_commands(0.02 * k, x0, tm, 0, sc.GAIT_START, (0.3, 0, 0, 0), horizon=prob["horizon"], from_previous=True); mpc2.enqueue(); out = mpc2.fetch(gains=True)
        tick.append(time.perf_counter() - t)
    print("horizon %d: solve latency median %.3f ms (min %.3f, p99 %.3f), back-to-back %.3f ms per solve; full tick (setup_commands + solve + fetch x,u,K) median %.3f ms"
          % (NI, 1e3 * np.median(lat), 1e3 * np.min(lat), 1e3 * np.percentile(lat, 99), 1e3 * thr, 1e3 * np.median(tick)))
