# Edge cases: shortest and longest horizons, an event exactly at t0 / at the final time.
import numpy as np, bipedal_control_amd as bp
from bipedal_control_amd import scenarios as sc
from tests import oracle_bridge as ob
itf = sc.h1_interface()
def rel(a, b): return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))
for NI, mx in ((1, 8), (2, 8), (3, 8), (450, 512)):
    prob = sc.trot_problem(itf, batch=2, n_intervals=NI)
    mpc = bp.BatchedSqpMpc(itf, max_batch=2, max_nodes=mx, return_gains=True)
    t, x, u, K, st = mpc.run(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"], gains=True)
    n = st[0].n_nodes
    xo, uo, Ko, so = ob.oracle_solve_like(prob, 1)
    print("N", NI, "nodes", n, "status", [s.status for s in st], "step", st[1].step_size, so[0][3], "x", rel(x[1, :n + 1], xo), "u", rel(u[1, :n], uo), "K", rel(K[1, :n], Ko))
# event times exactly on t0 and on the final time
for t0 in (0.175, 0.175 - 30 * sc.DT):
    sched = sc.gait_schedule(itf, "trot", t0, 30 * sc.DT)
    ev = np.asarray(sched.eventTimes)
    x0 = sc.perturbed_initial_states(itf, 1)
    tg = [itf.cmdVelToTargetTrajectories((0.3, 0, 0, 0), t0, x0[0], 30 * sc.DT)]
    prob = dict(t0=t0, x0=x0, schedule=sched, targets=tg, horizon=30 * sc.DT)
    mpc = bp.BatchedSqpMpc(itf, max_batch=1, max_nodes=48)
    t, x, u, _, st = mpc.run(t0, x0, sched, tg, horizon=30 * sc.DT)
    n = st[0].n_nodes
    xo, uo, _, so = ob.oracle_solve_like(prob, 0)
    print("t0", t0, "event on boundary:", bool(np.any(np.abs(ev - t0) < 1e-12) or np.any(np.abs(ev - (t0 + 30 * sc.DT)) < 1e-12)), "nodes", n, xo.shape[0] - 1, "x", rel(x[0, :n + 1], xo), "u", rel(u[0, :n], uo))
