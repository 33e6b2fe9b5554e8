import os, sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bipedal_control_amd as bp
from bipedal_control_amd import scenarios as sc
import oracle_bridge as ob
from tolerances import rel_x, rel_u, rel_K
itf = sc.interface("g1")
B = 260
prob = sc.trot_problem(itf, batch=B, n_intervals=45, gait="standing_trot")
for wave in ("0", "1"):
    os.environ["BPMPC_RICCATI_WAVE"] = wave
    mpc = bp.BatchedSqpMpc(itf, max_batch=B, max_nodes=72, sqp_iterations=2, return_gains=True)
    t, x, u, K, st = mpc.run(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"], gains=True)
    n = st[0].n_nodes
    for b in (0, 259):
        xo, uo, Ko, _ = ob.oracle_solve_like(prob, b, iterations=2, robot="g1")
        print("wave", wave, "problem", b, "rel x %.2e u %.2e K %.2e" % (rel_x(x[b, :n + 1], xo), rel_u(u[b, :n], uo), rel_K(K[b, :n], Ko)))
