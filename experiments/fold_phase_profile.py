# Needs libbpmpc.so built with -DBPMPC_FOLD_PROFILE (riccati_fold8.h): own work of the projector waves per stage and phase, in cycles.
import bipedal_control_amd as bp
from bipedal_control_amd import scenarios
itf = scenarios.h1_interface()
prob = scenarios.trot_problem(itf, batch=256, n_intervals=100)
mpc = bp.BatchedSqpMpc(itf, 256, 116)
mpc.setup(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
for st in ("linearize", "project", "riccati"):
    mpc.stage(st)
mpc.synchronize()
mpc.stage("riccati"); mpc.synchronize()
r = mpc.read("rprof").reshape(256, 8).mean(axis=0)
n = r[7]
print("stages", n, " cycles per stage: whole loop %.0f" % (r[6] / n))
print("P4 (dynamics side): P1 stage+request %.0f  P2 part one %.0f  P3 part two %.0f" % tuple(r[0:3] / n))
print("P5 (cost side):     P1 stage+request %.0f  P2 part one %.0f  P3 part two %.0f" % tuple(r[3:6] / n))
