# Needs libbpmpc.so built with -DBPMPC_RICCATI_PROFILE; cycles per phase of the wave-per-problem sweeps (riccati_wave.h, riccati_wave2.h).
# usage: python tools/riccati_wave_phase_profile.py [batch] [robot] [variant: 2 = riccati_wave.h, 4 = riccati_wave2.h]
import os, sys
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
robot = sys.argv[2] if len(sys.argv) > 2 else "h1"
variant = sys.argv[3] if len(sys.argv) > 3 else "2"
os.environ["BPMPC_RICCATI_WAVE"] = variant
import bipedal_control_amd as bp
from bipedal_control_amd import scenarios
itf = scenarios.interface(robot)
prob = scenarios.trot_problem(itf, batch=B, n_intervals=100, gait="standing_trot" if robot == "g1" else "trot")
mpc = bp.BatchedSqpMpc(itf, B, 116, pipeline_chunks=1)
lay = mpc.setup(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
for st in ("linearize", "project", "riccati"): mpc.stage(st)
mpc.synchronize()
mpc.stage("riccati"); mpc.synchronize()
r = mpc.read("rprof").reshape(-1, 8)[:B]
n = lay["n_nodes_max"]
if variant == "5":
    print("riccati_pair.h, wave 0, cycles per stage (products + tile + loads, wait B1, elimination, wait B2, updates + LDS writes, wait B3, S read):")
elif variant == "2":
    print("riccati_wave.h, cycles per stage by phase (top: stores + W loads, S W, B' SW + loads, A' SW + loads, tile + elimination, updates, outputs + loads):")
else:
    print("riccati_wave2.h, cycles per stage by phase (output stores, products by block column, tile + elimination, S update + symmetrisation, Acl, loads 1 + force rows of Pu, K, loads 2):")
print((r.mean(axis=0) / n).round(0)[:8], "total", (r.mean(axis=0) / n)[:8].sum().round(0))
