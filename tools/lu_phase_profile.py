# Needs libbpmpc.so built with -DBPMPC_LUS_PROFILE; cycles per phase of the structured elimination (project_lu_s.h), lane 0 of the first waves.
# usage: PYTHONPATH=. python tools/lu_phase_profile.py [batch] [robot]
import sys
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
robot = sys.argv[2] if len(sys.argv) > 2 else "h1"
import bipedal_control_amd as bp
from bipedal_control_amd import scenarios
itf = scenarios.interface(robot)
prob = scenarios.trot_problem(itf, batch=B, n_intervals=100, gait="standing_trot" if robot == "g1" else "trot")
mpc = bp.BatchedSqpMpc(itf, B, 116, pipeline_chunks=1)
mpc.setup(prob["t0"], prob["x0"], prob["schedule"], prob["targets"], horizon=prob["horizon"])
for st in ("linearize", "project"): mpc.stage(st)
mpc.synchronize()
mpc.stage("project"); mpc.synchronize()
r = mpc.read("rprof").reshape(-1, 8)[:B]
print("project_lu_s.h, cycles: loads, elimination, U + rank, back substitution, outputs")
print(r.mean(axis=0).round(0)[:5], "total", r.mean(axis=0)[:5].sum().round(0), " min", r.min(axis=0).round(0)[:5], " max", r.max(axis=0).round(0)[:5])
