# Needs libbpmpc.so built with -DBPMPC_RICCATI_PROFILE (see bipedal_control_amd/csrc/kernels/riccati_fast.h); prints cycles per phase of the backward sweep.
import bipedal_control_amd as bp
from bipedal_control_amd import scenarios
itf=scenarios.h1_interface()
prob=scenarios.trot_problem(itf,batch=256,n_intervals=100)
mpc=bp.BatchedSqpMpc(itf,256,116)
mpc.setup(prob["t0"],prob["x0"],prob["schedule"],prob["targets"],horizon=prob["horizon"])
for st in ("linearize","project","riccati"): mpc.stage(st)
mpc.synchronize()
mpc.stage("riccati"); mpc.synchronize()
r=mpc.read("rprof").reshape(-1,8)[:256]
print("cycles per stage by phase (P0 regs->LDS+sync, -, P1+sync, P2+sync, P3+sync, P4, P3 own work of wave 0 (Sn), P3 own work of wave 3 (elimination)):")
print((r.mean(axis=0)/107).round(0))
print("total per stage", (r.mean(axis=0)/107)[:6].sum())
