#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=.
cp bipedal_control_amd/libbpmpc.so /tmp/keep.so
for v in $1; do cp tools/probes/lib_$v.bin bipedal_control_amd/libbpmpc.so; echo $v; python tools/riccati_phase_profile.py 2>&1 | tail -3; python - <<PY
import bipedal_control_amd as bp
from bipedal_control_amd import scenarios
itf=scenarios.h1_interface()
prob=scenarios.trot_problem(itf,batch=256,n_intervals=100)
mpc=bp.BatchedSqpMpc(itf,256,116)
mpc.setup(prob["t0"],prob["x0"],prob["schedule"],prob["targets"],horizon=prob["horizon"])
for st in ("linearize","project","riccati"): mpc.stage(st)
mpc.synchronize(); mpc.stage("riccati"); mpc.synchronize()
r=mpc.read("rprof").reshape(-1,8)[:256]
print("rollout+norms cycles (whole horizon) [5]:", r[:,5].mean(), " E forward elimination per stage [7]:", r[:,7].mean()/107)
PY
done
cp /tmp/keep.so bipedal_control_amd/libbpmpc.so
