#!/bin/bash
# A/B/C... of compile-time variants on ONE GPU box (boxes of the pool differ by up to 30 %).
# usage: bash tools/ab_variants.sh "name:-DFLAG=1 -DOTHER=0" "name2:" ... -- [bench.py arguments]
# Builds one library per variant (BPMPC_EXTRA_FLAGS) in parallel, then a single gpurun call cycles through them twice.
set -e
VARIANTS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do VARIANTS+=("$1"); shift; done
[ "$1" == "--" ] && shift
NAMES=""
for v in "${VARIANTS[@]}"; do
  name="${v%%:*}"; flags="${v#*:}"
  NAMES="$NAMES $name"
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value $flags -o tools/probes/lib_$name.bin \
      bipedal_control_amd/csrc/{solver.hip,wbc.hip,capi.cpp,info_tree.cpp,urdf_tree.cpp,robot_model.cpp,reference_gen.cpp,device_model.cpp} > /tmp/ab_$name.log 2>&1 || echo "BUILD FAILED $name" ) &
done
wait
cp bipedal_control_amd/libbpmpc.so /tmp/libbpmpc_keep.so
# AB_PRE: optional command run first in the same call with the library as built in the tree (e.g. the GPU tests)
/usr/local/graft/bin/gpurun --timeout ${AB_TIMEOUT:-900} -- "${AB_PRE:-true}"'; for rep in 1 2; do for v in '"$NAMES"'; do cp tools/probes/lib_$v.bin bipedal_control_amd/libbpmpc.so; echo -n "$v "; timeout 200 python bench.py '"$*"' --steps 30 --warmup 3 --cpu-sample 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"ms_per_step\"], d[\"kernel_ms_per_step\"], (d.get(\"fused\") or {}).get(\"ms_per_step\"))"; done; done' 2>&1 | grep -v "^\[gpurun\]\|amdgpu.ids\|^----\|^$" | tail -${AB_TAIL:-20}
cp /tmp/libbpmpc_keep.so bipedal_control_amd/libbpmpc.so
for v in $NAMES; do rm -f tools/probes/lib_$v.bin; done
