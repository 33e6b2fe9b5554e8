#!/bin/bash
# bench lines (all kernel classes) of libraries tools/probes/lib_<name>.bin at several shapes on one box: bash tools/probes/ab_libs_shapes.sh "a b" "--batch 256|--batch 512"
export TMPDIR=/tmp PYTHONPATH=.
cp bipedal_control_amd/libbpmpc.so /tmp/keep.so
IFS='|' read -ra SHAPES <<< "${2:---batch 256|--batch 512|--batch 4096}"
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-fused >/dev/null 2>&1
for rep in 1 2; do for v in $1; do
  cp tools/probes/lib_$v.bin bipedal_control_amd/libbpmpc.so
  for ARGS in "${SHAPES[@]}"; do
    timeout 300 python bench.py $ARGS --cpu-sample 0 --no-fused --profile-all 2>/tmp/err.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v [$ARGS]', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
  done
done; done
cp /tmp/keep.so bipedal_control_amd/libbpmpc.so
