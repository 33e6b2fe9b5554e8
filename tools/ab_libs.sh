#!/bin/bash
# cycles prebuilt libraries tools/probes/lib_<name>.bin through bench lines on one box:  bash tools/ab_libs.sh "base xnackoff" "--batch 256|--batch 4096"
export TMPDIR=/tmp PYTHONPATH=.
cp bipedal_control_amd/libbpmpc.so /tmp/keep.so
IFS='|' read -ra SHAPES <<< "${2:---batch 256|--batch 4096}"
for rep in 1 2; do for v in $1; do
  cp tools/probes/lib_$v.bin bipedal_control_amd/libbpmpc.so
  for ARGS in "${SHAPES[@]}"; do
    timeout 300 python bench.py $ARGS --cpu-sample 0 2>/tmp/err.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$v $ARGS', d['value'], d['ms_per_step'], (d.get('fused') or {}).get('value'), d['kernel_ms_per_step'])
except Exception as e: print('$v $ARGS FAILED', e)"
  done
done; done
cp /tmp/keep.so bipedal_control_amd/libbpmpc.so
