#!/bin/bash
# A/B of whole-step variants on one box: bench lines (all kernel classes) for every library tools/probes/lib_<name>.bin
# usage: bash tools/probes/ab_step.sh "name1 name2 ..." [out tag] [shapes, |-separated]
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/${2:-abs}; mkdir -p $O
cp bipedal_control_amd/libbpmpc.so /tmp/keep.so
IFS='|' read -ra SHAPES <<< "${3:---batch 256|--batch 4096|--robot g1 --batch 1024}"
for rep in 1 2; do for v in $1; do
  cp tools/probes/lib_$v.bin bipedal_control_amd/libbpmpc.so
  for ARGS in "${SHAPES[@]}"; do
    timeout 300 python bench.py $ARGS --cpu-sample 0 2>/tmp/err.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$v [$ARGS]', d['value'], d['ms_per_step'], 'fused', (d.get('fused') or {}).get('value'), k)
except Exception as e: print('$v $ARGS FAILED', e)"
  done
done; done 2>&1 | tee $O/ab.txt
cp /tmp/keep.so bipedal_control_amd/libbpmpc.so
