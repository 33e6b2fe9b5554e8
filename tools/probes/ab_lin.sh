#!/bin/bash
# A/B of lineariser variants on one box: for every library tools/probes/lib_<name>.bin the LQ parity tests and the bench line at three shapes
# usage: bash tools/probes/ab_lin.sh "name1 name2 ..." [out tag] [shapes, |-separated]
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/${2:-abl}; mkdir -p $O
cp bipedal_control_amd/libbpmpc.so /tmp/keep.so
IFS='|' read -ra SHAPES <<< "${3:---batch 256|--batch 4096|--robot g1 --batch 1024}"
for v in $1; do
  cp tools/probes/lib_$v.bin bipedal_control_amd/libbpmpc.so
  timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_g1.py tests/test_gpu_hunter.py -m gpu -q -x -k "linearize or lq or fused or matches_oracle" 2>&1 | tail -2 | tr '\n' ' '; echo
  for rep in 1 2; do for ARGS in "${SHAPES[@]}"; do
    timeout 300 python bench.py $ARGS --cpu-sample 0 2>/tmp/err.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$v [$ARGS]', d['value'], d['ms_per_step'], 'fused', (d.get('fused') or {}).get('value'), 'lin', k.get('linearize'), 'frac', d['roofline']['frac'])
except Exception as e: print('$v $ARGS FAILED', e)"
  done; done
done 2>&1 | tee $O/ab.txt
cp /tmp/keep.so bipedal_control_amd/libbpmpc.so
