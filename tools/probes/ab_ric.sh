#!/bin/bash
# Riccati-only A/B of libraries tools/probes/lib_<name>.bin on one box: per-class kernel times of the headline bench (wrong results allowed: ablations)
# usage: bash tools/probes/ab_ric.sh "name1 name2 ..." [bench args]
export TMPDIR=/tmp PYTHONPATH=.
cp bipedal_control_amd/libbpmpc.so /tmp/keep.so
NAMES=$1; shift
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-fused >/dev/null 2>&1
for rep in 1 2; do for v in $NAMES; do
  cp tools/probes/lib_$v.bin bipedal_control_amd/libbpmpc.so
  timeout 300 python bench.py --cpu-sample 0 --no-fused --profile-all "$@" 2>/tmp/err.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$v', d['value'], d['ms_per_step'], k)
except Exception as e: print('$v FAILED', e); print(open('/tmp/err.log').read()[-400:])"
done; done
cp /tmp/keep.so bipedal_control_amd/libbpmpc.so
