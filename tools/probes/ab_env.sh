#!/bin/bash
# A/B of environment switches with the library in the tree on one box: bash tools/probes/ab_env.sh "A=1|A=2" [bench args]
export TMPDIR=/tmp PYTHONPATH=.
IFS='|' read -ra VARS <<< "$1"; shift
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-fused >/dev/null 2>&1
for rep in 1 2; do for v in "${VARS[@]}"; do
  env $v timeout 300 python bench.py --cpu-sample 0 --profile-all "$@" 2>/tmp/err.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$v] $*', d['value'], d['ms_per_step'], 'fused', (d.get('fused') or {}).get('value'), d['kernel_ms_per_step'])"
done; done
