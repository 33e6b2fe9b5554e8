#!/bin/bash
# A/B of environment switches on one box with the current library: bash tools/probes/ab_env2.sh "A=1|B=2" "shape1|shape2"   (first variant = no switch)
export TMPDIR=/tmp PYTHONPATH=.
IFS='|' read -ra VARS <<< "|$1"
IFS='|' read -ra SHAPES <<< "${2:---batch 4096|--robot g1 --batch 1024}"
for rep in 1 2; do for V in "${VARS[@]}"; do for ARGS in "${SHAPES[@]}"; do
  env $V timeout 300 python bench.py $ARGS --cpu-sample 0 2>/tmp/err.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('[$V] [$ARGS]', d['value'], d['ms_per_step'], 'fused', (d.get('fused') or {}).get('value'), k)
except Exception as e: print('[$V] $ARGS FAILED', e)"
done; done; done
