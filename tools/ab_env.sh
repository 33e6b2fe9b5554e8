#!/bin/bash
# bench lines for a list of environments on one box:  ABENV="A=1 A=2" ABARGS="--batch 256|--batch 512" bash tools/ab_env.sh <tag>
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/${1:-abe}; mkdir -p $O
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-fused >/dev/null 2>&1
IFS='|' read -ra SHAPES <<< "${ABARGS:---batch 256|--batch 4096}"
for rep in 1 2; do
for V in "X=0" ${ABENV:-}; do
for ARGS in "${SHAPES[@]}"; do
  env $V timeout 300 python bench.py $ARGS --cpu-sample 0 --no-fused > $O/line.json 2>$O/err.log
  python -c "
import json
d=json.loads(open('$O/line.json').read().strip().splitlines()[-1]); print('[$V] $ARGS', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])" 2>&1 | tail -1
done
done
done
