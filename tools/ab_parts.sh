#!/bin/bash
# A/B of the batch split over streams (BPMPC_BATCH_PARTS, BPMPC_BATCH_SKEW; solver.hip run_iterations) on ONE box:
#   gpurun --timeout 1500 -- 'bash tools/ab_parts.sh'      -> gpurun_out/r03j/
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r03j; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "batch_split" 2>&1 | tail -8 > $O/pytest_split.log
cat $O/pytest_split.log
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-fused >/dev/null 2>&1
for SK in 0 1; do
for P in ${PARTS:-1 2 3 4}; do
  for B in ${BATCHES:-256 512 4096}; do
    BPMPC_BATCH_SKEW=$SK BPMPC_BATCH_PARTS=$P timeout 300 python bench.py --batch $B --cpu-sample 0 > $O/b${B}_p${P}_s$SK.json 2>$O/err.log
    python -c "
import json
d=json.loads(open('$O/b${B}_p${P}_s$SK.json').read().strip().splitlines()[-1]); print('skew $SK parts $P batch $B', d['value'], d['ms_per_step'], (d.get('fused') or {}).get('value'), d['kernel_ms_per_step'])" 2>&1 | tail -1
  done
done
done
