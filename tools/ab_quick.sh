#!/bin/bash
# quick A/B on one box: GPU parity tests of the LQ model + bench lines at three shapes, optional environment in $ABENV ("A=1 B=2")
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/${1:-abq}; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > $O/pytest.log; cat $O/pytest.log
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-fused >/dev/null 2>&1
for V in "" ${ABENV:-}; do
for ARGS in "--batch 256" "--batch 4096" "--robot g1 --batch 1024"; do
  env $V timeout 300 python bench.py $ARGS --cpu-sample 0 > $O/line.json 2>$O/err.log
  python -c "
import json
d=json.loads(open('$O/line.json').read().strip().splitlines()[-1]); print('[$V] $ARGS', d['value'], d['ms_per_step'], (d.get('fused') or {}).get('value'), d['kernel_ms_per_step'], d['roofline']['frac'])" 2>&1 | tail -1
done
done
