# round 6: the eight-wave sweep in rounds beyond one problem per CU (default of the 22-state robots) against the regimes of rounds 3 to 5 (BPMPC_R8_ROUNDS=1)
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/${1:-r6r8}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee $O/tests.txt
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-fused >/dev/null 2>&1
for rep in 1 2; do for ARGS in "--batch 272" "--batch 384" "--batch 512" "--batch 640" "--batch 768" "--batch 1024" "--robot g1 --batch 512"; do for R in 1 0; do
  BPMPC_R8_ROUNDS=$R timeout 300 python bench.py $ARGS --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('[$ARGS] BPMPC_R8_ROUNDS=$R', d['value'], d['ms_per_step'], 'fused', (d.get('fused') or {}).get('value'), k)"
done; done; done 2>&1 | tee $O/ab.txt
