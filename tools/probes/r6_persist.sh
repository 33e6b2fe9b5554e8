export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/${1:-r6p}; mkdir -p $O
BPMPC_LIN_PERSIST=0 python tools/probes/sol_hash.py off 2>&1 | grep -v amdgpu | head -4 > $O/h0.txt
BPMPC_LIN_PERSIST=1 python tools/probes/sol_hash.py on 2>&1 | grep -v amdgpu | head -4 > $O/h1.txt
paste $O/h0.txt $O/h1.txt | awk '{print $0, ($6==$12 ? "same" : "DIFFERENT")}'
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-fused >/dev/null 2>&1
for rep in 1 2; do for C in 0 1; do for ARGS in "--batch 256" "--batch 4096" "--robot g1 --batch 1024"; do
BPMPC_LIN_PERSIST=$C timeout 300 python bench.py $ARGS --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('persist=$C [$ARGS]', d['value'], d['ms_per_step'], 'fused', (d.get('fused') or {}).get('value'), 'lin', k.get('linearize'), 'frac', d['roofline']['frac'])"
done; done; done 2>&1 | tee $O/ab.txt
