# round 6: event nodes out of the lineariser's way (BPMPC_LIN_COMPACT, default on) against the in-line mapping - bit identity, parity, timing
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/${1:-r6c}; mkdir -p $O
BPMPC_LIN_COMPACT=0 python tools/probes/sol_hash.py inline 2>&1 | grep -v amdgpu > $O/hash_inline.txt
python tools/probes/sol_hash.py compact 2>&1 | grep -v amdgpu > $O/hash_compact.txt
paste $O/hash_inline.txt $O/hash_compact.txt | awk '{print $0, ($6==$12 ? "same" : "DIFFERENT")}' | tee $O/hash.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_g1.py tests/test_gpu_api.py -m gpu -q -x 2>&1 | tail -3 | tee $O/pytest.txt
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-fused >/dev/null 2>&1
for rep in 1 2; do for C in 0 1; do for ARGS in "--batch 256" "--batch 4096" "--robot g1 --batch 1024"; do
BPMPC_LIN_COMPACT=$C timeout 300 python bench.py $ARGS --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('compact=$C [$ARGS]', d['value'], d['ms_per_step'], 'fused', (d.get('fused') or {}).get('value'), 'lin', k.get('linearize'), 'ls', k.get('linesearch'), 'frac', d['roofline']['frac'], (d['roofline'].get('write_roof') or {}).get('pattern_GBs'))"
done; done; done 2>&1 | tee $O/ab.txt
