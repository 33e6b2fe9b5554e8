# round 6 against the round-5 kernels (library built from commit 3ca7031) on ONE box: bench lines at four shapes, per-kernel times
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/${1:-r6v}; mkdir -p $O
cp bipedal_control_amd/libbpmpc.so /tmp/keep.so
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-fused >/dev/null 2>&1
for rep in 1 2; do for v in r05 head; do
  cp tools/probes/lib_$v.bin bipedal_control_amd/libbpmpc.so
  for ARGS in "--batch 256" "--batch 256 --gait-start -1.225" "--batch 512" "--batch 4096" "--robot g1 --batch 1024"; do
    timeout 300 python bench.py $ARGS --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$v [$ARGS]', d['value'], d['ms_per_step'], 'fused', (d.get('fused') or {}).get('value'), k, 'frac', d['roofline']['frac'])
except Exception as e: print('$v [$ARGS] FAILED', e)"
  done; done; done 2>&1 | tee $O/ab.txt
cp /tmp/keep.so bipedal_control_amd/libbpmpc.so
