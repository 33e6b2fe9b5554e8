export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/${1:-r6g}; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
cp bipedal_control_amd/libbpmpc.so /tmp/keep.so
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-fused >/dev/null 2>&1
for rep in 1 2; do for v in head kzero; do
  cp tools/probes/lib_$v.bin bipedal_control_amd/libbpmpc.so
  for ARGS in "--batch 256" "--batch 512" "--robot g1 --batch 256" "--batch 64"; do
    timeout 300 python bench.py $ARGS --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$v [$ARGS]', d['value'], d['ms_per_step'], 'fused', (d.get('fused') or {}).get('value'), 'ric', k['riccati'])"
  done; done; done 2>&1 | tee $O/ab.txt
cp /tmp/keep.so bipedal_control_amd/libbpmpc.so
