export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/${1:-r6t}; mkdir -p $O
bash tools/probes/ab_hash.sh "head trialc" 2>&1 | sort -k2,5 | awk '{print}' | tee $O/hash.txt | awk '{h[$2" "$3" "$4" "$5]=h[$2" "$3" "$4" "$5]" "$6} END{for(k in h){split(h[k],a," "); print k, (a[1]==a[2]?"same":"DIFFERENT")}}'
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -q -x 2>&1 | tail -2
cp bipedal_control_amd/libbpmpc.so /tmp/keep.so
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-fused >/dev/null 2>&1
for rep in 1 2; do for v in head trialc; do
  cp tools/probes/lib_$v.bin bipedal_control_amd/libbpmpc.so
  for ARGS in "--batch 256" "--batch 256 --gait-start -1.225" "--batch 4096"; do
    timeout 300 python bench.py $ARGS --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$v [$ARGS]', d['value'], d['ms_per_step'], 'fused', (d.get('fused') or {}).get('value'), 'lin', k['linearize'], 'ls', k['linesearch'])"
  done; done; done 2>&1 | tee $O/ab.txt
cp /tmp/keep.so bipedal_control_amd/libbpmpc.so
