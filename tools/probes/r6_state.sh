# round 6, state of the tree on one box: write roof, lineariser phases and timeline, GPU tests, bench lines at the three shapes
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/${1:-r6s}; mkdir -p $O
tools/probes/write_roof.bin 256 103 > $O/write_roof.json 2>&1; tools/probes/write_roof.bin 4096 103 >> $O/write_roof.json 2>&1; cat $O/write_roof.json
bash tools/probes/prof_lin.sh "linprof" 2>&1 | tee $O/prof_lin.txt
cp bipedal_control_amd/libbpmpc.so /tmp/keep2.so; cp tools/probes/lib_timeline.bin bipedal_control_amd/libbpmpc.so
python tools/lin_timeline.py 2>&1 | tail -25 | tee $O/timeline.txt
cp /tmp/keep2.so bipedal_control_amd/libbpmpc.so
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest.txt
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-fused >/dev/null 2>&1
for ARGS in "--batch 256" "--batch 256 --gait-start -1.225" "--batch 512" "--batch 4096" "--robot g1 --batch 1024"; do
timeout 300 python bench.py $ARGS --cpu-sample 0 2>$O/bench.err | tail -1 | tee -a $O/bench.jsonl | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$ARGS', d['value'], d['ms_per_step'], (d.get('fused') or {}).get('value'), d['kernel_ms_per_step'], d['roofline']['frac'])"
done
