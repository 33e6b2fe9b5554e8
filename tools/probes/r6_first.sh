export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r6a; mkdir -p $O
tools/probes/write_roof.bin 256 103 > $O/write_roof.json 2>&1; cat $O/write_roof.json
tools/probes/write_roof.bin 4096 103 >> $O/write_roof.json 2>&1; tail -1 $O/write_roof.json
bash tools/probes/prof_lin.sh "linprof evprof" 2>&1 | tee $O/prof_lin.txt
cp bipedal_control_amd/libbpmpc.so /tmp/keep2.so; cp tools/probes/lib_timeline.bin bipedal_control_amd/libbpmpc.so
python tools/lin_timeline.py 2>&1 | tail -25 | tee $O/timeline.txt
cp /tmp/keep2.so bipedal_control_amd/libbpmpc.so
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-fused >/dev/null 2>&1
timeout 300 python bench.py --cpu-sample 0 > $O/bench.json 2>$O/bench.err; tail -1 $O/bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], (d.get('fused') or {}).get('value'), d['kernel_ms_per_step'], d['roofline'])"
