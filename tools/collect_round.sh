#!/bin/bash
# Everything a round's evidence is made of, in ONE call on the GPU box (boxes of the pool differ in clocks, so numbers that are compared
# must come from the same box):   gpurun --timeout 2400 -- 'bash tools/collect_round.sh <tag>'
# Writes gpurun_out/<tag>/: GPU tests, bench lines (headline, G1 = configs[3], gait sweep = configs[4], batch 4096 = configs[2] on one GPU,
# batch 512 = its per-GPU share), batch-1 latency, WBC, and gpurun_out/<tag>_*: rocprofv3 kernel stats + PMC traffic for the headline and the
# three large shapes, SQ counters.  Copy what is judged into profiles/.
set -u
TAG=${1:-r03}
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $O/pytest.log
# counters first: the bench lines below carry `roofline.traffic` only if the committed counter files were collected on THIS tree (csrc hash), so the
# files of this run are put where profiles/traffic_index.json points (on the box; tools/publish_profiles.sh does the same at home afterwards)
bash tools/collect_profiles.sh $TAG > $O/profiles.log 2>&1
bash tools/collect_profiles.sh ${TAG}_b4096 --batch 4096 > $O/profiles_b4096.log 2>&1
bash tools/collect_profiles.sh ${TAG}_g1 --robot g1 --batch 1024 > $O/profiles_g1.log 2>&1
bash tools/collect_profiles.sh ${TAG}_sweep --workload gait-sweep --batch 4096 > $O/profiles_sweep.log 2>&1
bash tools/collect_counters.sh $TAG > $O/counters.log 2>&1
PUB=${2:-$(echo $TAG | sed 's/[a-z]*$//')}
for s in "" b4096_ g1_ sweep_; do cp gpurun_out/${TAG}_${s}traffic.json profiles/${PUB}_${s}traffic.json; done
cp gpurun_out/${TAG}_counters.json profiles/${PUB}_sq_counters.json
sed -i "s/r0[0-9][a-z]*_traffic/${PUB}_traffic/; s/r0[0-9][a-z]*_\(b4096\|g1\|sweep\)_traffic/${PUB}_\1_traffic/; s/r0[0-9][a-z]*_sq_counters/${PUB}_sq_counters/" profiles/traffic_index.json
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-fused > /dev/null 2>&1    # warm-up: the first bench after the test suite runs ~2 % slow (clocks)
timeout 600 python bench.py > $O/bench_h1.json 2> $O/bench_h1.err
timeout 600 python bench.py --robot g1 --batch 1024 --cpu-sample 16 > $O/bench_g1.json 2>> $O/bench_h1.err
timeout 600 python bench.py --workload gait-sweep --batch 4096 --cpu-sample 16 > $O/bench_sweep.json 2>> $O/bench_h1.err
timeout 600 python bench.py --batch 4096 --cpu-sample 0 > $O/bench_4096.json 2>> $O/bench_h1.err
timeout 600 python bench.py --batch 512 --cpu-sample 0 > $O/bench_512.json 2>> $O/bench_h1.err
timeout 600 python bench.py --robot hunter --cpu-sample 16 > $O/bench_hunter.json 2>> $O/bench_h1.err
timeout 600 python bench.py --robot h1:hard --cpu-sample 16 > $O/bench_h1_hard.json 2>> $O/bench_h1.err
timeout 600 python bench.py --gait-start -1.225 --cpu-sample 0 > $O/bench_h1_midswing.json 2>> $O/bench_h1.err      # round 4's input: nobody back-tracks
timeout 600 python bench.py --solver ddp --cpu-sample 2 > $O/bench_ddp.json 2>> $O/bench_h1.err                          # the reference's second solver at the configs[1] shape
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_ddp_stats -o run -- python bench.py --solver ddp --steps 5 --warmup 2 --cpu-sample 0 --no-fused > gpurun_out/${TAG}_ddp_bench_under_rocprof.json 2> gpurun_out/${TAG}_ddp_stats.log
python tools/summarize_rocpd.py "$(find gpurun_out/${TAG}_ddp_stats -name '*.db' | head -1)" gpurun_out/${TAG}_ddp_kernel_stats.csv > /dev/null
tools/probes/write_roof.bin 256 103 > $O/write_roof.json 2>&1; tools/probes/write_roof.bin 4096 103 >> $O/write_roof.json 2>&1; tools/probes/write_roof.bin 1024 103 24 >> $O/write_roof.json 2>&1
timeout 600 python tools/closed_loop_soak.py > $O/soak.log 2>&1
timeout 300 python tools/latency_probe.py > $O/latency.log 2>&1
timeout 300 python tools/wbc_probe.py > $O/wbc.log 2>&1
cat $O/pytest.log; tail -n 2 $O/latency.log; for f in h1 g1 sweep 4096 512 hunter h1_hard h1_midswing ddp; do python -c "
import json,sys
try:
    d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], (d.get('fused') or {}).get('value'), d['kernel_ms_per_step'])
except Exception as e: print('$f', 'FAILED', e)"; done
