#!/bin/bash
# round-4 baseline on one box: bench lines at three shapes + lineariser phase profiles (libs built with -DBPMPC_LINFAST_PROFILE / -DBPMPC_EVAL_PROFILE)
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r04a; mkdir -p $O
rocminfo | grep -E 'Compute Unit|Max Clock' | head -4 > $O/box.txt
for ARGS in "--batch 256" "--batch 4096" "--robot g1 --batch 1024"; do
  timeout 300 python bench.py $ARGS --cpu-sample 0 > $O/line.json 2>$O/err.log
  python -c "
import json
d=json.loads(open('$O/line.json').read().strip().splitlines()[-1]); print('$ARGS', d['value'], d['ms_per_step'], (d.get('fused') or {}).get('value'), d['kernel_ms_per_step'], d['roofline']['frac'])" 2>&1 | tail -1 | tee -a $O/lines.txt
done
cp bipedal_control_amd/libbpmpc.so /tmp/keep.so
for V in LINFAST EVAL; do
  cp tools/probes/lib_prof_$V.bin bipedal_control_amd/libbpmpc.so
  T=tools/linearize_phase_profile.py; [ $V = EVAL ] && T=tools/eval_phase_profile.py
  timeout 300 python $T 2>&1 | tail -3 | tee -a $O/phases.txt
done
cp /tmp/keep.so bipedal_control_amd/libbpmpc.so
