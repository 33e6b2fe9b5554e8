#!/bin/bash
# A/B on one GPU box (boxes differ by up to 30 %): builds the library of HEAD and of the working tree, runs bench.py alternately.
# usage: bash tools/ab_bench.sh [bench.py arguments, e.g. --batch 512]   (from the repository root, in the build container;
# it calls gpurun itself)
set -e
python -m bipedal_control_amd.build --force > /dev/null 2>&1
cp bipedal_control_amd/libbpmpc.so tools/probes/lib_work.bin
git stash -q
python -m bipedal_control_amd.build --force > /dev/null 2>&1
cp bipedal_control_amd/libbpmpc.so tools/probes/lib_head.bin
git stash pop -q
python -m bipedal_control_amd.build --force > /dev/null 2>&1
/usr/local/graft/bin/gpurun --timeout 600 -- 'for v in head work head work; do cp tools/probes/lib_$v.bin bipedal_control_amd/libbpmpc.so; echo $v; timeout 100 python bench.py '"$*"' --steps 30 --warmup 3 --cpu-sample 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"ms_per_step\"], d[\"kernel_ms_per_step\"])"; done' 2>&1 | tail -9
rm -f tools/probes/lib_work.bin tools/probes/lib_head.bin
