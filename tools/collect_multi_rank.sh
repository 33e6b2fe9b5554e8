#!/bin/bash
# Functional dry runs of the N = 8 code path on ONE device (no 8-GPU node in the builder's pool; gloo instead of RCCL, which refuses two ranks on
# one device): configs[2] (strong scaling, 4096 problems, 512 per rank, gather to rank 0) and configs[4] (gait sweep, gaits split over the ranks),
# and the kernel trace of the per-rank shape of configs[2] (batch 512).  Numbers are NOT scaling numbers: eight ranks share one GPU.
#   gpurun --timeout 1500 -- 'bash tools/collect_multi_rank.sh r05'
set -u
TAG=${1:-r05}
export TMPDIR=/tmp PYTHONPATH=. BPMPC_BENCH_ONE_DEVICE=1
O=gpurun_out; mkdir -p $O
timeout 900 python bench.py --gpus 8 --scaling strong --global-batch 4096 --steps 5 --warmup 2 --cpu-sample 0 --no-fused --gather-report > $O/${TAG}_8ranks_strong4096_one_device.json 2> $O/${TAG}_8ranks_strong.err
timeout 900 python bench.py --gpus 8 --workload gait-sweep --steps 3 --warmup 1 --cpu-sample 0 --no-fused --gather-report > $O/${TAG}_8ranks_sweep_root_one_device.json 2> $O/${TAG}_8ranks_sweep.err
unset BPMPC_BENCH_ONE_DEVICE
# RCCL itself, one rank (all this pool offers): its INFO log parsed into config.distributed.collective (bench.py --gather-report)
BPMPC_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-fused --gather-report > $O/${TAG}_rccl_one_rank_report.json 2> $O/${TAG}_rccl_one_rank.err
python -c "
import json
d=json.loads(open('$O/${TAG}_rccl_one_rank_report.json').read().strip().splitlines()[-1]); print('rccl 1 rank:', json.dumps(d['config']['distributed'])[:600])"
rocprofv3 --kernel-trace --stats -d $O/${TAG}_b512_stats -o run -- python bench.py --steps 10 --warmup 3 --cpu-sample 0 --batch 512 > $O/${TAG}_b512_bench_under_rocprof.json 2> $O/${TAG}_b512_stats.log
DB=$(find $O/${TAG}_b512_stats -name "*.db" | head -1)
python tools/summarize_rocpd.py "$DB" $O/${TAG}_b512_kernel_stats.csv > /dev/null
for f in strong4096 sweep_root; do python -c "
import json
d=json.loads(open('$O/${TAG}_8ranks_${f}_one_device.json').read().strip().splitlines()[-1]); c=d['config']
print('$f', d['value'], d['n_gpus'], c['distributed']['world_size_seen'], c['distributed']['gather'], c['job_report']['gather_consistent'], c['job_report']['failures'], c['distributed'].get('collective'))"; done
head -12 $O/${TAG}_b512_kernel_stats.csv
