#!/bin/bash
# Run on the GPU box from the repository root:  bash tools/collect_profiles.sh <tag> [bench.py arguments, e.g. --batch 4096]
# Produces under gpurun_out/<tag>_*: the rocprofv3 kernel-trace summary of the default bench command and two separate PMC passes
# (FETCH_SIZE, WRITE_SIZE; never combined with API tracing).  Copy what should be judged into profiles/.
set -u
TAG=${1:-r03}
shift || true
EXTRA="$*"
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
BENCH="python bench.py --steps 10 --warmup 3 --cpu-sample 0 $EXTRA"
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats -o run -- $BENCH > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_stats.log
DB=$(find $OUT/${TAG}_stats -name "*.db" | head -1)
python tools/summarize_rocpd.py "$DB" $OUT/${TAG}_kernel_stats.csv > /dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/${TAG}_pmc_$C -o run -- python bench.py --steps 3 --warmup 1 --settle 0 --cpu-sample 0 $EXTRA > /dev/null 2> $OUT/${TAG}_pmc_$C.log
done
python tools/summarize_pmc.py $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE $OUT/${TAG}_traffic.json "$EXTRA"
tail -1 $OUT/${TAG}_bench_under_rocprof.json
cat $OUT/${TAG}_kernel_stats.csv
