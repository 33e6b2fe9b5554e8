#!/bin/bash
# Copies the evidence of one tools/collect_round.sh run from gpurun_out/ (scratch) into profiles/ (tracked) under the round's names.
# usage: bash tools/publish_profiles.sh <run tag, e.g. r03b> <published tag, e.g. r03>
set -eu
RUN=$1; PUB=$2; G=gpurun_out; P=profiles
last() { tail -n 1 "$1"; }
cp $G/${RUN}_kernel_stats.csv $P/${PUB}_bench_kernel_stats.csv
last $G/${RUN}_bench_under_rocprof.json > $P/${PUB}_bench_line_under_rocprof.json
cp $G/${RUN}_traffic.json $P/${PUB}_traffic.json
for s in b4096 g1 sweep; do
  cp $G/${RUN}_${s}_kernel_stats.csv $P/${PUB}_${s}_kernel_stats.csv
  cp $G/${RUN}_${s}_traffic.json $P/${PUB}_${s}_traffic.json
  last $G/${RUN}_${s}_bench_under_rocprof.json > $P/${PUB}_${s}_bench_line_under_rocprof.json
done
for b in h1 g1 sweep 4096 512 hunter h1_hard h1_midswing ddp; do last $G/$RUN/bench_$b.json > $P/${PUB}_bench_line_$b.json; done
cp $G/$RUN/pytest.log $P/${PUB}_gpu_pytest.txt
tail -n 2 $G/$RUN/latency.log > $P/${PUB}_latency.txt
tail -n 1 $G/$RUN/wbc.log > $P/${PUB}_wbc.txt
cp $G/${RUN}_counters.csv $P/${PUB}_sq_counters.csv
cp $G/${RUN}_counters.json $P/${PUB}_sq_counters.json
for c in FETCH_SIZE WRITE_SIZE; do
  f=$(find $G/${RUN}_pmc_$c -name "*counter_collection.csv" | head -1)
  head -n 102 "$f" > $P/${PUB}_pmc_$(echo $c | tr A-Z a-z).csv
done
cp $G/${RUN}_ddp_kernel_stats.csv $P/${PUB}_ddp_kernel_stats.csv
cp $G/$RUN/write_roof.json $P/${PUB}_write_roof.json
tail -n 1 $G/$RUN/soak.log > $P/${PUB}_soak.txt
[ -f $G/parity_blocks.json ] && cp $G/parity_blocks.json $P/${PUB}_parity_blocks.json
sed -i "s/r0[0-9][a-z]*_traffic/${PUB}_traffic/; s/r0[0-9][a-z]*_\(b4096\|g1\|sweep\)_traffic/${PUB}_\1_traffic/; s/r0[0-9][a-z]*_sq_counters/${PUB}_sq_counters/" $P/traffic_index.json
echo published $RUN as $PUB
