#!/bin/bash
# Run on the GPU box from the repository root:  bash tools/collect_counters.sh <tag> [bench args]
# SQ counters of the hot kernels in separate rocprofv3 --pmc passes (8 SQ slots per pass; never combined with API tracing):
#   pass A: issue / stall split          SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
#   pass B: LDS and matrix core          SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM
# Output: gpurun_out/<tag>_counters.json / .csv (per kernel, per launch) - copy into profiles/.
set -u
TAG=${1:-r02}
shift || true
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --settle 0 --cpu-sample 0 $*"
A="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
B="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM"
rocprofv3 --kernel-trace --pmc $A --output-format csv -d $OUT/${TAG}_pmc_sqA -o run -- python bench.py $ARGS > /dev/null 2> $OUT/${TAG}_pmc_sqA.log || tail -5 $OUT/${TAG}_pmc_sqA.log
rocprofv3 --kernel-trace --pmc $B --output-format csv -d $OUT/${TAG}_pmc_sqB -o run -- python bench.py $ARGS > /dev/null 2> $OUT/${TAG}_pmc_sqB.log || tail -5 $OUT/${TAG}_pmc_sqB.log
python tools/summarize_counters.py $OUT/${TAG}_counters $OUT/${TAG}_pmc_sqA $OUT/${TAG}_pmc_sqB
