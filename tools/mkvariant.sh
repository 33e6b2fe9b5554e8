#!/bin/bash
# Builds a variant of libbpmpc.so in which ONE translation unit is compiled with extra flags (or from the working tree as it is) and the
# others are taken from csrc/build/default:   bash tools/mkvariant.sh <name> <unit, e.g. k_node> ["-DFLAG=1 ..."]
# -> tools/probes/lib_<name>.bin, cycled through bench lines on one box by tools/ab_libs.sh
set -e
NAME=$1; UNIT=${2:-k_node}; FLAGS=${3:-}
ROOT=$(cd $(dirname $0)/.. && pwd)
CSRC=$ROOT/bipedal_control_amd/csrc
python -m bipedal_control_amd.build > /dev/null
mkdir -p $CSRC/build/variant_$NAME
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $FLAGS -x hip -c $CSRC/$UNIT.hip -o $CSRC/build/variant_$NAME/$UNIT.o \
  -Rpass-analysis=kernel-resource-usage 2> $CSRC/build/variant_$NAME/$UNIT.remarks || { grep -E 'error' -A5 $CSRC/build/variant_$NAME/$UNIT.remarks; exit 1; }
OBJS=""
for o in $CSRC/build/default/*.o; do b=$(basename $o); if [ $b = $UNIT.o ]; then OBJS="$OBJS $CSRC/build/variant_$NAME/$UNIT.o"; else OBJS="$OBJS $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/probes/lib_$NAME.bin $OBJS
grep -E 'Function Name|VGPRs:|AGPRs|Scratch|Occupancy|LDS Size' $CSRC/build/variant_$NAME/$UNIT.remarks | grep -A5 "${4:-k_linearize_fast}" | sed 's/.*remark: //; s/ \[-Rpass.*//' | paste - - - - - - | cut -c1-260
