#!/bin/bash
# round 6: lineariser A/B on one box - parity of the LQ model, phase profiles, bench lines.  usage: bash tools/probes/r6_ab.sh "<libs>" "<prof libs>" [tag] [shapes]
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/${3:-r6b}; mkdir -p $O
[ -n "$2" ] && bash tools/probes/prof_lin.sh "$2" 2>&1 | tee $O/prof_lin.txt
bash tools/probes/ab_lin.sh "$1" ${3:-r6b} "${4:---batch 256|--batch 4096|--robot g1 --batch 1024}"
