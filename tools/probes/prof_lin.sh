#!/bin/bash
# phase cycles of lineariser variants: bash tools/probes/prof_lin.sh "name1 name2"  (names containing "evprof": built with -DBPMPC_EVAL_PROFILE, else -DBPMPC_LINFAST_PROFILE)
export TMPDIR=/tmp PYTHONPATH=.
cp bipedal_control_amd/libbpmpc.so /tmp/keep.so
for v in $1; do cp tools/probes/lib_$v.bin bipedal_control_amd/libbpmpc.so; echo $v; T=tools/linearize_phase_profile.py; [[ $v == *evprof* ]] && T=tools/eval_phase_profile.py; python $T 2>&1 | tail -3; done
cp /tmp/keep.so bipedal_control_amd/libbpmpc.so
