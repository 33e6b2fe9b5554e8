#!/bin/bash
# phase cycles of lineariser variants built with -DBPMPC_LINFAST_PROFILE: bash tools/probes/prof_lin.sh "name1 name2"
export TMPDIR=/tmp PYTHONPATH=.
cp bipedal_control_amd/libbpmpc.so /tmp/keep.so
for v in $1; do cp tools/probes/lib_$v.bin bipedal_control_amd/libbpmpc.so; echo $v; python tools/linearize_phase_profile.py 2>&1 | tail -2; done
cp /tmp/keep.so bipedal_control_amd/libbpmpc.so
