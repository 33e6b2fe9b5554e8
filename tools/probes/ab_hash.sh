#!/bin/bash
# solution hashes of libraries tools/probes/lib_<name>.bin on one box: bash tools/probes/ab_hash.sh "name1 name2"
export TMPDIR=/tmp PYTHONPATH=.
cp bipedal_control_amd/libbpmpc.so /tmp/keep.so
for v in $1; do cp tools/probes/lib_$v.bin bipedal_control_amd/libbpmpc.so; python tools/probes/sol_hash.py $v 2>&1 | grep -v amdgpu.ids; done
cp /tmp/keep.so bipedal_control_amd/libbpmpc.so
