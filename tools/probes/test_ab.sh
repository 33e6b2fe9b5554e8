#!/bin/bash
# runs a pytest selection under several prebuilt libraries: bash tools/probes/test_ab.sh "lib1 lib2" "<pytest args>"
export TMPDIR=/tmp PYTHONPATH=.
cp bipedal_control_amd/libbpmpc.so /tmp/keep.so
for v in $1; do cp tools/probes/lib_$v.bin bipedal_control_amd/libbpmpc.so; echo "== $v"; timeout 900 python -m pytest $2 -q -x 2>&1 | tail -2; done
cp /tmp/keep.so bipedal_control_amd/libbpmpc.so
