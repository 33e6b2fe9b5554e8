#!/bin/bash
# bitwise comparison of the lineariser outputs of two libraries on one box: bash tools/probes/lq_ab.sh libA libB
export TMPDIR=/tmp PYTHONPATH=.
cp bipedal_control_amd/libbpmpc.so /tmp/keep.so
for v in $1 $2; do cp tools/probes/lib_$v.bin bipedal_control_amd/libbpmpc.so; python tools/probes/lq_dump.py /tmp/lq_$v.npz 2>&1 | tail -1; done
cp /tmp/keep.so bipedal_control_amd/libbpmpc.so
python tools/probes/lq_dump.py /tmp/lq_$1.npz /tmp/lq_$2.npz | grep -v BITWISE | tail -40
