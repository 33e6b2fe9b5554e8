#!/bin/bash
# PMC traffic (FETCH_SIZE, WRITE_SIZE) + kernel time of the lineariser for library variants on one box: bash tools/probes/traffic_ab.sh "a b"
export TMPDIR=/tmp PYTHONPATH=.
cp bipedal_control_amd/libbpmpc.so /tmp/keep.so
for v in $1; do
  cp tools/probes/lib_$v.bin bipedal_control_amd/libbpmpc.so
  bash tools/collect_profiles.sh tab_$v ${2:-} > /dev/null 2>&1
  python - <<PY
import json
t=json.load(open("gpurun_out/tab_$v"+"_traffic.json"))
for k,v in t["all_kernels"].items():
    if "linearize" in k: print("$v", k, "bytes/launch %.1f MB  fetch %.1f MiB write %.1f MiB" % (v["hbm_bytes_per_launch"]/1e6, v["FETCH_SIZE_KiB_per_launch"]/1024, v["WRITE_SIZE_KiB_per_launch"]/1024))
import csv
for r in csv.DictReader(open("gpurun_out/tab_$v"+"_kernel_stats.csv")):
    if "linearize" in r["kernel"]: print("   ", r["kernel"][:60], r["avg_ns"])
PY
done
cp /tmp/keep.so bipedal_control_amd/libbpmpc.so
