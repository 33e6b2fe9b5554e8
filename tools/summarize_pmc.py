#!/usr/bin/env python3
"""Aggregate the counter CSVs of two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into per-kernel HBM bytes per launch.

usage: summarize_pmc.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json>
Units and corrections as prescribed by /opt/skills/guides/MI355X_MICROARCH.md: both counters are in KiB; on gfx950 FETCH_SIZE
reports half of the streamed read bytes (doubled here), WRITE_SIZE is taken as reported."""
import csv
import glob
import json
import os
import sys


def per_kernel(directory, counter):
    acc = {}
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                name = row["Kernel_Name"].split("(")[0].replace("void bpmpc::", "")
                e = acc.setdefault(name, [0.0, 0])
                e[0] += float(row["Counter_Value"])
                e[1] += 1
    return acc


def main(fetch_dir, write_dir, out, extra=""):
    fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    kernels = {}
    for name in sorted(set(fetch) | set(write)):
        if not name.startswith("k_"):
            continue
        f, nf = fetch.get(name, [0.0, 1])
        w, nw = write.get(name, [0.0, 1])
        kernels[name] = {"launches": nf, "FETCH_SIZE_KiB_per_launch": round(f / max(nf, 1), 1), "WRITE_SIZE_KiB_per_launch": round(w / max(nw, 1), 1),
                         "hbm_bytes_per_launch": int(1024 * (2.0 * f / max(nf, 1) + w / max(nw, 1)))}
    # the roofline kernel is the materialised lineariser (template argument `true`); per-step totals of both solve modes: every kernel
    # class runs once per step except the two linearisers, which split the steps between them
    import re
    lin = next((k for k in kernels if re.match(r"k_linearize_fast<\d+, true", k)), None) or next((k for k in kernels if k.startswith("k_linearize_fast")), None)
    lin_f = next((k for k in kernels if re.match(r"k_linearize_fast<\d+, false", k)), None)
    # the sweep: k_riccati_fast* (workgroup per problem) or k_riccati_wave (wavefront per problem; its roll-out k_riccati_rollout is a launch of its own)
    ric = next((k for k in kernels if k.startswith("k_riccati_fast") or k.startswith("k_riccati_wave")), None)
    steps = kernels[ric]["launches"] if ric else 0
    per_step_common = sum(v["hbm_bytes_per_launch"] * v["launches"] for k, v in kernels.items() if not k.startswith("k_linearize_fast") and not k.startswith("k_prepare")) / max(1, steps)
    import re
    mb, mi = re.search(r"--batch (\d+)", extra), re.search(r"--intervals (\d+)", extra)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from bipedal_control_amd.build import csrc_hash
    res = {"csrc_hash": csrc_hash(), "batch": int(mb.group(1)) if mb else 256, "intervals": int(mi.group(1)) if mi else (150 if "gait-sweep" in extra else 100), "bench_arguments": extra, "kernel": lin,
           "materialised_hbm_bytes_per_step": int(per_step_common + kernels[lin]["hbm_bytes_per_launch"]) if lin and steps else None,
           "fused_hbm_bytes_per_step": int(per_step_common + kernels[lin_f]["hbm_bytes_per_launch"]) if lin_f and steps else None,
           "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --output-format csv -- python bench.py --steps 3 --warmup 1 --cpu-sample 0 " + extra,
           "hbm_bytes_per_launch": kernels[lin]["hbm_bytes_per_launch"] if lin else None,
           "note": "FETCH_SIZE doubled (gfx950 reports half of the streamed read bytes, MI355X_MICROARCH.md); WRITE_SIZE as reported.",
           "all_kernels": kernels}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])
