#!/usr/bin/env python3
"""Per-kernel, per-launch averages of every counter found in the given rocprofv3 --pmc output directories.

usage: summarize_counters.py <out prefix> <pmc dir> [<pmc dir> ...]      -> <prefix>.json, <prefix>.csv
SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md); derived columns:
  valu_per_wave      SQ_INSTS_VALU / SQ_WAVES
  stall_parked       SQ_WAIT_ANY / SQ_WAVE_CYCLES        (wave parked on s_waitcnt / barrier)
  stall_issue        SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES   (issue stall: dependency / pipe busy)
  active             SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  lds_conflict_frac  SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE"""
import csv
import glob
import json
import os
import sys


def main(prefix, dirs):
    acc = {}
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    name = row["Kernel_Name"].split("(")[0].replace("void bpmpc::", "")
                    if not name.startswith("k_"):
                        continue
                    k = acc.setdefault(name, {"vgpr": row.get("VGPR_Count"), "agpr": row.get("Accum_VGPR_Count"), "lds": row.get("LDS_Block_Size"),
                                               "scratch": row.get("Scratch_Size"), "workgroup": row.get("Workgroup_Size"), "c": {}})
                    e = k["c"].setdefault(row["Counter_Name"], [0.0, 0])
                    e[0] += float(row["Counter_Value"])
                    e[1] += 1
    out = {}
    for name, k in sorted(acc.items()):
        c = {cn: v[0] / max(1, v[1]) for cn, v in k["c"].items()}
        launches = max(v[1] for v in k["c"].values())
        r = {"launches": launches, "vgpr": k["vgpr"], "agpr": k["agpr"], "lds_bytes": k["lds"], "scratch_bytes": k["scratch"], "workgroup": k["workgroup"]}
        r.update({cn: round(v, 1) for cn, v in sorted(c.items())})

        def ratio(a, b):
            return round(c[a] / c[b], 4) if a in c and b in c and c[b] > 0 else None
        r["valu_per_wave"] = ratio("SQ_INSTS_VALU", "SQ_WAVES")
        r["stall_parked"] = ratio("SQ_WAIT_ANY", "SQ_WAVE_CYCLES")
        r["stall_issue"] = ratio("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES")
        r["active"] = ratio("SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES")
        r["lds_conflict_frac"] = ratio("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")
        out[name] = r
    with open(prefix + ".json", "w") as f:
        json.dump(out, f, indent=1)
    cols = sorted({k for r in out.values() for k in r})
    with open(prefix + ".csv", "w") as f:
        f.write("kernel," + ",".join(cols) + "\n")
        for name, r in out.items():
            f.write('"%s",' % name + ",".join(str(r.get(cn, "")) for cn in cols) + "\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
