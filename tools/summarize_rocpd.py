#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd SQLite result (--kernel-trace --stats) into a small per-kernel CSV for profiles/."""
import sqlite3
import sys


def main(db_path, out_path):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(sgpr_count), "
                      "max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    with open(out_path, "w") as f:
        f.write("kernel,calls,total_ns,avg_ns,min_ns,max_ns,percent,vgpr,sgpr,lds_bytes,scratch_bytes,grid_x,workgroup_x\n")
        for r in rows:
            f.write('"%s",%d,%d,%.1f,%d,%d,%.2f,%s,%s,%s,%s,%s,%s\n' % (r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / total, r[6], r[7], r[8], r[9], r[10], r[11]))
    print(open(out_path).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
