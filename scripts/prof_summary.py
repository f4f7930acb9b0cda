#!/usr/bin/env python3
"""Summarise rocprofv3 sqlite outputs (gpurun_out/prof_<tag>/) into a text file for profiles/.
Usage: scripts/prof_summary.py gpurun_out/prof_<tag> > profiles/<name>.txt"""
import glob
import os
import sqlite3
import sys

root = sys.argv[1]


def q(db, sql):
    c = sqlite3.connect(db)
    try:
        return c.execute(sql).fetchall()
    finally:
        c.close()


tr = os.path.join(root, "trace", "trace_results.db")
if os.path.exists(tr):
    print("== rocprofv3 --kernel-trace --stats (per-kernel device time) ==")
    cols = [r[1] for r in q(tr, "pragma table_info(kernels)")]
    rows = q(tr, "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                 "group by name order by sum(end-start) desc")
    tot = sum(r[2] for r in rows) or 1
    print("%-60s %6s %12s %12s %12s %12s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for r in rows:
        print("%-60s %6d %12.1f %12.1f %12.1f %12.1f %6.1f" % (r[0][:60], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3,
                                                                r[5] / 1e3, 100.0 * r[2] / tot))
    kr = q(tr, "select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, workgroup_x, grid_x "
               "from kernels where name like 'rtx::%' group by name") if "vgpr_count" in cols else []
    for r in kr:
        print("   resources %-40s vgpr=%s agpr=%s sgpr=%s lds=%s scratch=%s wg=%s grid=%s" % (r[0][:40], *r[1:]))
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    db = os.path.join(d, "pmc_results.db")
    if not os.path.isdir(d) or not os.path.exists(db):
        continue
    print("\n== rocprofv3 --pmc pass: %s ==" % os.path.basename(d)[4:])
    try:
        cols = [r[1] for r in q(db, "pragma table_info(counters_collection)")]
        rows = q(db, "select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection "
                     "group by kernel_name, counter_name order by kernel_name, counter_name")
        print("%-50s %-28s %6s %18s %18s" % ("kernel", "counter", "disp", "sum", "avg/dispatch"))
        for r in rows:
            print("%-50s %-28s %6d %18.1f %18.1f" % (r[0][:50], r[1], r[2], r[3], r[4]))
    except Exception as e:  # schema differences between rocprofv3 builds
        print("could not read counters_collection:", e, "columns:", cols if 'cols' in dir() else None)
