"""Summarise rocprofv3 (rocpd sqlite) outputs of tools/profile.sh into text.

usage: python tools/prof_summary.py gpurun_out/prof_<tag> > profiles/<name>.txt
"""
import glob
import os
import sqlite3
import sys


def short(name):
    for k in ("trace_kernel", "extend_kernel", "shade_kernel", "raygen_kernel", "resolve_kernel", "first_hit_kernel",
              "to_float_kernel", "tile_list_kernel", "device_records_kernel"):
        if k in name:
            tail = name[name.index(k):].split("(")[0]
            return tail[:56].rstrip(">").rstrip()
    return name.split("(")[0][:60]


def main(root):
    trace = glob.glob(os.path.join(root, "trace", "*.db"))
    if trace:
        db = sqlite3.connect(trace[0])
        print("== kernel trace (rocprofv3 --kernel-trace --stats) ==")
        print(f"{'kernel':58s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s} {'vgpr':>5s} {'sgpr':>5s} {'lds':>6s}")
        rows = db.execute("select name, count(*), sum(duration), avg(duration), max(vgpr_count), "
                          "max(sgpr_count), max(lds_size) from kernels group by name "
                          "order by sum(duration) desc").fetchall()
        total = sum(r[2] for r in rows) or 1
        for n, c, tot, avg, vg, sg, lds in rows:
            print(f"{short(n):58s} {c:7d} {tot / 1e6:10.3f} {avg / 1e3:10.2f} {100 * tot / total:6.1f} {vg:5d} {sg:5d} {lds:6d}")
    for path in sorted(glob.glob(os.path.join(root, "pmc_*", "*.db"))):
        db = sqlite3.connect(path)
        print(f"\n== counters: {os.path.basename(os.path.dirname(path))} ==")
        try:
            rows = db.execute(
                "select name, counter_name, sum(counter_value), count(*) from pmc_events "
                "group by name, counter_name order by name").fetchall()
        except sqlite3.Error as e:
            cols = [c[1] for c in db.execute("pragma table_info(pmc_events)")]
            print("  (could not join:", e, cols, ")")
            continue
        for n, cname, val, cnt in rows:
            print(f"  {short(n):58s} {cname:28s} {val:20.0f}  ({cnt} dispatches)")


if __name__ == "__main__":
    main(sys.argv[1])
