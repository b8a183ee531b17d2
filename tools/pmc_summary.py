"""Summarises rocprofv3 outputs (rocpd sqlite .db or CSV): per kernel, per
counter totals, and kernel durations."""
import csv, glob, os, sys, collections, sqlite3
root = sys.argv[1]
for f in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
    c = sqlite3.connect(f)
    print("DB", os.path.relpath(f, root))
    try:
        rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name").fetchall()
        for r in rows:
            print("  KERNEL %-40s calls=%d total_ms=%.3f avg_ms=%.3f min_ms=%.3f max_ms=%.3f" % (
                r[0][:40], r[1], r[2] / 1e6, r[3] / 1e6, r[4] / 1e6, r[5] / 1e6))
    except Exception as e:
        print("  kernels view:", e)
    try:
        cols = [d[1] for d in c.execute("pragma table_info(counters_collection)")]
        kn = "kernel_name" if "kernel_name" in cols else "name"
        rows = c.execute("select %s, counter_name, count(*), sum(value) from counters_collection group by %s, counter_name" % (kn, kn)).fetchall()
        for r in rows:
            print("  PMC %-28s %-22s dispatches=%d total=%.6g per_dispatch=%.6g" % (r[0][:28], r[1], r[2], r[3], r[3] / r[2]))
    except Exception as e:
        print("  counters_collection:", e)
for f in glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True):
    print("STATS", f)
    print(open(f).read())
