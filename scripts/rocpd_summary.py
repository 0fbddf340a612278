#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd .db (kernel trace, optional PMC counters) as text.

    python scripts/rocpd_summary.py gpurun_out/prof_r01/bench_results.db > profiles/r01_kernel_stats.txt
"""
import json
import sqlite3
import sys


def main(path, json_out=None):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(
        "select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
        "from kernels group by %s order by sum(end-start) desc" % (name_col, name_col)).fetchall()
    total = sum(r[2] for r in rows) or 1
    print("# kernel-trace summary of %s" % path)
    print("%-90s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for name, calls, tot, avg, mn, mx in rows:
        print("%-90s %8d %12.1f %10.2f %10.2f %10.2f %6.1f" % (
            name[:90], calls, tot / 1e3, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
    try:
        pm = cur.execute(
            "select kernel_name, counter_name, count(*), avg(value), min(value), max(value) "
            "from counters_collection group by kernel_name, counter_name "
            "order by kernel_name, counter_name").fetchall()
    except sqlite3.Error:
        pm = []
    if json_out:
        doc = {}
        for name, calls, tot, avg, mn, mx in rows:
            doc.setdefault(name, {})["calls"] = calls
            doc[name]["avg_us"] = avg / 1e3
        for name, ctr, cnt, avg, mn, mx in pm:
            doc.setdefault(name, {}).setdefault("counters", {})[ctr] = avg
        with open(json_out, "w") as fh:
            json.dump(doc, fh, indent=1)
    if pm:
        print("\n# PMC counters per dispatch (avg / min / max over dispatches)")
        for name, ctr, cnt, avg, mn, mx in pm:
            if name.startswith("void at::") or name.startswith("__amd"):
                continue
            print("%-70s %-22s n=%-5d avg=%-14.1f min=%-14.1f max=%.1f" % (name[:70], ctr, cnt, avg, mn, mx))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
