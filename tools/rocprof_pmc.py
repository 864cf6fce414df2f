#!/usr/bin/env python
"""Per-kernel PMC counter values from a rocprofv3 --pmc run (rocpd sqlite db): mean per dispatch for each counter.
usage: python tools/rocprof_pmc.py <results.db> [kernel-name-substring]"""
import json
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    cur = db.cursor()
    views = [r[0] for r in cur.execute("select name from sqlite_master where type in ('view','table')")]
    if "counters_collection" not in views:
        print(json.dumps({"error": "no counters_collection view", "objects": views[:40]}))
        return
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
    rows = cur.execute("select %s, counter_name, avg(value), count(*), sum(value) from counters_collection "
                       "group by %s, counter_name" % (kcol, kcol)).fetchall()
    out = {}
    for k, c, avg, n, tot in rows:
        if pat and pat not in k:
            continue
        out.setdefault(k[:120], {})[c] = {"mean_per_dispatch": avg, "dispatches": n}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
