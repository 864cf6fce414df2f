#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite db or *_kernel_stats.csv) as text:
per-kernel calls / total / mean / min / max duration (us) and share of GPU time.
usage: python tools/rocprof_summary.py <results.db> [out.txt]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                       "max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = ["# rocprofv3 --kernel-trace --stats summary (durations in microseconds); total GPU kernel time %.1f us" % tot,
             "%-100s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "mean_us", "min_us", "max_us", "pct")]
    for r in rows:
        lines.append("%-100s %7d %12.1f %10.2f %10.2f %10.2f %6.2f" % (r[0][:100], r[1], r[2], r[3], r[4], r[5],
                                                                        100 * r[2] / tot))
    txt = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    else:
        sys.stdout.write(txt)


if __name__ == "__main__":
    main()
