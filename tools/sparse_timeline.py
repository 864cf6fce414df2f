#!/usr/bin/env python
"""Per-kernel start / end timeline of ONE pass of the sparse segment (7 rulebooks + 14 sparse convs) from a rocprofv3
--kernel-trace database (rocpd sqlite): which kernels overlap, what the critical path is.

    rocprofv3 --kernel-trace -d DIR -- python tools/run_sparse_only.py --config car --reps 5 [--graph]
    python tools/sparse_timeline.py DIR/**/*_results.db [out.txt]

The last pass is cut at the last `hash_build_kernel` (the first kernel of a pass); times are microseconds from its start."""
import glob
import sqlite3
import sys


def main():
    path = sys.argv[1]
    if "*" in path:
        path = sorted(glob.glob(path, recursive=True))[-1]
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("PRAGMA table_info(kernels)").fetchall()]
    extra = [c for c in ("queue_id", "stream_id", "queue", "stream") if c in cols]
    sel = "name, start, end" + "".join(", " + c for c in extra)
    rows = cur.execute("select %s from kernels order by start" % sel).fetchall()
    firsts = [i for i, r in enumerate(rows) if "hash_build" in r[0]]
    assert firsts, "no hash_build_kernel in the trace"
    # one pass = from the pyramid's hash build to the kernel before the next one (the last pass runs to the end)
    i0 = firsts[-1]
    seg = rows[i0:]
    # ... and ends with the 1x1x1 layer (spconv_pw_kernel): whatever the process launches afterwards is not part of it
    last = max((i for i, r in enumerate(seg) if "spconv_pw_kernel" in r[0]), default=len(seg) - 1)
    seg = seg[:last + 1]
    t0 = seg[0][1]
    out = ["# columns of `kernels`: %s" % ", ".join(cols),
           "# one pass of the sparse segment (last of %d), microseconds from the start of hash_build_kernel" % len(firsts),
           "%9s %9s %8s  %-14s %s" % ("start_us", "end_us", "dur_us", "/".join(extra) or "-", "kernel")]
    busy_end, idle, crit = 0.0, 0.0, []
    for r in seg:
        s, e = (r[1] - t0) / 1e3, (r[2] - t0) / 1e3
        if s > busy_end and busy_end > 0:
            idle += s - busy_end
        overl = " (overlaps the previous)" if s < busy_end - 0.05 else ""
        busy_end = max(busy_end, e)
        name = r[0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        out.append("%9.2f %9.2f %8.2f  %-14s %s%s" % (s, e, e - s, "/".join(str(v) for v in r[3:]) or "-", name[:90], overl))
    total = busy_end
    ksum = sum((r[2] - r[1]) / 1e3 for r in seg)
    out.append("# pass: %.1f us wall, %.1f us of kernel time summed, %.1f us with no kernel running (launch gaps / dependencies)"
               % (total, ksum, idle))
    txt = "\n".join(out) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    else:
        sys.stdout.write(txt)


if __name__ == "__main__":
    main()
