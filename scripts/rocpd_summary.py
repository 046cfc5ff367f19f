"""Summarise a rocprofv3 rocpd database (kernel trace): per-kernel totals and the timeline of the last step.

usage: python scripts/rocpd_summary.py <results.db> <steps_in_trace> [--timeline]
"""
import re
import sqlite3
import sys


def main():
    db, steps = sys.argv[1], int(sys.argv[2])
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"kernel time total {tot:.1f} us, per step {tot / steps:.1f} us over {steps} steps")
    print(f"{'kernel':72s} {'calls/step':>10s} {'us/step':>9s} {'avg us':>8s}")
    for name, n, t, avg in rows:
        nm = re.sub(r"lamp::", "", name)
        nm = re.sub(r"\(.*", "", nm)[:72]
        print(f"{nm:72s} {n / steps:10.1f} {t / steps:9.1f} {avg:8.1f}")
    if "--timeline" in sys.argv:
        tl = c.execute("select name,start,end,grid_x,workgroup_x from kernels order by start").fetchall()
        per = len(tl) // steps
        seg = tl[-per:]
        t0, prev = seg[0][1], None
        for i, (name, s, e, g, w) in enumerate(seg):
            nm = re.sub(r"lamp::", "", name)
            nm = re.sub(r"\(.*", "", nm)[:60]
            gap = (s - prev) / 1e3 if prev else 0.0
            print(f"{i:3d} {(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} gap{gap:6.1f} {g // w:6d} {nm}")
            prev = e


if __name__ == "__main__":
    main()
