#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) kernel trace into the --stats table (per-kernel calls, total,
average, min, max, % of GPU time) plus inter-kernel gap statistics.  Usage:
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [> profiles/x_kernel_stats.txt]
(`rocprofv3 --kernel-trace --stats` writes this database by default on ROCm 7.2; `-f csv` gives
the same numbers as *_kernel_stats.csv.)"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    c = db.cursor()
    cols = [r[1] for r in c.execute("pragma table_info('kernels')")]
    rows = c.execute("select name, start, end from kernels order by start").fetchall() if "name" in cols else []
    if not rows:
        rows = c.execute(
            "select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"
        ).fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print(f"# source: {path}\n# kernels: {len(rows)} dispatches, {tot/1e6:.3f} ms total kernel time, "
          f"span {(rows[-1][2]-rows[0][1])/1e6:.3f} ms")
    print(f"{'Name':<70} {'Calls':>7} {'Total(ms)':>11} {'Avg(us)':>10} {'Min(us)':>9} {'Max(us)':>9} {'%':>6}")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        short = name if len(name) <= 70 else name[:67] + "..."
        print(f"{short:<70} {a[0]:>7} {a[1]/1e6:>11.3f} {a[1]/a[0]/1e3:>10.2f} {a[2]/1e3:>9.2f} {a[3]/1e3:>9.2f} {100*a[1]/tot:>6.2f}")
    # gaps between consecutive lstm steps (launch-bound region)
    gaps = [rows[i + 1][1] - rows[i][2] for i in range(len(rows) - 1)
            if "lstm_step" in rows[i][0] and "lstm_step" in rows[i + 1][0]]
    if gaps:
        gaps.sort()
        print(f"# gap between consecutive lstm_step dispatches: median {gaps[len(gaps)//2]/1e3:.2f} us, "
              f"mean {sum(gaps)/len(gaps)/1e3:.2f} us, p90 {gaps[int(len(gaps)*0.9)]/1e3:.2f} us")


if __name__ == "__main__":
    main(sys.argv[1])
