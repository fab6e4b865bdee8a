#!/usr/bin/env python3
"""Summarise the rocpd SQLite database rocprofv3 (ROCm 7.2 default output) writes for
`rocprofv3 --kernel-trace --stats -- python3 bench.py ...` into a per-kernel CSV:
name, calls, total_us, avg_us, percent, ms_per_step, calls_per_step.

usage: rocpd_summary.py <results.db> <out.csv> <steps-in-the-run>
(steps = warmup + timed + 2 instrumented bench steps)
"""
import csv
import sqlite3
import sys


def main() -> None:
    db, out, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    con = sqlite3.connect(db)
    rows = con.execute(
        "select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc").fetchall()
    # top_kernels reports nanoseconds on some builds and microseconds on others: normalise through the raw table
    raw = con.execute("select sum(end - start) from rocpd_kernel_dispatch").fetchone()[0]
    scale = raw / 1e3 / sum(r[2] for r in rows)  # -> microseconds
    total = 0.0
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["name", "calls", "total_us", "avg_us", "percent", "ms_per_step", "calls_per_step"])
        for name, calls, dur, avg, pct in rows:
            us = dur * scale
            total += us
            w.writerow([name, calls, f"{us:.1f}", f"{us / calls:.2f}", f"{pct:.2f}", f"{us / steps / 1e3:.4f}", f"{calls / steps:.1f}"])
    print(f"{len(rows)} kernels, {total / steps / 1e3:.3f} ms of kernel time per step over {steps} steps -> {out}")
    for name, calls, dur, avg, pct in rows[:16]:
        print(f"  {dur * scale / steps / 1e3:7.3f} ms/step {calls / steps:6.1f} calls  {name[:100]}")


if __name__ == "__main__":
    main()
