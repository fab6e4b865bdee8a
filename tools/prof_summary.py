#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: ms/step and calls/step per kernel.
usage: prof_summary.py <dir-with-*_kernel_stats.csv> <steps-profiled> [top]"""
import csv
import glob
import sys

d, steps = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
f = glob.glob(f"{d}/**/*_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total {tot / 1e6 / steps:.3f} ms/step over {steps:g} steps ({f})")
for r in rows[:top]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{float(r['TotalDurationNs']) / 1e6 / steps:8.3f} ms/step {int(r['Calls']) / steps:7.1f} calls/step "
          f"avg {float(r['AverageNs']) / 1e3:8.1f} us {float(r['Percentage']):5.1f}%  {n[:80]}")
