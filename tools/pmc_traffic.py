#!/usr/bin/env python3
"""Fold two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE, each its own `--pmc` run with
`--output-format csv` of the same `bench.py` command) into profiles/<name>.json: average HBM-side bytes per
launch and kernel.  gfx950 corrections per MI355X_MICROARCH.md (HBM / rocprofv3 section): counter unit is KB;
FETCH_SIZE under-counts wide coalesced streams by 2x (128-B requests counted as 64 B), WRITE_SIZE is exact.

usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json> <optimiser steps in the profiled run> ["note"] ["workload key"]
(the workload key -- "<config basename>:b<batch>:<size>" -- and ``source_hash`` -- sha256 over csrc/, bench.source_hash()
-- are what bench.py matches before it attaches these numbers: a counter file of another workload OR another kernel
version is never paired with the live timings)
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    depth, out = 0, []
    for ch in name:          # drop the trailing argument list, keep template arguments
        if ch == "(" and depth == 0 and out and "".join(out).count("<") == "".join(out).count(">"):
            break
        out.append(ch)
    return "".join(out).strip()


def load(directory: str, counter: str):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] == counter:
                a = acc[short(row["Kernel_Name"])]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    return acc


def _source_hash() -> str:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    return bench.source_hash()


def main() -> None:
    fetch_dir, write_dir, out = sys.argv[1:4]
    steps = int(sys.argv[4])
    note = sys.argv[5] if len(sys.argv) > 5 else ""
    workload = sys.argv[6] if len(sys.argv) > 6 else ""
    fetch, write = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, [0.0, 0]), write.get(k, [0.0, 0])
        n = max(f[1], w[1], 1)
        kernels[k] = {"launches": n, "launches_per_step": round(n / steps, 2),
                      "fetch_bytes": round(2 * 1024 * f[0] / max(f[1], 1)), "write_bytes": round(1024 * w[0] / max(w[1], 1))}
    json.dump({"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes).  Counter unit KB; fetch_bytes = "
                        "2 x FETCH_SIZE (gfx950 counts 128-B requests as 64 B for wide coalesced streams, "
                        "MI355X_MICROARCH.md HBM section); write_bytes = WRITE_SIZE.  Averages per launch.  " + note,
               "workload": workload, "source_hash": _source_hash(), "steps_profiled": steps,
               "bytes_per_step": round(sum((k["fetch_bytes"] + k["write_bytes"]) * k["launches_per_step"] for k in kernels.values())),
               "kernels": kernels}, open(out, "w"), indent=1)
    print(f"{len(kernels)} kernels -> {out}")


if __name__ == "__main__":
    main()
