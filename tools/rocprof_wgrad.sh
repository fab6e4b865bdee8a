#!/bin/bash
# usage (on the GPU box, from the repo root): tools/rocprof_wgrad.sh <tag> [ENV=VAL ...]
# Runs tools/bench_wgrad.py under rocprofv3 --kernel-trace and prints the median KERNEL duration (not event time) of
# the weight-gradient partial kernel and of the slab reduction for every shape (23 dispatches per shape, in order).
tag=$1; shift
out=$PWD/gpurun_out/wgprof_$tag
rm -rf "$out"
root=$PWD
( cd /tmp && export TMPDIR=/tmp && env "$@" rocprofv3 --kernel-trace -d "$out" --output-format csv -- python3 "$root/tools/bench_wgrad.py" > "$out.log" 2>&1 )
python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
part = [r for r in rows if "wgrad_mfma" in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"]]
red = [r for r in rows if "wgrad_reduce" in r["Kernel_Name"]]
import os
shapes = ["32->32@256", "64->32@256", "64->64@128", "32->64@128", "128->64@128", "128->128@64", "64->128@64", "128->128@32"]
def med(rs):
    d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rs)
    return d[len(d) // 2]
tot = 0
for i, s in enumerate(shapes):
    p, r = part[i * 23:(i + 1) * 23][3:], red[i * 23:(i + 1) * 23][3:]
    if not p: break
    name = p[0]["Kernel_Name"].split("::")[-1].split("(")[0]
    print(f"{s:12s} {name:28s} partials {med(p):7.1f} us  reduce {med(r):5.1f} us  total {med(p) + med(r):7.1f}")
PY
