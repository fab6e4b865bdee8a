#!/bin/bash
# Run ON THE GPU BOX: tools/trace_kernels.sh <regex> [bench args] -- per-launch durations (us) of kernels matching <regex>
# in launch order over one short bench run (PTI_WGRAD_STREAM=0: one stream, no overlap stretching).
re=$1; shift
root=$PWD
out=$root/gpurun_out/ktrace
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
PTI_WGRAD_STREAM=0 rocprofv3 --kernel-trace -d "$out/t" --output-format csv -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> "$out/err"
f=$(find "$out/t" -name '*kernel_trace.csv' | head -1)
python3 - "$f" "$re" <<'PY'
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pat = re.compile(sys.argv[2])
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if pat.search(n):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        key = (n, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", ""))
        agg.setdefault(key, []).append(d)
for (n, g, w), ds in agg.items():
    print(f"{n[:48]:48s} grid={g:>9s} wg={w:>4s} launches={len(ds):3d} avg={sum(ds)/len(ds):8.1f}us min={min(ds):8.1f}")
PY
rm -rf "$out/t"
