#!/bin/bash
# Run ON THE GPU BOX: tools/prof_top.sh <out name> [bench args...] -- rocprofv3 kernel stats of bench.py, top rows printed
name=$1; shift
root=$PWD
out=$root/gpurun_out/$name
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/trace" --output-format csv -- python3 "$root/bench.py" --steps 6 --warmup 3 --no-cpu-baseline "$@" > "$out/bench.json" 2> "$out/bench.err"
f=$(find "$out/trace" -name '*kernel_stats.csv' | head -1)
cp "$f" "$out/kernel_stats.csv"
find "$out/trace" -name '*kernel_trace.csv' -delete
python3 - "$out/kernel_stats.csv" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]:
    n=r['Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0]
    print(f"{n[:64]:64s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1e3:8.1f}us pct={float(r['Percentage']):5.2f}")
PY
