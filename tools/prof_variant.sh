#!/bin/bash
# Run ON THE GPU BOX: tools/prof_variant.sh <bench args...> -- per-kernel ms per step of a bench variant (rocprofv3 --kernel-trace --stats)
root=$PWD
out=/tmp/prof_variant; rm -rf $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out --output-format csv -- python3 "$root/bench.py" --steps 6 --warmup 3 --no-cpu-baseline "$@" > /dev/null 2>&1
f=$(find $out -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 6 + 3 + 4 + 2
tot = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6
print(f"kernel time {tot:.3f} ms per step")
for r in rows[:24]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:72]
    print("%7.3f ms/step %6.1f x %8.1f us  %s" % (float(r["TotalDurationNs"]) / steps / 1e6, int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, n))
PY
