#!/bin/bash
# Run ON THE GPU BOX: tools/stream_timeline.sh [bench args] -- where the two streams of the training step are busy / idle.
# Kernel trace (rocprofv3 --kernel-trace) of a short bench run, default streams (weight gradients on the side stream);
# prints, for the LAST timed step, a 0.25-ms-bucket timeline of busy time per stream (queue) and which kernels run on each.
root=$PWD
out=$root/gpurun_out/timeline
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$out/t" --output-format csv -- python3 "$root/bench.py" --steps 6 --warmup 3 --no-cpu-baseline "$@" > "$out/bench.json" 2> "$out/err"
f=$(find "$out/t" -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
def nm(r): return r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", r.get("Stream_Id", "?")), nm(r)) for r in rows]
ev.sort()
# steps are delimited by adam_kernel launches; take the span between the 7th and 8th (inside the timed region)
adams = [e for e in ev if e[3] == "adam_kernel"]
k = min(7, len(adams) - 2)
t0, t1 = adams[k][1], adams[k + 1][1]
step = [e for e in ev if t0 <= e[0] < t1]
print(f"step span {(t1 - t0) / 1e6:.3f} ms, {len(step)} launches, queues: {collections.Counter(e[2] for e in step)}")
B = 250_000
nb = (t1 - t0) // B + 1
qs = sorted(set(e[2] for e in step), key=lambda q: -sum(e[1] - e[0] for e in step if e[2] == q))
for q in qs:
    busy = [0] * nb
    for s, e, qq, n in step:
        if qq != q: continue
        a = s
        while a < e:
            b = min(e, ((a - t0) // B + 1) * B + t0)
            busy[(a - t0) // B] += b - a
            a = b
    tot = sum(e[1] - e[0] for e in step if e[2] == q)
    print(f"queue {q}: busy {tot / 1e6:.3f} ms | per 0.25 ms: " + " ".join(f"{100 * x // B:3d}" for x in busy))
    top = collections.Counter()
    for s, e, qq, n in step:
        if qq == q: top[n] += e - s
    print("     " + ", ".join(f"{n} {v / 1e6:.2f}" for n, v in top.most_common(6)))
# first / last weight-gradient launch relative to the step, and when the last main-stream kernel ends
wg = [e for e in step if "wgrad" in e[3]]
main_q = qs[0]
last_main = max(e[1] for e in step if e[2] == main_q and e[3] != "adam_kernel")
print(f"first wgrad launch at +{(wg[0][0] - t0) / 1e6:.3f} ms, last wgrad ends +{(max(e[1] for e in wg) - t0) / 1e6:.3f} ms; "
      f"last non-Adam kernel of the main queue ends +{(last_main - t0) / 1e6:.3f} ms")
for s, e, qq, n in wg:
    print(f"   wgrad {n:24s} q{qq} +{(s - t0) / 1e6:7.3f} .. +{(e - t0) / 1e6:7.3f} ms ({(e - s) / 1e3:7.1f} us)")
# idle time of the main queue: gaps between consecutive kernels, by (previous -> next) pair, and the whole sequence
mq = [e for e in step if e[2] == main_q]
gaps = collections.Counter(); cnt = collections.Counter(); idle = 0
for a, b in zip(mq, mq[1:]):
    g = b[0] - a[1]
    if g > 0:
        idle += g; gaps[(a[3], b[3])] += g; cnt[(a[3], b[3])] += 1
print(f"main queue: {len(mq)} launches, idle between kernels {idle / 1e6:.3f} ms")
for (x, y), v in gaps.most_common(14):
    print(f"   gap {x:26s} -> {y:26s} n={cnt[(x, y)]:3d} total {v / 1e3:7.1f} us  avg {v / cnt[(x, y)] / 1e3:5.1f} us")
print("main queue sequence (start ms, duration us, gap before us):")
prev = None
for s, e, qq, n in mq:
    print(f"   +{(s - t0) / 1e6:7.3f} {(e - s) / 1e3:7.1f} {((s - prev) / 1e3 if prev else 0):6.1f}  {n}")
    prev = e
PY
find "$out/t" -name '*.csv' -delete
