#!/bin/bash
# Run ON THE GPU BOX: per-kernel time of the perceptual (LPIPS) term alone, 10 forward+backward passes at batch 32 x 256^2.
# The first (unprofiled) run fills MIOpen's find cache so that the profiled run holds steady-state kernels only.
root=$PWD
out=$root/gpurun_out/perc_kernels
rm -rf "$out"; mkdir -p "$out"
cat > /tmp/perc_loop.py <<PY
import sys, torch
sys.path.insert(0, "$root")
from pti_ldm_vae_amd.models import PerceptualLoss
dev = torch.device("cuda:0")
pl = PerceptualLoss(allow_random_init=True).to(dev)
x = torch.randn(32, 1, 256, 256, device=dev, requires_grad=True)
y = torch.randn(32, 1, 256, 256, device=dev)
for _ in range(10):
    torch.autograd.grad(pl(x, y), x)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
python3 /tmp/perc_loop.py
rocprofv3 --kernel-trace --stats -d "$out/trace" --output-format csv -- python3 /tmp/perc_loop.py > /dev/null 2> "$out/err"
f=$(find "$out/trace" -name '*kernel_stats.csv' | head -1)
cp "$f" "$out/kernel_stats.csv"
find "$out/trace" -name '*kernel_trace.csv' -delete
python3 - "$out/kernel_stats.csv" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f"total {tot/1e7:.3f} ms per pass")
for r in rows[:32]:
    n=r['Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0]
    print(f"{n[:72]:72s} calls/pass={int(r['Calls'])/10:6.1f} avg={float(r['AverageNs'])/1e3:8.1f}us ms/pass={float(r['TotalDurationNs'])/1e7:6.3f}")
PY
