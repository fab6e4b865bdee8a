#!/usr/bin/env python3
"""Fold the per-shape `rocprofv3 --pmc ... --kernel-trace` runs of tools/collect_profiles.sh into one text table:
per conv launch (average over the 5 launches of tools/pmc_conv.py): duration, MFMA-busy and VALU-busy fractions of the
launch per SIMD, share of wave time parked on waits, VALU instructions per wave."""
import collections
import csv
import glob
import os
import sys

out = sys.argv[1]
NSIMD = 256 * 4
print("rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU "
      "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -- python3 tools/pmc_conv.py <cin cout h w mode variant>")
print("batch 32, fp16 storage + fp16 forward operands (the training step's formats); averages over 5 launches.")
print("mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x clock); valu_busy = 4 x SQ_ACTIVE_INST_VALU (quad-cycles) / same;")
print("(quad-cycles x 4 = cycles; both are sums over the 1024 SIMDs)\n")
print(f"{'shape / variant':34s} {'kernel':46s} {'us':>7s} {'MFMA busy':>9s} {'VALU busy':>9s} {'wait':>6s} {'VALU/wave':>9s} {'TFLOP/s':>8s}")
for d in sorted(glob.glob(os.path.join(out, "pipes_*"))):
    if not os.path.isdir(d):
        continue
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not cc or not kt:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for row in csv.DictReader(open(cc[0])):
        k = row["Kernel_Name"]
        if "conv_mfma" not in k and "wgrad_mfma" not in k:
            continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "SQ_WAVE_CYCLES":
            cnt[k] += 1
    dur = collections.defaultdict(list)
    grid = {}
    for row in csv.DictReader(open(kt[0])):
        k = row["Kernel_Name"]
        if "conv_mfma" in k or "wgrad_mfma" in k:
            dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
            grid[k] = int(row["Grid_Size_X"]) * int(row["Grid_Size_Y"]) // 64
    spec = os.path.basename(d)[6:].split("_")
    cin, cout, h, w = (int(v) for v in spec[:4])
    for k, c in acc.items():
        n = max(cnt[k], 1)
        us = sorted(dur[k])[len(dur[k]) // 2]
        waves = grid[k]
        wave_cyc = c["SQ_WAVE_CYCLES"] / n * 4           # cycles summed over waves
        # duration in cycles per SIMD from the busy counter when present, else from wave cycles / resident waves
        busy = c["SQ_BUSY_CYCLES"] / n
        mfma = c["SQ_VALU_MFMA_BUSY_CYCLES"] / n         # cycles summed over SIMDs
        valu = c["SQ_ACTIVE_INST_VALU"] / n * 4
        # clock: MFMA cycles are exact (32 per 32x32x16 MFMA), so cycles/us = mfma cycles per SIMD / us x (1 / mfma_busy);
        # report fractions against duration x 1024 SIMDs x f, f from SQ_BUSY_CYCLES (per-SE counter) is unreliable -> use 2.0-2.4 GHz band
        lo, hi = (mfma / NSIMD) / (us * 2400), (mfma / NSIMD) / (us * 1900)
        vlo, vhi = (valu / NSIMD) / (us * 2400), (valu / NSIMD) / (us * 1900)
        name = k.replace("(anonymous namespace)::", "").replace("void ", "")
        name = name[:name.index("(")] if "(" in name else name
        flops = 2.0 * 32 * h * w * cin * cout * 9
        print(f"{' '.join(spec):34s} {name:46s} {us:7.1f} {100 * lo:4.0f}-{100 * hi:<3.0f}% {100 * vlo:4.0f}-{100 * vhi:<3.0f}% "
              f"{100 * c['SQ_WAIT_ANY'] / max(c['SQ_WAVE_CYCLES'], 1):5.0f}% {c['SQ_INSTS_VALU'] / n / max(waves, 1):9.0f} {flops / us / 1e6:8.0f}")
print("\nMFMA / VALU busy are given as a band: cycles per SIMD divided by the launch duration at 2.4 GHz (left) and 1.9 GHz (right) --")
print("the chip lowers its clock under MFMA load (MI355X_MICROARCH.md, DVFS give-back) and rocprofv3 gives no per-kernel clock.")
