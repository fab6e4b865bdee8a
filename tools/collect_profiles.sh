#!/bin/bash
# Run ON THE GPU BOX from the repo root (through gpurun): tools/collect_profiles.sh <tag, e.g. r02>
# Writes into gpurun_out/profiles_<tag>/ : kernel stats of the default bench command, HBM traffic per kernel from two
# separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of that same command folded by tools/pmc_traffic.py, and the pipe
# counters (MFMA busy / VALU busy / waits) of the final conv kernels per shape.  Copy what should be judged to profiles/.
set -e
tag=$1
root=$PWD
out=$root/gpurun_out/profiles_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
STEPS=6; WARM=3
echo "== kernel trace"; rocprofv3 --kernel-trace --stats -d "$out/trace" --output-format csv -- python3 "$root/bench.py" --steps $STEPS --warmup $WARM --no-cpu-baseline --detail-out "$out/${tag}_per_shape.json" > "$out/bench_trace.json" 2> "$out/bench_trace.err"
echo "== pmc fetch";   rocprofv3 --pmc FETCH_SIZE -d "$out/pmc_fetch" --output-format csv -- python3 "$root/bench.py" --steps $STEPS --warmup $WARM --no-cpu-baseline > /dev/null 2> "$out/pmc_fetch.err"
echo "== pmc write";   rocprofv3 --pmc WRITE_SIZE -d "$out/pmc_write" --output-format csv -- python3 "$root/bench.py" --steps $STEPS --warmup $WARM --no-cpu-baseline > /dev/null 2> "$out/pmc_write.err"
# steps in a run: warm-up + timed + 4 host-enqueue probes + 2 instrumented
python3 "$root/tools/pmc_traffic.py" "$out/pmc_fetch" "$out/pmc_write" "$out/${tag}_pmc_traffic.json" $((STEPS + WARM + 6)) "command: bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline (config A, batch 32, 256x256); steps profiled = warm-up + timed + 4 host-enqueue probes + 2 instrumented" "vae_dente_no_adv.json:b32:256"
cp "$(find "$out/trace" -name '*kernel_stats.csv' | head -1)" "$out/${tag}_bench_b32_kernel_stats.csv"
echo "== pipe counters per conv shape"
CNT="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
for spec in "32 32 256 256 0 full" "32 32 256 256 0 dgrad" "64 64 128 128 0 full" "64 64 128 128 0 dgrad" "128 128 64 64 0 full" "128 128 64 64 0 dgrad" "128 128 32 32 0 full" "128 128 32 32 0 dgrad" "128 128 128 128 0 plain" "128 128 128 128 0 wgrad" "128 128 64 64 0 wgrad" "64 64 128 128 0 wgrad" "32 32 256 256 0 wgrad"; do
  name=$(echo $spec | tr ' ' '_')
  rocprofv3 --pmc $CNT --kernel-trace -d "$out/pipes_$name" --output-format csv -- python3 "$root/tools/pmc_conv.py" $spec > /dev/null 2> "$out/pipes_$name.err" || echo "pipes $spec failed"
done
python3 "$root/tools/pipes_summary.py" "$out" > "$out/${tag}_pmc_conv_pipes.txt"
cat "$out/${tag}_pmc_conv_pipes.txt"
# keep the merged-back directory small: drop the raw per-dispatch tables
find "$out" -name '*counter_collection.csv' -delete; find "$out" -name '*kernel_trace.csv' -delete
