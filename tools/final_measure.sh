#!/bin/bash
# Run ON THE GPU BOX from the repo root: the round's closing measurements (GPU test suite, every bench configuration),
# results under gpurun_out/r3_final_*.json, one summary line each.
python -m pytest tests -m gpu -x -q 2>&1 | tail -1
A=config/ar_vae_dente_kl1e3.json; R=config/reg_edente_from_dente.json
python bench.py --steps 30 --warmup 8 --detail-out gpurun_out/r3_final_A_per_shape.json > gpurun_out/r3_final_A.json 2>/dev/null
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --channels 3 > gpurun_out/r3_final_A3.json 2>/dev/null
for b in 8 32; do
  python bench.py --steps 15 --warmup 4 --no-cpu-baseline --config $A --batch $b > gpurun_out/r3_final_AR$b.json 2>/dev/null
  python bench.py --steps 30 --warmup 8 --no-cpu-baseline --config $R --batch $b > gpurun_out/r3_final_reg$b.json 2>/dev/null
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --adv > gpurun_out/r3_final_A_adv.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --perceptual > gpurun_out/r3_final_A_perc.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --adv --perceptual > gpurun_out/r3_final_A_full.json 2>/dev/null
python bench.py --steps 15 --warmup 4 --no-cpu-baseline --adv --config $A --batch 8 > gpurun_out/r3_final_AR8_adv.json 2>/dev/null
python bench.py --steps 15 --warmup 4 --no-cpu-baseline --adv --perceptual --config $A --batch 8 > gpurun_out/r3_final_AR8_full.json 2>/dev/null
for f in A A3 AR8 AR32 reg8 reg32 A_adv A_perc A_full AR8_adv AR8_full; do
  python -c "
import json
d=json.loads(open('gpurun_out/r3_final_$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'], d.get('model_tflops_per_gpu'))"
done
