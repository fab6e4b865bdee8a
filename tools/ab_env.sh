#!/bin/bash
# usage: tools/ab_env.sh "ENV1=a ENV2=b" "ENV1=c" ...   -- interleaved A/B of bench.py step time on ONE box (2 rounds)
for r in 1 2; do
  for cfg in "$@"; do
    ms=$(env $cfg python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "round $r [$cfg] $ms ms"
  done
done
