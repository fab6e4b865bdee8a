#!/bin/bash
# usage: tools/ab_env_cfg.sh "<bench args>" "ENV1=a" "ENV1=b" ...  -- interleaved A/B of bench.py step time on ONE box (2 rounds)
args=$1; shift
for r in 1 2; do
  for cfg in "$@"; do
    ms=$(env $cfg python bench.py --steps 20 --warmup 6 --no-cpu-baseline $args 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
    echo "round $r [$args] [$cfg] ${ms:-FAILED} ms"
  done
done
