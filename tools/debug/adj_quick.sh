#!/bin/bash
# quick A/B of the adj step on the GPU box: bench lines (no profiler) at 65 536 and 384 rows; optional MATCHA_TUNE values
R=${GRAFT_REPO_ROOT:-$(pwd)}
B="--prof none --no-cpu-baseline --no-extras --front-end adj"
for t in "$@"; do
  echo "== MATCHA_TUNE=$t"
  MATCHA_TUNE=$t python $R/bench.py --steps 30 --warmup 5 $B | grep -o '"ms_per_step": [0-9.]*' | head -2 | tr '\n' ' '; echo
done
echo "== 384 rows"
python $R/bench.py --steps 200 --warmup 20 --rows 384 $B | grep -o '"ms_per_step": [0-9.]*' | head -2 | tr '\n' ' '; echo
