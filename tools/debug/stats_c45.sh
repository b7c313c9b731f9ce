#!/bin/bash
# rocprofv3 kernel stats of the d = 128 (configs[3]) and C5 (configs[4]) steps; prints the top kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}
for which in d128 c5; do
  if [ $which = d128 ]; then export BENCH_ARGS="--layout hg38_100kb --dim 128"; else export BENCH_ARGS="--layout c5 --dim 256 --ks 2,3,4,5,6,7,8 --rows 16384 --edges 10000000"; fi
  echo "== $which"
  TOPN=${TOPN:-22} $R/tools/debug/stats.sh
done
