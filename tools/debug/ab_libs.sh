#!/bin/bash
# A/B of whole libraries under matcha_amd/lib/abl/ in ONE gpurun call (boxes differ by up to 8 %: only same-call numbers compare):
# per library the in-step gather on the 4 GiB table (tools/front_gather_bench.py) and the headline step with its per-class times, REPS times.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out; LOG=gpurun_out/ab_libs.log; : > $LOG
for rep in $(seq 1 ${REPS:-2}); do
for lib in matcha_amd/lib/abl/*.so; do
  name=$(basename $lib .so); name=${name#libmatcha_hip_}
  if [ "${GATHER:-1}" = 1 ]; then
  MATCHA_HIP_LIB=$(pwd)/$lib python tools/front_gather_bench.py 2>/dev/null | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$name', d['kernel'], 'avg_launch_ms', d['avg_launch_ms'], 'frac', d['frac'])" | tee -a $LOG
  fi
  MATCHA_HIP_LIB=$(pwd)/$lib python bench.py --no-extras --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$name', 'step', d['ms_per_step'], d['kernel_class_ms_per_step'])" | tee -a $LOG
done
done
