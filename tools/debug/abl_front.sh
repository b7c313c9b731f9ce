#!/bin/bash
# In-situ ablations / variants of front_fwd2_kernel: the 4 GiB-table gather (tools/front_gather_bench.py) and, with STEP=1, the training step
#   build (here):   ABL_SRC=front_fused tools/debug/abl_fwd32.sh build "base:" "nodec:-DFF2_ABL=1" ...
#   run (GPU box):  [STEP=1] tools/debug/abl_front.sh
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out; LOG=gpurun_out/abl_front.log; : > $LOG
for lib in matcha_amd/lib/abl/*.so; do
  name=$(basename $lib .so); name=${name#libmatcha_hip_}
  MATCHA_HIP_LIB=$(pwd)/$lib python tools/front_gather_bench.py 2>/dev/null | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$name', d['kernel'], 'avg_launch_ms', d['avg_launch_ms'], 'frac', d['frac'])" | tee -a $LOG
  if [ "${STEP:-0}" = 1 ]; then
    MATCHA_HIP_LIB=$(pwd)/$lib python bench.py --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$name', 'step', d['ms_per_step'], d['kernel_class_ms_per_step'])" | tee -a $LOG
  fi
done
