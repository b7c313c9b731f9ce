#!/bin/bash
# In-situ ablations of front_fwd2_kernel on the 4 GiB table (tools/front_gather_bench.py): one line per variant library under matcha_amd/lib/abl/
#   build (here):   ABL_SRC=front_fused tools/debug/abl_fwd32.sh build "base:" "nodec:-DFF2_ABL=1" ...
#   run (GPU box):  tools/debug/abl_front.sh
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out; LOG=gpurun_out/abl_front.log; : > $LOG
for lib in matcha_amd/lib/abl/*.so; do
  name=$(basename $lib .so); name=${name#libmatcha_hip_}
  MATCHA_HIP_LIB=$(pwd)/$lib python tools/front_gather_bench.py 2>/dev/null | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$name', d['kernel'], 'avg_launch_ms', d['avg_launch_ms'], 'frac', d['frac'])" | tee -a $LOG
done
