#!/bin/bash
# kernel stats of the 384-row step (table and adj) on the GPU box
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for fe in table adj; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${1}_${fe}384_stats -- python $R/bench.py --steps 20 --warmup 5 --rows 384 --prof none --no-cpu-baseline --no-extras --windows 1 --front-end $fe > $R/gpurun_out/${1}_${fe}384.log 2>&1
done
