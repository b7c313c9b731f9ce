#!/bin/bash
# Run on the MI355X box (via gpurun): rocprofv3 kernel-trace stats (+ optional PMC passes) of the adj front end's step
# (the reference's own mode, Modules.py:176-201) at 65 536 rows and at the reference's 384-row batch.
# Usage: tools/collect_adj_profiles.sh <tag> [pmc]   -> gpurun_out/<tag>_adj{,384}_stats/..., <tag>_adj_{fetch,write,mfma1}/...
set -u
TAG=$1; PMC=${2:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
B="--prof none --no-cpu-baseline --no-extras --front-end adj"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_adj_stats -- python $R/bench.py --steps 20 --warmup 5 $B > $R/gpurun_out/${TAG}_adj_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_adj384_stats -- python $R/bench.py --steps 20 --warmup 5 --rows 384 $B > $R/gpurun_out/${TAG}_adj384_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_table384_stats -- python $R/bench.py --steps 20 --warmup 5 --rows 384 --prof none --no-cpu-baseline --no-extras > $R/gpurun_out/${TAG}_table384_stats.log 2>&1
if [ -n "$PMC" ]; then
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_adj_fetch -- python $R/bench.py --steps 2 --warmup 1 $B > $R/gpurun_out/${TAG}_adj_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_adj_write -- python $R/bench.py --steps 2 --warmup 1 $B > $R/gpurun_out/${TAG}_adj_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_adj_mfma1 -- python $R/bench.py --steps 2 --warmup 1 $B > $R/gpurun_out/${TAG}_adj_mfma1.log 2>&1
fi
echo "collected adj $TAG"
