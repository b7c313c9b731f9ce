#!/bin/bash
# Run on the MI355X box (via gpurun): rocprofv3 kernel-trace stats + the two PMC passes for HBM traffic of bench.py.
# Usage: tools/collect_profiles.sh <tag> [bench args...]   -> writes gpurun_out/<tag>_{stats,fetch,write,mfma1,mfma2}/...
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python $R/bench.py --steps 5 --warmup 2 --prof none --no-cpu-baseline "$@" > $R/gpurun_out/${TAG}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_fetch -- python $R/bench.py --steps 2 --warmup 1 --prof none --no-cpu-baseline "$@" > $R/gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_write -- python $R/bench.py --steps 2 --warmup 1 --prof none --no-cpu-baseline "$@" > $R/gpurun_out/${TAG}_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_mfma1 -- python $R/bench.py --steps 2 --warmup 1 --prof none --no-cpu-baseline "$@" > $R/gpurun_out/${TAG}_mfma1.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_mfma2 -- python $R/bench.py --steps 2 --warmup 1 --prof none --no-cpu-baseline "$@" > $R/gpurun_out/${TAG}_mfma2.log 2>&1
echo "collected $TAG"
