#!/bin/bash
# Run on the MI355X box (via gpurun): rocprofv3 kernel-trace stats + the two PMC passes for HBM traffic of bench.py.
# Usage: tools/collect_profiles.sh <tag> [bench args...]   -> writes gpurun_out/<tag>_{stats,fetch,write,mfma1,mfma2}/...
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python $R/bench.py --steps 20 --warmup 5 --prof none --no-cpu-baseline --no-extras "$@" > $R/gpurun_out/${TAG}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_fetch -- python $R/bench.py --steps 2 --warmup 1 --prof none --no-cpu-baseline --no-extras "$@" > $R/gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_write -- python $R/bench.py --steps 2 --warmup 1 --prof none --no-cpu-baseline --no-extras "$@" > $R/gpurun_out/${TAG}_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_mfma1 -- python $R/bench.py --steps 2 --warmup 1 --prof none --no-cpu-baseline --no-extras "$@" > $R/gpurun_out/${TAG}_mfma1.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_mfma2 -- python $R/bench.py --steps 2 --warmup 1 --prof none --no-cpu-baseline --no-extras "$@" > $R/gpurun_out/${TAG}_mfma2.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_mfma3 -- python $R/bench.py --steps 2 --warmup 1 --prof none --no-cpu-baseline --no-extras "$@" > $R/gpurun_out/${TAG}_mfma3.log 2>&1
echo "collected $TAG"
# the embedding gather alone (roofline_gather of the bench line): kernel stats + the two HBM-traffic passes
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_gather_stats -- python $R/tools/gather_bench.py > $R/gpurun_out/${TAG}_gather.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_gather_fetch -- python $R/tools/gather_bench.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_gather_write -- python $R/tools/gather_bench.py > /dev/null 2>&1
# BASELINE configs[3] (hg38 100 kb, d = 128) and configs[4] (C5: 1 M nodes, d = 256): kernel stats of the same step
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_d128_stats -- python $R/bench.py --steps 10 --warmup 3 --prof none --no-cpu-baseline --no-extras --layout hg38_100kb --dim 128 > $R/gpurun_out/${TAG}_d128.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_c5_stats -- python $R/bench.py --steps 10 --warmup 3 --prof none --no-cpu-baseline --no-extras --layout c5 --dim 256 --ks 2,3,4,5,6,7,8 --rows 16384 --edges 10000000 > $R/gpurun_out/${TAG}_c5.log 2>&1
# the gather the model step executes, on HBM-resident tables (bench.py roofline_gather_in_step): kernel stats + the HBM-read pass
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_front_stats -- python $R/tools/front_gather_bench.py > $R/gpurun_out/${TAG}_front.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_front_fetch -- python $R/tools/front_gather_bench.py > /dev/null 2>&1
echo "collected gather/d128/c5/front $TAG"
# the adj front end (the reference's own mode): kernel stats at 65 536 and 384 rows + the three PMC passes
bash $R/tools/collect_adj_profiles.sh $TAG pmc
