#!/bin/bash
# One rocprofv3 --pmc pass over a short bench.py run; prints the per-launch average of every counter for kernels whose name contains $1.
# Usage (on the MI355X box):  tools/pmc_kernel.sh fused_fwd32 SQ_WAVE_CYCLES SQ_BUSY_CYCLES ...      (BENCH_ARGS="..." for other workloads)
set -u
PAT=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$$
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT -- python $R/bench.py --steps 2 --warmup 1 --prof none --no-cpu-baseline --no-extras ${BENCH_ARGS:-} > $OUT.log 2>&1
python3 - "$OUT" "$PAT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")
if not f:
    print("no counter file; log tail:"); print(open(sys.argv[1] + ".log").read()[-1500:]); sys.exit(1)
acc, n = collections.defaultdict(float), collections.Counter()
for r in csv.DictReader(open(f[0])):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in acc:
    print(f"{k:40s} {acc[k] / n[k]:18.1f}   (avg over {n[k]} launches)")
PY
rm -rf $OUT $OUT.log
