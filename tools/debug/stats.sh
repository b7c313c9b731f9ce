#!/bin/bash
# rocprofv3 kernel stats of a short bench run; prints the top kernels (avg us per call)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/stats_$$
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $R/bench.py --steps 5 --warmup 2 --prof none --no-cpu-baseline --no-extras ${BENCH_ARGS:-} > $OUT.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(__import__("os").environ.get("TOPN", "28"))]:
    print(f"{r['Name'][:70]:70s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:9.1f} us  {100*float(r['TotalDurationNs'])/tot:5.1f} %")
PY
rm -rf $OUT $OUT.log
