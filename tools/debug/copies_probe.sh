#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for part in assemble step; do
  export PART=$part
  OUT=$R/gpurun_out/copies_$part
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $R/tools/debug/copies_probe.py > $OUT.log 2>&1
  echo "== $part"
  python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'][:80]:80s} calls {int(r['Calls']):4d} avg {float(r['AverageNs'])/1e3:7.1f} us")
PY
  rm -rf $OUT $OUT.log
done
