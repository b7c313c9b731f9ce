#!/bin/bash
# kernel order of the last iterations of tools/debug/copies_probe.py (PART=step|assemble)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export PART=${1:-step}
OUT=$R/gpurun_out/trace_$PART
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $R/tools/debug/copies_probe.py > $OUT.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
for r in rows[-45:]:
    print(f"{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f} us  {r['Kernel_Name'][:90]}")
PY
rm -rf $OUT $OUT.log
