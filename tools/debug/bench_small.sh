#!/bin/bash
# the reference's own batch (384 rows) and the headline step, eager
cd $GRAFT_REPO_ROOT
for rows in 384 65536; do
  python bench.py --no-extras --no-cpu-baseline --prof none --rows $rows --steps 50 --warmup 10 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print($rows, r['value'], r['ms_per_step'])"
done
