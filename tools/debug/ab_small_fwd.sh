#!/bin/bash
cd $GRAFT_REPO_ROOT
for kv in MATCHA_DISABLE_FWD32=0 MATCHA_DISABLE_FWD32=1 MATCHA_DISABLE_MERGED=1; do
  env $kv python bench.py --no-extras --no-cpu-baseline --prof none --rows 384 --steps 50 --warmup 10 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$kv', r['value'], r['ms_per_step'])"
done
