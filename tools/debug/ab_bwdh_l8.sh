#!/bin/bash
cd $GRAFT_REPO_ROOT
for kv in MATCHA_DISABLE_BWDH=0 MATCHA_DISABLE_BWDH=1; do
  env $kv python bench.py --no-extras --no-cpu-baseline --ks 2,3,4,5,6,7,8 --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$kv', r['value'], r['ms_per_step'], r['kernel_class_ms_per_step'])"
done
