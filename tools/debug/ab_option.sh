#!/bin/bash
# headline step with a library option set through the environment: tools/debug/ab_option.sh MATCHA_DISABLE_SIDE_STREAMS=2 [...]
cd $GRAFT_REPO_ROOT
for kv in "$@"; do
  echo "== $kv"
  env $kv python bench.py --no-extras --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/ab.json
  python -c "
import json; r=json.load(open('gpurun_out/ab.json')); print(r['value'], r['ms_per_step'], r['kernel_class_ms_per_step'])"
done
