#!/bin/bash
# configs[3] (hg38 100 kb, d = 128) and configs[4] (C5, d = 256): step time + per-class times
cd $GRAFT_REPO_ROOT
for args in "--layout hg38_100kb --dim 128" "--layout c5 --dim 256 --ks 2,3,4,5,6,7,8 --rows 16384 --edges 10000000"; do
  python bench.py --no-extras --no-cpu-baseline --steps 8 --warmup 3 $args 2>&1 | tail -1 > gpurun_out/quick.json
  python -c "
import json; r=json.load(open('gpurun_out/quick.json')); print(r['value'], r['ms_per_step'], r['kernel_class_ms_per_step'])"
done
