#!/bin/bash
# headline step + per-class times, nothing else
cd $GRAFT_REPO_ROOT && python bench.py --no-extras --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/quick.json
python -c "
import json; r=json.load(open('gpurun_out/quick.json')); print(r['value'], r['ms_per_step'], r['kernel_class_ms_per_step'])"
