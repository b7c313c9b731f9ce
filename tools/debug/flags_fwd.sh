#!/bin/bash
# A/B of compile-time variants of fused_fwd_kernel: each argument is a set of -D flags; rebuilds on the box and times the bench
cd $GRAFT_REPO_ROOT/matcha_amd/csrc
for f in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $f -c fused_fwd.hip -o ../../build/csrc/fused_fwd.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libmatcha_hip.so ../../build/csrc/*.o || exit 1
  echo "[$f]: $(cd $GRAFT_REPO_ROOT && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["kernel_class_ms_per_step"]["fused_fwd"])')"
done
