#!/bin/bash
# timing ablations of fused_bwd8_kernel: rebuild fused_bwd.o with -DFB8_ABL=<mask> on the box and time the bench's backward kernel
cd $GRAFT_REPO_ROOT/matcha_amd/csrc
for a in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DFB8_ABL=$a -c fused_bwd.hip -o ../../build/csrc/fused_bwd.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libmatcha_hip.so ../../build/csrc/*.o || exit 1
  echo "ABL $a: $(cd $GRAFT_REPO_ROOT && python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["kernel_class_ms_per_step"]["fused_bwd"])')"
done
