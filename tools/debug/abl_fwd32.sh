#!/bin/bash
# in-situ ablations of fused_fwd32_kernel: rebuild with -DF32_ABL=<bits> on the box and time the step (results are wrong on purpose)
for a in "$@"; do
  cd $GRAFT_REPO_ROOT/matcha_amd/csrc
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DF32_ABL=$a $EXTRA -c fused_fwd32.hip -o ../../build/csrc/fused_fwd32.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libmatcha_hip.so ../../build/csrc/*.o || exit 1
  cd $GRAFT_REPO_ROOT; echo "F32_ABL=$a $EXTRA"; tools/debug/quick_bench.sh 2>&1 | tail -1
done
