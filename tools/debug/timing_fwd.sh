#!/bin/bash
# per-phase wall clock of one workgroup of fused_fwd_kernel (-DFF_TIMING build on the box)
cd $GRAFT_REPO_ROOT/matcha_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DFF_TIMING $1 -c fused_fwd.hip -o ../../build/csrc/fused_fwd.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libmatcha_hip.so ../../build/csrc/*.o || exit 1
cd $GRAFT_REPO_ROOT && python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --prof none 2>&1 | grep "fused_fwd wg\|tail fwd" | tail -4
