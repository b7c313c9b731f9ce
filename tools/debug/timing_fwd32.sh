#!/bin/bash
# per-phase wall clock of one wavefront of fused_fwd32_kernel (-DFF_TIMING build on the box); $1 = extra compiler flags
cd $GRAFT_REPO_ROOT/matcha_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DFF_TIMING $1 -c fused_fwd32.hip -o ../../build/csrc/fused_fwd32.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libmatcha_hip.so ../../build/csrc/*.o || exit 1
cd $GRAFT_REPO_ROOT && python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --prof none 2>&1 | grep "fused_fwd32 wave" | tail -2
