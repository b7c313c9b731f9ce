#!/bin/bash
# per-phase wall clock of fused_bwd8_kernel's workgroup 0 (-DFB_TIMING build on the box)
cd $GRAFT_REPO_ROOT/matcha_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DFB_TIMING $1 -c fused_bwd.hip -o ../../build/csrc/fused_bwd.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libmatcha_hip.so ../../build/csrc/*.o || exit 1
cd $GRAFT_REPO_ROOT && python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --prof none 2>&1 | grep "fused_bwd8 wg0" | tail -4
