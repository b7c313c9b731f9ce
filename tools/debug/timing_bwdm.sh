#!/bin/bash
# per-phase wall clock of two wavefronts of workgroup 0 of fused_bwdm_kernel (-DFB_TIMING build on the box); $1 = extra compiler flags
cd $GRAFT_REPO_ROOT/matcha_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DFB_TIMING $1 -c fused_bwd.hip -o ../../build/csrc/fused_bwd.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libmatcha_hip.so ../../build/csrc/*.o || exit 1
cd $GRAFT_REPO_ROOT && python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --prof none 2>&1 | grep "fused_bwd. " | grep -v metric | tail -60
