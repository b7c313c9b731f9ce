#!/bin/bash
# In-situ ablations of the fused forward (results are wrong on purpose; only the kernel's time is read).
#   build (here):     tools/debug/abl_fwd32.sh build "<name>:<extra hipcc flags>" ...     e.g.  base:  noattn:-DF32_ABL=1  win12:-DF32_WIN=12
#   run (GPU box):    tools/debug/abl_fwd32.sh run      -> one line per variant: fused_fwd ms / step ms at 65 536 rows
#                     (ABL_SRC=enc128 ABL_BENCH_ARGS="--layout hg38_100kb --dim 128": the embed_dim 128 kernels, -DENC_ABL=<bits>)
# Variants are whole libraries under matcha_amd/lib/abl/ (git-ignored .so files travel to the GPU box; build/ does not).
set -u
cd "$(dirname "$0")/../.."
CS=matcha_amd/csrc
OUT=build/abl
LIBOUT=matcha_amd/lib/abl
FLAGS="--offload-arch=gfx950 --offload-compress -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-function -Wno-unused-variable -Wno-unused-parameter"
case ${1:-build} in
build)
  shift
  make -C $CS -j8 >/dev/null || exit 1
  mkdir -p $OUT $LIBOUT; rm -f $LIBOUT/*.so
  for spec in "$@"; do
    name=${spec%%:*}; extra=${spec#*:}
    src=${ABL_SRC:-fused_fwd32}                       # ABL_SRC=fused_bwd builds variants of the backward kernel's file instead
    /opt/rocm/bin/hipcc $FLAGS $extra -c $CS/$src.hip -o $OUT/${src}_$name.o || exit 1
    objs=""
    for o in build/csrc/*.o; do
      case $o in *-hip-*) ;; */$src.o) objs="$objs $OUT/${src}_$name.o";; *) objs="$objs $o";; esac
    done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $LIBOUT/libmatcha_hip_$name.so $objs || exit 1
    echo "built $name ($extra)"
  done
  ;;
run)
  mkdir -p gpurun_out; LOG=gpurun_out/abl_fwd32.log; : > $LOG
  for lib in $LIBOUT/*.so; do
    name=$(basename $lib .so); name=${name#libmatcha_hip_}
    MATCHA_HIP_LIB=$(pwd)/$lib python bench.py --no-extras --no-cpu-baseline --steps ${ABL_STEPS:-20} --warmup 5 ${ABL_BENCH_ARGS:-} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['roofline_by_kernel_class']
print('$name', 'fused_fwd', round(c['fused_fwd']['ms_per_step'],4), 'fused_bwd', round(c['fused_bwd']['ms_per_step'],4), 'step', round(d['ms_per_step'],4))" | tee -a $LOG
  done
  ;;
esac
