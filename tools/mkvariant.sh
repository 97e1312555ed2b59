#!/bin/bash
# build a variant of the library with extra flags on ONE translation unit (default stencil.hip):
#   [SRC=strang_fused] tools/mkvariant.sh <name> [-D...]   -> variants/lib_<name>.so
# (git-ignored; travels with gpurun).  A/B with tools/ab_many.sh.
set -e
NAME=$1; shift
SRC=${SRC:-stencil}
ROOT=$(cd $(dirname $0)/.. && pwd)
mkdir -p $ROOT/variants
C=$ROOT/pde_opt_amd/csrc
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -fno-slp-vectorize -Wno-unused-function -I/opt/rocm/include "$@" -c $C/$SRC.hip -o $ROOT/variants/${SRC}_$NAME.o
TAG=$(cd $ROOT && python -c "from pde_opt_amd.csrc.build import _flags_tag; print(_flags_tag([]))")
OBJS=""
for o in api stencil reduce spectral halo strang_fused comm jit; do
  if [ $o = $SRC ]; then OBJS="$OBJS $ROOT/variants/${SRC}_$NAME.o"; else OBJS="$OBJS $C/build/$o.$TAG.o"; fi
done
hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc -o $ROOT/variants/lib_$NAME.so $OBJS -L/opt/rocm/lib -lrocfft -ldl -Wl,-rpath,/opt/rocm/lib
rm $ROOT/variants/${SRC}_$NAME.o
echo built variants/lib_$NAME.so
