#!/bin/bash
# build a variant of the library with extra flags on stencil.hip: tools/mkvariant.sh <name> [-D...]
# -> variants/lib_<name>.so (git-ignored; travels with gpurun).  A/B with tools/ab.sh.
set -e
NAME=$1; shift
ROOT=$(cd $(dirname $0)/.. && pwd)
mkdir -p $ROOT/variants
C=$ROOT/pde_opt_amd/csrc
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function -I/opt/rocm/include "$@" -c $C/stencil.hip -o $ROOT/variants/stencil_$NAME.o
hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc -o $ROOT/variants/lib_$NAME.so $ROOT/variants/stencil_$NAME.o \
  $C/build/api.o $C/build/reduce.o $C/build/spectral.o $C/build/halo.o $C/build/strang_fused.o -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
rm $ROOT/variants/stencil_$NAME.o
echo built variants/lib_$NAME.so
