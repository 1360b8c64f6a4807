#!/bin/bash
# builds tools/_ab/libbigkrls_<name>.so: the library with eigen.hip (and gemm.hip when GEMM=1) compiled with extra -D flags
#   tools/build_variant.sh name "-DBK_RB_SLEEP=2"
set -e
cd "$(dirname "$0")/../bigkrls_amd/csrc"
name=$1; flags=$2
mkdir -p ../../tools/_ab
make -s -j8 >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $flags -c eigen.hip -o ../../tools/_ab/eigen_$name.o
objs="capi.o vecops.o solveforc.o deriv.o neff.o fit.o dist.o trace.o"
if [ "${GEMM:-0}" = "1" ]; then
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $flags -c gemm.hip -o ../../tools/_ab/gemm_$name.o
  objs="$objs ../../tools/_ab/gemm_$name.o"
else
  objs="$objs gemm.o"
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_ab/libbigkrls_$name.so $objs ../../tools/_ab/eigen_$name.o -ldl
echo built tools/_ab/libbigkrls_$name.so
