#!/bin/bash
# Development helper: builds tools/gemm_bench_<name> for several compile-time GEMM variants
# (each binary links its own copy of the library objects).
set -e
cd "$(dirname "$0")/.."
build() {
  name=$1; shift
  d=/tmp/bkvar_$name; mkdir -p $d
  for f in capi gemm vecops solveforc deriv eigen; do
    if [ $f = gemm ] || [ ! -f $d/$f.o ]; then
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude "$@" -c bigkrls_amd/csrc/$f.hip -o $d/$f.o &
    fi
  done
  wait
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -w -c tools/gemm_bench.hip -o $d/bench_main.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $d/*.o -o tools/gemm_bench_$name
}
build occ1 -DGEMM_OCC=1
build occ2 -DGEMM_OCC=2
