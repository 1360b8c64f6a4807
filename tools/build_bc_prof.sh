#!/bin/bash
# Profiling build of the library with per-phase shader-clock accounting inside bc_resident (-DBK_BC_PROF):
# tools/libbigkrls_bcprof.so, read by tools/bc_prof.py. Not shipped, not used by tests or bench.py.
set -e
cd "$(dirname "$0")/../bigkrls_amd/csrc"
mkdir -p /tmp/bcprof_obj
for f in capi gemm vecops solveforc deriv eigen neff fit dist; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DBK_BC_PROF -c $f.hip -o /tmp/bcprof_obj/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libbigkrls_bcprof.so /tmp/bcprof_obj/*.o -ldl
echo built tools/libbigkrls_bcprof.so
