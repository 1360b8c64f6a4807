# pq_chol phase by phase: one build of the library per phase (-DPQC_STOP=n compiles the kernel up to phase n only) with
# the bench entry point, timed by tools/pqc_bench.py. Development tool.
#   bash tools/pqc_bench.sh build   (here)      bash tools/pqc_bench.sh run [m ...]   (on the GPU box)
cd "$(dirname "$0")/.."
PH="1 2 3 4 5 6 7 8 9 10 11 12 13 0"
if [ "$1" = build ]; then
  mkdir -p tools/_ab/pqc
  for n in $PH; do
    ( cd bigkrls_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DBK_PQC_PROF -DPQC_STOP=$n -c eigen.hip -o ../../tools/_ab/pqc/eigen_$n.o 2>/dev/null &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_ab/pqc/libbigkrls_stop$n.so capi.o gemm.o vecops.o solveforc.o deriv.o ../../tools/_ab/pqc/eigen_$n.o neff.o fit.o dist.o -ldl && rm ../../tools/_ab/pqc/eigen_$n.o ) &
    if [ $(jobs -r | wc -l) -ge 4 ]; then wait -n; fi
  done
  wait
  ls tools/_ab/pqc
else
  shift
  python tools/pqc_bench.py "$@"
fi
