# Traced build of the library (time stamps in syrk_mirror_kernel) + the probe that reads them. Development only.
#   bash tools/syrk_trace.sh build      (here; the binaries travel to the GPU box)
#   bash tools/syrk_trace.sh run [out]  (on the GPU box)
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  make -C bigkrls_amd/csrc -j4 >/dev/null
  mkdir -p tools/_trace
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DBK_SYRK_TRACE -c bigkrls_amd/csrc/gemm.hip -o tools/_trace/gemm_trace.o
  (cd bigkrls_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_trace/libbigkrls_hip.so \
     capi.o ../../tools/_trace/gemm_trace.o vecops.o solveforc.o deriv.o eigen.o neff.o fit.o dist.o -ldl)
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/syrk_trace.hip -o tools/_trace/syrk_trace \
     -Ltools/_trace -lbigkrls_hip -Wl,-rpath,'$ORIGIN'
else
  O=${2:-gpurun_out/trace}; mkdir -p $O
  for k in 128 256 512; do tools/_trace/syrk_trace 20000 $k > $O/trace_k$k.log 2>&1; head -70 $O/trace_k$k.log; done
fi
