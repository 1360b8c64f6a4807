set -x
O=gpurun_out/r03l; mkdir -p $O
bash tools/syrk_trace.sh run $O > $O/trace_all.log 2>&1
for k in 128 256 512; do head -12 $O/trace_k$k.log; done
timeout 300 tools/syrk_k_probe 20000 q 2>&1 | grep -v check | head -24
