# round 5, call 15: the check through two combinations of ALL kept pairs + corruption errors softened into one redo
export TMPDIR=/tmp
O=gpurun_out/r05o; mkdir -p $O
timeout 900 python tests/_fault_inject.py > $O/fault_inject.log 2>&1; tail -2 $O/fault_inject.log
timeout 300 python tests/_dist_world_gpu.py 900 4 2 --eigtrunc 0.001 --garbage-rank 1 > $O/world_garbage.log 2>&1; grep -E "redoing|^rank" $O/world_garbage.log | cut -c1-260
timeout 300 python tools/knob_ab.py 20000 20 - "BIGKRLS_VERIFY=0" > $O/knob_verify_c3.log 2>&1; grep best $O/knob_verify_c3.log
timeout 200 python tools/knob_ab.py 5000 10 - "BIGKRLS_VERIFY=0" > $O/knob_verify_c2.log 2>&1; grep best $O/knob_verify_c2.log
rm -rf gpurun_out/trace_stress
timeout 900 python tools/world_trace_stress.py --no-trace --minutes ${M1:-11} > $O/stress_callbacks3.log 2>&1; grep -v "^round .* done" $O/stress_callbacks3.log | cut -c1-330 | tail -30; grep "^round .* done" $O/stress_callbacks3.log | tail -1
