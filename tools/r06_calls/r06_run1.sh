# round 6, GPU call 1: the GPU suite + smoke + C2/C3/C4 bench lines on the build with the replica broadcast, the check of
# Krylov fits against K, the divide & conquer's pinned uploads -- then the stage-1 panel loop as a captured hipGraph
# (BIGKRLS_S1_GRAPH: =1 timed capture / instantiate / launch, =2 cached) against the plain loop, same box, fresh processes.
O=gpurun_out/${EVID:-r06b}; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --durations=8 > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
python bench.py --config C2 --steps 10 --warmup 3 --no-cpu-baseline 2>$O/bench_C2.err | tail -1 > $O/bench_C2.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>$O/bench_C3.err | tail -1 > $O/bench_C3_10steps.json
python bench.py --config C4 --steps 4 --warmup 2 --no-cpu-baseline 2>$O/bench_C4.err | tail -1 > $O/bench_C4.json
tail -6 $O/gpu_tests.log; tail -2 $O/smoke.log
for f in $O/bench_*.json; do python -c "
import json
d=json.load(open('$f')); r=d['roofline']; print('$f', d['value'], r.get('kernel','')[:40], r.get('frac'), r.get('fit_frac'), d['phases_s'])"; done
BIGKRLS_VERBOSE=1 python tools/fit_bench.py 5000 10 > $O/verbose_C2.log 2>&1
BIGKRLS_VERBOSE=1 python tools/fit_bench.py 20000 20 > $O/verbose_C3.log 2>&1
grep -E "d&c|divide|stage" $O/verbose_C3.log | tail -22
BIGKRLS_VERBOSE=1 BIGKRLS_S1_GRAPH=1 timeout 300 python tools/fit_bench.py 5000 10 > $O/graph_verbose_C2.log 2>&1; grep -E "graph|stage 1|rep" $O/graph_verbose_C2.log | tail -8
BIGKRLS_VERBOSE=1 BIGKRLS_S1_GRAPH=1 timeout 300 python tools/fit_bench.py 20000 20 > $O/graph_verbose_C3.log 2>&1; grep -E "graph|stage 1|rep" $O/graph_verbose_C3.log | tail -8
KNOB_AB_OWN_STREAM=1 timeout 600 python tools/knob_ab.py 5000 10 BIGKRLS_S1_GRAPH=2 - > $O/graph_ab_C2.log 2>&1; cat $O/graph_ab_C2.log | grep best
KNOB_AB_OWN_STREAM=1 timeout 600 python tools/knob_ab.py 20000 20 BIGKRLS_S1_GRAPH=2 - > $O/graph_ab_C3.log 2>&1; cat $O/graph_ab_C3.log | grep best
# the trailing update at three workgroups per CU (-DSYRK64_LDS_EXACT=1) behind a gate that lets the panel factorisation's
# workgroups become resident first (-DBK_S1_GATE): head / gate at two per CU / gate at three / three without the gate
timeout 900 python tools/fit_ab.py 20000 20 bigkrls_amd/libbigkrls_hip.so tools/_ab/libbigkrls_gate2.so tools/_ab/libbigkrls_gate3.so tools/_ab/libbigkrls_exact3.so > $O/gate_ab_C3.log 2>&1; grep best $O/gate_ab_C3.log
timeout 600 python tools/fit_ab.py 5000 10 bigkrls_amd/libbigkrls_hip.so tools/_ab/libbigkrls_gate2.so tools/_ab/libbigkrls_gate3.so > $O/gate_ab_C2.log 2>&1; grep best $O/gate_ab_C2.log
# T factor from pq_chol itself + the panel chain on one stream below 6 144 rows (round 6): as built / without the stream
# change / as before (V'V chain, two stream changes)
timeout 600 python tools/knob_ab.py 5000 10 - BIGKRLS_S1_SWAP_M=0 BIGKRLS_S1_TPQ=0,BIGKRLS_S1_SWAP_M=0 > $O/tpq_ab_C2.log 2>&1; grep best $O/tpq_ab_C2.log
timeout 900 python tools/knob_ab.py 20000 20 - BIGKRLS_S1_SWAP_M=0 BIGKRLS_S1_TPQ=0,BIGKRLS_S1_SWAP_M=0 > $O/tpq_ab_C3.log 2>&1; grep best $O/tpq_ab_C3.log
# one C2 panel step kernel by kernel with the T factor from pq_chol and the chain on one stream (compare profiles/r05/r05b_C2_panel_chain_timeline.log)
rocprofv3 --kernel-trace --output-format csv -d $O/kt_C2 -o run -- python3 tools/eig_once.py 5000 10 > /dev/null 2>&1
python tools/panel_chain_timeline.py $O/kt_C2 > $O/C2_panel_chain_timeline.log 2>&1; head -30 $O/C2_panel_chain_timeline.log | cut -c1-150
f=$(find $O/kt_C2 -name "*kernel_trace.csv" | head -1); rm -rf $O/kt_C2
