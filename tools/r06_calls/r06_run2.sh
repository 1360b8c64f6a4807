# round 6, GPU call 2: the build with the gate + three update workgroups per CU as the default, the T block on four chains,
# the divide & conquer's own zero fill -- tests, bench lines, the two-per-CU control, the divide & conquer kernel by
# kernel, the stream-swap threshold, the captured graph on top, and the one arm of DESIGN.md section 8 that needed this
# build (every upload of the eigensolver through pinned memory).
O=gpurun_out/${EVID:-r06c}; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --durations=8 > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
tail -6 $O/gpu_tests.log; tail -2 $O/smoke.log
for c in C2 C4 C5; do python bench.py --config $c --steps 4 --warmup 2 --no-cpu-baseline 2>$O/bench_$c.err | tail -1 > $O/bench_$c.json; done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>$O/bench_C3.err | tail -1 > $O/bench_C3_10steps.json
for f in $O/bench_*.json; do python -c "
import json
d=json.load(open('$f')); r=d['roofline']; print('$f', d['value'], r.get('kernel','')[:40], r.get('frac'), r.get('fit_frac'), [(p['kernel'][-12:], p['frac'], p['total_ms_per_fit']) for p in r.get('parts', [])])"; done
timeout 900 python tools/fit_ab.py 20000 20 bigkrls_amd/libbigkrls_hip.so tools/_ab/libbigkrls_two_per_cu.so > $O/three_vs_two_per_cu_ab_C3.log 2>&1; grep best $O/three_vs_two_per_cu_ab_C3.log
timeout 600 python tools/fit_ab.py 5000 10 bigkrls_amd/libbigkrls_hip.so tools/_ab/libbigkrls_two_per_cu.so > $O/three_vs_two_per_cu_ab_C2.log 2>&1; grep best $O/three_vs_two_per_cu_ab_C2.log
timeout 900 python tools/knob_ab.py 20000 20 - BIGKRLS_S1_SWAP_M=4096 BIGKRLS_S1_SWAP_M=8192 BIGKRLS_S1_SWAP_M=10240 > $O/swap_threshold_ab_C3.log 2>&1; grep best $O/swap_threshold_ab_C3.log
# the captured graph (default from the third decomposition of a size, n >= 8 192) against the plain loop: default-stream contexts
timeout 600 python tools/knob_ab.py 20000 20 - BIGKRLS_S1_GRAPH=0 > $O/graph_ab_C3.log 2>&1; grep best $O/graph_ab_C3.log
timeout 600 python tools/knob_ab.py 10000 10 - BIGKRLS_S1_GRAPH=0 > $O/graph_ab_N10000.log 2>&1; grep best $O/graph_ab_N10000.log
timeout 600 python tools/knob_ab.py 14000 10 - BIGKRLS_S1_GRAPH=0 > $O/graph_ab_N14000.log 2>&1; grep best $O/graph_ab_N14000.log
BIGKRLS_VERBOSE=1 python tools/eig_once.py 20000 20 > $O/eig_verbose_20000_20.log 2>&1; grep -E "d&c|divide" $O/eig_verbose_20000_20.log | tail -14 | cut -c1-220
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_eig -o run -- python3 tools/eig_once.py 20000 20 > /dev/null 2>&1
f=$(find $O/prof_eig -name "*kernel_stats.csv" | head -1); cp "$f" $O/eig_20000_20_kernel_stats.csv; rm -rf $O/prof_eig
python tools/kstats.py $O/eig_20000_20_kernel_stats.csv 2>/dev/null | head -40
# DESIGN.md section 8, the pinned-upload arm: the single-GPU control (own-stream contexts, no check against K) on this build
OVERSUB_OWN_STREAM=1 BIGKRLS_VERIFY=0 timeout 560 python tools/oversub_single.py --minutes 8 --procs 32 --reps 8 --small --no-trace > $O/oversub_single_pinned_uploads.log 2>&1
grep -v "^round .* done" $O/oversub_single_pinned_uploads.log | cut -c1-600 | tail -20
