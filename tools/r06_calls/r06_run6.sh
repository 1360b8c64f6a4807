# round 6, GPU calls 6 and 7: the precompute for the back-transforms on the look-ahead stream (BIGKRLS_BG=1: on a lowest-priority stream; call 6 had the default the other way round) -- tests, same-box A/B at three sizes, phase times; call 7 with bt2_build_t as one wave per task
O=gpurun_out/${EVID:-r06i}; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --durations=8 > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
tail -4 $O/gpu_tests.log; tail -2 $O/smoke.log
timeout 900 python tools/knob_ab.py 20000 20 - BIGKRLS_BG=1 > $O/bg_stream_ab_C3.log 2>&1; grep best $O/bg_stream_ab_C3.log
timeout 600 python tools/knob_ab.py 5000 10 - BIGKRLS_BG=1 > $O/bg_stream_ab_C2.log 2>&1; grep best $O/bg_stream_ab_C2.log
timeout 600 python tools/knob_ab.py 10000 10 - BIGKRLS_BG=1 > $O/bg_stream_ab_N10000.log 2>&1; grep best $O/bg_stream_ab_N10000.log
BIGKRLS_VERBOSE=1 python tools/eig_once.py 20000 20 > $O/eig_verbose_20000_20.log 2>&1; grep -E "divide|stage|back|gather" $O/eig_verbose_20000_20.log | tail -7 | cut -c1-220
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>$O/bench_C3.err | tail -1 > $O/bench_C3_10steps.json
python bench.py --config C2 --steps 10 --warmup 3 --no-cpu-baseline 2>$O/bench_C2.err | tail -1 > $O/bench_C2.json
for f in $O/bench_*.json; do python -c "
import json
d=json.load(open('$f')); r=d['roofline']; print('$f', d['value'], r.get('frac'), r.get('fit_frac'), d['phases_s']['eigen'])"; done
