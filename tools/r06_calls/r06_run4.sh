# round 6, GPU calls 4 and 5: the look-ahead stream builds the stage-2 back-transform's T factors first and the merged stage-1
# blocks behind them (the stage-2 back-transform no longer waits for the latter) -- tests, bench lines, phase times
O=gpurun_out/${EVID:-r06f}; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --durations=8 > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
tail -4 $O/gpu_tests.log; tail -2 $O/smoke.log
python bench.py --config C2 --steps 10 --warmup 3 --no-cpu-baseline 2>$O/bench_C2.err | tail -1 > $O/bench_C2.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>$O/bench_C3.err | tail -1 > $O/bench_C3_10steps.json
python bench.py --config C4 --steps 4 --warmup 2 --no-cpu-baseline 2>$O/bench_C4.err | tail -1 > $O/bench_C4.json
for f in $O/bench_*.json; do python -c "
import json
d=json.load(open('$f')); r=d['roofline']; print('$f', d['value'], r.get('kernel','')[:40], r.get('frac'), r.get('fit_frac'), d['phases_s']['eigen'])"; done
BIGKRLS_VERBOSE=1 python tools/eig_once.py 20000 20 > $O/eig_verbose_20000_20.log 2>&1; grep -E "divide|stage|back|gather" $O/eig_verbose_20000_20.log | tail -7 | cut -c1-220
BIGKRLS_VERBOSE=1 python tools/eig_once.py 5000 10 > $O/eig_verbose_5000_10.log 2>&1; grep -E "divide|stage|back|gather" $O/eig_verbose_5000_10.log | tail -7 | cut -c1-220
BIGKRLS_VERBOSE=1 python tools/eig_once.py 10000 10 > $O/eig_verbose_10000_10.log 2>&1; grep -E "divide|stage|back|gather" $O/eig_verbose_10000_10.log | tail -7 | cut -c1-220
