# round 6, GPU call 15: block Lanczos -- the sample check against K left to the fit's check of all pairs, and the first
# Gram-Schmidt pass against the last two blocks only: tests first, then the same-box A/B at C4 and C5, bench lines
O=gpurun_out/${EVID:-r06q}; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x --durations=5 -k "lanczos or krylov or c4 or c5 or fault or recovery or watchdog" > $O/gpu_tests_kry.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests_kry.log
tail -4 $O/gpu_tests_kry.log
python tools/kry_ab.py 50000 20 512 2 > $O/kry_ab_C4.log 2>&1; cat $O/kry_ab_C4.log
python tools/kry_ab.py 100000 50 1024 1 > $O/kry_ab_C5.log 2>&1; cat $O/kry_ab_C5.log
python -m pytest tests -m gpu -q --durations=5 > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
tail -4 $O/gpu_tests.log; tail -2 $O/smoke.log
for c in C4 C5; do python bench.py --config $c --steps 4 --warmup 2 --no-cpu-baseline 2>$O/bench_$c.err | tail -1 > $O/bench_$c.json; done
for f in $O/bench_*.json; do python -c "
import json
d=json.load(open('$f')); r=d['roofline']; print('$f', d['value'], r.get('frac'), r.get('fit_frac'), d['phases_s'])"; done
