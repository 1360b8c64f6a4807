# round 6, GPU calls 12-14: fitted values and the check against K out of the marginal-effects pass; call 14: the O(N P) host loops of the fit on up to eight threads -- tests, bench lines
O=gpurun_out/${EVID:-r06p}; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --durations=5 > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
tail -4 $O/gpu_tests.log; tail -2 $O/smoke.log
for c in C2 C4 C5; do python bench.py --config $c --steps 4 --warmup 2 --no-cpu-baseline 2>$O/bench_$c.err | tail -1 > $O/bench_$c.json; done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>$O/bench_C3.err | tail -1 > $O/bench_C3_10steps.json
for f in $O/bench_*.json; do python -c "
import json
d=json.load(open('$f')); r=d['roofline']; print('$f', d['value'], r.get('frac'), r.get('fit_frac'), d['phases_s'], [k['kernel'][:12] for k in d['other_kernels']])"; done
