# round 6, GPU call 17: the final library (split count of the long Lanczos products, dead tiles of the 128 x 128 Cholesky): tests, C4 / C5 lines, the A/B once more
O=gpurun_out/${EVID:-r06s}; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --durations=5 > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
tail -4 $O/gpu_tests.log; tail -2 $O/smoke.log
for c in C4 C5; do python bench.py --config $c --steps 4 --warmup 2 --no-cpu-baseline 2>$O/bench_$c.err | tail -1 > $O/bench_$c.json; done
for f in $O/bench_*.json; do python -c "
import json
d=json.load(open('$f')); r=d['roofline']; print('$f', d['value'], r.get('frac'), r.get('fit_frac'), d['phases_s'])"; done
python tools/kry_ab.py 50000 20 512 1 2>&1 | grep -v amdgpu.ids > $O/kry_ab_C4.log; cat $O/kry_ab_C4.log
