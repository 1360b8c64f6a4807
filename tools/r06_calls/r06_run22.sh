# round 6, GPU call 25: the block Lanczos on numerically low-rank kernels after the fix (early check on an invariant Krylov space, ill-conditioned blocks re-orthogonalised, honest breakdown), then everything else again
O=gpurun_out/${EVID:-r06x}; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_level1.py -m gpu -q -x -k "low_rank" > $O/gpu_tests_lowrank.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests_lowrank.log
grep -v amdgpu.ids $O/gpu_tests_lowrank.log | tail -30
( python tools/lowrank_check.py 20000 2 512; python tools/lowrank_check.py 24000 1 256; python tools/lowrank_check.py 50000 2 512 ) > $O/lowrank_check.log 2>&1
grep -v amdgpu.ids $O/lowrank_check.log | tail -60
python tools/kry_sweep.py 2>&1 | grep -v amdgpu.ids | grep "^n=\|^====\|estimate\|re-orth" > $O/kry_sweep.log; cat $O/kry_sweep.log
python -m pytest tests -m gpu -q --durations=5 > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
tail -4 $O/gpu_tests.log; tail -2 $O/smoke.log
for c in C4 C5; do python bench.py --config $c --steps 4 --warmup 2 --no-cpu-baseline 2>$O/bench_$c.err | tail -1 > $O/bench_$c.json; done
for f in $O/bench_*.json; do python -c "
import json
d=json.load(open('$f')); r=d['roofline']; print('$f', d['value'], r.get('frac'), r.get('fit_frac'))"; done
