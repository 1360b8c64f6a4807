O=gpurun_out/r04g; mkdir -p $O
for c in C4 C5; do python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline 2>$O/bench_$c.err | tail -1 > $O/bench_$c.json; done
BIGKRLS_VERBOSE=1 python bench.py --config C4 --steps 1 --warmup 1 --no-cpu-baseline > $O/c4_verbose.log 2>&1
for f in $O/bench_C4.json $O/bench_C5.json; do python -c "
import json
d=json.load(open('$f')); r=d['roofline']; print('$f', d['value'], r['kernel'][:40], r['frac'], r.get('fit_frac'), d['phases_s'])"; done
grep "block Lanczos\|eigen n=" $O/c4_verbose.log | tail -40
