set -x
O=gpurun_out/r03j; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q --durations=6 > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log; tail -12 $O/gpu_tests.log
for c in C4 C5; do BIGKRLS_VERBOSE=1 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline 2>$O/bench_$c.err | tail -1 > $O/bench_$c.json; grep "check: steps" $O/bench_$c.err | tail -4; done
for c in C3 C4 C5; do python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --force-dist 2>$O/bench_${c}_dist.err | tail -1 > $O/bench_${c}_forcedist.json; done
for f in $O/bench_*.json; do echo $f; python -c "
import json
d=json.load(open('$f')); print(d['value'], d['phases_s'])"; done
