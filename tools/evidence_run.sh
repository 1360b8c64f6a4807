set -x
O=gpurun_out/${EVID:-r02s}; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q --durations=12 > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
for c in C2 C4 C5; do python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline 2>$O/bench_$c.err | tail -1 > $O/bench_$c.json; done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>$O/bench_C3_10.err | tail -1 > $O/bench_C3_10steps.json
python bench.py 2>$O/bench_default.err | tail -1 > $O/bench_C3_default_with_cpu_baseline.json
for c in C3 C4 C5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$c -o run -- python3 bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline 2>$O/prof_$c.err | tail -1 > $O/bench_${c}_under_rocprof.json
  f=$(find $O/prof_$c -name "*kernel_stats.csv" | head -1); cp "$f" $O/${c}_kernel_stats.csv; rm -rf $O/prof_$c
done
tail -3 $O/gpu_tests.log; cat $O/smoke.log | tail -2; for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(d['value'], d['roofline']['kernel'][:50], d['roofline']['frac'], d.get('cpu_baseline',{}).get('value'))"; done
