set -x
O=gpurun_out/r03d; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q --durations=8 > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
tail -30 $O/gpu_tests.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>$O/bench_C3.err | tail -1 > $O/bench_C3.json
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --force-dist 2>$O/bench_C3_dist.err | tail -1 > $O/bench_C3_forcedist.json
python bench.py --config C4 --steps 3 --warmup 1 --no-cpu-baseline 2>$O/bench_C4.err | tail -1 > $O/bench_C4.json
python bench.py --config C4 --steps 3 --warmup 1 --no-cpu-baseline --force-dist 2>$O/bench_C4_dist.err | tail -1 > $O/bench_C4_forcedist.json
for f in $O/bench_*.json; do echo $f; python -c "
import json
d=json.load(open('$f')); print(d['value'], d['phases_s']); print([(o['kernel'][:28], o['achieved'], o['unit'], o.get('avg_launch_us')) for o in d['other_kernels'] if o['kernel'][:3] in ('sf_','der','gem','syr')])"; done
tail -5 $O/*.err
