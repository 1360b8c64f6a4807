export TMPDIR=/tmp
O=gpurun_out/r05v; mkdir -p $O
python bench.py --cpu-budget-s 60 > $O/bench_default_short_cpu_budget.json 2> $O/bench_default.err; echo "rc=$?"; tail -c 1500 $O/bench_default_short_cpu_budget.json; python -c "
import json; d=json.loads(open('$O/bench_default_short_cpu_budget.json').read().strip().splitlines()[-1]); print(len(json.dumps(d)), d['value'], d['roofline']['kernel_gemm'], list(d['cpu_baseline'].keys()))"
