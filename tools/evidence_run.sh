set -x
O=gpurun_out/${EVID:-r04m}; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q --durations=12 > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
for c in C2 C4 C5; do python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline 2>$O/bench_$c.err | tail -1 > $O/bench_$c.json; done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>$O/bench_C3_10.err | tail -1 > $O/bench_C3_10steps.json
for c in C3 C4 C5; do python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --force-dist 2>$O/bench_${c}_dist.err | tail -1 > $O/bench_${c}_forcedist.json; done
for c in C3 C4 C5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$c -o run -- python3 bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline 2>$O/prof_$c.err | tail -1 > $O/bench_${c}_under_rocprof.json
  f=$(find $O/prof_$c -name "*kernel_stats.csv" | head -1); cp "$f" $O/${c}_kernel_stats.csv; rm -rf $O/prof_$c
done
# fabric traffic of the kernel build at the three sizes (separate passes)
for cfg in "20000 20" "50000 20" "100000 50"; do
  tag=$(echo $cfg | tr ' ' '_')
  for cnt in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $O/pmc_kb_${tag}_$cnt -o run -- python3 tools/kb_bench.py $cfg > /dev/null 2>&1
    f=$(find $O/pmc_kb_${tag}_$cnt -name "*counter_collection.csv" | head -1)
    python - "$f" "$cfg" $cnt >> $O/kernel_build_pmc.log <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "kernel_block" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[3]]
vals = [float(r["Counter_Value"]) for r in rows]
print(sys.argv[2], sys.argv[3], rows[0]["Kernel_Name"][:60] if rows else "-", "launches", len(vals), "mean bytes", 1024.0 * sum(vals) / max(len(vals), 1))
PY
    rm -rf $O/pmc_kb_${tag}_$cnt
  done
done
if [ -n "$WITH_CPU" ]; then python bench.py 2>$O/bench_default.err | tail -1 > $O/bench_C3_default_with_cpu_baseline.json; fi
tail -3 $O/gpu_tests.log; cat $O/smoke.log | tail -2; cat $O/kernel_build_pmc.log; for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(d['value'], d['roofline']['kernel'][:50], d['roofline']['frac'], d['kernel_gemm']['hbm_write_gbs'], d['kernel_gemm']['tflops'], d.get('cpu_baseline',{}).get('value'))"; done
