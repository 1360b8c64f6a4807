set -x
O=gpurun_out/${EVID:-r06z}; mkdir -p $O
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
# fabric traffic of the kernel build at the three sizes (separate passes; KB_PMC=1)
for cfg in ${KB_PMC:+"20000 20" "50000 20" "100000 50"}; do
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
# ---- round 4 extras: phase times, stage-2 profiles, MFMA utilisation ----------------------------------------------
if [ -n "$R04_EXTRAS" ]; then
  for cfg in "5000 10" "20000 20"; do
    tag=$(echo $cfg | tr ' ' '_')
    BIGKRLS_VERBOSE=1 python tools/eig_once.py $cfg > $O/eig_verbose_$tag.log 2>&1
  done
  if [ -f tools/libbigkrls_bcprof.so ]; then
    BIGKRLS_BC_PROF_FLIGHT=1 python tools/bc_prof.py 20000 > $O/bc_prof_regwin.log 2>&1
    BIGKRLS_BC=lds python tools/bc_prof.py 20000 > $O/bc_prof_lds.log 2>&1
    python tools/bc_trace.py 20000 > $O/bc_trace.log 2>&1
  fi
  tools/pingpong > $O/pingpong.log 2>&1
  tools/ic_read_probe 40 > $O/ic_read_probe.log 2>&1
  CMD="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline"
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmcA -o run -- $CMD > /dev/null 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmcB -o run -- $CMD > /dev/null 2>&1
  python tools/mfma_pmc.py $O/pmcA $O/pmcB $O/mfma_pmc_C3.json "$CMD" > $O/mfma_pmc_C3.log 2>&1
  rm -rf $O/pmcA $O/pmcB
fi
# ---- round 6: fabric traffic of the trailing update's launches INSIDE the fit (SYRK_PMC=1; two separate passes) ------
if [ -n "$SYRK_PMC" ]; then
  CMD="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcF -o run -- $CMD > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcW -o run -- $CMD > /dev/null 2>&1
  python tools/syrk_infit_pmc.py $O/pmcF $O/pmcW $O/syrk_infit_traffic_pmc.json "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- $CMD" > $O/syrk_infit_traffic_pmc.log 2>&1
  rm -rf $O/pmcF $O/pmcW
  cat $O/syrk_infit_traffic_pmc.log | tail -2
fi
if [ -n "$WITH_CPU" ]; then python bench.py 2>$O/bench_default.err | tail -1 > $O/bench_C3_default_with_cpu_baseline.json; fi
tail -3 $O/gpu_tests.log; cat $O/smoke.log | tail -2; cat $O/kernel_build_pmc.log; for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); kg=d['roofline'].get('kernel_gemm') or d.get('kernel_gemm') or {}; print(len(json.dumps(d)), d['value'], d['roofline']['kernel'][:50], d['roofline']['frac'], d['roofline'].get('fit_frac'), kg.get('ms'), kg.get('hbm_write_gbs'), kg.get('tflops'), d.get('cpu_baseline',{}).get('value'))"; done
