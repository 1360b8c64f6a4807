set -x
O=gpurun_out/r03c; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q --durations=8 > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>$O/bench_C3.err | tail -1 > $O/bench_C3.json
for c in C3 C4; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmcA_$c -o run -- python3 bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>$O/pmcA_$c.err
  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmcB_$c -o run -- python3 bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>$O/pmcB_$c.err
  python tools/mfma_pmc.py $O/pmcA_$c $O/pmcB_$c $O/mfma_pmc_$c.json "python3 bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline" > $O/mfma_pmc_$c.log 2>&1
  rm -rf $O/pmcA_$c $O/pmcB_$c
done
tail -5 $O/gpu_tests.log; tail -2 $O/smoke.log; cat $O/bench_C3.json; cat $O/mfma_pmc_C3.log $O/mfma_pmc_C4.log
