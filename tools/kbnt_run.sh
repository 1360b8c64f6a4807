python tools/kb_bench.py 20000 20 2>&1 | tail -1
python tools/kb_bench.py 50000 20 2>&1 | tail -1
python tools/kb_bench.py 100000 20 2>&1 | tail -1
python -m pytest tests/test_gpu_level1.py -x -q -k "kernel" 2>&1 | tail -2
mkdir -p gpurun_out/kbnt
for cnt in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d gpurun_out/kbnt/pmc_$cnt -o run -- python3 tools/kb_bench.py 50000 20 > /dev/null 2>&1
done
for c in C4 C5; do python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/kbnt/bench_$c.json; done
