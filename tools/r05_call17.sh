export TMPDIR=/tmp
O=gpurun_out/r05w; mkdir -p $O
python -m pytest tests -m gpu -x -q --durations=6 > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log; tail -12 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
