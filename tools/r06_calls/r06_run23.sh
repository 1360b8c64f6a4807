# round 6, GPU call 26: the same with the early check fired once only: low-rank cases, the sweep, the whole suite
O=gpurun_out/${EVID:-r06x2}; mkdir -p $O
export TMPDIR=/tmp
( python tools/lowrank_check.py 20000 2 512; python tools/lowrank_check.py 30000 3 512; python tools/lowrank_check.py 50000 2 512 ) > $O/lowrank_check.log 2>&1
grep -v amdgpu.ids $O/lowrank_check.log | grep "eigen:\|fit:\|check:\|max |theta\|----\|re-orth" 
python -m pytest tests -m gpu -q --durations=5 > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
tail -4 $O/gpu_tests.log; tail -2 $O/smoke.log
