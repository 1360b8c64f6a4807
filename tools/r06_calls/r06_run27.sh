# round 6, GPU call 30: Neig above the numerical rank of the kernel (random blocks on the invariant subspace), then the Lanczos tests and the whole suite
O=gpurun_out/${EVID:-r06x6}; mkdir -p $O
export TMPDIR=/tmp
( python tools/lowrank_check.py 20000 2 1024; python tools/lowrank_check.py 20000 2 2048; python tools/lowrank_check.py 50000 2 1024 ) > $O/lowrank_check.log 2>&1
grep -v amdgpu.ids $O/lowrank_check.log | grep "eigen:\|fit:\|check:\|max |theta\|----\|re-orth\|invariant\|rror"
python -m pytest tests -m gpu -q --durations=5 > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
tail -6 $O/gpu_tests.log; tail -2 $O/smoke.log
