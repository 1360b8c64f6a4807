# round 6, GPU call 34: the whole GPU suite and smoke() on the round's last commit
O=gpurun_out/${EVID:-r06x9}; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --durations=3 > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
grep -v amdgpu.ids $O/gpu_tests.log | tail -7; tail -2 $O/smoke.log
