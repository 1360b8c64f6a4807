# round 6, GPU call 27: the multi-rank cases again with the low-rank world-2 Lanczos case added
O=gpurun_out/${EVID:-r06x3}; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_dist_world.py -m gpu -q -x --durations=5 > $O/gpu_tests_world.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests_world.log
grep -v amdgpu.ids $O/gpu_tests_world.log | tail -30
