# round 6, GPU call 28: fits on the boundary shapes of the block-Lanczos path against the dense decomposition
O=gpurun_out/${EVID:-r06x4}; mkdir -p $O
python tools/fit_kry_sweep.py > $O/fit_kry_sweep.log 2>&1
grep -v amdgpu.ids $O/fit_kry_sweep.log | tail -30
