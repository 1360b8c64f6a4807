# round 6, GPU call 21: the block Lanczos of the final build over a spread of shapes, with and without the estimate
O=gpurun_out/${EVID:-r06v}; mkdir -p $O
python tools/kry_sweep.py > $O/kry_sweep.log 2>&1
grep -v amdgpu.ids $O/kry_sweep.log | tail -120
