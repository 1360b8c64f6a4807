# round 6, GPU call 29: decomposition residuals over the sizes that cross the size-dependent switches, final build
O=gpurun_out/${EVID:-r06x5}; mkdir -p $O
( python tools/size_sweep.py; python tools/size_sweep.py large; python tools/size_sweep.py r6 ) > $O/size_sweep.log 2>&1
grep -v amdgpu.ids $O/size_sweep.log | tail -70
