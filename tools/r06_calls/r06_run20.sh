# round 6, GPU call 23: the dense eigensolver on small graded-spectrum matrices
O=gpurun_out/${EVID:-r06w}; mkdir -p $O
python tools/graded_spectrum_check.py > $O/graded_spectrum_check.log 2>&1
grep -v amdgpu.ids $O/graded_spectrum_check.log | tail -60
