# round 6, GPU call 32: soak of the final build -- 8 rounds of seven mixed fits on one context, lambda bitwise stable, no slow fit
O=gpurun_out/${EVID:-r06x7}; mkdir -p $O
timeout 420 python tools/soak.py 8 > $O/soak.log 2>&1; echo "rc=$?" >> $O/soak.log
grep -v amdgpu.ids $O/soak.log | tail -20
