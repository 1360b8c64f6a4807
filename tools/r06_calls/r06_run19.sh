# round 6, GPU call 22: numerically low-rank kernels (P = 2, 3) through the block Lanczos and the dense path
O=gpurun_out/${EVID:-r06w}; mkdir -p $O
( python tools/lowrank_check.py 20000 2 512; python tools/lowrank_check.py 20000 2 128; python tools/lowrank_check.py 20000 3 512; python tools/lowrank_check.py 24000 1 256 ) > $O/lowrank_check.log 2>&1
grep -v amdgpu.ids $O/lowrank_check.log | tail -80
