O=gpurun_out/r03p; mkdir -p $O
timeout 900 python tools/soak.py 3 > $O/soak.log 2>&1; tail -12 $O/soak.log
timeout 600 python tools/eig_stress.py > $O/eig_stress.log 2>&1; tail -15 $O/eig_stress.log
timeout 300 python tools/determinism.py > $O/determinism.log 2>&1; tail -5 $O/determinism.log
