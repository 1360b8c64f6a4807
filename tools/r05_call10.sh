# round 5, call 10: every workspace slab poisoned (all-ones bytes = NaN / -1) when allocated and at the start of every
# fit: anything that reads workspace before writing it turns the result into NaN at once
export TMPDIR=/tmp BIGKRLS_POISON=1
O=gpurun_out/r05j; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_poison.log 2>&1; tail -2 $O/smoke_poison.log
BIGKRLS_SKIP_WORLD_RUNS=1 timeout 1500 python -m pytest tests/test_gpu_fit.py tests/test_gpu_level1.py tests/test_gpu_fit_capi.py tests/test_gpu_golden_and_properties.py -q -x > $O/tests_poison.log 2>&1; tail -5 $O/tests_poison.log
BIGKRLS_SKIP_WORLD_RUNS=1 timeout 1500 python -m pytest tests/test_gpu_configs.py -q -x -k "C2 or c2 or C3 or c3 or 33000 or aggregated" > $O/tests_poison_configs.log 2>&1; tail -5 $O/tests_poison_configs.log
BIGKRLS_PQ=steps BIGKRLS_BC=wavefront timeout 600 python tests/_dist_world_gpu.py 2500 6 3 > $O/world3_poison.log 2>&1; tail -3 $O/world3_poison.log | cut -c1-300
timeout 600 python tests/_dist_world_gpu.py 13500 6 2 --eigtrunc 0.001 --rccl-mock --default-knobs > $O/world2_groups_poison.log 2>&1; tail -2 $O/world2_groups_poison.log | cut -c1-300
timeout 600 python tests/_dist_world_gpu.py 17000 10 2 --krylov 60 > $O/world2_krylov_poison.log 2>&1; tail -2 $O/world2_krylov_poison.log | cut -c1-300
