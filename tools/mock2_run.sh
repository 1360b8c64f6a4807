# bench.py --gpus 2 on ONE GPU over tests/mock_rccl (function check of the multi-rank path at the bench sizes; the two
# ranks share the device, so the times are not a scaling measurement)
export TMPDIR=/tmp BIGKRLS_BENCH_SHARE_GPU=1 BIGKRLS_RCCL_LIB=$PWD/tests/mock_rccl/libmock_rccl.so
O=gpurun_out/mock2; mkdir -p $O
for c in C3 C4; do
  timeout 900 python bench.py --gpus 2 --config $c --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?"
  tail -1 $O/bench_$c.json | cut -c1-300
done
