# round 6, GPU call 16: the block-Lanczos product W = K B_j in isolation -- split counts, HBM stream against a cache-resident operand
O=gpurun_out/${EVID:-r06r}; mkdir -p $O
export TMPDIR=/tmp
( ./tools/kb_shape_probe 50000
  for s in 1 3 4 5 6 7 9 13; do BIGKRLS_GEMM_SPLITS=$s ./tools/kb_shape_probe 50000 | grep -v "64 columns"; done
  ./tools/kb_shape_probe 100000
  for s in 3 5 7 10; do BIGKRLS_GEMM_SPLITS=$s ./tools/kb_shape_probe 100000 | grep -v "64 columns"; done ) > $O/kb_shape_probe.log 2>&1
cat $O/kb_shape_probe.log
