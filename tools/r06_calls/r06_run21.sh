# round 6, GPU call 24: conditioning of the new block at every block-Lanczos step (pivots of its Gram matrix), healthy and low-rank kernels
O=gpurun_out/${EVID:-r06w}; mkdir -p $O
export BIGKRLS_KRY_PIVOTS=1 BIGKRLS_VERBOSE=1
for cfg in "50000 20 512 104" "100000 50 1024 105" "40000 15 300 7" "17000 10 60 3" "50000 2 512 7" "30000 3 512 7" "24000 4 256 7"; do
  echo "==== $cfg"; python tools/kry_history.py $cfg 2>&1 | grep -i "lanczos\|lastkeeper\|rror" | awk '!seen[$0]++'
done > $O/kry_pivots.log 2>&1
BIGKRLS_EIGK=krylov python tools/kry_history.py 20000 2 512 7 2>&1 | grep -i "lanczos\|lastkeeper\|rror" | awk '!seen[$0]++' >> $O/kry_pivots.log
cat $O/kry_pivots.log | cut -c1-200
