python tests/_fault_inject.py 2>&1 | tail -2
for knobs in "BIGKRLS_PQ=pessimistic" "" "BIGKRLS_BT1_GRP=8"; do
  echo "== $knobs"
  for cfg in "5000 10" "20000 20"; do env $knobs BIGKRLS_VERBOSE=1 timeout 300 python tools/eig_once.py $cfg 2>&1 | grep "^rep\|stage 1 (dense\|back-transform stage 1" | tail -3; done
done
