# stress: the world-3 ragged case beside heavy neighbours; keeps the logs of the rounds that fail
export TMPDIR=/tmp
O=gpurun_out/flake2; rm -rf $O; mkdir -p $O
fails=0
for i in $(seq 1 ${ROUNDS:-30}); do
  BIGKRLS_VERBOSE=1 BIGKRLS_PQ=steps BIGKRLS_BC=wavefront python tests/_dist_world_gpu.py 2500 6 3 > $O/a$i.log 2>&1 & pa=$!
  python tests/_dist_world_gpu.py 13500 6 2 --eigtrunc 0.001 --rccl-mock > $O/b$i.log 2>&1 & pb=$!
  BIGKRLS_PQ=steps BIGKRLS_BC=wavefront python tests/_dist_world_gpu.py 11700 6 3 --eigtrunc 0.001 > $O/c$i.log 2>&1 & pc=$!
  BIGKRLS_PQ=steps BIGKRLS_BC=wavefront python tests/_dist_world_gpu.py 3000 8 2 > $O/d$i.log 2>&1 & pd=$!
  python tests/_dist_world_gpu.py 17000 10 2 --krylov 60 --rccl-mock > $O/e$i.log 2>&1 & pe=$!
  wait $pa; ra=$?; wait $pb; rb=$?; wait $pc; rc=$?; wait $pd; rd=$?; wait $pe; re=$?
  if [ "$ra$rb$rc$rd$re" != "00000" ]; then fails=$((fails+1)); echo "round $i FAILED: rc $ra $rb $rc $rd $re"; else rm -f $O/*$i.log; fi
done
echo "rounds ${ROUNDS:-30} failures $fails"
