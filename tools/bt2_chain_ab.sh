# stage-2 back-transform: chains (BIGKRLS_BT2=chain, BIGKRLS_BT2_SEG groups per ticket) vs one ticket per task (BIGKRLS_BT2=tasks)
export TMPDIR=/tmp
O=gpurun_out/bt2chain; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_level1.py -x -q -k "eigen or bt2 or back" > $O/tests.log 2>&1; tail -3 $O/tests.log
for n in "20000 20" "10000 20" "7000 10"; do
  for mode in tasks chain16 chain32 chain64; do
    case $mode in
      tasks) export BIGKRLS_BT2=tasks; unset BIGKRLS_BT2_SEG;;
      chain*) export BIGKRLS_BT2=chain; export BIGKRLS_BT2_SEG=${mode#chain};;
    esac
    echo "== $n $mode"; BIGKRLS_VERBOSE=1 timeout 300 python tools/eig_once.py $n 2>&1 | grep -E "back-transform stage 2" | tail -2
  done
done
