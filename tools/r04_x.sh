for cfg in "5000 10" "20000 20"; do BIGKRLS_VERBOSE=1 timeout 300 python tools/eig_once.py $cfg 2>&1 | grep "^rep\|back-transform stage 2" | tail -3; done
for r in 0 100000; do echo "KB_R=$r"; BIGKRLS_KB_R=$r python tools/kb_bench.py 20000 20 2>&1 | tail -1; BIGKRLS_KB_R=$r python tools/kb_bench.py 50000 20 2>&1 | tail -1; done
timeout 300 python tools/contention_check.py 5000 2>&1 | tail -3
