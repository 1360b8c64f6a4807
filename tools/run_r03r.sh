python tools/knob_ab.py 5000 10 BIGKRLS_PQ=householder - 2>&1 | grep best
python tools/knob_ab.py 20000 20 BIGKRLS_PQ=householder - 2>&1 | grep best
