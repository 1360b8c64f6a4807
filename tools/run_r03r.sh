python tools/_ab/chk.py > /tmp/chk.log 2>&1
grep -v amdgpu.ids /tmp/chk.log | grep -v "left panel" | head -4 | cut -c1-200
grep -c "left panel" /tmp/chk.log
python tools/knob_ab.py 5000 10 BIGKRLS_PQ=householder - 2>&1 | grep best
python tools/fit_ab.py 20000 20 tools/libbigkrls_head.so bigkrls_amd/libbigkrls_hip.so 2>&1 | grep best
timeout 600 python tools/eig_stress.py 2>&1 | tail -16
