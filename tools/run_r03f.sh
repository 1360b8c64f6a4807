set -x
O=gpurun_out/r03f; mkdir -p $O
export TMPDIR=/tmp
timeout 120 tools/store_run_probe 20000 > $O/store_run_probe.log 2>&1
timeout 120 tools/store_run_probe 50000 >> $O/store_run_probe.log 2>&1
timeout 1500 python tools/knob_ab.py 20000 20 "-" "BIGKRLS_PQ_LDS_PAD=8192" "BIGKRLS_S1AGG_MIN=12288" "BIGKRLS_S1AGG_MIN=10240,BIGKRLS_PQ_LDS_PAD=8192" "BIGKRLS_PQ_LDS_PAD=40000" > $O/knob_ab.log 2>&1
cat $O/store_run_probe.log; grep -v amdgpu $O/knob_ab.log
