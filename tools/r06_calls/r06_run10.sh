# round 6, GPU call 10: the aggregation thresholds of stage 1 again, now that the update runs at three workgroups per CU
# behind the gate and the look-ahead stream carries less (round 5 found lower thresholds neutral to slower)
O=gpurun_out/${EVID:-r06m}; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python tools/knob_ab.py 20000 20 - BIGKRLS_S1AGG4_MIN=15000 BIGKRLS_S1AGG4_MIN=16500 BIGKRLS_S1AGG4_MIN=18000 BIGKRLS_S1AGG=2 > $O/agg_threshold2_ab_C3.log 2>&1; grep best $O/agg_threshold2_ab_C3.log
