# round 6, GPU call 9: eight instead of four panels per merged stage-1 back-transform block above n = 10 000, now that the
# blocks are built in batched launches (round 4 measured: steps faster, precompute as much slower)
O=gpurun_out/${EVID:-r06j}; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/knob_ab.py 20000 20 - BIGKRLS_BT1_GRP=8 > $O/bt1_grp8_ab_C3.log 2>&1; grep best $O/bt1_grp8_ab_C3.log
timeout 600 python tools/knob_ab.py 14000 10 - BIGKRLS_BT1_GRP=8 > $O/bt1_grp8_ab_N14000.log 2>&1; grep best $O/bt1_grp8_ab_N14000.log
BIGKRLS_BT1_GRP=8 BIGKRLS_VERBOSE=1 python tools/eig_once.py 20000 20 > $O/eig_verbose_20000_20_grp8.log 2>&1; grep -E "divide|stage|back|gather" $O/eig_verbose_20000_20_grp8.log | tail -7 | cut -c1-220
