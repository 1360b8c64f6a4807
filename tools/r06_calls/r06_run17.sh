# round 6, GPU call 20: kernel-by-kernel timeline of the stage-1 panel chain at N = 20 000 (where does the empty pq_resident launch wait?)
O=gpurun_out/${EVID:-r06u}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --kernel-trace --output-format csv -d /tmp/r06u_trace -- python3 tools/eig_once.py 20000 20 > $O/eig_once.log 2>&1
for f in 0.03 0.15 0.3 0.45 0.6 0.75 0.9; do python tools/panel_chain_timeline.py /tmp/r06u_trace $f > $O/C3_panel_chain_timeline_$f.log 2>&1; done
head -70 $O/C3_panel_chain_timeline_0.15.log
