# round 5, call 9: the barrier probe with a rounding-mode check that tells nearest from truncation; then the evidence set
export TMPDIR=/tmp
O=gpurun_out/r05i; mkdir -p $O
tools/barrier_probe 3 3 > $O/barrier_probe_idle.log 2>&1; tail -1 $O/barrier_probe_idle.log
timeout 400 python tools/cwsr_probe_run.py --barrier --procs 40 --seconds 150 --ms 2 --load > $O/barrier_probe_40.log 2>&1; tail -6 $O/barrier_probe_40.log
EVID=r05z bash tools/evidence_run.sh > $O/evidence_run.log 2>&1; tail -25 $O/evidence_run.log
