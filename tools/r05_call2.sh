# round 5, call 2: the pageable-source probes, the regression test of the divide & conquer fix, tests of what changed,
# A/B of knobs and of the poll back-off builds, kernel trace of one C2 decomposition, the single-process control again
export TMPDIR=/tmp
O=gpurun_out/r05b; mkdir -p $O
tools/pageable_h2d_probe > $O/pageable_h2d_probe.log 2>&1; cat $O/pageable_h2d_probe.log
timeout 600 python tools/dc_async_source_probe.py > $O/dc_async_source_probe.log 2>&1; grep BIGKRLS_FAULT $O/dc_async_source_probe.log
timeout 900 python tests/_fault_inject.py > $O/fault_inject.log 2>&1; tail -3 $O/fault_inject.log
timeout 900 python -m pytest tests/test_gpu_level1.py -q -x -k "deriv" > $O/test_deriv.log 2>&1; tail -3 $O/test_deriv.log
BIGKRLS_SKIP_WORLD_RUNS=1 timeout 900 python -m pytest tests/test_gpu_configs.py -q -x -k "c3 or C3" > $O/test_c3.log 2>&1; tail -3 $O/test_c3.log
timeout 600 python tools/knob_ab.py 20000 20 - "BIGKRLS_DERIV48=0" "BIGKRLS_KB_R=1024" "BIGKRLS_KB_NT=1" "BIGKRLS_S1AGG4_MIN=10752" "BIGKRLS_S1AGG_MIN=8704,BIGKRLS_S1AGG4_MIN=10752" > $O/knob_c3.log 2>&1; cat $O/knob_c3.log
timeout 600 python tools/fit_ab.py 20000 20 bigkrls_amd/libbigkrls_hip.so tools/_ab/libbigkrls_sleep1.so tools/_ab/libbigkrls_sleep4.so > $O/ab_sleep_c3.log 2>&1; cat $O/ab_sleep_c3.log
timeout 300 python tools/fit_ab.py 5000 10 bigkrls_amd/libbigkrls_hip.so tools/_ab/libbigkrls_sleep1.so tools/_ab/libbigkrls_sleep4.so > $O/ab_sleep_c2.log 2>&1; cat $O/ab_sleep_c2.log
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_c2 -o run -- python3 $R/tools/eig_once.py 5000 10 > $R/$O/eig_prof_c2.log 2>&1
cd $R
f=$(find $O/prof_c2 -name "*kernel_stats.csv" | head -1); cp "$f" $O/eig_5000_10_kernel_stats.csv
t=$(find $O/prof_c2 -name "*kernel_trace.csv" | head -1); python tools/trace_timeline.py "$t" > $O/eig_5000_10_timeline.log 2>&1
python - "$t" > $O/c2_panel_chain.log 2>&1 <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, nm in enumerate(names) if "pq_chol" in nm]
k = idx[len(idx) - 40]           # a panel of the last decomposition, 40 panels before its end (m ~ 2500)
t0 = int(rows[k]["Start_Timestamp"])
for r in rows[k - 2 : k + 30]:
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} {(int(r["End_Timestamp"]) - t0) / 1e3:9.1f} us  q{r.get("Queue_Id","?")} {r["Kernel_Name"][:90]}')
PY
head -36 $O/c2_panel_chain.log
gzip -c "$t" > $O/eig_5000_10_kernel_trace.csv.gz; rm -rf $O/prof_c2
timeout 600 python bench.py --config C3 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; python -c "
import json; d=json.load(open('$O/bench_c3.json')); print(len(json.dumps(d)), d['value'], d['roofline']['frac'], d['roofline']['kernel_gemm'], [ (k['kernel'],k['achieved'],k.get('avg_launch_us')) for k in d['other_kernels']])"
rm -rf gpurun_out/oversub_single
timeout 900 python tools/oversub_single.py --minutes ${SINGLE_MIN:-8} > $O/single_after_fix.log 2>&1
tail -12 $O/single_after_fix.log
