# round 5, call 19: fabric traffic of the trailing update once more (the k = 512 whole-triangle shape carries 65 % of the
# kernel's time since round 4 and had last been measured in round 3): FETCH_SIZE and WRITE_SIZE in separate passes
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05u; mkdir -p $O
R=$PWD
cd /tmp
for cnt in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $O/pmc_$cnt -o run -- $R/tools/syrk_k_probe 20000 q > $O/probe_$cnt.log 2>&1
  f=$(find $O/pmc_$cnt -name "*counter_collection.csv" | head -1)
  python3 - "$f" $cnt >> $O/syrk_traffic_pmc.log <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "syrk_mirror" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2]]
print(sys.argv[2], "launches", len(rows))
for i, r in enumerate(rows):
    print(i, r["Kernel_Name"][:40], "grid", r.get("Grid_Size"), "bytes", 1024.0 * float(r["Counter_Value"]))
PY
  rm -rf $O/pmc_$cnt
done
cd $R
cat $O/probe_FETCH_SIZE.log | head -40; cat $O/syrk_traffic_pmc.log | head -80
