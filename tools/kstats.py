"""Print the top rows of a rocprofv3 kernel_stats.csv (development helper)."""
import csv, glob, sys
f = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob('gpurun_out/**/*kernel_stats.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f, 'total kernel ms', round(tot / 1e6, 2))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    print(f"{r['Name'][:72]:72s} {int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):5.1f}%")
