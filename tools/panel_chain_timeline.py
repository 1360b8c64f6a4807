"""One stage-1 panel step kernel by kernel, from a `rocprofv3 --kernel-trace --output-format csv` directory of
`tools/eig_once.py N P`: the launches around a pq_chol launch in the middle of the last decomposition (start and end in
microseconds relative to that launch, queue, kernel), three panel periods long.   python tools/panel_chain_timeline.py DIR [which]
(`which`: fraction of the way through the pq_chol launches of the last decomposition, default 0.5)"""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
pq = [i for i, r in enumerate(rows) if "pq_chol" in r[3]]
# decompositions are separated by long gaps between pq_chol launches
last = [pq[0]]
for a, b in zip(pq, pq[1:]):
    if rows[b][0] - rows[a][0] > 20_000_000:
        last = []
    last.append(b)
i0 = last[int(frac * (len(last) - 1))]
i3 = last[min(int(frac * (len(last) - 1)) + 3, len(last) - 1)]
t0 = rows[i0][0]
queues = {}
print(f"# {len(last)} pq_chol launches in the last decomposition; panel period here {(rows[i3][0] - t0) / 3e3:.1f} us")
for s, e, q, name in rows:
    if t0 - 60_000 <= s <= rows[i3][0] + 20_000:
        qi = queues.setdefault(q, len(queues) + 1)
        print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} us  q{qi} {name[:100]}")
