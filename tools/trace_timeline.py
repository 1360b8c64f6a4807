"""Main-stream timeline of stage 1 from a rocprofv3 kernel trace (development tool):
  rocprofv3 --kernel-trace --output-format csv -d DIR -o run -- python3 tools/eig_once.py 20000 20
  python tools/trace_timeline.py DIR/.../run_kernel_trace.csv
Splits the LAST decomposition's stage 1 (from the first pq_resident to bc_resident) into what the GPU was doing:
busy time per kernel family on each queue, time where only the panel QR (side stream) ran, and idle gaps."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
def fam(n):
    for k in ("pq_resident", "syrk_mirror", "bc_resident", "bt_build_t", "splitk_reduce", "s1_fused_yt", "s1_fused_s", "s1_fused_z",
              "s1_extract", "bt2_wy", "dc_", "copyBuffer", "fillBuffer"):
        if k in n: return k
    if "gemm_kernel" in n:
        return "gemm<" + ("T" if "<true" in n else "N") + ("T" if ", true" in n.split(">")[0] else "N") + "," + n.split(",")[2].split(">")[0].strip() + ">"
    return n[:30]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), fam(r["Kernel_Name"]), r.get("Queue_Id", r.get("Stream_Id", "?"))) for r in rows]
ev.sort()
bcs = [e for e in ev if e[2] == "bc_resident"]
t_end = bcs[-1][0]
t_prev = bcs[-2][1] if len(bcs) > 1 else ev[0][0]
s1 = [e for e in ev if e[0] >= t_prev and e[1] <= t_end]
first_pq = min(e[0] for e in s1 if e[2] == "pq_resident")
s1 = [e for e in s1 if e[0] >= first_pq]
T0, T1 = first_pq, t_end
print(f"stage 1 window: {(T1 - T0) / 1e6:.1f} ms, {len(s1)} kernels")
# sweep line: classify every instant
pts = []
for a, b, f, q in s1:
    pts.append((a, 1, f)); pts.append((b, -1, f))
pts.sort()
active = collections.Counter()
cur = T0
acc = collections.Counter()
for t, d, f in pts:
    if t > cur:
        fams = tuple(sorted(k for k, v in active.items() if v > 0))
        acc[fams] += t - cur
        cur = t
    active[f] += d
tot = T1 - T0
for fams, v in acc.most_common(18):
    print(f"  {100 * v / tot:5.1f} %  {v / 1e6:7.2f} ms  {' + '.join(fams) if fams else '(idle)'}")
per = collections.Counter()
for a, b, f, q in s1: per[f] += b - a
print("kernel time by family:")
for f, v in per.most_common(14): print(f"  {v / 1e6:7.2f} ms  {f}")
