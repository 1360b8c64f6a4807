"""Per stage-1 panel, from a rocprofv3 kernel trace (see trace_timeline.py for the command): how long after the
fused kernel that releases the next panel's columns (s1_fused_z) do the panel QR (look-ahead stream) and the trailing
update (main stream) start, how long do they run, and which of them ends last. Development tool."""
import csv, sys, statistics as st
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
bcs = [e for e in ev if "bc_resident" in e[2]]
t_end = bcs[-1][0]
t_prev = bcs[-2][1] if len(bcs) > 1 else ev[0][0]
s1 = [e for e in ev if e[0] >= t_prev and e[1] <= t_end]
zs = [e for e in s1 if "s1_fused_z" in e[2]]
out = []
for i, z in enumerate(zs):
    nxt = zs[i + 1][0] if i + 1 < len(zs) else t_end
    pq = [e for e in s1 if ("pq_chol" in e[2] or "pq_resident" in e[2]) and z[1] <= e[0] < nxt]
    if pq: pq = [(pq[0][0], max(e[1] for e in pq), "pq")]
    sy = [e for e in s1 if "syrk_mirror" in e[2] and z[1] <= e[0] < nxt]
    av = [e for e in s1 if "gemm_kernel<false, false, 64>" in e[2] and z[1] <= e[0] < nxt]
    if not pq or not sy: continue
    out.append(dict(i=i, pq_delay=(pq[0][0] - z[1]) / 1e3, sy_delay=(sy[0][0] - z[1]) / 1e3, pq_us=(pq[0][1] - pq[0][0]) / 1e3,
                    sy_us=sum(e[1] - e[0] for e in sy) / 1e3, pq_after_sy=(pq[0][1] - sy[-1][1]) / 1e3,
                    av_wait=((av[0][0] - max(pq[0][1], sy[-1][1])) / 1e3) if av else float("nan"), step=(nxt - z[1]) / 1e3))
print(f"{len(out)} panels with both launches")
print("panel  QR starts  update starts  QR runs  update runs  QR ends after update  next A22 V starts after both  step (us)")
for o in out[::max(1, len(out) // 40)]:
    print(f"{o['i']:5d}  {o['pq_delay']:9.1f}  {o['sy_delay']:13.1f}  {o['pq_us']:7.1f}  {o['sy_us']:11.1f}  {o['pq_after_sy']:20.1f}  {o['av_wait']:28.1f}  {o['step']:8.1f}")
for k in ("pq_delay", "sy_delay", "pq_us", "sy_us", "pq_after_sy"):
    v = [o[k] for o in out]
    print(f"{k:12s} mean {st.mean(v):8.1f}  median {st.median(v):8.1f}  sum {sum(v) / 1e3:8.2f} ms")
print("sum over panels of max(0, QR ends after update): %.2f ms" % (sum(max(0.0, o["pq_after_sy"]) for o in out) / 1e3))
