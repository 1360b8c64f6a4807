"""Reads the diagnostic traces of one multi-rank run (BIGKRLS_TRACE_DIR: csrc/trace.hip writes pid<pid>.trace, the
callback collectives of bigkrls_amd/dist.py write pid<pid>.pytrace) and names the first buffer that is not what it
should be:

  * a collective's result ("C:" lines) or a replicated intermediate ("R:" lines) that differs between the ranks;
  * with the callback transport: a buffer whose hash on the host differs from its hash on the device before the
    copy (device -> host) or after it (host -> device);
  * `--repeat`: the fits of ONE process (the same fit repeated) compared with its first fit.

    python tools/trace_diff.py DIR [--repeat] [--quiet]

Exit code 0: consistent; 1: an inconsistency was found (printed)."""
import glob
import os
import sys


def load(dirname):
    procs = []
    for path in sorted(glob.glob(os.path.join(dirname, "pid*.trace"))):
        pid = os.path.basename(path)[3:-6]
        lines = []
        for ln in open(path):
            f = ln.split()
            if len(f) != 6:
                continue
            lines.append((f[1], int(f[2]), f[3], int(f[4]), int(f[5])))     # tag, count, hash, extra, free MiB
        rank, world, kind = None, None, None
        fits = []
        for rec in lines:
            if rec[0].startswith("L:comm_"):
                rank, world, kind = rec[3] // 1000, rec[3] % 1000, rec[0][7:]
            elif rec[0] == "L:fit_begin":
                fits.append([])
            elif fits:
                fits[-1].append(rec)
        py = []
        pp = os.path.join(dirname, f"pid{pid}.pytrace")
        if os.path.exists(pp):
            for ln in open(pp):
                f = ln.split()
                if len(f) == 5:
                    py.append((f[1], int(f[2]), f[3], f[4]))                # kind, count, hash as fetched, hash as stored
        procs.append(dict(pid=pid, rank=rank, world=world, kind=kind, fits=fits, py=py, lines=lines))
    return procs


def shared(recs):
    return [r for r in recs if r[0][:2] in ("C:", "R:")]


def first_diff(seqs, names):
    """first position at which the sequences (lists of records) disagree in (tag, count, hash)"""
    n = min(len(s) for s in seqs)
    for i in range(n):
        keys = {(s[i][0], s[i][1], s[i][2]) for s in seqs}
        if len(keys) > 1:
            return i, [f"{nm}: {s[i][0]} count={s[i][1]} hash={s[i][2]} extra={s[i][3]}" for nm, s in zip(names, seqs)]
    if len({len(s) for s in seqs}) > 1:
        return n, [f"{nm}: {len(s)} records" for nm, s in zip(names, seqs)]
    return None


def check_transport(p):
    """callback transport of one rank: device hash before the copy == host hash after it, and back"""
    bad = []
    recs = [r for f in p["fits"] for r in f] if p["kind"] == "callbacks" else []
    # also the collectives outside a fit (none today) are ignored: the python trace counts every callback
    ins = [r for r in p["lines"] if r[0] in ("L:ar_in", "L:ag_in", "L:bc_in", "L:bc_in_root")]
    outs = [r for r in p["lines"] if r[0] in ("C:ar_out", "C:ag_out", "C:bc_out")]
    if p["kind"] != "callbacks":
        return bad
    for i, cb in enumerate(p["py"]):
        if i < len(ins) and ins[i][2] != cb[2]:
            bad.append(f"rank {p['rank']} collective #{i} ({cb[0]}, {cb[1]} doubles): device -> host copy changed the data "
                       f"(device {ins[i][2]}, host {cb[2]})")
        if i < len(outs) and outs[i][2] != cb[3]:
            bad.append(f"rank {p['rank']} collective #{i} ({cb[0]}, {cb[1]} doubles): host -> device copy changed the data "
                       f"(host {cb[3]}, device {outs[i][2]})")
        if len(bad) >= 4:
            break
    return bad


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    repeat = "--repeat" in sys.argv
    quiet = "--quiet" in sys.argv
    procs = load(args[0])
    if not procs:
        print("no traces in", args[0])
        return 2
    found = []
    if repeat:
        for p in procs:
            if len(p["fits"]) < 2:
                continue
            # the RESULTS of a fit ("R:fit_*": eigenvalues, Q, a, lambda, c, yhat, D, var) must repeat bit for bit. The
            # records before them may differ in number: a fired watchdog of a persistent kernel makes the call redo the
            # decomposition with the launch-per-step kernels (its first, abandoned pass is in the trace too) -- reported
            # as a retry, not as an inconsistency.
            res = [[r for r in f if r[0].startswith("R:fit_")] for f in p["fits"]]
            for j, f in enumerate(p["fits"][1:], 1):
                d = first_diff([res[0], res[j]], ["fit 0", f"fit {j}"])
                if d:
                    msg = (f"pid {p['pid']}: the results of fit {j} differ from fit 0 at result record {d[0]} (min free "
                           f"memory {min(r[4] for r in f)} MiB):\n    " + "\n    ".join(d[1]))
                    # where the two fits part ways: the first record (of all, not only the results) that differs from the
                    # majority of the process's fits with the same number of records
                    same = [g for g in p["fits"] if len(g) == len(f)]
                    if len(same) >= 3:
                        from collections import Counter
                        for i in range(len(f)):
                            maj = Counter(g[i][2] for g in same).most_common(1)[0][0]
                            if f[i][2] != maj:
                                prev = f[i - 1] if i > 0 else None
                                msg += (f"\n    first record of fit {j} that deviates from the other fits: #{i} {f[i][0]} extra={f[i][3]} "
                                        f"count={f[i][1]}" + (f" (after {prev[0]} extra={prev[3]}, still equal)" if prev else ""))
                                break
                    found.append(msg)
                elif len(f) != len(p["fits"][0]) and not quiet:
                    print(f"pid {p['pid']}: fit {j} has {len(f)} records, fit 0 {len(p['fits'][0])}: a decomposition was redone "
                          "(watchdog retry), results identical")
    else:
        ranks = sorted([p for p in procs if p["rank"] is not None], key=lambda p: p["rank"])
        if ranks:
            nf = min(len(p["fits"]) for p in ranks)
            for j in range(nf):
                seqs = [shared(p["fits"][j]) for p in ranks]
                if j > 0:     # the single-process fit every rank runs afterwards: the same computation everywhere
                    seqs = [[r for r in p["fits"][j] if not r[0].startswith("L:comm")] for p in ranks]
                d = first_diff(seqs, [f"rank {p['rank']}" for p in ranks])
                if d:
                    what = "the distributed fit" if j == 0 else "the single-process fit that follows"
                    prev = seqs[0][d[0] - 1] if d[0] > 0 else None
                    found.append(f"{what}: ranks differ at shared record {d[0]}" +
                                 (f" (the record before it, equal on all ranks: {prev[0]} count={prev[1]} extra={prev[3]})" if prev else "") +
                                 ":\n    " + "\n    ".join(d[1]))
            for p in ranks:
                found += check_transport(p)
    lo = min((r[4] for p in procs for r in p["lines"]), default=-1)
    if not quiet or found:
        print(f"{args[0]}: {len(procs)} processes, lowest free device memory seen {lo} MiB, "
              f"{sum(len(p['lines']) for p in procs)} records")
    for f in found:
        print("INCONSISTENT:", f)
    if not found and not quiet:
        print("consistent")
    return 1 if found else 0


if __name__ == "__main__":
    sys.exit(main())
