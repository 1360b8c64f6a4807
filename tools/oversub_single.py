"""Control for the rare wrong results of multi-rank runs under oversubscription: the SAME load --
many processes with their own contexts and N x N workspaces on one GPU, beside a C4 fit -- but every process runs plain
single-GPU fits (no communicator, no collectives, no host round trips), the same fit several times. Every fit logs the
hashes of its intermediate results (BIGKRLS_TRACE_DIR) and must reproduce its process's first fit bit for bit
(tools/trace_diff.py --repeat).

    python tools/oversub_single.py [--minutes M] [--procs P] [--reps K] [--small] [--no-trace] [--arms "ENV=1|-"]
"""
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
SHAPES = [(3000, 8, None), (2500, 6, None), (300, 4, 0.001), (1200, 5, 0.01), (11700, 6, 0.001), (13500, 6, 0.001), (900, 4, 0.001),
          (17000, 10, None)]
if "--small" in sys.argv:     # many short fits (more fits per minute): the sizes of the multi-rank cases that failed
    SHAPES = [(3000, 8, None), (2500, 6, None), (300, 4, 0.001), (1200, 5, 0.01), (900, 4, 0.001), (600, 4, 0.001), (3000, 8, None),
              (11700, 6, 0.001)]


def arg(name, default):
    return type(default)(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


def worker():
    sys.path.insert(0, ROOT)
    import bigkrls_amd as bk
    from bigkrls_amd.synth import synth
    n, p, trunc, reps = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
    # (OVERSUB_OWN_STREAM=1: a stream of the context's own instead of the process's default (NULL) stream, on which
    #  every other Python-driven fit of this repository runs)
    ctx = bk.Context(0, own_stream=bool(os.environ.get("OVERSUB_OWN_STREAM")))
    X, y = synth(n, p, 103)
    kw = {} if trunc == "None" else dict(eigtrunc=float(trunc))
    if n == 17000:
        kw["Neig"] = 60
    lam = []
    for _ in range(reps):
        out = bk.bigKRLS(y, X, ctx=ctx, noisy=False, **kw)
        lam.append((out["lambda"], out["lastkeeper"], float(out["coeffs"][0])))
    ok = all(v == lam[0] for v in lam)
    # (the context's recovery counters tell a replay after a fired watchdog -- launch-per-step kernels, last-digit
    #  differences that eigtrunc = 0 amplifies into another kept-pair count, quirk Q7 -- from a silent fault)
    print(f"n={n} p={p}: {reps} fits, lambda/lastkeeper/c[0] {'identical' if ok else 'DIFFER: ' + repr(lam)} counters {ctx.counters()}", flush=True)
    sys.exit(0 if ok else 1)


def main():
    minutes, nprocs, reps = arg("--minutes", 10.0), arg("--procs", 24), arg("--reps", 4)
    # --arms "A=1;B=2|C=3|-": process i runs with the environment of arm i mod len(arms) ("-" = nothing set): same load,
    # same moment, the failures counted per arm
    arms = arg("--arms", "-").split("|")
    arm_fits = {a: 0 for a in arms}
    arm_bad = {a: 0 for a in arms}
    out = os.path.join(ROOT, "gpurun_out", "oversub_single")
    os.makedirs(out, exist_ok=True)
    t_end = time.time() + 60.0 * minutes
    fits = bad = 0
    r = 0
    while time.time() < t_end:
        procs = []
        for i in range(nprocs):
            n, p, trunc = SHAPES[i % len(SHAPES)]
            tdir = os.path.join(out, f"r{r}_p{i}_n{n}")
            shutil.rmtree(tdir, ignore_errors=True)
            os.makedirs(tdir)
            env = dict(os.environ, BIGKRLS_TRACE_DIR=tdir)
            if "--no-trace" in sys.argv:
                env.pop("BIGKRLS_TRACE_DIR")
            if (i // len(arms)) % 2 == 0:
                env.update(BIGKRLS_PQ="steps", BIGKRLS_BC="wavefront")
            arm = arms[i % len(arms)]
            if arm != "-":
                for kv in arm.split(";"):
                    k, v = kv.split("=")
                    env[k] = v
            log = open(os.path.join(tdir, "log.txt"), "w")
            procs.append((subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(n), str(p), str(trunc), str(reps)],
                                           stdout=log, stderr=subprocess.STDOUT, cwd=ROOT, env=env), log, tdir, arm))
        load = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C4", "--steps", "4", "--warmup", "1",
                                 "--no-cpu-baseline"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=ROOT)
        for p, log, tdir, arm in procs:
            rc = p.wait()
            log.close()
            fits += reps
            arm_fits[arm] += reps
            d = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_diff.py"), tdir, "--repeat", "--quiet"] if
                               "--no-trace" not in sys.argv else [sys.executable, "-c", "pass"], capture_output=True, text=True)
            caught = [ln for ln in open(os.path.join(tdir, "log.txt"), errors="replace") if "NONDETERMINISM" in ln]
            for ln in caught[:6]:
                print(f"round {r}: {os.path.basename(tdir)}: {ln.strip()}", flush=True)
            if rc != 0 or d.returncode != 0 or caught:
                bad += 1
                arm_bad[arm] += 1
                print(f"round {r}: {os.path.basename(tdir)} arm [{arm}] rc={rc} trace_diff={d.returncode}\n  " +
                      "\n  ".join(open(os.path.join(tdir, "log.txt")).read().splitlines()[-3:]) + "\n" + d.stdout, flush=True)
            else:
                shutil.rmtree(tdir, ignore_errors=True)
        load.wait()
        print(f"round {r} done: {fits} single-GPU fits so far, {bad} bad processes, {time.time() - (t_end - 60 * minutes):.0f} s", flush=True)
        r += 1
    print(f"single-GPU fits {fits}, bad processes {bad}")
    for a in arms:
        print(f"arm [{a}]: {arm_fits[a]} fits, {arm_bad[a]} processes with a wrong fit")


if __name__ == "__main__":
    if "--worker" in sys.argv:
        sys.argv.remove("--worker")
        sys.argv.insert(1, "--w")
        worker()
    else:
        main()
