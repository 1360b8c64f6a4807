#!/usr/bin/env python3
"""bench.py -- the bigKRLS() fit on MI355X, BASELINE.json's metric and configs.

    python bench.py --gpus N --steps K --warmup W [--config C2|C3|C4|C5]

A step = one full bigKRLS() fit (kernel -> eigen -> lambda search -> coefficients ->
variance matrices -> marginal effects). The default workload is configs[2] (C3), the
one BASELINE.json's metric is quoted on: N=20000, P=20, fp64, eigtrunc=0.001, synthetic
G(N,P,seed=103) inputs already resident in HBM when the timed region starts (the N x P
input is 3.2 MB; the PCIe-inclusive figure is in DESIGN.md). `--config` selects another
BASELINE.json configuration (C2 N=5000 P=10; C4 N=50000 P=20 Neig=512; C5 N=100000 P=50
Neig=1024 which.derivatives=c(1,3,5)); --n/--p/--seed/--neig/--eigtrunc/--which-derivatives
override single fields. Rank 0 prints ONE JSON line. For N > 1 the same fit is row-block
partitioned over the ranks (strong scaling): see bigkrls_amd/dist.py.

`roofline` is for the dominant kernel of the fit -- whichever of the profiled kernels takes
the most time on the critical path (dense path: the stage-1 band update / A22 V GEMMs, bulge
chasing, or the one-stage symv; Neig << N: the K B_j product of the block Lanczos) --
measured live with HIP events on the launch stream; the panel QR, which runs concurrently
on the look-ahead stream, is listed in `other_kernels`; `roofline.kernel_gemm` reports the
Gaussian-kernel GEMM the metric names (time, TFLOP/s, HBM-write GB/s). The line is compact
(< 5 KB: the driver keeps the tail of stdout); `--long-json` adds every entry's definition
(DESIGN.md section 5 has them).
`cpu_baseline` times the oracle's literal restatement of the reference on the host cores at
the bench size (oracle/cpu_baseline.py, a child process started before the GPU is touched
and released after the GPU timing; rank 0, N=1 only), bounded by --cpu-budget-s.
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
FP64_MFMA_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 32 FLOP/clk x 2.4 GHz (v_mfma_f64_16x16x4_f64: 64 cyc)

# BASELINE.json configs[1..4] (configs[0], C1, is the CPU-only plumbing case: tests/, not a bench line)
CONFIGS = {
    "C2": dict(n=5000, p=10, seed=102, neig=None, eigtrunc=None, which=None,
               desc="N=5000 P=10 fp64, full eigendecomp + all derivatives"),
    "C3": dict(n=20000, p=20, seed=103, neig=None, eigtrunc=None, which=None,
               desc="N=20000 P=20 fp64, eigtrunc=0.001"),
    "C4": dict(n=50000, p=20, seed=104, neig=512, eigtrunc=None, which=None,
               desc="N=50000 P=20 fp64, Neig=512"),
    "C5": dict(n=100000, p=50, seed=105, neig=1024, eigtrunc=None, which=[1, 3, 5],
               desc="N=100000 P=50 fp64, Neig=1024, which.derivatives subset"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="C3")
    ap.add_argument("--n", type=int, default=None)
    ap.add_argument("--p", type=int, default=None)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--neig", type=int, default=None)
    ap.add_argument("--eigtrunc", type=float, default=None)
    ap.add_argument("--which-derivatives", type=str, default=None,
                    help="comma-separated 1-based columns, e.g. 1,3,5")
    ap.add_argument("--cpu-n", type=int, default=None,
                    help="rows of the CPU-baseline sample (default: the bench N for C2/C3)")
    ap.add_argument("--cpu-budget-s", type=float, default=880.0,
                    help="wall-clock bound of the CPU baseline; what is measured until then is reported")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="route a single-GPU run through the row-block path (bigkrls_amd.dist) under an RCCL group "
                         "of size 1: exercises exactly the code of --gpus N > 1 on one GPU")
    ap.add_argument("--long-json", action="store_true",
                    help="every roofline / other_kernels entry with its definition (`note`) and per-launch figures")
    ap.add_argument("--dry-launch", action="store_true",
                    help="start the --gpus N rank processes, let them rendezvous over gloo on the CPU and build the "
                         "library's rank objects over host buffers -- no GPU, no fit: checks the launch path only")
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    custom = False
    for key, val in (("n", args.n), ("p", args.p), ("seed", args.seed), ("neig", args.neig),
                     ("eigtrunc", args.eigtrunc)):
        if val is not None:
            custom = custom or cfg[key] != val
            cfg[key] = val
    if args.which_derivatives is not None:
        cfg["which"] = [int(t) for t in args.which_derivatives.split(",") if t]
        custom = True
    cfg["name"] = args.config + ("*" if custom else "")
    args.cfg = cfg
    return args


def symv_traffic(alg_bytes_total, launches):
    """Average HBM bytes per symv launch: algorithmic bytes x the traffic/algorithmic ratio that
    rocprofv3 PMC passes measured for this kernel pair (profiles/r01_symv_pmc.json)."""
    path = os.path.join(ROOT, "profiles", "r01_symv_pmc.json")
    if launches <= 0 or not os.path.exists(path):
        return None
    try:
        ratio = float(json.load(open(path))["traffic_over_algorithmic"])
    except Exception:
        return None
    return round(alg_bytes_total / launches * ratio, 0)


def pmc_ratio(kernel):
    """traffic / algorithmic HBM bytes of a kernel from the rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE in separate passes, gfx950 correction and calibration as described in the files):
    profiles/r04/r04_traffic_pmc.json (the kernel build in round 4's tile order), profiles/r03/r03_traffic_pmc.json
    (the trailing update in round 3's tile order), then
    profiles/r02/r02_traffic_pmc.json (A22 V), else profiles/r01_traffic_pmc.json; keys matched by prefix."""
    for rel in (("profiles", "r04", "r04_traffic_pmc.json"), ("profiles", "r03", "r03_traffic_pmc.json"),
                ("profiles", "r02", "r02_traffic_pmc.json"),
                ("profiles", "r01_traffic_pmc.json")):
        try:
            d = json.load(open(os.path.join(ROOT, *rel)))
        except Exception:
            continue
        for key, val in d.items():
            if key.startswith(kernel) and isinstance(val, dict) and "traffic_over_algorithmic" in val:
                return float(val["traffic_over_algorithmic"])
    return None


class CpuBaseline:
    """oracle/cpu_baseline.py as a child process (kind "port": the oracle's literal restatement of
    the reference). Started before this process touches the GPU -- a process that has initialised
    the GPU must not spawn programs on the GPU boxes -- it waits on stdin until `release()` is called
    after the GPU timing, so the two never share the host cores. `collect()` waits until the child is
    done or the budget is spent (then the child, this exact PID, is killed) and assembles the
    `cpu_baseline` object from the phases measured until then."""

    def __init__(self, n, p, seed, eigtrunc, small_n=2000, budget_s=700.0):
        cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--n", str(n), "--p", str(p),
               "--seed", str(seed), "--small-n", str(small_n), "--budget-s", str(budget_s)]
        if eigtrunc is not None:
            cmd += ["--eigtrunc", str(eigtrunc)]
        self.n, self.p = n, p
        self.lines = []
        self.proc = subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, bufsize=1)
        self.reader = threading.Thread(target=self._read, daemon=True)
        self.reader.start()

    def _read(self):
        for line in self.proc.stdout:
            line = line.strip()
            if line.startswith("{"):
                try:
                    self.lines.append(json.loads(line))
                except ValueError:
                    pass

    def release(self):
        try:
            self.proc.stdin.write("go\n")
            self.proc.stdin.flush()
        except Exception:
            pass
        self.t_go = time.perf_counter()

    def abandon(self):
        if self.proc.poll() is None:
            self.proc.kill()

    def collect(self, budget_s):
        try:
            self.proc.wait(timeout=max(1.0, budget_s))
            timed_out = False
        except subprocess.TimeoutExpired:
            self.proc.kill()
            self.proc.wait()
            timed_out = True
        self.reader.join(timeout=5.0)
        ph = {d["phase"]: d for d in self.lines if "phase" in d}
        cores = ph.get("ready", {}).get("cores", os.cpu_count() or 1)
        host_cpus = ph.get("ready", {}).get("cpu_count", os.cpu_count() or 1)
        small = ph.get("small")
        res = {"unit": "s per fit", "cores": int(cores), "kind": "port"}
        lit_keys = ["kernel", "eigen", "lambda", "coeffs", "vcov_c", "vcov_fitted", "derivatives"]
        if "done" in ph:
            d = ph["done"]
            ex = [k for k in lit_keys if ph.get(k, {}).get("extrapolated")]
            # seconds of `value` that were scaled from timed samples rather than timed (the child reports them per phase)
            ex_s = sum(ph[k].get("extrapolated_s", ph[k]["s"]) for k in ex)
            lam_ph, der_ph = ph.get("lambda", {}), ph.get("derivatives", {})
            res.update({
                "value": d["literal_s"],
                "extrapolated": bool(ex),
                "efficient_port_s": d["efficient_s"],
                "sample": (f"literal restatement of the reference at N={self.n}, P={self.p}, {cores} BLAS threads on a "
                           f"{host_cpus}-CPU host (hand loops single-threaded like the reference); timed in full: kernel, "
                           f"dsyevd, V, V_yhat (4N^3), {lam_ph.get('probes_timed', 1)} of {d['probes']} solveforc probes, "
                           f"{der_ph.get('columns_timed', 1)} of {self.p} derivative columns (the rest at their mean: "
                           f"{100.0 * ex_s / max(d['literal_s'], 1e-9):.0f} % of value); efficient_port_s = the O(N^2 K) "
                           "identities, in full"),
                "phases_s": {k: ph[k]["s"] for k in lit_keys if k in ph},
                "extrapolated_phases": ex, "extrapolated_share": round(ex_s / max(d["literal_s"], 1e-9), 3),
                "host_cpus": int(host_cpus),
                "efficient_phases_s": d["efficient_phases_s"],
                "lastkeeper": d["lastkeeper"], "lambda": d["lam"],
            })
        else:
            got = {k: ph[k]["s"] for k in lit_keys if k in ph}
            res.update({
                "value": small["literal_s"] if small else None,
                "extrapolated": False,
                "efficient_port_s": small["efficient_s"] if small else None,
                "sample": (f"the N={self.n} sample did not finish within the {budget_s:.0f} s budget"
                           f"{' (killed)' if timed_out else ''}; phases measured at N={self.n} until then are in "
                           f"partial_phases_s; value = the full literal fit at N={small['n'] if small else '?'}, "
                           f"P={self.p} (BLAS threads = {cores})"),
                "partial_phases_s": got,
            })
        if small:
            res["small_sample"] = {"n": small["n"], "literal_s": small["literal_s"],
                                   "efficient_s": small["efficient_s"], "phases_s": small["phases_s"]}
        return res


def _short_kernel(name):
    """`kernel_name<...>` of an entry whose "kernel" field is `name: what it computes` (the long form is --long-json)."""
    return name.split(":")[0].split(" (")[0].strip()[:64]


_KEEP = ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "total_ms_per_fit", "launches_sampled",
         "launches", "concurrent", "tflops", "hbm_write_gbs", "fit_flops", "fit_frac")


def _compact_entry(e):
    out = {"kernel": _short_kernel(e["kernel"])}
    out.update({k: e[k] for k in _KEEP if k in e and e[k] is not None or k == "traffic" and k in e})
    if "parts" in e:
        out["parts"] = [{"kernel": _short_kernel(q["kernel"]), "frac": q["frac"], "achieved": q["achieved"],
                         "avg_launch_us": q["avg_launch_us"], "total_ms_per_fit": q["total_ms_per_fit"],
                         "traffic": q["traffic"]} for q in e["parts"]]
    return out


TRAFFIC_SOURCE = ("model: algorithmic bytes x the (FETCH_SIZE+WRITE_SIZE)/algorithmic ratio of separate rocprofv3 --pmc "
                  "passes (profiles/r03/, r02/, r04/ per launch shape; cross-checked over all in-fit launches: "
                  "profiles/r06/r06z_syrk_infit_traffic_pmc.json, 1.066), not a counter read in this run")


def compact_line(res):
    """The one JSON line in its short form (the driver keeps only the tail of stdout): every definition that used to
    travel as a `note` is in DESIGN.md section 5; `kernel_gemm` -- the GEMM half of BASELINE.json's metric -- sits
    inside `roofline`. `--long-json` prints the entries with their definitions."""
    out = {k: v for k, v in res.items() if k not in ("roofline", "other_kernels", "kernel_gemm", "cpu_baseline", "config")}
    cfg = dict(res["config"])
    out["config"] = cfg
    kg = res.get("kernel_gemm")
    if kg:
        kg = {k: v for k, v in kg.items() if k != "note"}
    # `roofline` always exists in the line, and `kernel_gemm` -- half of BASELINE.json's metric -- always with it (also
    # when no candidate kernel was sampled); a three-number copy stays at the top level for readers of older lines
    out["roofline"] = _compact_entry(res["roofline"]) if res.get("roofline") else {}
    out["roofline"]["kernel_gemm"] = kg
    if out["roofline"].get("traffic") is not None:
        out["roofline"]["traffic_source"] = res["roofline"].get("traffic_source", TRAFFIC_SOURCE)
    if kg:
        out["kernel_gemm"] = {k: kg[k] for k in ("tflops", "hbm_write_gbs", "ms") if k in kg}
    out["other_kernels"] = [_compact_entry(e) for e in res.get("other_kernels", [])]
    cb = res.get("cpu_baseline")
    if cb:
        cb = dict(cb)
        cb.pop("small_sample", None)
        cb.pop("extrapolated_phases", None)
        out["cpu_baseline"] = cb
    out["definitions"] = "DESIGN.md section 5"
    return out


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _visible_gpus():
    """Devices a rank process will see, counted by a short-lived child (this process never loads the HIP runtime)."""
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                           capture_output=True, text=True, timeout=600)
        return int(r.stdout.strip().splitlines()[-1])
    except Exception:
        return 0


def _share_gpu():
    """BIGKRLS_BENCH_SHARE_GPU=1 (tests only, with BIGKRLS_RCCL_LIB = tests/mock_rccl): all ranks on device 0, gloo for
    this program's own barrier -- the launcher, the rank processes and the library's RCCL code path on a 1-GPU box."""
    return os.environ.get("BIGKRLS_BENCH_SHARE_GPU") == "1"


def launch_ranks(args):
    """`python bench.py --gpus N` (N > 1) outside a launcher: this process starts the N ranks itself -- the reference's
    parallel path also starts its own workers (R/bigKRLS.R:340-343, makeCluster(Ncores)) -- as N children of THIS
    program with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set (what torch.distributed.run would set), relays rank 0's
    JSON line as its last line of stdout and exits non-zero if any rank fails. It never touches the GPU and never
    re-executes itself. Fewer visible devices than N is an error, not a silent one-GPU run."""
    n = args.gpus
    if not args.dry_launch and not _share_gpu():
        have = _visible_gpus()
        if have < n:
            print(f"bench.py: --gpus {n} but only {have} GPU(s) visible "
                  f"(HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES')}); not running on fewer", file=sys.stderr)
            return 2
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0",
                   BIGKRLS_BENCH_SELF_LAUNCHED="1")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0) or None))
    out0 = []
    reader = threading.Thread(target=lambda: out0.extend(procs[0].stdout), daemon=True)
    reader.start()
    # a rank that dies leaves its peers inside a collective: end them (these exact PIDs) as soon as one has failed
    failed = None
    live = set(range(n))
    while live and failed is None:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is not None:
                live.discard(r)
                if rc != 0:
                    failed = (r, rc)
        time.sleep(0.05)
    if failed is not None:
        t_end = time.time() + 15.0
        for r in sorted(live):
            while procs[r].poll() is None and time.time() < t_end:
                time.sleep(0.1)
            if procs[r].poll() is None:
                procs[r].kill()
            procs[r].wait()
    reader.join(timeout=10.0)
    lines = [ln.rstrip("\n") for ln in out0 if ln.strip()]
    json_line = next((ln for ln in reversed(lines) if ln.lstrip().startswith("{")), None)
    for ln in lines:
        if ln is not json_line:
            print(ln, file=sys.stderr)
    if failed is not None:
        print(f"bench.py: rank {failed[0]} exited with code {failed[1]}", file=sys.stderr)
        return 1
    if json_line is None:
        print("bench.py: rank 0 printed no result line", file=sys.stderr)
        return 1
    print(json_line, flush=True)
    return 0


def dry_rank(args, world, rank):
    """One rank of `--dry-launch`: rendezvous over gloo, the library's rank object over host buffers
    (bigkrls_comm_create_callbacks with no context), its rank count read back, one collective through it, and the
    rows this rank would own in the fit. No GPU is touched."""
    import ctypes as C
    import numpy as np
    import torch.distributed as dist
    from bigkrls_amd import _lib, dist as bkdist

    if os.environ.get("BIGKRLS_DRY_FAIL_RANK") == str(rank):      # tests: a rank that dies before the rendezvous
        return 3
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = bkdist.comm_callbacks(None)
    r, w = comm.rank_count()
    count = 3
    buf = np.zeros((4 + world) * count)
    buf[0:count] = 1.0
    _lib.call("bigkrls_comm_check", comm.handle, buf.ctypes.data_as(C.c_void_p), count)
    cfg = args.cfg
    opt = _lib.FitOptions()
    opt.struct_bytes = C.sizeof(_lib.FitOptions)
    opt.neig = cfg["neig"] or 0
    r0, r1 = C.c_int64(-1), C.c_int64(-1)
    _lib.call("bigkrls_fit_dist_rows", comm.handle, cfg["n"], C.byref(opt), C.byref(r0), C.byref(r1))
    rows = [None] * world
    dist.all_gather_object(rows, [int(r0.value), int(r1.value)])
    comm.close()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "comm_nranks": int(w), "comm_rank": int(r),
                          "ranks_in_allreduce": int(buf[0]), "rows": rows,
                          "self_launched": bool(os.environ.get("BIGKRLS_BENCH_SELF_LAUNCHED")),
                          "config": {"name": cfg["name"], "n": cfg["n"], "p": cfg["p"]}}), flush=True)
    return 0


def main():
    args = parse()
    cfg = args.cfg
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))           # before anything here imports torch or touches the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        # a launcher that started another number of ranks than --gpus says: refuse, a line with the wrong n_gpus is worse
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if args.dry_launch:
        sys.exit(dry_rank(args, world, rank))
    # ---- CPU-baseline child: started before anything here touches the GPU ------------------------
    cpu = None
    cpu_n = args.cpu_n if args.cpu_n is not None else (cfg["n"] if cfg["n"] <= 20000 else None)
    if world == 1 and rank == 0 and not args.no_cpu_baseline and cpu_n is not None:
        cpu = CpuBaseline(cpu_n, cfg["p"], cfg["seed"], cfg["eigtrunc"], budget_s=args.cpu_budget_s)

    import numpy as np
    import torch
    import torch.distributed as dist

    share = world > 1 and _share_gpu()
    if share:
        local_rank = 0
    if world > 1 and torch.cuda.device_count() <= local_rank:
        print(f"bench.py: rank {rank} has no GPU (LOCAL_RANK={local_rank}, {torch.cuda.device_count()} visible)",
              file=sys.stderr)
        sys.exit(2)
    if world > 1 or args.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29571")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if share:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)

    try:
        res = run(args, cfg, world, rank, local_rank, np, torch, dist)
    except BaseException:
        if cpu is not None:
            cpu.abandon()
        raise
    if rank == 0:
        if cpu is not None:
            cpu.release()
            res["cpu_baseline"] = cpu.collect(args.cpu_budget_s)
        elif world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = {
                "value": None, "unit": "s per fit", "cores": os.cpu_count(), "kind": "port",
                "sample": (f"not timed at N={cfg['n']}: the literal CPU path is infeasible at this size (dsyevd of a "
                           f"{8e-9 * cfg['n'] ** 2:.0f} GB matrix, hours; SURVEY.md section 8(d)); pass --cpu-n to time "
                           "the restatement on a smaller sample of the same generator")}
        # RCCL writes a version banner to the C-level stdout, which is flushed at exit: flush it now so
        # that the JSON line is the LAST line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(res if args.long_json else compact_line(res)), flush=True)
    if world > 1 or args.force_dist:
        from bigkrls_amd import dist as bkdist
        bkdist.release_comms()
        dist.destroy_process_group()


def run(args, cfg, world, rank, local_rank, np, torch, dist):
    import bigkrls_amd as bk
    from bigkrls_amd.synth import synth

    ctx = bk.Context(local_rank if world > 1 else 0)
    n, p = cfg["n"], cfg["p"]
    X, y = synth(n, p, cfg["seed"])
    fit_kw = dict(Neig=cfg["neig"], eigtrunc=cfg["eigtrunc"], which_derivatives=cfg["which"])

    if world > 1 or args.force_dist:
        from bigkrls_amd import dist as bkdist

        comm = bkdist.get_comm(ctx, "rccl")               # the library's own RCCL communicator, whatever torch's group is
        comm_rank, comm_nranks = comm.rank_count()        # read back from the library's rank object (bigkrls_comm_rank)
        if (comm_rank, comm_nranks) != (rank, world):
            raise RuntimeError(f"communicator says rank {comm_rank} of {comm_nranks}, launcher says {rank} of {world}")

        def one_fit(timings):
            # the same outputs as the single-GPU fit (the variance matrices are computed only when asked for): this
            # rank's column blocks of K, vcov.est.c, vcov.est.fitted
            return bkdist.bigKRLS_dist(y, X, ctx=ctx, comm=comm, timings=timings, keep_outputs=True, **fit_kw)
    else:
        comm_nranks = None

        def one_fit(timings):
            return bk.bigKRLS(y, X, ctx=ctx, timings=timings, **fit_kw)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = one_fit({})
        del out
    ctx.set_profile(True)
    phase_sum = {}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        T = {}
        out = one_fit(T)
        for k, v in T.items():
            phase_sum[k] = phase_sum.get(k, 0.0) + v
        lastkeeper, lam = out["lastkeeper"], out["lambda"]
        del out
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else ctx.device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    sec_per_fit = dt / args.steps

    prof = {name: ctx.get_profile(name) for name in
            ("symv", "kernel_block", "trailing_update", "band_update", "band_update2", "band_update4", "band_av", "bulge_chase",
             "panel_qr", "lanczos_kb", "lanczos_cgs2", "solveforc_probe", "deriv_rows", "yhat_gemv", "vcov_syrk")}
    ctx.set_profile(False)

    res = None
    if rank == 0:
        phases = {k: round(v / args.steps, 4) for k, v in phase_sum.items()}
        kb_ms, kb_flops, kb_n = prof["kernel_block"]

        STRIDE = 8   # the library brackets every 8th stage-1 panel (S1_PROF_STRIDE): totals are x8

        def mfma_entry(name, kernel, note, traffic=None, stride=STRIDE):
            ms, fl, cnt = prof[name]
            if ms <= 0:
                return None
            tf = fl / (ms / 1e3) / 1e12
            return {"kernel": kernel, "bound": "mfma", "achieved": round(tf, 3), "peak": FP64_MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(tf / FP64_MFMA_PEAK_TFLOPS, 4),
                    "traffic": traffic(fl / max(cnt, 1)) if traffic else None,
                    "launches_sampled": cnt, "avg_launch_us": round(ms * 1e3 / max(cnt, 1), 2),
                    "total_ms_per_fit": round(ms * stride / args.steps, 2),
                    "avg_algorithmic_flops_per_launch": round(fl / max(cnt, 1), 0),
                    "note": note + ("; every 8th panel is bracketed (total_ms_per_fit = 8 x the sampled time)"
                                    if stride == STRIDE else "")}

        def symv_entry():
            ms, by, cnt = prof["symv"]
            if ms <= 0:
                return None
            gbs = (by / 1e9) / (ms / 1e3)
            return {"kernel": "trd_symv_tiles(+trd_symv_reduce): Householder symv over the lower triangle of "
                              "the trailing matrix (one-stage tridiagonalisation, BIGKRLS_EIG=1stage)",
                    "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": symv_traffic(by, cnt),
                    "launches_sampled": cnt, "avg_launch_us": round(ms * 1e3 / max(cnt, 1), 2),
                    "total_ms_per_fit": round(ms * 8 / args.steps, 2),
                    "avg_algorithmic_bytes_per_launch": round(by / max(cnt, 1), 0),
                    "note": "achieved = 8*L(L+1)/2 algorithmic bytes per launch / HIP-event duration on the launch "
                            "stream over every 8th column; traffic = algorithmic x the PMC ratio of "
                            "profiles/r01_symv_pmc.json (FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE)"}

        def bulge_entry():
            ms, by, cnt = prof["bulge_chase"]
            if ms <= 0:
                return None
            gbs = (by / 1e9) / (ms / 1e3)
            resident = ms * 1e3 / max(cnt, 1) > 1000.0   # one persistent launch per decomposition vs sampled ~10 us wavefront launches
            if resident:
                return {"kernel": "bc_regwin: location-resident bulge chasing (stage 2, band b=64 -> tridiagonal), one "
                                  "persistent launch per fit, one 512-thread workgroup per band location with its 64 x 128 "
                                  "window in registers (BIGKRLS_BC=lds: bc_resident, the window in LDS)",
                        "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                        "launches": cnt, "avg_launch_us": round(ms * 1e3 / max(cnt, 1), 2),
                        "total_ms_per_fit": round(ms / args.steps, 2),
                        "avg_algorithmic_bytes_per_launch": round(by / max(cnt, 1), 0),
                        "note": "achieved = algorithmic HBM bytes (band read once, 16 N b, + stored reflectors, "
                                "4 N^2) / HIP-event duration. The band lives in registers for the whole stage; the kernel "
                                "is bound by the 2 message hops per sweep between neighbouring workgroups (~0.9 us each) "
                                "and the ~1.1 us the two locations compute between an arrival and their send (N sweeps x "
                                "~3.1 us), not by HBM"}
            return {"kernel": "bc_wavefront: one anti-diagonal wavefront of bulge-chasing tasks (stage 2 of the "
                              "two-stage tridiagonalisation, band b=64 -> tridiagonal), ~2N launches per fit",
                    "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                    "launches_sampled": cnt, "avg_launch_us": round(ms * 1e3 / max(cnt, 1), 2),
                    "total_ms_per_fit": round(ms * 64 / args.steps, 2),
                    "avg_algorithmic_bytes_per_launch": round(by / max(cnt, 1), 0),
                    "note": "achieved = algorithmic bytes per launch (each task reads+writes its 64x64 off-diagonal "
                            "block and the lower triangle of its 64x64 diagonal block: 98.8 KB per task, <= N/127 "
                            "tasks per launch) / HIP-event duration on the launch stream, every 64th launch sampled. "
                            "The band (16 N b bytes = 20 MB) stays in L2/MALL, so the kernel is bound by the "
                            "dependent-launch latency of the 2N-long wavefront chain, not by HBM bandwidth"}

        def panel_entry():
            ms, by, cnt = prof["panel_qr"]
            if ms <= 0:
                return None
            gbs = (by / 1e9) / (ms / 1e3)
            return {"kernel": "pq_chol (+ pq_resident as its fallback): register-resident factorisation of one m x 64 panel "
                              "(stage 1) by CholeskyQR2 + Householder reconstruction, one launch per panel on the "
                              "look-ahead stream, concurrent with syrk_mirror_kernel",
                    "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                    "launches_sampled": cnt, "avg_launch_us": round(ms * 1e3 / max(cnt, 1), 2),
                    "total_ms_per_fit": round(ms * STRIDE / args.steps, 2), "concurrent": True,
                    "avg_algorithmic_bytes_per_launch": round(by / max(cnt, 1), 0),
                    "note": "achieved = algorithmic bytes (panel read + written once, V written once: 24 m b) / "
                            "HIP-event duration on the look-ahead stream (both launches: pq_resident returns at once "
                            "unless pq_chol left the panel to it). The kernel is a chain of latencies -- two "
                            "all-reduces of a 64 x 64 Gram matrix and one broadcast between its workgroups, three "
                            "64 x 64 factorisations inside each -- not of bytes; beside the trailing update it starts "
                            "only where an update workgroup has ended (256 registers per lane, 107 KB of LDS)"}

        def hbm_entry(name, kernel, note, flops_per_byte=None):
            ms, by, cnt = prof[name]
            if ms <= 0:
                return None
            gbs = (by / 1e9) / (ms / 1e3)
            e = {"kernel": kernel, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None, "launches_sampled": cnt,
                 "avg_launch_us": round(ms * 1e3 / max(cnt, 1), 2), "total_ms_per_fit": round(ms / args.steps, 3),
                 "avg_algorithmic_bytes_per_launch": round(by / max(cnt, 1), 0), "note": note}
            if flops_per_byte:
                e["tflops"] = round(gbs * flops_per_byte / 1e3, 2)
            return e

        n_probes_note = ("one probe of the golden-section search per launch; the launch first folds the previous probe's "
                         "loss and takes R's branch on the device (no read-back inside the search)")
        extra = [
            hbm_entry("solveforc_probe", "sf_probe_kernel: one solveforc probe, c_i = sum_k Q_ik a_k/(d_k+lambda), "
                      "g_i = sum_k Q_ik^2/(d_k+lambda), Le = sum (c_i/g_i)^2 (src/solveforc.cpp:13-65 in O(NK))",
                      "achieved = 8 N K algorithmic bytes (Q read once per probe, K = lastkeeper) / HIP-event duration of "
                      "one sampled probe launch; " + n_probes_note + "; Q (8 N K = "
                      f"{8e-6 * n * int(lastkeeper):.0f} MB) stays in the Infinity Cache between probes", 0.5),
            hbm_entry("deriv_rows", "deriv_rows: K [1, c, x_j, x_j o c | b_j, b_j o c] for all columns as one skinny fp64 "
                      "MFMA GEMM (gemm_kernel<N,N,64>) + O(NP) finalize (src/bigderiv_v3.cpp:13-111 in O(N^2))",
                      "achieved = 8 N^2 algorithmic bytes (K read once for ALL columns) / HIP-event duration of the whole "
                      "pass (column min/max, operand build, GEMM, finalize)",
                      (2.0 + 2.0 * (p if cfg["which"] is None else len(cfg["which"]))) / 4.0),
            hbm_entry("yhat_gemv", "gemv_n: yfitted = K c over the full kernel (R/bigKRLS.R:291)",
                      "achieved = 8 N^2 algorithmic bytes / HIP-event duration", 0.25),
        ]
        vms, vfl, vcnt = prof["vcov_syrk"]
        if vms > 0:
            vtf = vfl / (vms / 1e3) / 1e12
            extra.append({"kernel": "syrk_mirror_kernel<128,false>: vcov.est.c / vcov.est.fitted = Q diag(w) Q' on the lower "
                                    "tile triangle, stored twice (R/bigKRLS.R:299-307)",
                          "bound": "mfma", "achieved": round(vtf, 3), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                          "frac": round(vtf / FP64_MFMA_PEAK_TFLOPS, 4), "traffic": None, "launches_sampled": vcnt,
                          "avg_launch_us": round(vms * 1e3 / max(vcnt, 1), 2),
                          "total_ms_per_fit": round(vms / args.steps, 3),
                          "hbm_write_gbs": round(8.0 * n * n / 1e9 / (vms / max(vcnt, 1) / 1e3), 1),
                          "note": "achieved = N (N+1) K algorithmic flops per matrix (K = lastkeeper; the lower tiles are "
                                  "computed, the mirror is stored) / HIP-event duration; at small K the launch is bound by "
                                  "its 8 N^2 bytes of stores (hbm_write_gbs)"})
        extra = [e for e in extra if e]

        cands = [
            symv_entry(),
            bulge_entry(),
            panel_entry(),
            mfma_entry("band_update", "syrk_mirror_kernel<64> at k = 128: A22 -= [V Z][Z V]' on the lower tile triangle + mirrored "
                       "store (stage 1 of the two-stage tridiagonalisation), one launch per 64-column panel",
                       "achieved = m(m+1)*2b algorithmic flops per launch (lower triangle incl. diagonal tiles, "
                       "b = 64, m = trailing size minus the next panel's 64 columns, which the preceding fused "
                       "kernel s1_fused_z updates) / HIP-event duration on the launch stream; traffic = algorithmic "
                       "HBM bytes of the launch (read 4 m^2 + write 8 m^2) x the FETCH_SIZE+WRITE_SIZE / algorithmic "
                       "ratio of profiles/r03/r03_traffic_pmc.json for the 128x64 tiles in the XCD-aware band-major order (1.002; 1.149 in "
                       "round 2's column order)",
                       traffic=(lambda flops: round(12.0 * flops / 128.0 * pmc_ratio("syrk_mirror_kernel<64> k=128"), 0))
                       if pmc_ratio("syrk_mirror_kernel<64> k=128") else None),
            mfma_entry("band_update2", "syrk_mirror_kernel<64> at k = 256: A22 -= [V Z V Z][Z V Z V]' for a GROUP of two "
                       "panels, applied as two pieces of equal area (the columns right / left of a cut), one per panel "
                       "step, each concurrent with the next panel's QR (stage 1 while the trailing matrix has 10752 .. 12800 rows)",
                       "achieved = algorithmic flops of the piece's columns of the lower triangle, 2*256*sum_c (m - c), / "
                       "HIP-event duration on the launch stream; traffic = algorithmic HBM bytes of the piece (read 4 + write 8 "
                       "bytes per lower-triangle entry, mirrored: 12 bytes per 256 flops) x the FETCH_SIZE+WRITE_SIZE / "
                       "algorithmic ratio measured for the two pieces (profiles/r03/r03_traffic_pmc.json, 1.014)",
                       traffic=(lambda flops: round(12.0 * flops / 256.0 * pmc_ratio("syrk_mirror_kernel<64> k=256, the two"), 0))
                       if pmc_ratio("syrk_mirror_kernel<64> k=256, the two") else None),
            mfma_entry("band_update4", "syrk_mirror_kernel<64> at k = 512: A22 -= U U'^T for a GROUP of four panels, the "
                       "whole lower tile triangle in one launch at the end of the group's fourth step, concurrent with the "
                       "next panel's factorisation (stage 1 while the trailing matrix has >= 12800 rows)",
                       "achieved = 512 m^2 algorithmic flops per launch (lower triangle) / HIP-event duration on the launch "
                       "stream, every second group bracketed (total_ms_per_fit = 2 x the sampled time); traffic = algorithmic "
                       "HBM bytes (12 bytes per 512 flops) x the whole-triangle k = 512 ratio of profiles/r05/r05u_syrk_traffic_pmc.json (1.146; 1.145 in round 3)",
                       traffic=(lambda flops: round(12.0 * flops / 512.0 * 1.146, 0)),
                       stride=2),
            mfma_entry("band_av", "gemm_kernel<N,N,64>: Y = A22 V (stage 1), one launch per panel",
                       "achieved = 2 m^2 b flops per launch (b = 64) / HIP-event duration; traffic = algorithmic HBM bytes of "
                       "the launch (A22 read once, 8 m^2, + V and Y, 16 m b: 1/16 byte per flop) x the ratio of "
                       "profiles/r02/r02_traffic_pmc.json (1.07: FETCH_SIZE doubled per the gfx950 correction for its "
                       "16-B-per-lane operand loads, + WRITE_SIZE, + the split-K reduction)",
                       traffic=(lambda flops: round((flops / 16.0 + 16.0 * (flops / 128.0) ** 0.5 * 64.0)
                                                    * pmc_ratio("gemm_kernel<N,N,64>"), 0))
                       if pmc_ratio("gemm_kernel<N,N,64>") else None),
            mfma_entry("lanczos_kb", "gemm_kernel<N,N,128>: W = K B_j, the N x N x 128 product of one block-Lanczos "
                       "step (Neig << N, the reference's eigs_sym branch src/eigen.cpp:18-22)",
                       "achieved = 2 N^2 b flops (b = 128) per step / HIP-event duration on the launch stream, every "
                       "step bracketed; reads K once per step (8 N^2 B, AI = 32 flop/B)", stride=1),
            mfma_entry("lanczos_cgs2", "gemm_kernel<T,N> + gemm_kernel<N,N>: classical Gram-Schmidt, first against the last "
                       "two blocks, then against all of them (C = B'W, W -= B C), four skinny GEMMs per step",
                       "achieved = 4 N (dim + 256) b flops per step / HIP-event duration, every step bracketed", stride=1),
        ]
        cands = [c for c in cands if c]
        # The two trailing-update entries are launches of ONE kernel (syrk_mirror_kernel<64>, one row in a rocprofv3
        # listing) at two ranks: it competes for "dominant kernel" with its total time, at its blended rate.
        upd = [c for c in cands if c["kernel"].startswith("syrk_mirror_kernel")]
        if len(upd) >= 2:
            ms = [c["total_ms_per_fit"] for c in upd]
            fl = [c["achieved"] * c["total_ms_per_fit"] for c in upd]          # TFLOP/s x ms
            ln = [c["launches_sampled"] for c in upd]
            tf = sum(fl) / sum(ms)
            tr = ([c["traffic"] * c["launches_sampled"] for c in upd] if all(c["traffic"] for c in upd) else None)
            merged = {"kernel": "syrk_mirror_kernel<64>: the stage-1 trailing update A22 -= [V Z][Z V]' (lower tile "
                                "triangle + mirrored store): k = 512 for groups of four panels while the trailing matrix "
                                "has >= 12800 rows, k = 256 pieces for groups of two down to 10752, k = 128 per panel below",
                      "bound": "mfma", "achieved": round(tf, 3), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                      "frac": round(tf / FP64_MFMA_PEAK_TFLOPS, 4),
                      "traffic": round(sum(tr) / sum(ln), 0) if tr else None,
                      "launches_sampled": sum(ln),
                      "avg_launch_us": round(sum(c["avg_launch_us"] * c["launches_sampled"] for c in upd) / sum(ln), 2),
                      "total_ms_per_fit": round(sum(ms), 2),
                      "avg_algorithmic_flops_per_launch": round(
                          sum(c["avg_algorithmic_flops_per_launch"] * c["launches_sampled"] for c in upd) / sum(ln), 0),
                      "note": "time-weighted over the kernel's launch shapes (see `parts`); achieved = algorithmic "
                              "flops / HIP-event duration on the launch stream, sampled launches (see the parts); traffic = "
                              "launch-weighted mean of the parts' figures",
                      "parts": upd}
            cands = [c for c in cands if c not in upd] + [merged]
        # The dominant kernel is the one with the most time on the critical path: pq_resident runs on
        # the look-ahead stream concurrently with (and hidden behind) syrk_mirror_kernel, so its kernel
        # time -- the largest in a rocprofv3 listing -- is not wall time; it is reported in other_kernels.
        crit = [c for c in cands if not c.get("concurrent")]
        roof = max(crit or cands, key=lambda c: c["total_ms_per_fit"]) if cands else None
        tu_ms, tu_flops, tu_n = prof["trailing_update"]
        # fit-level roofline: the flops the eigensolver cannot avoid / wall-clock of the WHOLE fit / fp64 MFMA peak.
        # Dense path: tridiagonalisation 4N^3/3 + the two back-transforms of the kept columns 2 * 2 N^2 lastkeeper.
        # Neig << N: the K B_j products and the re-orthogonalisation actually issued (from the library's counters).
        if roof is not None:
            if prof["lanczos_kb"][0] > 0:
                fit_flops = (prof["lanczos_kb"][1] + prof["lanczos_cgs2"][1]) / args.steps
                fit_flops_note = "sum of the K B_j and CGS2 flops of the block Lanczos (library counters)"
            else:
                fit_flops = 4.0 * n ** 3 / 3.0 + 2.0 * n * n * float(lastkeeper) * 2.0
                fit_flops_note = "4N^3/3 (tridiagonalisation) + 2 * 2 N^2 lastkeeper (two back-transforms of the kept columns)"
            roof = dict(roof)
            roof["fit_flops"] = fit_flops
            roof["fit_frac"] = round(fit_flops / sec_per_fit / (FP64_MFMA_PEAK_TFLOPS * 1e12), 4)
            roof["fit_frac_note"] = ("fit_frac = fit_flops / value / fp64 MFMA peak, fit_flops = " + fit_flops_note +
                                     "; the whole fit's wall-clock incl. everything that runs no MFMA")
        res = {
            # BASELINE.json: "bigKRLS() fit wall-clock + kernel-GEMM fp64 GFLOP/s at N=20000,P=20";
            # `value` is the wall-clock half, `kernel_gemm.gflops` the GEMM half
            "metric": f"bigKRLS() fit wall-clock at N={n},P={p} (kernel-GEMM fp64 GFLOP/s: roofline.kernel_gemm.gflops)",
            "value": round(sec_per_fit, 4),
            "unit": "s",
            "n_gpus": world,
            "comm_nranks": comm_nranks,      # bigkrls_comm_rank of the RCCL communicator the fit ran on (null: bigkrls_fit)
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(sec_per_fit * 1e3, 2),
            "higher_is_better": False,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": (f"{cfg['name']} ({cfg['desc']}): bigKRLS() full fit, N={n}, P={p}, fp64, "
                                    f"Neig={'N' if cfg['neig'] is None else cfg['neig']}, "
                                    f"eigtrunc={cfg['eigtrunc'] if cfg['eigtrunc'] is not None else (0.001 if n > 3000 else 0)}, "
                                    f"{'all derivatives' if cfg['which'] is None else 'which.derivatives=' + str(cfg['which'])}, "
                                    "vcov.est=TRUE; synthetic G(N,P,seed)"),
                       "name": cfg["name"], "n": n, "p": p, "seed": cfg["seed"], "neig": cfg["neig"],
                       "which_derivatives": cfg["which"], "lastkeeper": int(lastkeeper),
                       "lambda": float(lam),
                       "parallelism": ("1 GPU" if world == 1 and not args.force_dist
                                       else f"row-block x{world}, RCCL broadcast / all-gather")},
            "phases_s": phases,
            "kernel_gemm": {
                "gflops": round(kb_flops / (kb_ms / 1e3) / 1e9, 1) if kb_ms > 0 else None,
                "tflops": round(kb_flops / (kb_ms / 1e3) / 1e12, 3) if kb_ms > 0 else None,
                "frac_of_fp64_mfma_peak": round(kb_flops / (kb_ms / 1e3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4) if kb_ms > 0 else None,
                "hbm_write_gbs": round(8.0 * (kb_flops / (2.0 * p)) / 1e9 / (kb_ms / 1e3), 1) if kb_ms > 0 else None,
                "frac_of_hbm_peak": round(8.0 * (kb_flops / (2.0 * p)) / 1e9 / (kb_ms / 1e3) / HBM_PEAK_GBS, 4) if kb_ms > 0 else None,
                "ms": round(kb_ms / max(kb_n, 1), 4), "launches": kb_n,
                "traffic": (round(8.0 * n * n * pmc_ratio("kernel_block_sym_kernel"), 0)
                            if pmc_ratio("kernel_block_sym_kernel") else None),
                "note": "kernel_block_sym_kernel: 2*N^2*P algorithmic flops per launch (the symmetric variant "
                        "executes half of them and mirrors); for P <~ 40 the build is HBM-write bound (8 N^2 bytes, "
                        "AI = P/4 flop/B), so the binding roofline is hbm_write_gbs / 8000"},
            "roofline": roof,
            "other_kernels": [c for c in cands if c["kernel"] != (roof or {}).get("kernel")] + extra,
        }
        if tu_ms > 0:
            res["trailing_update"] = {"tflops": round(tu_flops / (tu_ms / 1e3) / 1e12, 3), "launches": tu_n}
    return res


if __name__ == "__main__":
    main()
