#!/usr/bin/env python3
"""bench.py -- the bigKRLS() fit on MI355X, BASELINE.json's metric and config.

    python bench.py --gpus N --steps K --warmup W

A step = one full bigKRLS() fit (kernel -> eigen -> lambda search -> coefficients ->
variance matrices -> marginal effects) of configs[2]: N=20000, P=20, fp64,
eigtrunc=0.001, synthetic G(N,P,seed=103) inputs already resident in HBM when the
timed region starts (the N x P input is 3.2 MB; the PCIe-inclusive figure is in
DESIGN.md).  Rank 0 prints ONE JSON line.  For N > 1 the same fit is row-block
partitioned over the ranks (strong scaling): see bigkrls_amd/dist.py.

`roofline` is for the dominant kernel of the fit -- whichever of the profiled
eigensolver kernels (the stage-1 band update / A22 V GEMMs, bulge chasing, or the
one-stage symv) takes the most time on the critical path -- measured live with HIP
events on the launch stream; the panel QR, which runs concurrently on the look-ahead
stream, is listed in `other_kernels`; `kernel_gemm` reports the Gaussian-kernel GEMM
the metric names.
`cpu_baseline` times the oracle's literal restatement of the reference on the
host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
FP64_MFMA_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 32 FLOP/clk x 2.4 GHz (v_mfma_f64_16x16x4_f64: 64 cyc)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=20000)
    ap.add_argument("--p", type=int, default=20)
    ap.add_argument("--seed", type=int, default=103)
    ap.add_argument("--cpu-n", type=int, default=2000, help="rows of the bounded CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


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
    """traffic / algorithmic HBM bytes of a kernel from the rocprofv3 PMC passes in
    profiles/r01_traffic_pmc.json (FETCH_SIZE + WRITE_SIZE, separate passes, calibrated there)."""
    path = os.path.join(ROOT, "profiles", "r01_traffic_pmc.json")
    try:
        return float(json.load(open(path))[kernel]["traffic_over_algorithmic"])
    except Exception:
        return None


def cpu_baseline(p, n_cpu, seed):
    """Literal CPU restatement (oracle, kind 'port') of the same fit on a bounded sample."""
    import numpy as np
    from oracle import krls_oracle as orc
    try:
        from threadpoolctl import threadpool_info
        threads = max([d.get("num_threads", 1) for d in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    X, y = orc.synth(n_cpu, p, seed)
    T = {}
    t0 = time.perf_counter()
    orc.fit(y, X, literal=True, timings=T, return_squares=False)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    orc.fit(y, X, literal=False, return_squares=False)
    dt_fast = time.perf_counter() - t0
    return {
        "value": round(dt, 3), "unit": "s per fit", "cores": int(threads), "kind": "port",
        # the same fit with the O(N^2 K) identities the HIP path uses (so that the speed-up is not
        # inflated by the reference's avoidable N^3 terms), same host, same sample
        "efficient_port_s": round(dt_fast, 3),
        "sample": (f"full literal fit (reference loop structure: N^2K/2-per-probe solveforc, 4N^3 V_yhat, "
                   f"4N^3-per-column derivatives; LAPACK dsyevd/BLAS via scipy OpenBLAS) at N={n_cpu}, "
                   f"P={p}: 1/{(20000 // n_cpu) ** 3} of the N^3 work of the N=20000 workload; "
                   f"hand loops single-threaded like the reference, BLAS threads={threads}"),
        "phases_s": {k: round(v, 3) for k, v in T.items()},
    }


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)

    import bigkrls_amd as bk
    from bigkrls_amd.synth import synth

    ctx = bk.Context(local_rank if world > 1 else 0)
    X, y = synth(args.n, args.p, args.seed)

    if world > 1:
        from bigkrls_amd import dist as bkdist

        def one_fit(timings):
            return bkdist.bigKRLS_dist(y, X, ctx=ctx, timings=timings, keep_outputs=False)
    else:
        def one_fit(timings):
            return bk.bigKRLS(y, X, ctx=ctx, timings=timings)

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
        tmax = torch.tensor([dt], dtype=torch.float64, device=ctx.device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    sec_per_fit = dt / args.steps

    prof = {name: ctx.get_profile(name) for name in
            ("symv", "kernel_block", "trailing_update", "band_update", "band_av", "bulge_chase", "panel_qr")}
    ctx.set_profile(False)

    if rank == 0:
        n, p = args.n, args.p
        phases = {k: round(v / args.steps, 4) for k, v in phase_sum.items()}
        kb_ms, kb_flops, kb_n = prof["kernel_block"]

        STRIDE = 8   # the library brackets every 8th stage-1 panel (S1_PROF_STRIDE): totals are x8

        def mfma_entry(name, kernel, note, traffic=None):
            ms, fl, cnt = prof[name]
            if ms <= 0:
                return None
            tf = fl / (ms / 1e3) / 1e12
            return {"kernel": kernel, "bound": "mfma", "achieved": round(tf, 3), "peak": FP64_MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(tf / FP64_MFMA_PEAK_TFLOPS, 4),
                    "traffic": traffic(fl / max(cnt, 1)) if traffic else None,
                    "launches_sampled": cnt, "avg_launch_us": round(ms * 1e3 / max(cnt, 1), 2),
                    "total_ms_per_fit": round(ms * STRIDE / args.steps, 2),
                    "avg_algorithmic_flops_per_launch": round(fl / max(cnt, 1), 0),
                    "note": note + "; every 8th panel is bracketed (total_ms_per_fit = 8 x the sampled time)"}

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
            resident = cnt <= args.steps          # one persistent launch per fit vs sampled wavefront launches
            if resident:
                return {"kernel": "bc_resident: LDS-resident bulge chasing (stage 2, band b=64 -> tridiagonal), one "
                                  "persistent launch per fit, one workgroup per band location",
                        "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                        "launches": cnt, "avg_launch_us": round(ms * 1e3 / max(cnt, 1), 2),
                        "total_ms_per_fit": round(ms / args.steps, 2),
                        "avg_algorithmic_bytes_per_launch": round(by / max(cnt, 1), 0),
                        "note": "achieved = algorithmic HBM bytes (band read once, 16 N b, + stored reflectors, "
                                "4 N^2) / HIP-event duration. The band lives in LDS for the whole stage; the kernel "
                                "is bound by the 2 message hops per sweep between neighbouring workgroups "
                                "(N sweeps x ~6 us), not by HBM"}
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
            return {"kernel": "pq_resident: register-resident Householder QR of one m x 64 panel (stage 1), one launch "
                              "per panel on the look-ahead stream, concurrent with syrk_mirror_kernel",
                    "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                    "launches_sampled": cnt, "avg_launch_us": round(ms * 1e3 / max(cnt, 1), 2),
                    "total_ms_per_fit": round(ms * STRIDE / args.steps, 2), "concurrent": True,
                    "avg_algorithmic_bytes_per_launch": round(by / max(cnt, 1), 0),
                    "note": "achieved = algorithmic bytes (panel read + written once, V written once: 24 m b) / "
                            "HIP-event duration on the look-ahead stream. The kernel is bound by 64 dependent "
                            "all-to-all exchanges of partial sums between its workgroups (~6-11 us each), and its "
                            "duration is hidden behind the trailing update it runs concurrently with"}

        cands = [
            symv_entry(),
            bulge_entry(),
            panel_entry(),
            mfma_entry("band_update", "syrk_mirror_kernel: A22 -= [V Z][Z V]' on the lower tile triangle + mirrored "
                       "store (stage 1 of the two-stage tridiagonalisation), one launch per 64-column panel",
                       "achieved = m(m+1)*2b algorithmic flops per launch (lower triangle incl. diagonal tiles, "
                       "b = 64, m = trailing size minus the next panel's 64 columns, which the preceding fused "
                       "kernel s1_fused_z updates) / HIP-event duration on the launch stream; traffic = algorithmic "
                       "HBM bytes of the launch (read 4 m^2 + write 8 m^2) x the FETCH_SIZE+WRITE_SIZE / algorithmic "
                       "ratio of profiles/r01_traffic_pmc.json (1.011)",
                       traffic=(lambda flops: round(12.0 * flops / 128.0 * pmc_ratio("syrk_mirror_kernel"), 0))
                       if pmc_ratio("syrk_mirror_kernel") else None),
            mfma_entry("band_av", "gemm_kernel<N,N,64>: Y = A22 V (stage 1), one launch per panel",
                       "achieved = 2 m^2 b flops per launch / HIP-event duration"),
        ]
        cands = [c for c in cands if c]
        # The dominant kernel is the one with the most time on the critical path: pq_resident runs on
        # the look-ahead stream concurrently with (and hidden behind) syrk_mirror_kernel, so its kernel
        # time -- the largest in a rocprofv3 listing -- is not wall time; it is reported in other_kernels.
        crit = [c for c in cands if not c.get("concurrent")]
        roof = max(crit or cands, key=lambda c: c["total_ms_per_fit"]) if cands else None
        tu_ms, tu_flops, tu_n = prof["trailing_update"]
        res = {
            # BASELINE.json: "bigKRLS() fit wall-clock + kernel-GEMM fp64 GFLOP/s at N=20000,P=20";
            # `value` is the wall-clock half, `kernel_gemm.gflops` the GEMM half
            "metric": f"bigKRLS() fit wall-clock at N={n},P={p} (kernel-GEMM fp64 GFLOP/s in kernel_gemm.gflops)",
            "value": round(sec_per_fit, 4),
            "unit": "s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(sec_per_fit * 1e3, 2),
            "higher_is_better": False,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"C3: bigKRLS() full fit, N={n}, P={p}, fp64, Neig=N, eigtrunc=0.001, "
                                   "all derivatives, vcov.est=TRUE; G(N,P,seed) = sin(X beta)+0.25 eps",
                       "n": n, "p": p, "seed": args.seed, "lastkeeper": int(lastkeeper),
                       "lambda": float(lam),
                       "parallelism": "1 GPU" if world == 1 else f"row-block x{world}, RCCL all-gather"},
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
                        "executes half of them and mirrors); at P=20 the build is HBM-write bound (8 N^2 bytes, "
                        "AI = P/4 flop/B), so the binding roofline is hbm_write_gbs / 8000"},
            "roofline": roof,
            "other_kernels": [c for c in cands if c is not roof],
        }
        if tu_ms > 0:
            res["trailing_update"] = {"tflops": round(tu_flops / (tu_ms / 1e3) / 1e12, 3), "launches": tu_n}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(p, args.cpu_n, args.seed)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
