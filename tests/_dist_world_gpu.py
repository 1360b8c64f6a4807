"""bigkrls_fit_dist at WORLD_SIZE > 1 on ONE GPU.

RCCL refuses two ranks on one device, and this pool hands out single-GPU boxes, so the multi-rank code of the
library (column blocks with offsets, the partitioned stage 1, the split back-transform, the sharded block Lanczos,
the row-block lambda search) is driven here through its callback table (bigkrls_comm_create_callbacks): every
rank is its own process with its own context on device 0, the collectives stage the device buffers through host
memory and a gloo group (bigkrls_amd.dist.comm_callbacks; test only, the product path uses RCCL).
Checks every rank's result -- and its column blocks of K and vcov.est.c -- against the single-process fit.

    python tests/_dist_world_gpu.py [N] [P] [WORLD] [--krylov NEIG] [--rccl-mock]

--rccl-mock: the communicator is built the way the product builds it -- bigkrls_comm_unique_id on rank 0, the id
broadcast over the process group, bigkrls_comm_create: csrc/dist.hip's dlopen'd function table and its stream-ordered,
asynchronous collectives -- with BIGKRLS_RCCL_LIB pointing at tests/mock_rccl/libmock_rccl.so, a stand-in for librccl
that accepts several ranks on one device (shared memory between the rank processes, same stream semantics).

The launcher never touches the GPU: it only starts the rank processes. tests/conftest.py starts the launchers at
session start (before the pytest process itself initialises the GPU) and tests/test_gpu_dist_world.py collects them.
"""
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


PORT_TAKEN = 97      # exit code of a rank whose rendezvous port was taken between the launcher's probe and its own bind


def launcher():
    import socket
    import time
    args = [a for a in sys.argv[1:]]
    pos = [a for a in args if a.isdigit()][:3]
    world = int(pos[2]) if len(pos) > 2 else 2
    for attempt in range(4):
        # a port of its own (a scheme by PID collides when thread IDs -- the same number space -- push the launchers' PIDs
        # a multiple of the table size apart). Probing for a free port and binding it later is a race with every other
        # launcher and with gloo's own ephemeral connections: when rank 0 finds the port taken (before any library code
        # has run) the rendezvous -- only the rendezvous -- is started again on another port.
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=port, BIGKRLS_DIST_WORKER="1")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + args, env=env))
        # a rank that dies leaves its peers inside a collective (or inside the rendezvous) for ever: once one has failed,
        # the others get a minute and are then ended (these exact PIDs) -- a failure, never a hang
        codes = [None] * world
        t_fail = None
        while any(c is None for c in codes):
            for i, p in enumerate(procs):
                if codes[i] is None:
                    codes[i] = p.poll()
            if t_fail is None and any(c not in (None, 0) for c in codes):
                t_fail = time.time()
            if t_fail is not None and (PORT_TAKEN in codes or time.time() - t_fail > 60.0):
                for i, p in enumerate(procs):
                    if codes[i] is None:
                        p.kill()
                        codes[i] = p.wait() or 1
                break
            time.sleep(0.05)
        if PORT_TAKEN in codes and attempt < 3:
            print(f"[launcher] rendezvous port {port} was taken before rank 0 could bind it; starting the ranks again", flush=True)
            continue
        rc = 0
        for c in codes:
            rc |= (c if c is not None else 1)
        sys.exit(1 if rc else 0)


def worker():
    selftest = os.environ.get("BIGKRLS_LAUNCHER_SELFTEST")
    if selftest:
        # CPU self-test of the launcher (tests/test_world_supervisor_cpu.py): "port": rank 0 finds its port taken once;
        # "dead": rank 1 dies and rank 0 would wait for it for ever
        import time
        rank = int(os.environ["RANK"])
        marker = os.environ["BIGKRLS_LAUNCHER_MARKER"]
        if selftest == "port":
            if rank == 0 and not os.path.exists(marker):
                open(marker, "w").close()
                sys.exit(PORT_TAKEN)
            time.sleep(0.3 if os.path.exists(marker) else 30.0)
            sys.exit(0)
        if rank == 1:
            sys.exit(3)
        time.sleep(600.0)
    sys.path.insert(0, ROOT)
    import time
    import numpy as np
    import torch
    import torch.distributed as dist
    import bigkrls_amd as bk
    from bigkrls_amd import dist as bkdist
    from bigkrls_amd.synth import synth

    pos = [a for a in sys.argv[1:] if not a.startswith("--") and a.isdigit()][:3]
    n = int(pos[0]) if pos else 3000
    p = int(pos[1]) if len(pos) > 1 else 8
    neig = None
    if "--krylov" in sys.argv:
        neig = int(sys.argv[sys.argv.index("--krylov") + 1])
    # (n <= 3000 defaults to eigtrunc = 0, where the cut is decided by the signs of round-off-level eigenvalues:
    #  quirk Q7; the small cases pass a threshold)
    trunc = float(sys.argv[sys.argv.index("--eigtrunc") + 1]) if "--eigtrunc" in sys.argv else None
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    fault_rank = int(sys.argv[sys.argv.index("--fault-rank") + 1]) if "--fault-rank" in sys.argv else None
    watchdog_rank = int(sys.argv[sys.argv.index("--watchdog-rank") + 1]) if "--watchdog-rank" in sys.argv else None
    if watchdog_rank is not None:
        # the test build again: ONE rank reports a fired persistent-kernel watchdog after the distributed stage 1; all
        # ranks must agree on it, replay the decomposition with the per-step kernels and finish with the right answer
        import bigkrls_amd._lib as L
        L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "capi", "libbigkrls_hip_fault.so")
        if rank == watchdog_rank:
            os.environ["BIGKRLS_FAULT"] = "watchdog"
        os.environ["BIGKRLS_VERBOSE"] = "1"
    garbage_rank = int(sys.argv[sys.argv.index("--garbage-rank") + 1]) if "--garbage-rank" in sys.argv else None
    if garbage_rank is not None:
        # the test build once more: ONE rank's slice of the eigenvectors comes back wrong WITHOUT an error (its last kept
        # column scaled by 1.001). Every rank checks the decomposition against its own rows of K, the failure is agreed
        # on, all ranks redo the decomposition, and the fit ends with the right answer
        import bigkrls_amd._lib as L
        L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "capi", "libbigkrls_hip_fault.so")
        if rank == garbage_rank:
            os.environ["BIGKRLS_FAULT"] = "eig_garbage"
        os.environ["BIGKRLS_REPORT_REDO"] = "1"
    ulp_rank = int(sys.argv[sys.argv.index("--ulp-rank") + 1]) if "--ulp-rank" in sys.argv else None
    if ulp_rank is not None:
        # the test build: ONE rank's copy of the replicated eigenvalues is off by one unit in the last place of one kept
        # value -- a valid decomposition that passes the check against K. The lock-step lambda search must still see the
        # same values on every rank (rank 0's are broadcast, csrc/fit.hip): identical lambda, coefficients and
        # eigenvalues on all ranks, bit for bit, and the deviating rank counts the event
        import bigkrls_amd._lib as L
        L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "capi", "libbigkrls_hip_fault.so")
        if rank == ulp_rank:
            os.environ["BIGKRLS_FAULT"] = "vals_ulp"
        os.environ["BIGKRLS_REPORT_REDO"] = "1"
    if fault_rank is not None:
        # the test build of the library (fault-injection hooks compiled in); the fault itself only in ONE rank's process
        import bigkrls_amd._lib as L
        L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "capi", "libbigkrls_hip_fault.so")
        if rank == fault_rank:
            os.environ["BIGKRLS_FAULT"] = "s1_open"
    mock = "--rccl-mock" in sys.argv
    if mock:
        os.environ["BIGKRLS_RCCL_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mock_rccl", "libmock_rccl.so")
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    except Exception as e:              # (torch.distributed.DistNetworkError: EADDRINUSE on rank 0)
        if "EADDRINUSE" in str(e) or "address already in use" in str(e):
            sys.exit(PORT_TAKEN)
        raise
    ctx = bk.Context(0)
    if mock:
        comm = bkdist.get_comm(ctx, "rccl")               # bigkrls_comm_unique_id / bigkrls_comm_create: the product's path
        assert (comm.world, comm.rank, comm.kind) == (world, rank, "rccl")
        assert comm.rank_count() == (rank, world)
    else:
        comm = bkdist.get_comm(ctx, "host")               # bigkrls_comm_create_callbacks: collectives staged through gloo
        assert (comm.world, comm.rank, comm.kind) == (world, rank, "callbacks")
    X, y = synth(n, p, 103)
    kw = dict(Neig=neig) if neig else {}
    if trunc is not None:
        kw["eigtrunc"] = trunc
    if "--options" in sys.argv:
        # a binary last column (first differences, src/bigderiv_v3.cpp:31-87), a which.derivatives subset that
        # includes it (quirk Q6 in the rescaling) -- the row-block derivative pass with its all-gathers
        X[:, -1] = (X[:, -1] > 0.12345).astype(float)
        kw["which_derivatives"] = [2, p]
    if fault_rank is not None:
        # a rank that fails locally must not leave its peers waiting in a collective: EVERY rank gets the error
        from bigkrls_amd._lib import BigKRLSError, ENOMEM
        try:
            bkdist.bigKRLS_dist(y, X, comm=comm, **kw)
            ok = False
            msg = "no error raised"
        except BigKRLSError as e:
            msg = str(e)
            ok = e.code == ENOMEM and (("injected fault" in msg) if rank == fault_rank else ("another rank" in msg))
        print(f"rank {rank}/{world} fault injected in rank {fault_rank}: {msg[:120]}" + (" OK" if ok else " MISMATCH"), flush=True)
        dist.barrier()
        bkdist.release_comms()
        dist.destroy_process_group()
        sys.exit(0 if ok else 1)
    T = {}
    t0 = time.perf_counter()
    out = bkdist.bigKRLS_dist(y, X, comm=comm, timings=T, keep_outputs=True, **kw)
    ctx.sync()
    dt = time.perf_counter() - t0
    cnt = ctx.counters()
    replicas_ok = True
    if ulp_rank is not None:
        import hashlib
        os.environ.pop("BIGKRLS_FAULT", None)              # (the single-process fit below is the undisturbed one)
        mine = (float(out["lambda"]), int(out["lastkeeper"]),
                hashlib.sha256(np.ascontiguousarray(out["K.eigenvalues"], dtype=np.float64).tobytes()).hexdigest(),
                hashlib.sha256(np.ascontiguousarray(out["coeffs"], dtype=np.float64).tobytes()).hexdigest())
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        replicas_ok = all(e == everyone[0] for e in everyone) and cnt["replica_diff"] == (1 if rank == ulp_rank else 0)
        print(f"rank {rank}/{world} one-ulp deviation injected in rank {ulp_rank}: replicas "
              f"{'bitwise identical' if all(e == everyone[0] for e in everyone) else 'DIFFER ' + repr(everyone)}, counters {cnt}", flush=True)
    one = bk.bigKRLS(y, X, ctx=ctx, **kw)

    def rel(a, b):
        a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
        return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))

    checks = {
        "lambda": rel(out["lambda"], one["lambda"]),
        "coeffs": rel(out["coeffs"], one["coeffs"]),
        "yfitted": rel(out["yfitted"], one["yfitted"]),
        "avgderivatives": rel(out["avgderivatives"], one["avgderivatives"]),
        "derivatives": rel(out["derivatives"], one["derivatives"]),
        "var.avgderivatives": rel(out["var.avgderivatives"], one["var.avgderivatives"]),
        "K.eigenvalues[:lastkeeper]": rel(np.asarray(out["K.eigenvalues"])[: out["lastkeeper"]],
                                          np.asarray(one["K.eigenvalues"])[: one["lastkeeper"]]),
    }
    r0, r1 = out["rows"]
    if r1 > r0:
        checks["K.cols"] = rel(out["K.cols"].to_numpy()[:, : r1 - r0], one["K"].to_numpy()[:, r0:r1] if hasattr(one["K"], "to_numpy") else one["K"][:, r0:r1])
        vc = one["vcov.est.c"]
        vc = vc.to_numpy() if hasattr(vc, "to_numpy") else vc
        checks["vcov.est.c.cols"] = rel(out["vcov.est.c.cols"].to_numpy()[:, : r1 - r0], vc[:, r0:r1])
    ok = out["lastkeeper"] == one["lastkeeper"] and all(v < 1e-7 for v in checks.values()) and replicas_ok
    print(f"rank {rank}/{world} N={n} P={p} neig={neig}: {dt:.2f} s lastkeeper {out['lastkeeper']} vs {one['lastkeeper']} "
          + " ".join(f"{k}={v:.1e}" for k, v in checks.items())
          + f" [redone={cnt['redone']} replayed={cnt['replayed']} replica_diff={cnt['replica_diff']}]" + (" OK" if ok else " MISMATCH"), flush=True)
    dist.barrier()
    bkdist.release_comms()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    if os.environ.get("BIGKRLS_DIST_WORKER") == "1":
        worker()
    else:
        launcher()
