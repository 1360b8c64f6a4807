import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


# Multi-rank runs of the HIP backend on one GPU (tests/_dist_world_gpu.py): N, P, WORLD, extra arguments.
WORLD_CASES = {
    "dense_world2": ["3000", "8", "2"],
    "dense_world3_ragged": ["2500", "6", "3"],
    "krylov_world2": ["17000", "10", "2", "--krylov", "60"],
    "dense_world4_empty_rank": ["300", "4", "4", "--eigtrunc", "0.001"],          # blocks of 128 columns: the fourth rank owns nothing
    "replicated_world2": ["200", "3", "2", "--eigtrunc", "0.001"],
    "dense_world3_binary_which": ["1200", "5", "3", "--eigtrunc", "0.01", "--options"],
    # a local failure in ONE rank (test build of the library): every rank must return the error, none may hang
    "dense_world2_rank_failure": ["600", "4", "2", "--eigtrunc", "0.001", "--fault-rank", "1"],                # n <= 256: K gathered, Q by all-reduce
    # ONE rank's persistent-kernel watchdog fires (test build): agreed on by all ranks, decomposition replayed everywhere
    "dense_world2_watchdog_replay": ["900", "4", "2", "--eigtrunc", "0.001", "--watchdog-rank", "1"],
    # ONE rank's eigenvector slice comes back wrong without an error (test build): the fit's check against K catches it on
    # every rank's own rows, the failure is agreed on and the decomposition redone everywhere
    "dense_world2_garbage_redo": ["900", "4", "2", "--eigtrunc", "0.001", "--garbage-rank", "1"],
    # ONE rank's replicated eigenvalues deviate by one unit in the last place (test build): rank 0's are used everywhere,
    # lambda / c / eigenvalues come back bitwise identical on both ranks, the deviating rank counts the event
    "dense_world2_ulp_replica": ["900", "4", "2", "--eigtrunc", "0.001", "--ulp-rank", "1"],
    "krylov_world2_ulp_replica": ["17000", "10", "2", "--krylov", "60", "--ulp-rank", "1"],
    # a numerically low-rank kernel (P = 2) through the sharded block Lanczos: ill-conditioned blocks re-orthogonalised with
    # all-reduced coefficients, the check as soon as the Krylov space is invariant -- the same decisions on both ranks
    "krylov_world2_low_rank": ["17000", "2", "2", "--krylov", "512"],
    # The product's own communicator path (unique id -> bigkrls_comm_create -> the dlopen'd function table, collectives
    # asynchronous on the context's stream) over tests/mock_rccl, with the library's DEFAULT kernels ("--default-knobs":
    # the persistent panel factorisation / bulge chasing beside the collectives; a watchdog that fires because the rank
    # processes of this session crowd one GPU is agreed on and replayed)
    "rccl_mock_dense_world2": ["3000", "8", "2", "--rccl-mock", "--default-knobs"],
    "rccl_mock_dense_world3_ragged": ["2500", "6", "3", "--rccl-mock"],
    "rccl_mock_krylov_world2": ["17000", "10", "2", "--krylov", "60", "--rccl-mock", "--default-knobs"],
    "rccl_mock_dense_world4_empty_rank": ["300", "4", "4", "--eigtrunc", "0.001", "--rccl-mock"],
    # n large enough for the partitioned stage 1 to aggregate panels per trailing update: pairs (trailing matrix >= 10 752
    # rows: the first seven pairs at n = 11 700), three ranks with a ragged last block; groups of four (>= 12 800 rows: two
    # groups at n = 13 500) followed by pairs, two ranks over the RCCL path
    "dense_world3_pairs": ["11700", "6", "3", "--eigtrunc", "0.001"],
    "rccl_mock_dense_world2_groups": ["13500", "6", "2", "--eigtrunc", "0.001", "--rccl-mock", "--default-knobs"],
}
# `python bench.py --gpus 2` end to end on the one GPU: its own launcher, two rank processes, the library's RCCL
# communicator over tests/mock_rccl (BIGKRLS_BENCH_SHARE_GPU=1: both ranks on device 0, gloo for the program's barrier)
BENCH_CASES = {
    "bench_gpus2_c2": ["--gpus", "2", "--config", "C2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
}
_world_runs = {}
_supervisor = None


def pytest_sessionstart(session):
    """The rank processes have to be started before this process initialises the GPU (a process that holds
    the GPU must not start other programs on this pool), i.e. here, not inside a test. Counting devices does
    not initialise anything."""
    expr = session.config.getoption("markexpr", "") or ""
    if "not gpu" in expr or os.environ.get("BIGKRLS_SKIP_WORLD_RUNS"):
        return
    try:
        import torch
        if torch.cuda.device_count() == 0:
            return
    except Exception:
        return
    import json
    import subprocess
    import tempfile
    # the stand-in for librccl is built by __graft_entry__.build(); build it here if it did not travel (no GPU needed)
    mock = os.path.join(ROOT, "tests", "mock_rccl", "libmock_rccl.so")
    if not os.path.exists(mock):
        subprocess.run(["make", "-C", os.path.dirname(mock)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    have_mock = os.path.exists(mock)
    jobs = []
    tmpdir = tempfile.mkdtemp(prefix="bigkrls_world_")
    for name, args in WORLD_CASES.items():
        if "--rccl-mock" in args and not have_mock:
            continue                                   # (the test of this case skips)
        # The rank processes share the one GPU with each other and with this session: the persistent kernels of the
        # eigensolver spin on messages between workgroups that must be co-resident, which nothing guarantees here --
        # most cases use the launch-per-step kernels (a fired watchdog is replayed, but each replay costs seconds);
        # "--default-knobs" cases run the library as shipped.
        env = {"BIGKRLS_PQ": "steps", "BIGKRLS_BC": "wavefront"}
        if "--default-knobs" in args:
            env = {"BIGKRLS_PQ": None, "BIGKRLS_BC": None}
        log = os.path.join(tmpdir, f"{name}.log")
        jobs.append([name, [sys.executable, os.path.join(ROOT, "tests", "_dist_world_gpu.py")] + args, env, log])
        _world_runs[name] = log
    for name, args in BENCH_CASES.items():
        if not have_mock:
            continue
        log = os.path.join(tmpdir, f"{name}.log")
        env = {"BIGKRLS_BENCH_SHARE_GPU": "1", "BIGKRLS_RCCL_LIB": mock, "WORLD_SIZE": None}
        jobs.append([name, [sys.executable, os.path.join(ROOT, "bench.py")] + args, env, log])
        _world_runs[name] = log
    # one supervisor runs them a few at a time (tests/_world_supervisor.py); it is started here, before this process
    # touches the GPU, and never touches it itself
    global _supervisor
    _supervisor = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_world_supervisor.py"), json.dumps(jobs)],
                                   cwd=ROOT)


def wait_world_run(log, timeout=1500.0):
    """(exit code, log text) of a supervised job; waits for its .rc file."""
    import time
    t_end = time.time() + timeout
    while not os.path.exists(log + ".rc"):
        if time.time() > t_end or (_supervisor is not None and _supervisor.poll() is not None and not os.path.exists(log + ".rc")):
            text = open(log).read() if os.path.exists(log) else ""
            return -999, text + "\n[conftest] the job did not finish (supervisor gone or timeout)"
        time.sleep(0.2)
    return int(open(log + ".rc").read().strip()), open(log).read()


def pytest_sessionfinish(session, exitstatus):
    if _supervisor is not None and _supervisor.poll() is None:
        _supervisor.kill()                       # (its own PID; running rank processes end with their collectives' peers)


@pytest.fixture(scope="session")
def world_runs():
    return _world_runs


@pytest.fixture(scope="session")
def lib():
    """ctypes handle of libbigkrls_hip.so; the GPU tests call through this C ABI."""
    from bigkrls_amd import _lib
    return _lib.load()


_session_ctx = None


@pytest.fixture(scope="session")
def ctx():
    import bigkrls_amd as bk
    global _session_ctx
    _session_ctx = bk.Context(0)
    return _session_ctx


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """How often the session's shared context took a recovery path (bigkrls_ctx_get_counters): a rising rate of
    redone / replayed decompositions must be visible even though each of them ends in a right answer."""
    if _session_ctx is not None:
        try:
            terminalreporter.write_line(f"bigkrls recovery counters of the session context: {_session_ctx.counters()}")
        except Exception as e:          # (never turn a green session red from here)
            terminalreporter.write_line(f"bigkrls recovery counters unavailable: {e}")
