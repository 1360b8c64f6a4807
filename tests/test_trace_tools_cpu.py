"""The diagnostic trace tooling of DESIGN.md section 8, on the CPU: the host-side hash of bigkrls_amd.dist equals the
definition the device kernel implements (csrc/trace.hip: sum over the elements of a position-dependent 64-bit mix of
their bits, modulo 2^64), and tools/trace_diff.py finds the first inconsistent record of a multi-rank run, a
device/host transport mismatch, and a fit that differs from its repetitions -- while a redone decomposition (more
records, same results) is not reported as a fault."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _mix(bits, i):
    m = (1 << 64) - 1
    x = bits ^ ((i * 0x9E3779B97F4A7C15 + 1) & m)
    x ^= x >> 30
    x = (x * 0xBF58476D1CE4E5B9) & m
    x ^= x >> 27
    x = (x * 0x94D049BB133111EB) & m
    x ^= x >> 31
    return x


def test_host_hash_matches_the_definition():
    from bigkrls_amd.dist import trace_hash
    rng = np.random.default_rng(3)
    a = rng.standard_normal(257)
    a[5] = -0.0
    ref = sum(_mix(int(b), i) for i, b in enumerate(a.view(np.uint64))) & ((1 << 64) - 1)
    assert trace_hash(a) == ref
    assert trace_hash(a.reshape(1, -1)) == ref and trace_hash(a[::-1]) != ref          # position-dependent
    b = a.copy()
    b[5] = 0.0
    assert trace_hash(b) != ref                                                          # bits, not values


def _write(path, records):
    with open(path, "w") as f:
        for i, (tag, count, h, extra) in enumerate(records):
            f.write(f"{i} {tag} {count} {h:016x} {extra} 100000\n")


def _run(d, *flags):
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_diff.py"), str(d), *flags], capture_output=True, text=True)


def test_trace_diff_multi_rank_and_transport(tmp_path):
    fit = [("L:fit_begin", 0, 0, 300), ("L:ar_in", 64, 11, 0), ("C:ar_out", 64, 12, 0), ("R:s1_V", 100, 13, 0),
           ("L:bc_in", 8, 14, 1), ("C:bc_out", 8, 15, 1), ("R:fit_vals", 300, 16, 90), ("R:fit_c", 300, 17, 0)]
    for r in range(2):
        _write(tmp_path / f"pid{100 + r}.trace", [("L:comm_callbacks", 0, 0, r * 1000 + 2)] + fit)
        with open(tmp_path / f"pid{100 + r}.pytrace", "w") as f:
            f.write(f"0 ar 64 {11:016x} {12:016x}\n1 bc 8 {14:016x} {15:016x}\n")
    ok = _run(tmp_path)
    assert ok.returncode == 0 and "consistent" in ok.stdout, ok.stdout
    # rank 1's replicated factor differs: named with the record before it
    bad = list(fit)
    bad[3] = ("R:s1_V", 100, 99, 0)
    _write(tmp_path / "pid101.trace", [("L:comm_callbacks", 0, 0, 1002)] + bad)
    r = _run(tmp_path)
    assert r.returncode == 1 and "ranks differ at shared record 1" in r.stdout and "R:s1_V" in r.stdout, r.stdout
    # a host copy that changed the data on rank 0
    _write(tmp_path / "pid101.trace", [("L:comm_callbacks", 0, 0, 1002)] + fit)
    with open(tmp_path / "pid100.pytrace", "w") as f:
        f.write(f"0 ar 64 {77:016x} {12:016x}\n1 bc 8 {14:016x} {15:016x}\n")
    r = _run(tmp_path)
    assert r.returncode == 1 and "device -> host copy changed the data" in r.stdout, r.stdout


def test_trace_diff_repeated_fits_tolerate_a_redone_decomposition(tmp_path):
    res = [("R:fit_vals", 300, 21, 90), ("R:fit_Q", 27000, 22, 2), ("R:fit_c", 300, 23, 0)]
    eig = [("R:eig_band", 1000, 1, 0), ("R:eig_d", 300, 2, 0), ("R:eig_vals", 300, 3, 90), ("L:eig_Qpart", 27000, 4, 0)]
    begin = [("L:fit_begin", 0, 0, 300)]
    garbage = [("R:eig_band", 1000, 1, 0), ("R:eig_d", 300, 2, 0), ("R:eig_vals", 300, 3, 90), ("L:eig_Qpart", 27000, 666, 0)]
    _write(tmp_path / "pid7.trace", begin + eig + res + begin + garbage + eig + res + begin + eig + res)
    r = _run(tmp_path, "--repeat")
    assert r.returncode == 0 and "redone" in r.stdout and "results identical" in r.stdout, r.stdout
    wrong = [("R:fit_vals", 300, 21, 87), ("R:fit_Q", 27000, 55, 2), ("R:fit_c", 300, 56, 0)]
    _write(tmp_path / "pid7.trace", begin + eig + res + begin + eig + wrong + begin + eig + res)
    r = _run(tmp_path, "--repeat")
    assert r.returncode == 1 and "the results of fit 1 differ from fit 0" in r.stdout, r.stdout
