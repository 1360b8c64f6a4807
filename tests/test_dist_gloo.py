"""The N > 1 path (row-block partition + collectives of bigkrls_amd/dist.py) on CPU:
world_size 2 and 3 under gloo, local kernels replaced by a numpy test double."""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world,n,p,binary", [(2, 101, 3, 0), (2, 160, 4, 1), (3, 100, 3, 1)])
def test_row_block_fit_matches_oracle(world, n, p, binary):
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = "2"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(HERE, "_dist_worker.py"), str(n), str(p), str(binary)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("OK") == world


@pytest.mark.parametrize("world,n,p,neig", [(2, 1100, 3, 12), (3, 900, 2, 8)])
def test_sharded_block_lanczos_matches_lapack(world, n, p, neig):
    """SURVEY 8(e) "Eigen, partial": rank r multiplies its own rows of K, one all-gather of an
    N x block matrix per Lanczos step; everything else replicated."""
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = "2"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(HERE, "_dist_worker.py"), "krylov", str(n), str(p), str(neig)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("OK") == world


@pytest.mark.parametrize("world,n,p,neig", [(2, 300, 3, 300), (3, 333, 2, 333), (2, 275, 3, 40), (1, 290, 3, 290)])
def test_sharded_dense_eigen_matches_lapack(world, n, p, neig):
    """SURVEY 8(e) "Eigen, dense tridiagonalisation": stage 1 partitioned by column blocks -- per
    64-column panel one broadcast of the panel strip and one all-gather of A22 V; the reduced matrix is
    replicated, the back-transformed eigenvector columns are all-gathered. n not a multiple of 64 (ragged
    last block, empty last rank at world 3), Neig = N and Neig < N."""
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = "2"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(HERE, "_dist_worker.py"), "dense", str(n), str(p), str(neig)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("OK") == world


def test_row_block_fit_with_sharded_dense_eigen_matches_oracle():
    """The whole row-block fit at a size that takes the sharded dense eigensolver (n > 256): K is never
    gathered, Q arrives by an all-gather of column blocks."""
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = "2"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(HERE, "_dist_worker.py"), "280", "8", "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("OK") == 2
