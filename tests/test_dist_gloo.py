"""The N > 1 plumbing of the multi-GPU fit on CPU: world sizes 2 and 3 under gloo through the library's own entry
points (bigkrls_comm_create_callbacks / bigkrls_comm_check / bigkrls_fit_dist_rows). The numerics of
bigkrls_fit_dist need a GPU: tests/test_gpu_dist_world.py runs world sizes 2 and 3 on one MI355X through the same
callback table, tests/test_gpu_fit.py the RCCL communicator at world size 1."""
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


@pytest.mark.parametrize("world", [1, 2, 3])
def test_callback_communicator_and_row_partition(world):
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = "2"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(HERE, "_dist_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count(" OK") == world
