"""`python bench.py --gpus N` starts its own N ranks (the reference's parallel path starts its own workers too,
R/bigKRLS.R:340-343): the launch path on CPU through `--dry-launch` (gloo rendezvous, the library's rank object over
host buffers, no GPU), the failure of one rank, a launcher/--gpus mismatch and too few devices."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    e.update(OMP_NUM_THREADS="2", **env)
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=600)


@pytest.mark.parametrize("world,config", [(2, "C3"), (3, "C4")])
def test_gpus_n_launches_n_ranks_that_rendezvous(world, config):
    r = run(["--gpus", str(world), "--dry-launch", "--config", config])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    line = json.loads(last)                                     # the JSON line is the LAST line of stdout
    assert line["n_gpus"] == world and line["comm_nranks"] == world and line["self_launched"] is True
    assert line["ranks_in_allreduce"] == world                  # every rank took part in a collective of the library
    rows = line["rows"]
    n = line["config"]["n"]
    assert rows[0][0] == 0 and rows[-1][1] == n
    assert all(rows[i][1] == rows[i + 1][0] for i in range(world - 1))


def test_a_failing_rank_fails_the_run_and_leaves_no_rank_behind():
    r = run(["--gpus", "2", "--dry-launch"], BIGKRLS_DRY_FAIL_RANK="1")
    assert r.returncode == 1
    assert "rank 1 exited with code 3" in r.stderr
    assert not any(ln.lstrip().startswith("{") for ln in r.stdout.splitlines())


def test_mismatch_between_launcher_and_gpus_is_refused():
    r = run(["--gpus", "3", "--dry-launch"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr


def test_fewer_devices_than_gpus_is_an_error_not_a_one_gpu_run():
    r = run(["--gpus", "2", "--steps", "1", "--warmup", "0"], HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    assert r.returncode == 2 and "GPU(s) visible" in r.stderr and r.stdout.strip() == ""
