"""bigkrls_fit_dist at WORLD_SIZE 2, 3 and 4 on one GPU through the callback communicator
(tests/_dist_world_gpu.py): the partitioned dense stage 1 with column-block offsets (a ragged last block, a rank
that owns nothing), the split back-transform, the sharded block Lanczos, the replicated decomposition of a tiny
problem, the row-block lambda search; every rank compared with the single-process fit of the same data. The rank
processes are started at session start (conftest.py: one supervisor process that runs the cases four at a time). RCCL itself refuses two ranks on one device: it is covered
at world size 1 (test_gpu_fit.py), and the library's RCCL code path -- unique id, bigkrls_comm_create, the dlopen'd
function table, collectives asynchronous on the context's stream, default kernels -- at world sizes 2-4 over
tests/mock_rccl (the "rccl_mock_*" cases and `bench.py --gpus 2`)."""
import pytest

import json

from conftest import BENCH_CASES, WORLD_CASES, wait_world_run

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(WORLD_CASES))
def test_world_n_hip_backend_matches_single_fit(world_runs, name):
    if name not in world_runs:
        pytest.skip("rank processes were not started (no GPU at session start)")
    rc, text = wait_world_run(world_runs[name])
    world = int(WORLD_CASES[name][2])
    assert rc == 0, text[-4000:]
    ok_lines = [ln for ln in text.splitlines() if ln.startswith("rank ") and ln.rstrip().endswith(" OK")]
    assert len(ok_lines) == world, text[-4000:]
    if name.endswith("garbage_redo"):
        # every rank must have redone the decomposition (the wrong slice was one rank's; the decision is agreed)
        assert text.count("redoing the decomposition") == world, text[-4000:]
    if name.endswith("watchdog_replay"):
        # the fault fired in ONE rank; every rank must have replayed (the decision is agreed, not local)
        assert text.count("replaying the distributed decomposition") == world, text[-4000:]


@pytest.mark.parametrize("name", sorted(BENCH_CASES))
def test_bench_gpus2_through_the_rccl_path(world_runs, name):
    """`python bench.py --gpus 2`: the launcher starts two ranks, each builds the library's RCCL communicator
    (bigkrls_comm_unique_id / bigkrls_comm_create; librccl replaced by tests/mock_rccl, which accepts two ranks on one
    device and keeps the stream-ordered semantics) and runs bigkrls_fit_dist with the default kernels; the last line
    of stdout is rank 0's JSON line with n_gpus = 2 and the rank count read back from the library."""
    if name not in world_runs:
        pytest.skip("rank processes were not started (no GPU at session start)")
    rc, text = wait_world_run(world_runs[name])
    assert rc == 0, text[-4000:]
    line = [ln for ln in text.splitlines() if ln.lstrip().startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["comm_nranks"] == 2 and res["scaling"] == "strong", line[:400]
    assert res["value"] > 0 and "N=5000" in res["config"]["workload"], line[:400]
