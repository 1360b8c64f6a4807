"""bigkrls_fit_dist at WORLD_SIZE 2, 3 and 4 on one GPU through the callback communicator
(tests/_dist_world_gpu.py): the partitioned dense stage 1 with column-block offsets (a ragged last block, a rank
that owns nothing), the split back-transform, the sharded block Lanczos, the replicated decomposition of a tiny
problem, the row-block lambda search; every rank compared with the single-process fit of the same data. The rank
processes are started by conftest.py at session start; RCCL itself is covered at world size 1 (test_gpu_fit.py)
because it refuses two ranks on one device."""
import pytest

from conftest import WORLD_CASES

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(WORLD_CASES))
def test_world_n_hip_backend_matches_single_fit(world_runs, name):
    if name not in world_runs:
        pytest.skip("rank processes were not started (no GPU at session start)")
    proc, log = world_runs[name]
    try:
        rc = proc.wait(timeout=900)
    finally:
        if proc.poll() is None:
            proc.kill()
    text = open(log).read()
    world = int(WORLD_CASES[name][2])
    assert rc == 0, text[-4000:]
    ok_lines = [ln for ln in text.splitlines() if ln.startswith("rank ") and ln.rstrip().endswith(" OK")]
    assert len(ok_lines) == world, text[-4000:]
    if name.endswith("watchdog_replay"):
        # the fault fired in ONE rank; every rank must have replayed (the decision is agreed, not local)
        assert text.count("replaying the distributed decomposition") == world, text[-4000:]
