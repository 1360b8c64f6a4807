"""tests/_world_supervisor.py, the process that runs the multi-rank cases of the GPU session a few at a time: exit codes
land in <log>.rc, every job runs exactly once (a failing job is NOT run again: its first exit code is the verdict),
environment overrides are applied (None removes a variable). Dummy jobs, no GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_supervisor_runs_every_job_once_and_records_its_code(tmp_path):
    flag = str(tmp_path / "flag")
    jobs = [
        # would pass on a second run: the supervisor must not give it one
        ["flaky", [sys.executable, "-c",
                   f"import os,sys; p={flag!r}; ok=os.path.exists(p); open(p,'w').close(); print('second' if ok else 'first'); "
                   "sys.exit(0 if ok else 3)"], {}, str(tmp_path / "a.log")],
        ["bad", [sys.executable, "-c", "import os,sys; print(os.environ.get('FOO'), os.environ.get('BAR')); sys.exit(5)"],
         {"FOO": "1", "BAR": None}, str(tmp_path / "b.log")],
        ["good", [sys.executable, "-c", "print('fine')"], {}, str(tmp_path / "c.log")],
    ]
    env = dict(os.environ, BAR="set-by-parent")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_world_supervisor.py"), json.dumps(jobs)],
                       capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode == 0, p.stderr
    rc = {n: int(open(tmp_path / f"{n}.log.rc").read()) for n in "abc"}
    assert rc == {"a": 3, "b": 5, "c": 0}
    assert open(tmp_path / "a.log").read().strip() == "first"
    assert open(tmp_path / "b.log").read().strip() == "1 None"           # FOO added, BAR removed
    assert not any(f.name.endswith(".attempt1") for f in tmp_path.iterdir())


def test_rank_launcher_restarts_the_rendezvous_on_a_taken_port_and_never_hangs(tmp_path, monkeypatch):
    """tests/_dist_world_gpu.py's launcher (no GPU here: the ranks are its self-test stubs): the port it probed can be
    taken before rank 0 binds it -- with some thirty launchers and gloo's own connections on one host that happens --
    and then ONLY the rendezvous is started again; a rank that dies must end the run as a failure within the grace
    period instead of leaving its peers (and the session) waiting for ever."""
    import time
    script = os.path.join(ROOT, "tests", "_dist_world_gpu.py")
    env = dict(os.environ, BIGKRLS_LAUNCHER_SELFTEST="port", BIGKRLS_LAUNCHER_MARKER=str(tmp_path / "m1"))
    env.pop("BIGKRLS_DIST_WORKER", None)
    p = subprocess.run([sys.executable, script, "100", "2", "2"], capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode == 0 and "starting the ranks again" in p.stdout, p.stdout + p.stderr
    env["BIGKRLS_LAUNCHER_SELFTEST"] = "dead"
    t0 = time.time()
    p = subprocess.run([sys.executable, script, "100", "2", "2"], capture_output=True, text=True, timeout=200, env=env)
    assert p.returncode == 1 and 55.0 < time.time() - t0 < 120.0, (p.returncode, time.time() - t0)
