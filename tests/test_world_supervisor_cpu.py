"""tests/_world_supervisor.py, the process that runs the multi-rank cases of the GPU session a few at a time: exit codes
land in <log>.rc, a job that fails is run once more with its first output kept as <log>.attempt1, environment
overrides are applied (None removes a variable). Dummy jobs, no GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_supervisor_runs_jobs_records_codes_and_retries_once(tmp_path):
    flag = str(tmp_path / "flag")
    jobs = [
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
    assert rc == {"a": 0, "b": 5, "c": 0}
    assert open(tmp_path / "a.log").read().strip() == "second"
    assert open(tmp_path / "a.log.attempt1").read().strip() == "first"
    assert open(tmp_path / "b.log").read().strip() == "1 None"           # FOO added, BAR removed
    assert os.path.exists(tmp_path / "b.log.attempt1") and not os.path.exists(tmp_path / "c.log.attempt1")
    assert p.stdout.count("running it once more") == 2
