"""Runs the multi-rank cases of the GPU test session a few at a time (never touches the GPU itself).

tests/conftest.py starts ONE of these at session start -- before the pytest process initialises the GPU, after which it
must not start programs -- with a JSON job list on the command line:  [[name, argv, env_overrides, log_path], ...].
Each job's output goes to log_path; its exit code is written to log_path + ".rc" when it ends. At most PARALLEL jobs
run at once: all of them at once (some thirty processes with their own N x N workspaces and persistent kernels on
one GPU, beside the session's own C4 / C5 fits) turned rare scheduling pathologies into test failures."""
import json
import os
import subprocess
import sys
import time

PARALLEL = 4


def main():
    jobs = json.loads(sys.argv[1])
    running = []          # (proc, log_path, log_file)
    pending = list(jobs)
    while pending or running:
        while pending and len(running) < PARALLEL:
            name, argv, env_over, log_path = pending.pop(0)
            env = dict(os.environ)
            for k, v in env_over.items():
                if v is None:
                    env.pop(k, None)
                else:
                    env[k] = v
            lf = open(log_path, "w")
            p = subprocess.Popen(argv, stdout=lf, stderr=subprocess.STDOUT, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            running.append((p, log_path, lf))
        still = []
        for p, log_path, lf in running:
            rc = p.poll()
            if rc is None:
                still.append((p, log_path, lf))
            else:
                lf.close()
                with open(log_path + ".rc.tmp", "w") as f:
                    f.write(str(rc))
                os.replace(log_path + ".rc.tmp", log_path + ".rc")
        running = still
        time.sleep(0.1)


if __name__ == "__main__":
    main()
