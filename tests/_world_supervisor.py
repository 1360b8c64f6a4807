"""Runs the multi-rank cases of the GPU test session a few at a time (never touches the GPU itself).

tests/conftest.py starts ONE of these at session start -- before the pytest process initialises the GPU, after which it
must not start programs -- with a JSON job list on the command line:  [[name, argv, env_overrides, log_path], ...].
Each job's output goes to log_path; its exit code is written to log_path + ".rc" when it ends. At most PARALLEL jobs
run at once. Every job runs ONCE: its first exit code is the verdict (a job that fails is not run again)."""
import json
import os
import subprocess
import sys
import time

PARALLEL = 4


def main():
    jobs = json.loads(sys.argv[1])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    running = []          # (proc, job, log_file)
    pending = list(jobs)

    def start(job):
        name, argv, env_over, log_path = job
        env = dict(os.environ)
        for k, v in env_over.items():
            if v is None:
                env.pop(k, None)
            else:
                env[k] = v
        lf = open(log_path, "a")
        p = subprocess.Popen(argv, stdout=lf, stderr=subprocess.STDOUT, env=env, cwd=root)
        running.append((p, job, lf))

    while pending or running:
        while pending and len(running) < PARALLEL:
            start(pending.pop(0))
        still = []
        for p, job, lf in running:
            rc = p.poll()
            if rc is None:
                still.append((p, job, lf))
                continue
            log_path = job[3]
            lf.close()
            with open(log_path + ".rc.tmp", "w") as f:
                f.write(str(rc))
            os.replace(log_path + ".rc.tmp", log_path + ".rc")
        running = still
        time.sleep(0.1)


if __name__ == "__main__":
    main()
