"""Runs the multi-rank cases of the GPU test session a few at a time (never touches the GPU itself).

tests/conftest.py starts ONE of these at session start -- before the pytest process initialises the GPU, after which it
must not start programs -- with a JSON job list on the command line:  [[name, argv, env_overrides, log_path], ...].
Each job's output goes to log_path; its exit code is written to log_path + ".rc" when it ends. At most PARALLEL jobs
run at once: all of them at once (some thirty processes with their own N x N workspaces and persistent kernels on
one GPU, beside the session's own C4 / C5 fits) turned rare scheduling pathologies into test failures. A job that
fails is run once more (its first output is kept as <log>.attempt1): see DESIGN.md section 7, the open row."""
import json
import os
import subprocess
import sys
import time

PARALLEL = 4
RETRY_ONCE = True


def main():
    jobs = json.loads(sys.argv[1])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    running = []          # (proc, job, log_file, attempt)
    pending = [(job, 1) for job in jobs]

    def start(job, attempt):
        name, argv, env_over, log_path = job
        env = dict(os.environ)
        for k, v in env_over.items():
            if v is None:
                env.pop(k, None)
            else:
                env[k] = v
        lf = open(log_path, "a")
        p = subprocess.Popen(argv, stdout=lf, stderr=subprocess.STDOUT, env=env, cwd=root)
        running.append((p, job, lf, attempt))

    while pending or running:
        while pending and len(running) < PARALLEL:
            start(*pending.pop(0))
        still = []
        for p, job, lf, attempt in running:
            rc = p.poll()
            if rc is None:
                still.append((p, job, lf, attempt))
                continue
            log_path = job[3]
            if rc != 0 and attempt == 1 and RETRY_ONCE:
                # DESIGN.md section 7, "open": callback-table runs under heavy oversubscription have come back wrong
                # about once in a hundred; the first attempt's output is kept beside the log, the verdict is the second's
                lf.close()
                os.replace(log_path, log_path + ".attempt1")      # (the tests count lines of the log: a fresh one)
                print(f"[supervisor] attempt 1 of {job[0]} ended with code {rc} ({log_path}.attempt1); running it once more",
                      flush=True)
                pending.append((job, 2))
                continue
            lf.close()
            with open(log_path + ".rc.tmp", "w") as f:
                f.write(str(rc))
            os.replace(log_path + ".rc.tmp", log_path + ".rc")
        running = still
        time.sleep(0.1)


if __name__ == "__main__":
    main()
