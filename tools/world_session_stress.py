"""Start every world case of tests/conftest.py at once (as the GPU test session does) beside a foreground load, several
times; report the cases that fail and keep their logs (development probe).  python tools/world_session_stress.py [rounds]"""
import os, subprocess, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import WORLD_CASES
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
out = os.path.join(ROOT, "gpurun_out", "flake3"); os.makedirs(out, exist_ok=True)
bad = 0
for r in range(rounds):
    procs = {}
    for name, args in WORLD_CASES.items():
        if os.environ.get("STRESS_ALL_MOCK") and "--rccl-mock" not in args and "--fault-rank" not in args:
            args = list(args) + ["--rccl-mock"]          # every case through the RCCL code path (tests/mock_rccl)
        env = dict(os.environ, BIGKRLS_PQ="steps", BIGKRLS_BC="wavefront", BIGKRLS_VERBOSE="1")
        if "--default-knobs" in args:
            env = dict(os.environ, BIGKRLS_VERBOSE="1")
        log = open(os.path.join(out, f"r{r}_{name}.log"), "w")
        procs[name] = (subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_world_gpu.py")] + args,
                                        stdout=log, stderr=subprocess.STDOUT, cwd=ROOT, env=env), log)
    load = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C4", "--steps", "4", "--warmup", "1",
                             "--no-cpu-baseline"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=ROOT)
    for name, (p, log) in procs.items():
        rc = p.wait()
        log.close()
        expect_fail = False
        if rc != 0:
            bad += 1
            print(f"round {r}: {name} rc={rc}", flush=True)
        else:
            os.remove(log.name)
    load.wait()
print(f"rounds {rounds}, failing cases {bad}")
