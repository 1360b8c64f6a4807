"""bench.py prints ONE compact JSON line: the driver keeps only the tail of stdout, so the line must stay small and
carry BASELINE.json's second metric half -- the kernel GEMM -- inside `roofline`. Checked on a recorded long-form
line of round 4 (profiles/r04/, a data file) without a GPU."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_compact_line_keeps_the_contract_and_stays_small():
    import bench
    long_form = json.load(open(os.path.join(ROOT, "profiles", "r04", "r04z_bench_C3_default_with_cpu_baseline.json")))
    assert len(json.dumps(long_form)) > 10000          # what the driver's tail used to cut
    line = bench.compact_line(long_form)
    text = json.dumps(line)
    assert len(text) < 6000
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "fit_frac", "kernel_gemm"):
        assert key in roof, key
    kg = roof["kernel_gemm"]
    assert kg["tflops"] > 0 and kg["hbm_write_gbs"] > 0 and kg["ms"] > 0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    cb = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert "note" not in text or '"note"' not in text
    assert line["config"]["workload"].startswith("C3")
    # the parts of the dominant kernel survive in short form
    assert all(set(p) >= {"kernel", "frac", "total_ms_per_fit"} for p in roof.get("parts", []))
