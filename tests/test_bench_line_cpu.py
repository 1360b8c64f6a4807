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


def test_compact_line_without_a_sampled_roofline_still_carries_the_kernel_gemm():
    """No candidate kernel sampled (roofline = None): the line keeps a `roofline` object with `kernel_gemm` in it, and the
    short top-level copy."""
    import bench
    long_form = json.load(open(os.path.join(ROOT, "profiles", "r04", "r04z_bench_C3_default_with_cpu_baseline.json")))
    kg = long_form["kernel_gemm"]
    long_form["roofline"] = None
    line = bench.compact_line(long_form)
    assert line["roofline"]["kernel_gemm"]["tflops"] == kg["tflops"]
    assert line["kernel_gemm"]["tflops"] == kg["tflops"] and line["kernel_gemm"]["ms"] == kg["ms"]
    assert "traffic_source" not in line["roofline"]


def test_compact_line_says_where_the_traffic_figure_comes_from():
    import bench
    long_form = json.load(open(os.path.join(ROOT, "profiles", "r04", "r04z_bench_C3_default_with_cpu_baseline.json")))
    line = bench.compact_line(long_form)
    assert line["roofline"]["traffic"] is not None and line["roofline"]["traffic_source"].startswith("model:")


def test_cpu_baseline_object_is_assembled_from_the_child_lines_and_fits_the_line():
    """CpuBaseline.collect() on recorded child output (no child process, no GPU): the required keys, the share of
    `value` that is scaled from timed samples, and the whole compact line still under 6 KB with it."""
    import bench
    cb = bench.CpuBaseline.__new__(bench.CpuBaseline)
    cb.n, cb.p = 20000, 20

    class Done:
        def wait(self, timeout=None):
            return 0

        def kill(self):
            pass

        def join(self, timeout=None):
            pass

    cb.proc = cb.reader = Done()
    cb.lines = [
        {"phase": "ready", "cores": 64, "cpu_count": 256},
        {"phase": "small", "n": 2000, "literal_s": 15.5, "efficient_s": 1.9, "phases_s": {"kernel": 0.05}},
        {"phase": "kernel", "s": 5.3}, {"phase": "eigen", "s": 288.8},
        {"phase": "lambda", "s": 230.4, "extrapolated": True, "extrapolated_s": 194.0, "probes_timed": 6},
        {"phase": "coeffs", "s": 6.0}, {"phase": "vcov_c", "s": 0.4}, {"phase": "vcov_fitted", "s": 29.1},
        {"phase": "derivatives", "s": 653.1, "extrapolated": True, "extrapolated_s": 392.0, "columns_timed": 8},
        {"phase": "done", "literal_s": 1213.4, "efficient_s": 299.5, "probes": 38,
         "efficient_phases_s": {"kernel": 5.3, "eigen": 288.8}, "lastkeeper": 250, "lam": 36.3},
    ]
    res = cb.collect(700.0)
    assert res["kind"] == "port" and res["cores"] == 64 and res["value"] == 1213.4 and res["unit"] == "s per fit"
    assert abs(res["extrapolated_share"] - (194.0 + 392.0) / 1213.4) < 1e-3
    assert "6 of 38" in res["sample"] and "8 of 20" in res["sample"] and len(res["sample"]) < 500
    long_form = json.load(open(os.path.join(ROOT, "profiles", "r04", "r04z_bench_C3_default_with_cpu_baseline.json")))
    long_form["cpu_baseline"] = res
    assert len(json.dumps(bench.compact_line(long_form))) < 6000
