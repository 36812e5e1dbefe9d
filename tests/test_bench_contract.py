"""The bench line the driver records: the committed run of `python bench.py` (profiles/r01_bench_default.json, produced on an
MI355X) carries every field of the bench contract, with the types and the internal consistency the contract asks for.  CPU
only: nothing is executed, the JSON a real run printed is checked."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["r01_bench_default.json", "r01_bench_cube.json", "r01_bench_dodge.json", "r01_bench_wavy_cfg4.json"])
def test_committed_bench_line_has_the_contract_fields(name):
    d = json.load(open(os.path.join(ROOT, "profiles", name)))
    for key, typ in [("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)]:
        assert isinstance(d[key], typ), key
    assert d["vs_baseline"] is None                      # BASELINE.md publishes no number for this metric
    assert d["metric"] == "Mrays/s" and d["unit"] == "Mrays/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["scaling"] in ("weak", "strong") and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = rays of the frame / time of a step
    assert abs(d["value"] - d["rays_per_frame"] / (d["ms_per_step"] * 1e-3) / 1e6) <= 1e-3 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-3 * max(1.0, r["frac"])
    assert r["traffic"] is None or r["traffic"] > 0
    # achieved = algorithmic bytes per launch / average launch duration (HIP events inside the timed region)
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) <= 2e-3 * r["achieved"]
    assert "HIP events" in r["timing_source"]
    v = r["valu_issue"]
    assert v is not None and 0.0 < v["frac"] < 1.0 and abs(v["frac"] - v["achieved_per_simd_per_ns"] / v["peak_per_simd_per_ns"]) < 1e-3
    c = d["cpu_baseline"]
    assert c["unit"] == "Mrays/s" and c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)


def test_profiles_hold_the_rocprof_summaries_the_numbers_come_from():
    for name in ("r01_cube_kernel_stats.csv", "r01_dodge_kernel_stats.csv", "traffic.json", "valu.json"):
        assert os.path.getsize(os.path.join(ROOT, "profiles", name)) > 100, name
    head = open(os.path.join(ROOT, "profiles", "r01_cube_kernel_stats.csv")).readline()
    assert "AverageNs" in head and "Calls" in head
    body = open(os.path.join(ROOT, "profiles", "r01_cube_kernel_stats.csv")).read()
    assert "k_shadow" in body and "k_shade" in body and "k_trace" in body
