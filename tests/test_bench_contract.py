"""The bench line the driver records: the committed run of `python bench.py` (profiles/r03_bench_default.json, produced on an MI355X)
carries every field of the bench contract, with the types and the internal consistency the contract asks for.  CPU only: nothing is
executed, the JSON a real run printed is checked."""
import csv
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_default.json")))


def records():
    out = [("cube headline", DEFAULT)]
    out += list(DEFAULT["tree_scenes"].items())
    return out


def test_headline_has_the_contract_fields():
    d = DEFAULT
    for key, typ in [("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                     ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)]:
        assert isinstance(d[key], typ), key
    assert d["vs_baseline"] is None                      # BASELINE.md publishes no number for this metric
    assert d["metric"] == "Mrays/s" and d["unit"] == "Mrays/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["scaling"] in ("weak", "strong") and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["scene"] == "cube.obj" and (d["config"]["width"], d["config"]["height"], d["config"]["max_depth"], d["config"]["samples"]) == (1920, 1080, 4, 64)
    # the tree path is timed by the same invocation
    assert set(DEFAULT["tree_scenes"]) == {"dodge_1920x1080_d4_s64", "cfg4_wavy_3840x2160_d8_s256"}
    assert DEFAULT["tree_scenes"]["cfg4_wavy_3840x2160_d8_s256"]["config"]["tree"]["nodes"] > 1000


def rooflines(d):
    """`roofline` is the kernel group that takes most of the frame (shadow group or k_shade); the other one is kept beside it (the trace group
    has a record of its own: test_trace_group_walked_rays_launches_and_cpu_ratios)"""
    out = [d["roofline"]]
    for k in ("roofline_shadow", "roofline_shade"):
        if k in d:
            out.append(d[k])
    return out


@pytest.mark.parametrize("name,d", records())
def test_every_record_is_consistent_and_its_roofline_physical(name, d):
    # value = rays of the frame / time of a step
    assert abs(d["value"] - d["rays_per_frame"] / (d["ms_per_step"] * 1e-3) / 1e6) <= 1e-3 * d["value"]
    groups = {r["group"] for r in rooflines(d)}
    assert groups == {"shadow", "shade"}
    ms = d["roofline"]["ms_per_frame"]
    assert d["roofline"]["group"] == ("shade" if ms["shade"] > ms["shadow"] else "shadow")       # the dominant group carries the name `roofline`
    for r in rooflines(d):
        assert r["bound"] == "valu" and r["unit"] == "wave-instructions/SIMD/ns" and r["peak"] == 0.967
        # PHYSICAL: 0 < frac <= 1, frac = achieved / peak, and re-derivable from the line itself: modelled instructions / group time
        assert 0.0 <= r["useful_frac"] <= r["frac"] <= 1.0 and r["frac"] > 0.0
        assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 2e-3
        total = r["modelled_valu_wave_instructions_per_frame"]["total"]
        assert abs(r["achieved"] - total / (r["ms_per_frame"][r["group"]] * 1e6) / 1024) <= 2e-3 * max(1.0, r["achieved"])
        c = r["cost_per_step"]
        if r["group"] == "shadow":
            # the model is steps x cost, both in the line
            w = r["work"]["shadow"]
            assert w["units"] > 0
            assert total >= w["tri_steps_lanes_triangles"] * c["tri_lanes_triangles"] + w["tri_steps_lanes_rays"] * c["tri_lanes_rays"]
            assert r["reference_semantics_bytes"]["per_frame_k_shadow"] > 0
        else:
            w = r["work"]["shade"]
            assert total == w["tiles_of_64_hits"] * (w["samples"] * c["shade_sample"] + c["shade_tile"])
        # PMC constants: quoted (then physical too) or explicitly null
        ev = r["executed_valu"]
        assert ev["constant"] is True and (ev["frac"] is None or r["frac"] * 0.5 <= ev["frac"] <= 1.0)
        if ev["frac"] is not None:
            # the clock-independent form beside the per-ns one: instructions / GPU cycles / 1024 SIMDs / 0.5 per cycle
            assert 0.0 < ev["frac_clock_independent"] <= 1.0 and "GRBM_GUI_ACTIVE" in ev["clock_independent_source"]
            if r["group"] == "shade":
                assert 0.0 < ev["wave_lifetime_share_executing_valu"] <= 1.0 and ev["valu_active_quadcycles_per_instruction"] >= 0.9
        assert r["traffic"] is None or (r["traffic"] > 0 and 0.0 < r["hbm_frac"] <= 1.0)
        assert "HIP events" in r["timing_source"]
    c = d["cpu_baseline"]
    assert c["unit"] == "Mrays/s" and c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)


@pytest.mark.parametrize("name,d", records())
def test_trace_group_walked_rays_launches_and_cpu_ratios(name, d):
    """Round-3 evidence fields: a roofline record for the trace group whose counters carry the trace region's own names; the rays actually formed and
    walked beside the queries resolved; launches per frame; both CPU ratios with their hosts."""
    t = d["roofline_trace"]
    assert t["group"] == "trace" and t["bound"] == "valu" and t["peak"] == 0.967
    w = t["work"]["trace"]
    assert set(w) >= {"tri_steps_lanes_rays", "tri_steps_lanes_triangles", "box_steps_stack_walk", "units", "tri_cone_tests"}
    assert not any(k.startswith("beam") or k.startswith("shaft") for k in w)            # no shadow-kernel names on the trace region
    assert w["units"] > 0 and 0.0 < t["frac"] <= 1.0 and t["useful_frac"] <= t["frac"]
    assert abs(t["achieved"] - t["modelled_valu_wave_instructions_per_frame"]["total"] / (t["ms_per_frame"]["trace"] * 1e6) / 1024) <= 2e-3 * max(1.0, t["achieved"])
    # queries resolved vs rays actually walked
    assert 0 < d["rays_walked_per_frame"] <= d["rays_per_frame"]
    assert abs(d["Mrays_walked_per_s"] - d["rays_walked_per_frame"] / (d["ms_per_step"] * 1e-3) / 1e6) <= 1e-3 * d["Mrays_walked_per_s"]
    if name == "cube headline":
        # the counters of the group the headline roofline is about (k_shade) and of the trace group: HBM bytes per launch from the PMC passes
        assert d["roofline"]["traffic"] > 0 and 0.0 < d["roofline"]["hbm_frac"] < 0.5 and d["roofline_trace"]["traffic"] > 0
        assert d["rays_walked_per_frame"] < 0.2 * d["rays_per_frame"]       # the beam test decides most sample segments without forming them
        assert d["launches_per_frame"] <= 12                                # memset + 4 + 4 + k_deep + resolve (22 before round 3)
    assert d["launches_per_frame"] >= 6
    ratios = d["cpu_baseline"]["ratios"]
    assert ratios["vs_port_same_host"]["value"] == d["speedup_vs_cpu_port"] and "this host" in ratios["vs_port_same_host"]["cpu"]
    if "cfg4" not in name:
        r = ratios["vs_reference_as_surveyed"]
        assert r["value"] > ratios["vs_port_same_host"]["value"] and "another host" in r["cpu"]


@pytest.mark.parametrize("scene,kernel", [("cube", "k_shadow<false, true, false>"), ("dodge", "k_shadow_shaft"), ("wavy", "k_shadow_shaft")])
def test_rocprof_kernel_stats_agree_with_the_event_times(scene, kernel):
    """profiles/r03_<scene>_kernel_stats.csv (rocprofv3 --kernel-trace --stats over bench.py) against the HIP-event time of the SAME run
    (profiles/r03_bench_<scene>_under_rocprof.json): the shadow group's average duration per frame agrees within 10 % (+ 7 us per launch of the
    group for the gaps the event interval contains: under the profiler a launch boundary costs 5-6 us)."""
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", f"r03_{scene}_kernel_stats.csv"))))
    # the shadow GROUP of a level: beam test + walking launch (+ leaf-task launch); "k_shadow<true, ..." are the COUNT variants of the statistics frames
    shadow = [r for r in rows if ("k_shadow" in r["Name"] or "k_beam" in r["Name"] or "k_pair_beam" in r["Name"]) and "k_shadow<true" not in r["Name"]]
    assert any(kernel in r["Name"] for r in shadow)
    bench = json.load(open(os.path.join(ROOT, "profiles", f"r03_bench_{scene}_under_rocprof.json")))
    rs = bench["roofline"] if bench["roofline"]["group"] == "shadow" else bench["roofline_shadow"]
    # one WALKING launch per level (not the beam test, not the leaf-task launches k_shadow_shaft<true, ...> / k_shadow<false, false, true>)
    walking = [r for r in shadow if "k_shadow" in r["Name"] and "k_shadow_shaft<true" not in r["Name"] and "k_shadow<false, false, true>" not in r["Name"]]
    frames = sum(int(r["Calls"]) for r in walking) / rs["launches_per_frame"]
    ms_rocprof = sum(float(r["TotalDurationNs"]) for r in shadow) / frames / 1e6
    ms_events = rs["ms_per_frame"]["shadow"]
    # the event interval of a group also holds the gaps between its launches (the cube's group is ten launches of a few microseconds each)
    launches = sum(int(r["Calls"]) for r in shadow) / frames
    assert -0.10 * ms_events <= ms_events - ms_rocprof <= 0.10 * ms_events + launches * 0.007, (ms_rocprof, ms_events, launches)


def test_pmc_constants_are_stamped_with_the_kernels_source():
    for name in ("traffic.json", "valu.json"):
        j = json.load(open(os.path.join(ROOT, "profiles", name)))
        for sc in ("cube", "dodge", "wavy"):
            assert len(j[sc]["kernels_sha256"]) == 64, (name, sc)
