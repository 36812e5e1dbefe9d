"""Pins the CPU oracle (oracle/rt_oracle.c) to the reference outputs recorded in SURVEY.md Appendix A.

These are the reference's own results (whole-frame result.ppm md5s, traceRay/screenToWorld/tree values); the oracle
must reproduce them bit-for-bit before it is trusted as the checker for the HIP path.  CPU only.
"""
import hashlib
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
KA = json.load(open(os.path.join(HERE, "golden", "survey_known_answers.json")))


def f32(x):
    return np.asarray(x, np.float32)


def close9(a, b):
    """Known answers are printed with 9 significant digits, which identify a float32 uniquely."""
    a, b = f32(a), f32(b)
    return all(np.float32(float(f"{x:.9g}")) == np.float32(float(f"{y:.9g}")) for x, y in zip(a.ravel(), b.ravel()))


@pytest.fixture(scope="module")
def cube(oracle, scenes):
    s = oracle.load_scene(os.path.join(scenes, "cube.obj"))
    yield s
    s.close()


@pytest.fixture(scope="module")
def dodge(oracle, scenes):
    s = oracle.load_scene(os.path.join(scenes, "dodgeColorTest.obj"))
    yield s
    s.close()


def test_camera_known_answers(oracle):
    cam = oracle.camera(256, 256)
    assert list(cam.center) == KA["center"]
    for key, want in KA["screen_to_world_256"].items():
        i, j = [int(v) for v in key.split(",")]
        assert close9(oracle.screen_to_world(cam, i, j), want), key


def test_area_light_samples(oracle):
    l = oracle.lights(area=True)
    pts = oracle.light_samples(l, (-1.0, 1.0, 1.0))
    ka = KA["area_samples_light_-1_1_1"]
    assert len(pts) == ka["count"]
    assert close9(pts[0], ka["first"]) and close9(pts[-1], ka["last"])


def test_cube_tree(cube):
    ka = KA["cube_root_box"]
    root = cube.node(0)
    assert cube.nnodes == ka["nodes"] and root["nfaces"] == ka["faces"] and root["is_leaf"] == 1
    assert close9(root["box"][:3], ka["min"]) and close9(root["box"][3:], ka["max"])


def test_dodge_tree(dodge):
    ka = KA["dodge_tree"]
    nodes = [dodge.node(i) for i in range(dodge.nnodes)]
    leaves = [n for n in nodes if n["is_leaf"] and not n["is_empty"]]
    assert dodge.nnodes == ka["nodes"]
    assert len(leaves) == ka["leaves"]
    assert sum(n["nfaces"] for n in leaves) == ka["face_refs"]
    assert max(n["nfaces"] for n in leaves) == ka["max_leaf"]
    assert max(n["depth"] for n in nodes) == ka["depth"]
    assert close9(nodes[0]["box"][:3], ka["root_min"]) and close9(nodes[0]["box"][3:], ka["root_max"])
    reachable = set()
    for n in leaves:
        reachable.update(n["faces"].tolist())
    assert dodge.nfaces - len(reachable) == ka["lost_faces"]


@pytest.mark.parametrize("scene_name,key", [("cube", "cube_trace_ray_256_area25"), ("dodge", "dodge_trace_ray_256_area25")])
def test_trace_ray_known_answers(oracle, cube, dodge, scene_name, key):
    scene = cube if scene_name == "cube" else dodge
    cam = oracle.camera(256, 256)
    l = oracle.lights(area=True)
    o = f32(list(cam.center))
    for px, want in KA[key].items():
        i, j = [int(v) for v in px.split(",")]
        scr = oracle.screen_to_world(cam, i, j)
        # raytraceScene culls against the root box first (flyscene.cpp:576-581); traceRay gives the same BACKGROUND
        got = scene.trace_ray(l, o, scr - o)
        assert close9(got, want), (px, got, want)


def test_dodge_tree_loses_triangles(oracle, dodge):
    """the reference's octree returns face 16302 at px (134,134), not the brute-force closest face."""
    ka = KA["dodge_tree_vs_brute_256_stride2"]
    cam = oracle.camera(256, 256)
    o = f32(list(cam.center))
    scr = oracle.screen_to_world(cam, 134, 134)
    face, t = dodge.closest_hit(o, scr - o)
    assert face == ka["px_134_134"]["face"]
    assert close9([t], [ka["px_134_134"]["t"]])
    hits = 0
    for j in range(0, 256, 2):
        for i in range(0, 256, 2):
            scr = oracle.screen_to_world(cam, i, j)
            f, _ = dodge.closest_hit(o, scr - o)
            hits += f >= 0
    assert hits == ka["tree_hits"]


def test_cube_64_histogram(oracle, cube):
    ka = KA["cube_64_area_histogram"]
    rgb, _, _ = cube.render(oracle.camera(64, 64), oracle.lights(area=True), 64, 64)
    q = oracle.quantise(rgb).reshape(-1, 3)
    white = int((q == 255).all(axis=1).sum())
    assert white == ka["white_pixels"] and len(q) - white == ka["other_pixels"]
    assert len({tuple(r) for r in q.tolist()}) == ka["distinct_colours"]


@pytest.mark.parametrize("case", KA["result_ppm_md5"], ids=lambda c: f"{c['scene']}-{c['size']}-{'area' if c['area'] else 'point'}")
def test_result_ppm_md5(oracle, cube, dodge, case, tmp_path):
    """Whole-frame parity with the unmodified reference: byte-identical result.ppm."""
    scene = cube if case["scene"] == "cube.obj" else dodge
    n = case["size"]
    rgb, _, _ = scene.render(oracle.camera(n, n), oracle.lights(area=bool(case["area"])), n, n, max_depth=-1, threads=8)
    out = tmp_path / "result.ppm"
    assert scene.write_ppm(out, rgb) == 1
    assert hashlib.md5(out.read_bytes()).hexdigest() == case["md5"]
