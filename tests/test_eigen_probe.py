"""Pins the oracle's (and, through tests/test_host_scene.py, the product's) float conventions against the
reference's REAL Eigen 3.3.7 and GL-free Tucano headers: tests/golden/eigen_probe.json was produced by
oracle/ref_probe.cpp compiled in place against /root/reference (generating script committed; the binary is not).
CPU only."""
import ctypes as C
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
PROBE = json.load(open(os.path.join(HERE, "golden", "eigen_probe.json")))


def h2f(hexes):
    return np.array([int(h, 16) for h in hexes], np.uint32).view(np.float32)


def test_vector_conventions_match_eigen(oracle):
    lib = oracle.lib
    lib.orc_vec_ops.argtypes = [C.POINTER(C.c_float)] * 3 + [C.c_float] * 4 + [C.c_int, C.c_int, C.c_float, C.POINTER(C.c_float)]
    order = [("dot", 1), ("sqn", 1), ("cross", 3), ("normalized", 3), ("head3_diff_normalized", 3), ("head3_minus_fixed_norm", 1),
             ("blend_015_085", 3), ("blend_010_090", 3), ("blend_02_08", 3), ("reflect", 3), ("bary", 3), ("phong_accum", 3),
             ("area_scale", 3), ("octant_max", 3), ("world_vertex", 3)]
    for i, rec in enumerate(PROBE["vec"]):
        a, b, c = h2f(rec["a"]), h2f(rec["b"]), h2f(rec["c"])
        u, v, w = (h2f([rec[k]])[0] for k in ("u", "v", "w"))
        steps = int(h2f([rec["steps"]])[0])
        scale = h2f([rec["scale"]])[0]
        out = np.zeros(64, np.float32)
        fp = lambda x: x.ctypes.data_as(C.POINTER(C.c_float))
        lib.orc_vec_ops(fp(a), fp(b), fp(c), float(u), float(v), float(w), float(i % 26), steps, i, float(scale), fp(out))
        k = 0
        for name, n in order:
            want = h2f(rec[name]) if n == 3 else h2f([rec[name]])
            got = out[k:k + n]
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (i, name, got, want)
            k += n
        assert rec["normalize_inplace"] == rec["normalized"]


def test_dynamic_and_fixed_reductions_differ_somewhere():
    """The head(3)-block normalisation really uses a different summation order than Vector3f (why the oracle
    keeps two normalise helpers): at least one probe vector shows different bits."""
    diff = 0
    for rec in PROBE["vec"]:
        a, b = h2f(rec["a"]), h2f(rec["b"])
        d = (a - b).astype(np.float32)
        f = np.float32
        z_fixed = f(d[0] * d[0]) + f(f(d[1] * d[1]) + f(d[2] * d[2]))
        z_dyn = f(f(d[0] * d[0]) + f(d[1] * d[1])) + f(d[2] * d[2])
        diff += int(f(z_fixed) != f(z_dyn))
    assert diff > 0


def test_default_view_matches_eigen(oracle):
    dv = PROBE["default_view"]
    cam = oracle.camera(640, 480)
    assert np.array_equal(h2f(dv["center"]), np.array(list(cam.center), np.float32))
    assert np.array_equal(np.array(dv["inverse_rows"], np.float32).ravel(), np.array(list(cam.inv_view), np.float32))


def test_area_light_matches_reference_class(oracle):
    for rec in PROBE["arealight"]:
        l = oracle.lights(area=True, usteps=rec["steps"], vsteps=rec["steps"])
        got = oracle.light_samples(l, h2f(rec["corner"]))
        want = np.stack([h2f(p) for p in rec["points"]])
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), rec["steps"]


@pytest.mark.parametrize("mtl,obj", [("cube.mtl", "cube.obj"), ("dodgeColorTest.mtl", "dodgeColorTest.obj")])
def test_mtl_loader_matches_reference_loader(oracle, scenes, mtl, obj):
    osc = oracle.load_scene(os.path.join(scenes, obj))
    mats = osc.materials()
    want = PROBE["mtl"][mtl]
    assert len(mats) == len(want)
    for (f, il), w in zip(mats, want):
        assert np.array_equal(f[:3].view(np.uint32), h2f(w["kd"]).view(np.uint32))
        assert np.array_equal(f[3:6].view(np.uint32), h2f(w["ks"]).view(np.uint32))
        assert f[6] == h2f([w["ns"]])[0] and f[7] == h2f([w["ni"]])[0] and il == w["illum"]
    osc.close()


def test_orphan_mtl_files_parse_like_the_reference(oracle, scenes, tmp_path):
    """planecube/simplemirror/mirror/glassstraw .mtl (their .obj were never shipped): load through a one-triangle OBJ."""
    for mtl in ("planecube.mtl", "simplemirror.mtl", "mirror.mtl", "glassstraw.mtl"):
        d = tmp_path / mtl.replace(".", "_")
        d.mkdir()
        (d / mtl).write_bytes(open(os.path.join(scenes, mtl), "rb").read())
        (d / "t.obj").write_text(f"mtllib {mtl}\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
        osc = oracle.load_scene(str(d / "t.obj"))
        want = PROBE["mtl"][mtl]
        mats = osc.materials()
        assert len(mats) == len(want)
        for (f, il), w in zip(mats, want):
            assert np.array_equal(f[:3].view(np.uint32), h2f(w["kd"]).view(np.uint32))
            assert np.array_equal(f[3:6].view(np.uint32), h2f(w["ks"]).view(np.uint32))
            assert f[6] == h2f([w["ns"]])[0] and f[7] == h2f([w["ni"]])[0] and il == w["illum"]
        osc.close()
