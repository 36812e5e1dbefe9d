"""Round-3 parity additions.

* The reference's own full-size `result.ppm` md5s (src/flyscene.cpp:641 writes the file) reproduced by the HIP path:
  cube.obj 1000^2 point / area(25), cube.obj 1440^2 area(25), dodgeColorTest.obj 1440^2 area(25) -- the pixel-count twins of the headline
  configuration at the reference's 25 samples.  The oracle reproduces the same seven md5s on the CPU (tests/test_oracle_golden.py).
* bunny.ply (resources/models/bunny.ply, 69,451 faces, 361 nodes): the one deep-tree PLY the reference ships, through loader, device tree
  build and a whole frame.
* Light grids the earlier tests never rendered: usteps != vsteps, multi-pass grids that are not blocks of 8 x 8, pass / block-row / light-slot
  counts that are not powers of two (k_shadow_shaft's float-reciprocal division, rt_kernels.hip: udiv).
"""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SCENES = os.path.join(HERE, "golden", "scenes")
KA = json.load(open(os.path.join(HERE, "golden", "survey_known_answers.json")))


def render_gpu(rt, ctx, cam, L, w, h, depth):
    p = rt.make_params(w, h, depth)
    rgb = np.zeros((h, w, 3), np.float32)
    hits = np.zeros((h, w), np.int32)
    st = ctx.lib.rt_render(ctx.handle, C.byref(cam), C.byref(L), C.byref(p), rgb.ctypes.data_as(C.c_void_p), hits.ctypes.data_as(C.c_void_p), None)
    rt.capi.check(ctx.lib, ctx.handle, st, "rt_render")
    return rgb, hits


def assert_exact(rgb, hits, ref, rhits):
    assert np.array_equal(hits, rhits), f"{int((hits != rhits).sum())} closest-hit face ids differ"
    assert np.array_equal(rgb.view(np.uint32), ref.view(np.uint32)), f"max |RGB - oracle| = {float(np.abs(rgb - ref).max())}"
    assert (rhits >= 0).sum() > 0.02 * rhits.size, "the frame must actually show the object"


# ------------------------------------------------------------------------------------------ the reference's full-size md5s on the GPU
@pytest.mark.gpu
@pytest.mark.parametrize("scene,size,area", [("cube.obj", 1000, 0), ("cube.obj", 1000, 1), ("cube.obj", 1440, 1), ("dodgeColorTest.obj", 1440, 1)])
def test_reference_full_size_md5(rt, scene, size, area, tmp_path):
    """raytraceScene(W, W) + writePPMImage at the reference's own settings (5 x 5 samples or the point light, natural depth): the bytes of
    result.ppm hash to the value the unmodified reference produced."""
    want = [c["md5"] for c in KA["result_ppm_md5"] if c["scene"] == scene and c["size"] == size and c["area"] == area]
    assert len(want) == 1
    hs = rt.HostScene(os.path.join(SCENES, scene), 1000, 15)
    ctx = rt.Context(0)
    ctx.upload(hs)
    L = rt.make_lights(area=bool(area), usteps=5, vsteps=5)
    rgb, hits = render_gpu(rt, ctx, rt.default_camera(size, size), L, size, size, -1)
    out = tmp_path / "result.ppm"
    assert ctx.lib.rt_write_ppm(str(out).encode(), rgb.ctypes.data_as(C.c_void_p), size, size) == 0
    assert hashlib.md5(out.read_bytes()).hexdigest() == want[0]
    ctx.close(); hs.close()


# ------------------------------------------------------------------------------------------ bunny.ply
def test_bunny_ply_host_scene_equals_oracle(rt, oracle):
    """Product loader + host tree build vs the oracle's, array for array (69,451 faces -> 361 nodes / 288 leaves / 81,299 face refs)."""
    path = os.path.join(SCENES, "bunny.ply")
    hs = rt.HostScene(path, 1000, 15)
    osc = oracle.load_scene(path)
    a, oa = hs.arrays(), osc.arrays()
    tri = oa["wverts"][oa["face_vid"].reshape(-1)].reshape(-1, 9)
    assert np.array_equal(a["tri_verts"].view(np.uint32), tri.view(np.uint32))
    assert np.array_equal(a["vert_normal"].view(np.uint32), oa["normals"].view(np.uint32))
    assert np.array_equal(a["face_normal"].view(np.uint32), oa["face_normal"].view(np.uint32))
    info = hs.info()
    assert (info["nodes"], info["leaves"], info["face_refs"]) == (361, 288, 81299) and info["nodes"] == osc.nnodes
    osc.close(); hs.close()


@pytest.mark.gpu
@pytest.mark.parametrize("u,depth", [(5, 2), (8, 2)])
def test_bunny_ply_renders_like_the_oracle(rt, oracle, u, depth):
    """A frame of the deep-tree PLY: 25 samples (two pairs per wave, stack walk) and 64 samples (shaft walk).  Parity unpinned beyond the
    oracle: the reference's PLY importer never builds faces (plyimporter.hpp:186-262)."""
    path = os.path.join(SCENES, "bunny.ply")
    hs = rt.HostScene(path, 1000, 15)
    ctx = rt.Context(0)
    ctx.upload(hs)
    osc = oracle.load_scene(path)
    w, h = 240, 168
    rgb, hits = render_gpu(rt, ctx, rt.default_camera(w, h), rt.make_lights(area=True, usteps=u, vsteps=u), w, h, depth)
    ref, rhits, _ = osc.render(oracle.camera(w, h), oracle.lights(area=True, usteps=u, vsteps=u), w, h, max_depth=depth, threads=8, want_hits=True)
    assert_exact(rgb, hits, ref, rhits)
    osc.close(); ctx.close(); hs.close()


@pytest.mark.gpu
def test_bunny_ply_gpu_octree_build_equals_host_build(rt):
    path = os.path.join(SCENES, "bunny.ply")
    cpu = rt.HostScene(path, 1000, 15)
    gpu = rt.HostScene(path, 1000, 15)
    ctx = rt.Context(0)
    gpu.build_gpu(ctx, 1000, 15)
    a, b = cpu.arrays(), gpu.arrays()
    assert cpu.info() == gpu.info() and cpu.info()["nodes"] == 361
    assert np.array_equal(a["node_box"].view(np.uint32), b["node_box"].view(np.uint32))
    for k in ("node_first", "node_count_flags", "face_refs"):
        assert np.array_equal(a[k], b[k]), k
    ctx.close(); cpu.close(); gpu.close()


# ------------------------------------------------------------------------------------------ light grids
GRIDS = [(12, 12), (24, 24), (8, 16), (16, 8), (5, 8), (9, 7)]


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["cube.obj", "dodgeColorTest.obj"])
@pytest.mark.parametrize("us,vs", GRIDS)
def test_light_grids_that_are_not_square_powers_of_two(rt, oracle, scene, us, vs):
    """usteps x vsteps = 12 x 12 (P = 3 passes of 64, no blocks), 24 x 24 (blocks: 9 passes, 3 blocks per row), 8 x 16 / 16 x 8 (2 passes,
    blocks, bpr 2 / 1: i and j must not be swapped), 5 x 8 and 9 x 7 (40 / 63 samples in one word; 9 x 7: two-pair packing does not apply).
    Frame == oracle bit for bit."""
    path = os.path.join(SCENES, scene)
    hs = rt.HostScene(path, 1000, 15)
    ctx = rt.Context(0)
    ctx.upload(hs)
    osc = oracle.load_scene(path)
    w, h, depth = 176, 120, 2
    rgb, hits = render_gpu(rt, ctx, rt.default_camera(w, h), rt.make_lights(area=True, usteps=us, vsteps=vs), w, h, depth)
    ref, rhits, _ = osc.render(oracle.camera(w, h), oracle.lights(area=True, usteps=us, vsteps=vs), w, h, max_depth=depth, threads=8, want_hits=True)
    assert_exact(rgb, hits, ref, rhits)
    osc.close(); ctx.close(); hs.close()


@pytest.mark.gpu
@pytest.mark.parametrize("scene,us,vs,beam_trees", [("cube.obj", 12, 12, False), ("dodgeColorTest.obj", 12, 12, False), ("dodgeColorTest.obj", 12, 12, True),
                                                    ("dodgeColorTest.obj", 24, 24, False), ("dodgeColorTest.obj", 16, 8, True)])
def test_three_lights_with_non_power_of_two_passes(rt, oracle, scene, us, vs, beam_trees, monkeypatch):
    """Three lights (lslots = 3) at 12 x 12 (P = 3): both run-time divisions of a shaft unit go through the float-reciprocal path; 24 x 24
    adds bpr = 3.  Once more with the beam test on the tree (RT_BEAM_TREES=1)."""
    if beam_trees:
        monkeypatch.setenv("RT_BEAM_TREES", "1")
    pts = [(-1.0, 1.0, 1.0), (0.8, 0.4, 1.5), (0.0, 0.0, 2.0)]
    path = os.path.join(SCENES, scene)
    hs = rt.HostScene(path, 1000, 15)
    ctx = rt.Context(0)
    ctx.upload(hs)
    osc = oracle.load_scene(path)
    w, h, depth = 144, 96, 2
    rgb, hits = render_gpu(rt, ctx, rt.default_camera(w, h), rt.make_lights(points=pts, area=True, usteps=us, vsteps=vs), w, h, depth)
    ref, rhits, _ = osc.render(oracle.camera(w, h), oracle.lights(area=True, usteps=us, vsteps=vs, points=pts), w, h, max_depth=depth, threads=8, want_hits=True)
    assert_exact(rgb, hits, ref, rhits)
    osc.close(); ctx.close(); hs.close()


# ------------------------------------------------------------------------------------------ captured graphs own their sphere offsets
@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["cube.obj", "dodgeColorTest.obj"])
def test_sphere_graph_survives_later_sphere_calls(rt, scene):
    """A hipGraph captured in RT_LIGHT_SPHERE mode keeps rendering ITS offsets after later eager sphere-mode calls with different -- and
    with more -- offsets (the context's buffer is rewritten, then reallocated: a graph that held that pointer would render other samples or
    read freed memory)."""
    path = os.path.join(SCENES, scene)
    hs = rt.HostScene(path, 1000, 15)
    ctx = rt.Context(0)
    ctx.upload(hs)
    w, h = 160, 96
    cam = rt.default_camera(w, h)
    p = rt.make_params(w, h, 2)
    # (size the frame buffers for four visibility words per pair first: a later growth would -- rightly -- invalidate the graph)
    render_gpu(rt, ctx, cam, rt.make_lights(area=True, usteps=16, vsteps=16), w, h, 2)
    off_a = rt.sphere_offsets(65, 1.0, 64)
    L_a = rt.set_sphere(rt.make_lights(area=False), off_a)
    want, _ = render_gpu(rt, ctx, cam, L_a, w, h, 2)
    out = rt.hipmem.DeviceBuffer(h * w * 3 * 4)
    g = rt.FrameGraph(ctx, L_a, p, out.address, 0)
    g.launch(cam); g.stats()
    assert np.array_equal(out.to_numpy(np.float32, (h, w, 3)).view(np.uint32), want.view(np.uint32))
    # same size, other offsets: the context's buffer is rewritten in place
    L_b = rt.set_sphere(rt.make_lights(area=False), rt.sphere_offsets(7, 0.5, 64))
    other, _ = render_gpu(rt, ctx, cam, L_b, w, h, 2)
    assert not np.array_equal(other, want)
    g.launch(cam); g.stats()
    assert np.array_equal(out.to_numpy(np.float32, (h, w, 3)).view(np.uint32), want.view(np.uint32))
    # more offsets: the context's buffer is reallocated
    L_c = rt.set_sphere(rt.make_lights(area=False), rt.sphere_offsets(9, 1.0, 256))
    render_gpu(rt, ctx, cam, L_c, w, h, 2)
    g.launch(cam); g.stats()
    assert np.array_equal(out.to_numpy(np.float32, (h, w, 3)).view(np.uint32), want.view(np.uint32))
    g.close(); out.free(); ctx.close(); hs.close()


# ------------------------------------------------------------------------------------------ k_deep: levels 2.. of flat scenes in one launch
@pytest.mark.gpu
@pytest.mark.parametrize("which,u,depth,lights", [("mixed", 5, 4, 1), ("mixed", 8, 8, 1), ("mixed", 4, 6, 3), ("cube", 8, 4, 1), ("mixed", 12, 5, 2)])
def test_deep_levels_in_one_launch_equal_the_wide_kernels_and_the_oracle(rt, oracle, tmp_path, monkeypatch, which, u, depth, lights):
    """Flat scenes run their bounce levels >= 2 in ONE launch (k_deep: a wave carries 64 rays through all remaining levels, every sample's
    lightStrikes segment walked in place).  The frame, the hit ids and the ray counters equal the four-launches-per-level path (RT_NO_DEEP=1)
    and the oracle bit for bit -- on the mixed-material scene (illum 2/3/5/6/7/9: populated deep levels, own light lists after a mirror
    bounce, pass-through and refraction chains), with one word, several words (12 x 12) and several lights."""
    import scenes_gen
    path = scenes_gen.mixed_materials(str(tmp_path)) if which == "mixed" else os.path.join(SCENES, "cube.obj")
    pts = [(-1.0, 1.0, 1.0), (0.8, 0.4, 1.5), (0.0, 0.0, 2.0)][:lights]
    w, h = 224, 152
    cam, L = rt.default_camera(w, h, 0.4 if which == "mixed" else 0.0), rt.make_lights(points=pts, area=True, usteps=u, vsteps=u)
    hs = rt.HostScene(path, 1000, 15)
    frames, stats = [], []
    for no_deep in (False, True):
        if no_deep:
            monkeypatch.setenv("RT_NO_DEEP", "1")
        ctx = rt.Context(0)
        ctx.upload(hs)
        p = rt.make_params(w, h, depth)
        rgb = np.zeros((h, w, 3), np.float32)
        hits = np.zeros((h, w), np.int32)
        st = rt.capi.rt_stats()
        rc = ctx.lib.rt_render(ctx.handle, C.byref(cam), C.byref(L), C.byref(p), rgb.ctypes.data_as(C.c_void_p), hits.ctypes.data_as(C.c_void_p), C.byref(st))
        rt.capi.check(ctx.lib, ctx.handle, rc, "rt_render")
        frames.append((rgb, hits))
        stats.append((st.rays_primary, st.rays_bounce, st.rays_centre, st.rays_sample, st.shaded_hits, st.launches_total))
        ctx.close()
    assert np.array_equal(frames[0][0].view(np.uint32), frames[1][0].view(np.uint32)) and np.array_equal(frames[0][1], frames[1][1])
    assert stats[0][:5] == stats[1][:5]
    if depth >= 3:
        assert stats[0][5] < stats[1][5]                       # fewer launches per frame
    osc = oracle.load_scene(path)
    ref, rhits, ost = osc.render(oracle.camera(w, h, 0.4 if which == "mixed" else 0.0), oracle.lights(area=True, usteps=u, vsteps=u, points=pts), w, h,
                                 max_depth=depth, threads=8, want_hits=True)
    assert_exact(frames[0][0], frames[0][1], ref, rhits)
    assert (stats[0][1], stats[0][2], stats[0][3]) == (ost.rays_bounce, ost.rays_centre, ost.rays_sample)
    if which == "mixed":
        assert ost.rays_bounce > 0
    osc.close(); hs.close()


# ------------------------------------------------------------------------------------------ BASELINE cfg5 on one GPU, full size
@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["cube.obj", "dodgeColorTest.obj"])
def test_cfg5_animation_120_frames_1080p_graph_replay(rt, oracle, scene):
    """BASELINE cfg5's per-GPU share at full size: 120 frames at 1920 x 1080, depth 4, 64 samples, the fly camera yawing 2 pi / 120 per frame
    (flycamera.hpp:166-191), ONE captured hipGraph replayed per frame.  Every frame is produced without an error; frames 0, 7, 60 and 113 (the object in view at 0, 7 and 113; the camera looks away from it at 60) equal the
    oracle's frame of that camera bit for bit (float RGB) and their 8-bit rows equal writePPMImage's quantisation; frame 120 (yaw 2 pi in
    float) reproduces the eager frame of the same camera."""
    w, h, frames = 1920, 1080, 120
    path = os.path.join(SCENES, scene)
    hs = rt.HostScene(path, 1000, 15)
    ctx = rt.Context(0)
    ctx.upload(hs)
    L = rt.make_lights(area=True, usteps=8, vsteps=8)
    p = rt.make_params(w, h, 4)
    out = rt.hipmem.DeviceBuffer(h * w * 3 * 4)
    out8 = rt.hipmem.DeviceBuffer(h * w * 3)
    g = rt.FrameGraph(ctx, L, p, out.address, out8.address)
    osc = oracle.load_scene(path)
    shas = []
    for f in range(frames + 1):
        yaw = float(np.float32(2.0 * np.pi * f / frames))
        g.launch(rt.default_camera(w, h, yaw))
        if f in (0, 7, 60, 113, frames):
            g.stats()                                   # synchronises
            got = out.to_numpy(np.float32, (h, w, 3))
            got8 = out8.to_numpy(np.uint8, (h, w, 3))
            shas.append(hashlib.sha256(got8.tobytes()).hexdigest())
            if f < frames:
                ref, _, _ = osc.render(oracle.camera(w, h, yaw), oracle.lights(area=True, usteps=8, vsteps=8), w, h, max_depth=4, threads=8)
                assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (scene, f, float(np.abs(got - ref).max()))
                assert np.array_equal(got8, np.clip(oracle.quantise(ref), 0, 255).astype(np.uint8))
            else:
                eager, _ = render_gpu(rt, ctx, rt.default_camera(w, h, yaw), L, w, h, 4)
                assert np.array_equal(got.view(np.uint32), eager.view(np.uint32))
    st = g.stats()
    assert st.total_rays() > 0 and len(set(shas[:4])) == 4          # the camera really moved
    g.close(); out.free(); out8.free(); osc.close(); ctx.close(); hs.close()


# ------------------------------------------------------------------------------------------ the per-hit beam test of tree scenes (k_pair_beam)
def _render_stats(rt, ctx, cam, L, w, h, depth):
    p = rt.make_params(w, h, depth)
    rgb = np.zeros((h, w, 3), np.float32)
    hits = np.zeros((h, w), np.int32)
    st = rt.capi.rt_stats()
    rc = ctx.lib.rt_render(ctx.handle, C.byref(cam), C.byref(L), C.byref(p), rgb.ctypes.data_as(C.c_void_p), hits.ctypes.data_as(C.c_void_p), C.byref(st))
    rt.capi.check(ctx.lib, ctx.handle, rc, "rt_render")
    return rgb, hits, st


PAIR_CASES = [("dodgeColorTest.obj", "grid", 16, 16, 1, 3),      # blocks of 8 x 8, four passes: the case it is on for by default
              ("dodgeColorTest.obj", "grid", 10, 10, 2, 2),      # 100 samples in two strips (no blocks), two lights
              ("dodgeColorTest.obj", "grid", 8, 8, 1, 4),        # one pass: only with RT_ITEM_BEAM=2; mirror bounces carry light lists of their own
              ("dodgeColorTest.obj", "sphere", 100, 0, 1, 2),    # seeded sphere samples (their box comes from the offsets)
              ("bunny.ply", "grid", 16, 8, 1, 2),                # a deep tree (361 nodes: groups beyond the LDS copy of the top)
              ("wavy", "grid", 16, 16, 1, 2),                    # the height field of cfg4 (grazing light: many beams blocked)
              ("mixed", "grid", 12, 12, 2, 5)]                   # every material branch, deep bounces


@pytest.mark.gpu
@pytest.mark.parametrize("scene,kind,us,vs,n_lights,depth", PAIR_CASES)
def test_pair_beam_on_equals_off_equals_oracle(rt, oracle, tmp_path, monkeypatch, scene, kind, us, vs, n_lights, depth):
    """k_pair_beam (one beam per lit hit in front of k_shadow_shaft) decides most (hit, light) pairs without forming a sample ray; the frame,
    the closest-hit ids and the sample-ray count must not depend on it: RT_ITEM_BEAM=2 (always) == RT_ITEM_BEAM=0 (never) == the oracle."""
    import scenes_gen
    if scene == "wavy":
        path = scenes_gen.wavy_grid(str(tmp_path), n=96)
    elif scene == "mixed":
        path = scenes_gen.mixed_materials(str(tmp_path))
    else:
        path = os.path.join(SCENES, scene)
    pts = [(-1.0, 1.0, 1.0), (0.8, 0.4, 1.5)][:n_lights]
    w, h = 160, 100
    if kind == "sphere":
        off = rt.sphere_offsets(77, 0.3, us)
        L = rt.set_sphere(rt.make_lights(points=pts, area=False), off)
        oL = oracle.lights(area=False, points=pts)
        oL.mode = 2
        oL.n_offsets = us
        oL.offsets = off.ctypes.data_as(C.POINTER(C.c_float))
    else:
        L = rt.make_lights(points=pts, area=True, usteps=us, vsteps=vs)
        oL = oracle.lights(area=True, usteps=us, vsteps=vs, points=pts)
    # a small leaf capacity makes a real tree of the generated scenes too (mixed at 16: 25 nodes, one of them lost, two unreachable faces)
    cap = 1000 if scene.endswith((".obj", ".ply")) else (16 if scene == "mixed" else 64)
    hs = rt.HostScene(path, cap, 15)
    frames = {}
    for mode in ("2", "0"):
        monkeypatch.setenv("RT_ITEM_BEAM", mode)
        ctx = rt.Context(0)
        ctx.upload(hs)
        frames[mode] = _render_stats(rt, ctx, rt.default_camera(w, h), L, w, h, depth)
        ctx.close()
    (rgb, hits, st), (rgb0, hits0, st0) = frames["2"], frames["0"]
    assert np.array_equal(hits, hits0) and np.array_equal(rgb.view(np.uint32), rgb0.view(np.uint32))
    assert st.rays_sample == st0.rays_sample and st.rays_bounce == st0.rays_bounce and st.shaded_hits == st0.shaded_hits
    assert st.rays_sample_walked < st0.rays_sample_walked, "the beams must decide some pairs"
    osc = oracle.load_scene(path, cap, 15)
    ref, rhits, _ = osc.render(oracle.camera(w, h), oL, w, h, max_depth=depth, threads=8, want_hits=True)
    assert_exact(rgb, hits, ref, rhits)
    osc.close(); hs.close()


@pytest.mark.gpu
def test_graph_replay_with_pair_beams_in_the_captured_frame(rt, oracle):
    """A tree scene under a 16 x 16 light: the captured launch sequence holds k_pair_beam + k_shadow_shaft over the survivor list at every level.
    Twelve replays with a yawing camera: frames 0, 5 and 11 equal the oracle bit for bit, and a replay equals the eager frame of its camera."""
    w, h, depth = 240, 136, 3
    path = os.path.join(SCENES, "dodgeColorTest.obj")
    hs = rt.HostScene(path, 1000, 15)
    ctx = rt.Context(0)
    ctx.upload(hs)
    L = rt.make_lights(area=True, usteps=16, vsteps=16)
    p = rt.make_params(w, h, depth)
    out = rt.hipmem.DeviceBuffer(h * w * 3 * 4)
    g = rt.FrameGraph(ctx, L, p, out.address, 0)
    osc = oracle.load_scene(path)
    for f in range(12):
        yaw = float(np.float32(0.05 * f))
        g.launch(rt.default_camera(w, h, yaw))
        if f in (0, 5, 11):
            st = g.stats()                                   # synchronises
            got = out.to_numpy(np.float32, (h, w, 3))
            ref, _, _ = osc.render(oracle.camera(w, h, yaw), oracle.lights(area=True, usteps=16, vsteps=16), w, h, max_depth=depth, threads=8)
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (f, float(np.abs(got - ref).max()))
            assert 0 < st.rays_sample_walked < st.rays_sample            # the beams decided pairs inside the replayed graph too
    eager, _ = render_gpu(rt, ctx, rt.default_camera(w, h, float(np.float32(0.05 * 11))), L, w, h, depth)
    assert np.array_equal(out.to_numpy(np.float32, (h, w, 3)).view(np.uint32), eager.view(np.uint32))
    g.close(); out.free(); osc.close(); ctx.close(); hs.close()


LIGHT_SPOTS = [(0.0, 0.0, 0.3), (0.3, 0.1, 0.05), (-0.4, 0.2, -0.1), (0.0, 0.45, 0.0), (40.0, 25.0, 60.0), (-0.9, -0.9, 0.02), (0.05, 0.02, 1.0e-3)]


@pytest.mark.gpu
@pytest.mark.parametrize("us", [8, 16])
@pytest.mark.parametrize("spot", LIGHT_SPOTS)
def test_lights_inside_close_to_and_far_from_the_model(rt, oracle, spot, us, monkeypatch):
    """The shaft constructions (tangent planes through the hit, the near box without its tip, the far cone, the chunk slabs) with the light where
    they degenerate: inside the model's bounding box, a hair above a surface, level with the hits along an axis (no tip to cut, no tangent in a
    projection), far outside.  Culled == RT_NO_CULL=1 == oracle, one pass (units only) and four passes (k_pair_beam)."""
    path = os.path.join(SCENES, "dodgeColorTest.obj")
    hs = rt.HostScene(path, 1000, 15)
    w, h, depth = 128, 80, 2
    L = rt.make_lights(points=[spot], area=True, usteps=us, vsteps=us)
    frames = []
    for no_cull in (False, True):
        if no_cull:
            monkeypatch.setenv("RT_NO_CULL", "1")
        ctx = rt.Context(0)
        ctx.upload(hs)
        frames.append(render_gpu(rt, ctx, rt.default_camera(w, h), L, w, h, depth))
        ctx.close()
    monkeypatch.delenv("RT_NO_CULL", raising=False)
    assert np.array_equal(frames[0][1], frames[1][1]) and np.array_equal(frames[0][0].view(np.uint32), frames[1][0].view(np.uint32))
    osc = oracle.load_scene(path)
    ref, rhits, _ = osc.render(oracle.camera(w, h), oracle.lights(area=True, usteps=us, vsteps=us, points=[spot]), w, h, max_depth=depth, threads=8, want_hits=True)
    assert_exact(frames[0][0], frames[0][1], ref, rhits)
    osc.close(); hs.close()


@pytest.mark.gpu
@pytest.mark.parametrize("us,vs,n_lights", [(32, 32, 1), (12, 12, 25), (16, 24, 3)])
def test_pair_beam_extremes_many_passes_many_lights(rt, oracle, us, vs, n_lights):
    """16 passes per pair (32 x 32 samples: lane = pass writes 16 visibility words), the reference's maximum of 25 lights (25 light slots per
    hit, three passes each), a non-square block grid (16 x 24: six passes, three blocks per row): dodge tree == oracle bit for bit."""
    rng = np.random.default_rng(5)
    pts = [(-1.0, 1.0, 1.0)] + [tuple(float(x) for x in rng.uniform(-1.2, 1.2, 3) + np.array([0.0, 0.0, 1.3])) for _ in range(n_lights - 1)]
    path = os.path.join(SCENES, "dodgeColorTest.obj")
    hs = rt.HostScene(path, 1000, 15)
    ctx = rt.Context(0)
    ctx.upload(hs)
    osc = oracle.load_scene(path)
    w, h, depth = 64, 40, 1
    rgb, hits, st = _render_stats(rt, ctx, rt.default_camera(w, h), rt.make_lights(points=pts, area=True, usteps=us, vsteps=vs), w, h, depth)
    ref, rhits, _ = osc.render(oracle.camera(w, h), oracle.lights(area=True, usteps=us, vsteps=vs, points=pts), w, h, max_depth=depth, threads=8, want_hits=True)
    assert_exact(rgb, hits, ref, rhits)
    assert 0 < st.rays_sample_walked < st.rays_sample
    osc.close(); ctx.close(); hs.close()


@pytest.mark.gpu
def test_synchronize_after_asynchronous_frames(rt, oracle):
    """rt_render_device without a stats record is asynchronous; rt_synchronize waits for it.  Forty frames back to back on the context's stream,
    one wait, the device buffer equals the oracle's frame."""
    w, h, depth = 320, 200, 3
    path = os.path.join(SCENES, "dodgeColorTest.obj")
    hs = rt.HostScene(path, 1000, 15)
    ctx = rt.Context(0)
    ctx.upload(hs)
    cam, L, p = rt.default_camera(w, h), rt.make_lights(area=True, usteps=8, vsteps=8), rt.make_params(w, h, depth)
    out = rt.hipmem.DeviceBuffer(h * w * 3 * 4)
    for _ in range(40):
        rt.capi.check(ctx.lib, ctx.handle, ctx.lib.rt_render_device(ctx.handle, C.byref(cam), C.byref(L), C.byref(p), C.c_void_p(out.address), None, None, None, None), "render")
    rt.capi.check(ctx.lib, ctx.handle, ctx.lib.rt_synchronize(ctx.handle), "rt_synchronize")
    got = out.to_numpy(np.float32, (h, w, 3))
    osc = oracle.load_scene(path)
    ref, _, _ = osc.render(oracle.camera(w, h), oracle.lights(area=True, usteps=8, vsteps=8), w, h, max_depth=depth, threads=8)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    osc.close(); out.free(); ctx.close(); hs.close()
