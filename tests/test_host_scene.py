"""Host-side scene preparation of the PRODUCT (raytracer-in-cpp_amd/csrc/host_scene.cpp, through the C ABI)
against the CPU oracle: loader, normalisation, vertex/face normals, bug-compatible octree, flattening.  CPU only;
no compute entry point is called."""
import os

import numpy as np
import pytest

RT_NODE_LEAF = 0x80000000


def flat_tree_from_oracle(osc):
    """Flatten the oracle's pointer tree the way the reference's BFS walks it: isEmpty children are never visited."""
    order, out = [0], []
    head = 0
    while head < len(order):
        n = osc.node(order[head])
        head += 1
        if n["is_leaf"] and not n["is_empty"]:
            out.append(("leaf", n["box"], n["faces"]))
        elif n["is_empty"]:
            out.append(("inner", n["box"], 0))
        else:
            live = [c for c in n["children"][: n["nchildren"]] if not osc.node(c)["is_empty"]]
            out.append(("inner", n["box"], len(live)))
            order.extend(live)
    return out


@pytest.mark.parametrize("name", ["cube.obj", "dodgeColorTest.obj"])
def test_scene_arrays_match_oracle(rt, oracle, scenes, name):
    path = os.path.join(scenes, name)
    hs = rt.HostScene(path)
    osc = oracle.load_scene(path)
    a, o = hs.arrays(), osc.arrays()
    assert a["tri_vid"].shape[0] == osc.nfaces
    assert np.array_equal(a["tri_vid"], o["face_vid"])
    assert np.array_equal(a["mat_id"], o["face_mat"])
    # bit-exact floats
    assert np.array_equal(a["face_normal"].view(np.uint32), o["face_normal"].view(np.uint32))
    assert np.array_equal(a["vert_normal"].view(np.uint32), o["normals"].view(np.uint32))
    world = o["wverts"][o["face_vid"].astype(np.int64)].reshape(-1, 9)
    assert np.array_equal(a["tri_verts"].view(np.uint32), world.view(np.uint32))
    mats = osc.materials()
    assert len(mats) == a["mat_f"].shape[0]
    for i, (f, il) in enumerate(mats):
        assert np.array_equal(a["mat_f"][i].view(np.uint32), f.view(np.uint32)) and a["mat_illum"][i] == il
    # flattened octree == BFS of the oracle's tree
    want = flat_tree_from_oracle(osc)
    assert len(want) == a["node_box"].shape[0]
    for i, (kind, box, payload) in enumerate(want):
        assert np.array_equal(a["node_box"][i].view(np.uint32), box.view(np.uint32)), i
        cf = int(a["node_count_flags"][i])
        if kind == "leaf":
            assert cf & RT_NODE_LEAF
            first, cnt = int(a["node_first"][i]), cf & 0x7FFFFFFF
            assert np.array_equal(a["face_refs"][first:first + cnt].astype(np.int32), payload)
        else:
            assert not (cf & RT_NODE_LEAF) and (cf & 0x7FFFFFFF) == payload
    info = hs.info()
    assert info["nodes"] == osc.nnodes
    hs.close()
    osc.close()


def test_small_capacity_tree_matches_oracle(rt, oracle, scenes):
    """Deeper trees (capacity 200 -> depth up to 15 bound) exercise the split/classify recursion and lost-face rule."""
    path = os.path.join(scenes, "dodgeColorTest.obj")
    hs = rt.HostScene(path, leaf_capacity=200, max_depth=15)
    osc = oracle.load_scene(path, capacity=200, maxdepth=15)
    a = hs.arrays()
    want = flat_tree_from_oracle(osc)
    assert len(want) == a["node_box"].shape[0]
    nleaf = 0
    for i, (kind, box, payload) in enumerate(want):
        assert np.array_equal(a["node_box"][i].view(np.uint32), box.view(np.uint32))
        if kind == "leaf":
            nleaf += 1
            first, cnt = int(a["node_first"][i]), int(a["node_count_flags"][i]) & 0x7FFFFFFF
            assert np.array_equal(a["face_refs"][first:first + cnt].astype(np.int32), payload)
    assert nleaf > 148
    hs.close()
    osc.close()


@pytest.mark.parametrize("seed,n_tri,cap", [(7, 3000, 1000), (8, 800, 250), (10, 1500, 250), (12, 40, 1000), (17, 1200, 300), (21, 700, 200)])
def test_random_soup_scene_matches_oracle(rt, oracle, tmp_path, seed, n_tri, cap):
    """Seeded triangle soups (slivers, zero-area and duplicated triangles, shared vertices, six materials): loader,
    normalisation, normals and the bug-compatible octree of the product are bit-identical to the oracle's."""
    import scenes_gen
    path = scenes_gen.random_soup(str(tmp_path), seed, n_tri)
    hs = rt.HostScene(path, leaf_capacity=cap, max_depth=15)
    osc = oracle.load_scene(path, capacity=cap, maxdepth=15)
    a, o = hs.arrays(), osc.arrays()
    assert np.array_equal(a["tri_vid"], o["face_vid"]) and np.array_equal(a["mat_id"], o["face_mat"])
    assert np.array_equal(a["face_normal"].view(np.uint32), o["face_normal"].view(np.uint32))      # NaN normals of zero-area faces included
    assert np.array_equal(a["vert_normal"].view(np.uint32), o["normals"].view(np.uint32))
    world = o["wverts"][o["face_vid"].astype(np.int64)].reshape(-1, 9)
    assert np.array_equal(a["tri_verts"].view(np.uint32), world.view(np.uint32))
    want = flat_tree_from_oracle(osc)
    assert len(want) == a["node_box"].shape[0]
    for i, (kind, box, payload) in enumerate(want):
        assert np.array_equal(a["node_box"][i].view(np.uint32), box.view(np.uint32)), i
        cf = int(a["node_count_flags"][i])
        if kind == "leaf":
            first, cnt = int(a["node_first"][i]), cf & 0x7FFFFFFF
            assert cf & RT_NODE_LEAF and np.array_equal(a["face_refs"][first:first + cnt].astype(np.int32), payload)
        else:
            assert not (cf & RT_NODE_LEAF) and (cf & 0x7FFFFFFF) == payload
    hs.close()
    osc.close()


def test_camera_and_lights_match_oracle(rt, oracle):
    import ctypes as C
    lib = rt.load_library()
    for (w, h, yaw) in [(256, 256, 0.0), (1920, 1080, 0.0), (640, 360, 0.7), (333, 517, -2.1)]:
        cam = rt.default_camera(w, h, yaw)
        ocam = oracle.camera(w, h, yaw)
        assert list(cam.center) == list(ocam.center) and list(cam.inv_view) == list(ocam.inv_view)
        for (i, j) in [(0, 0), (w - 1, h - 1), (w // 2, h // 2), (7, h - 3), (w - 5, 11)]:
            out = (C.c_float * 3)()
            lib.rt_screen_to_world(C.byref(cam), float(i), float(j), out)
            assert np.array_equal(np.array(out, np.float32).view(np.uint32), oracle.screen_to_world(ocam, i, j).view(np.uint32))
    fs = rt.Flyscene()
    for (u, v) in [(5, 5), (8, 8), (16, 16), (3, 7)]:
        fs.usteps, fs.vsteps = u, v
        ol = oracle.lights(area=True, usteps=u, vsteps=v)
        for p in [(-1.0, 1.0, 1.0), (0.3, -0.2, 0.9), (0.0, 0.0, 0.0)]:
            assert np.array_equal(fs.createSpherePoint(p).view(np.uint32), oracle.light_samples(ol, p).view(np.uint32))


def test_ppm_writer_byte_exact(rt, oracle, scenes, tmp_path):
    import ctypes as C
    rng = np.random.default_rng(7)
    rgb = rng.random((37, 53, 3), dtype=np.float32) * 1.2
    rgb[0, 0] = (1.0, 0.0, 0.999999)
    lib = rt.load_library()
    p1, p2 = tmp_path / "a.ppm", tmp_path / "b.ppm"
    assert lib.rt_write_ppm(str(p1).encode(), rgb.ctypes.data_as(C.c_void_p), 53, 37) == 0
    osc = oracle.load_scene(os.path.join(scenes, "cube.obj"))
    assert osc.write_ppm(p2, rgb) == 1
    assert p1.read_bytes() == p2.read_bytes()
    osc.close()


def test_pfm_side_channel_round_trips(rt, tmp_path):
    import ctypes as C
    rng = np.random.default_rng(8)
    rgb = rng.random((19, 31, 3), dtype=np.float32) * 1.5
    lib = rt.load_library()
    p = tmp_path / "a.pfm"
    assert lib.rt_write_pfm(str(p).encode(), rgb.ctypes.data_as(C.c_void_p), 31, 19) == 0
    raw = p.read_bytes()
    head = b"PF\n31 19\n-1.0\n"
    assert raw.startswith(head) and len(raw) == len(head) + 19 * 31 * 12
    back = np.frombuffer(raw[len(head):], dtype="<f4").reshape(19, 31, 3)[::-1]
    assert np.array_equal(back.view(np.uint32), rgb.view(np.uint32))
    assert lib.rt_write_pfm(b"/nonexistent_dir/x.pfm", rgb.ctypes.data_as(C.c_void_p), 31, 19) == rt.capi.RT_ERR_IO


def test_bad_inputs_are_rejected(rt, scenes, tmp_path):
    import ctypes as C
    lib = rt.load_library()
    h = C.c_void_p()
    assert lib.rt_host_scene_load(b"/nonexistent/file.obj", 1000, 15, C.byref(h)) == rt.capi.RT_ERR_IO
    quad = tmp_path / "quad.obj"
    quad.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\n")
    assert lib.rt_host_scene_load(str(quad).encode(), 1000, 15, C.byref(h)) == rt.capi.RT_ERR_IO
    assert lib.rt_host_scene_load(None, 1000, 15, C.byref(h)) == rt.capi.RT_ERR_INVALID
    p = rt.make_params(64, 48, row0=0, row1=48, stripe=8, rank=1, nranks=3)
    rows = [y for y in range(48) if (y // 8) % 3 == 1]
    assert lib.rt_local_rows(C.byref(p)) == len(rows)


def test_octree_growth_guard_and_lost_faces(rt, oracle, scenes):
    """Capacity 100 on dodge: depth-9 tree with 'lost' faces (children holding exactly `capacity` faces are neither leaf
    nor split, boxTree.cpp:140-145) -- product == oracle.  Capacity 64 makes the reference's construction explode
    (normalised-vector SAT accepts faces in every octant): the product refuses instead of exhausting memory."""
    import ctypes as C
    path = os.path.join(scenes, "dodgeColorTest.obj")
    hs = rt.HostScene(path, leaf_capacity=100, max_depth=15)
    info = hs.info()
    assert info["depth"] >= 6 and info["lost_nodes"] >= 1 and info["unreachable_faces"] >= 1
    osc = oracle.load_scene(path, capacity=100, maxdepth=15)
    want = flat_tree_from_oracle(osc)
    a = hs.arrays()
    assert len(want) == a["node_box"].shape[0]
    for i, (kind, box, payload) in enumerate(want):
        assert np.array_equal(a["node_box"][i].view(np.uint32), box.view(np.uint32))
        if kind == "leaf":
            first, cnt = int(a["node_first"][i]), int(a["node_count_flags"][i]) & 0x7FFFFFFF
            assert np.array_equal(a["face_refs"][first:first + cnt].astype(np.int32), payload)
        else:
            assert (int(a["node_count_flags"][i]) & 0x7FFFFFFF) == payload
    lib = rt.load_library()
    out = (C.c_int32 * 4)()
    assert lib.rt_debug_chunk_stats(C.byref(hs.view), out) == 0 and out[0] >= info["leaves"] and out[2] == info["leaves"]
    h = C.c_void_p()
    assert lib.rt_host_scene_load(path.encode(), 64, 15, C.byref(h)) == rt.capi.RT_ERR_UNSUPPORTED
    hs.close()
    osc.close()


def test_chunk_bounds_skip_triangles_the_reference_can_never_accept(rt, scenes):
    import ctypes as C
    """dodgeColorTest.obj has 15 collinear triangles.  Four of them can never be accepted by rayTriangleIntersection as the reference
    computes it (three have a zero face normal: dn == 0 for every ray; one has a float denominator of exactly 0: 1/denom = inf): they need
    no bound, so the chunks that hold only such triangles stay cullable.  The other eleven (a denominator that is one rounding error
    instead of zero: their barycentrics are noise) keep their chunks un-cullable.  CPU only: the counts rt_upload_scene works from."""
    lib = rt.load_library()
    hs = rt.HostScene(os.path.join(scenes, "dodgeColorTest.obj"), 1000, 15)
    out = (C.c_int32 * 4)()
    assert lib.rt_debug_chunk_stats(C.byref(hs.view), out) == 0
    chunks, cullable = out[0], out[1]
    never = chunks - cullable
    assert chunks > 400 and 0 < never < 23, (chunks, cullable)         # 23 when every collinear triangle counted; the live ones sit in 22 chunks
    hs.close()


def test_malformed_ply_files_are_refused(rt, tmp_path):
    """The PLY loader does not trust the file (host_scene.cpp: load_ply): unparsable header lines, element counts the file cannot hold
    (an element without properties may claim none), list lengths past the end of the file, and face indices that are negative,
    non-finite or >= 2^32 (a cast that is undefined behaviour) all fail with RT_ERR_IO -- quickly, without allocating for the claim."""
    import ctypes as C
    import struct
    import time
    lib = rt.load_library()
    h = C.c_void_p()
    tri = "0 0 0\n1 0 0\n0 1 0\n"
    head = "ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\n"
    good = head + "element face 1\nproperty list uchar int vertex_indices\nend_header\n" + tri + "3 0 1 2\n"
    cases = {
        "ok.ply": (good, True),
        "neg_index.ply": (good.replace("3 0 1 2", "3 0 1 -2"), False),
        "nan_index.ply": (good.replace("property list uchar int", "property list uchar float").replace("3 0 1 2", "3 0 1 nan"), False),
        "huge_index.ply": (good.replace("property list uchar int", "property list uchar double").replace("3 0 1 2", "3 0 1 4294967296"), False),
        "count_text.ply": (good.replace("element face 1", "element face many"), False),
        "count_negative.ply": (good.replace("element vertex 3", "element vertex -3"), False),
        "no_props_huge.ply": ("ply\nformat ascii 1.0\nelement junk 18446744073709551615\nend_header\n", False),
        "no_props_big.ply": ("ply\nformat ascii 1.0\nelement junk 4000000000\n" + head[len("ply\nformat ascii 1.0\n"):] + "end_header\n" + tri, False),
        "vertex_claim.ply": (good.replace("element vertex 3", "element vertex 3000000000"), False),
        "prop_noname.ply": (good.replace("property float z", "property float"), False),
        "list_noname.ply": (good.replace("property list uchar int vertex_indices", "property list uchar int"), False),
    }
    for name, (text, ok) in cases.items():
        f = tmp_path / name
        f.write_text(text)
        t0 = time.time()
        st = lib.rt_host_scene_load(str(f).encode(), 1000, 15, C.byref(h))
        assert time.time() - t0 < 5.0, name
        assert (st == 0) == ok, (name, st)
        if st == 0:
            lib.rt_host_scene_free(h)
    # binary: a list count larger than the rest of the file
    b = tmp_path / "list_len.ply"
    hdr = ("ply\nformat binary_little_endian 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\n"
           "element face 1\nproperty list uint int vertex_indices\nend_header\n").encode()
    b.write_bytes(hdr + struct.pack("<9f", 0, 0, 0, 1, 0, 0, 0, 1, 0) + struct.pack("<I3i", 0xFFFFFFF0, 0, 1, 2))
    assert lib.rt_host_scene_load(str(b).encode(), 1000, 15, C.byref(h)) == rt.capi.RT_ERR_IO


@pytest.mark.parametrize("name", ["dodgeColorTest.obj", "bunny.ply", "wavy"])
def test_chunk_boxes_and_slabs_contain_their_triangles(rt, scenes, tmp_path, name):
    """What the exact-culling rules rely on (rt_capi.cpp: build_chunk_bounds): every cullable chunk's inflated box AND its slab along the
    chunk's mean normal contain all vertices of its hittable triangles, with at least `infl` (per axis, resp. sum |sn_k| infl) to spare --
    the computed point of an accepted hit lies within infl of its triangle.  And the slab is what it is for: on smooth meshes it is several
    times thinner than the box is along the slab's direction.  CPU only."""
    import ctypes as C
    import scenes_gen
    lib = rt.load_library()
    path = scenes_gen.wavy_grid(str(tmp_path), n=96) if name == "wavy" else os.path.join(scenes, name)
    hs = rt.HostScene(path, 1000, 15)
    v = hs.view
    n = C.c_int32()
    assert lib.rt_debug_chunk_bounds(C.byref(v), None, 0, C.byref(n), None, None) == 0 and n.value > 0
    b = np.zeros((n.value, 16), np.float32)
    chunk0 = np.zeros(v.n_nodes, np.uint32)
    refs = np.zeros(v.n_face_refs, np.uint32)
    assert lib.rt_debug_chunk_bounds(C.byref(v), b.ctypes.data_as(C.POINTER(C.c_float)), n.value, C.byref(n), chunk0.ctypes.data_as(C.POINTER(C.c_uint32)),
                                     refs.ctypes.data_as(C.POINTER(C.c_uint32))) == 0
    a = hs.arrays()
    tv = a["tri_verts"].reshape(-1, 3, 3).astype(np.float64)
    thin = []
    checked = 0
    for ni in range(v.n_nodes):
        cf = int(a["node_count_flags"][ni])
        if not cf & 0x80000000:
            continue
        cnt, first = cf & 0x7FFFFFFF, int(a["node_first"][ni])
        for c in range((cnt + 63) // 64):
            cb = b[int(chunk0[ni]) + c].astype(np.float64)
            if cb[6] >= 1.5:
                assert cb[11] <= -1e38 and cb[12] >= 1e38                    # never cullable: the slab rejects nothing either
                continue
            faces = refs[first + 64 * c:first + min(cnt, 64 * c + 64)]
            P = tv[faces].reshape(-1, 3)
            edges = np.abs(tv[faces][:, 1] - tv[faces][:, 0]).sum(1) + np.abs(tv[faces][:, 2] - tv[faces][:, 0]).sum(1)
            P = P[np.repeat(edges > 0, 3)] if (edges > 0).any() else P       # (triangles that can never be hit need no bound; keep it simple: degenerate points)
            infl, sn = cb[7], cb[8:11]
            assert abs(np.linalg.norm(sn) - 1.0) < 1e-5
            d = P @ sn
            sinfl = np.abs(sn).sum() * infl
            if len(P):
                assert (P >= cb[0:3] + 0.99 * infl).all() and (P <= cb[3:6] - 0.99 * infl).all()
                assert d.min() >= cb[11] + 0.99 * sinfl and d.max() <= cb[12] - 0.99 * sinfl
                checked += 1
                # the box's extent along sn against the slab's thickness
                thin.append((np.abs(sn) * (cb[3:6] - cb[0:3])).sum() / max(cb[12] - cb[11], 1e-30))
    assert checked > 20
    print(name, "box extent along sn / slab thickness: median %.2f, p90 %.2f" % (np.median(thin), np.percentile(thin, 90)))
    if name != "dodgeColorTest.obj":
        assert np.median(thin) > 1.5, np.median(thin)                          # smooth surfaces: thin plates in fat boxes
    hs.close()
