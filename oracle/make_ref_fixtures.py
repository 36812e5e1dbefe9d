#!/usr/bin/env python3
"""Generates tests/golden/ref_pins.npz: OUTPUTS of the reference's own code (oracle/_ref/ref_probe2 = src/boundingBox.cpp +
src/boxTree.cpp + camera.hpp + ppmIO.hpp compiled in place, no stand-ins) on the inputs of tests/ref_inputs.py.

Runs only where /root/reference exists (this container): `make -C oracle ref && python oracle/make_ref_fixtures.py`.
The fixture holds numbers only -- decisions, face-id sets, float bit patterns, the bytes of one small .ppm -- plus a sha256 of every
input array.  Test infrastructure: nothing under raytracer-in-cpp_amd/ touches this.
"""
import hashlib
import os
import struct
import subprocess
import sys
import tempfile
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib          # noqa: E402
import ref_inputs as RI    # noqa: E402

PROBE = os.path.join(ROOT, "oracle", "_ref", "ref_probe2")
SCENE = os.path.join(ROOT, "tests", "golden", "scenes", "dodgeColorTest.obj")


def run(*args):
    subprocess.check_call([PROBE] + [str(a) for a in args])


def write_tree(path, nodes):
    with open(path, "wb") as f:
        f.write(struct.pack("<i", len(nodes)))
        for nd in nodes:
            f.write(np.asarray(nd["box"], np.float32).tobytes())
            ch = [c for c in nd["children"]]
            f.write(struct.pack("<3i", int(nd["is_leaf"]), int(nd["is_empty"]), int(nd["nchildren"])))
            f.write(struct.pack("<8i", *ch))
            f.write(struct.pack("<i", int(nd["nfaces"])))
            f.write(np.asarray(nd["faces"], np.int32).tobytes())


def read_sets(path, n):
    raw = np.fromfile(path, np.int32)
    counts, ids, pos = np.zeros(n, np.int32), [], 0
    for i in range(n):
        c = int(raw[pos]); pos += 1
        counts[i] = c
        ids.append(raw[pos:pos + c].copy()); pos += c
    assert pos == raw.size
    return counts, ids


def main():
    if not os.path.exists(PROBE):
        raise SystemExit("oracle/_ref/ref_probe2 missing: run `make -C oracle ref` (needs /root/reference)")
    orc = oracle_lib.load()
    sc = orc.load_scene(SCENE)
    nodes = [sc.node(i) for i in range(sc.nnodes)]
    arr = sc.arrays()
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        T = lambda name: os.path.join(tmp, name)      # noqa: E731
        # 1. BoundingBox::boxIntersect
        box = RI.box_cases()
        box.tofile(T("box.in"))
        run("box", T("box.in"), T("box.out"))
        dec = np.fromfile(T("box.out"), np.uint8)
        assert dec.size == box.shape[0]
        out["box_in_sha"] = RI.sha(box); out["box_hit"] = np.packbits(dec)
        # 2. BoxTree::intersect on the dodge tree re-assembled through the public fields
        cam = orc.camera(256, 256)
        scr = np.array([orc.screen_to_world(cam, i, j) for j in range(0, 256, 8) for i in range(0, 256, 8)], np.float32)
        rays = RI.tree_rays(arr["wverts"], scr)
        rays.tofile(T("rays.in"))
        write_tree(T("tree.bin"), nodes)
        run("tree", T("tree.bin"), T("rays.in"), T("faces.out"), "faces")
        run("tree", T("tree.bin"), T("rays.in"), T("leaves.out"), "leaves")
        fc, fids = read_sets(T("faces.out"), rays.shape[0])
        lc, lids = read_sets(T("leaves.out"), rays.shape[0])
        out["tree_in_sha"] = RI.sha(rays)
        out["tree_face_count"] = fc
        out["tree_face_crc"] = np.array([zlib.crc32(x.tobytes()) for x in fids], np.uint32)
        out["tree_face_ids_first512"] = np.concatenate(fids[:512]) if fids[:512] else np.zeros(0, np.int32)
        out["tree_leaf_count"] = lc
        out["tree_leaf_ids"] = np.concatenate(lids)
        # 3. clasifyFace decisions (flow restated in the probe, arithmetic by the reference's members) + the members on their own
        pairs = RI.sat_pairs(nodes, arr["wverts"], arr["face_vid"])
        pairs.tofile(T("sat.in"))
        run("sat", T("sat.in"), T("sat.out"))
        sd = np.fromfile(T("sat.out"), np.uint8)
        assert sd.size == pairs.shape[0]
        out["sat_in_sha"] = RI.sha(pairs); out["sat_decision"] = np.packbits(sd)
        prim = RI.prim_cases()
        prim.tofile(T("prim.in"))
        run("prim", T("prim.in"), T("prim.out"))
        raw = np.fromfile(T("prim.out"), np.uint8)
        n = prim.shape[0]
        out["prim_in_sha"] = RI.sha(prim)
        out["prim_dec"] = raw[:n * 12].reshape(n, 12)[:, :8].copy()
        out["prim_minmax"] = raw[n * 12:].view(np.float32).reshape(n, 2).copy()
        # 4. Camera::screenToWorld on every pixel
        for k, (W, H, yaw) in enumerate(RI.CAMERAS):
            bits = "%08x" % struct.unpack("<I", struct.pack("<f", yaw))[0]
            run("cam", W, H, bits, T("cam.out"))
            v = np.fromfile(T("cam.out"), np.float32)
            out[f"cam{k}_center"] = v[:3].copy()
            pts = v[3:]
            out[f"cam{k}_sha"] = hashlib.sha256(pts.tobytes()).hexdigest()
            out[f"cam{k}_every1009"] = pts[::1009].copy()
        # 5. writePPMImage
        img = RI.ppm_image()
        img.tofile(T("ppm.in"))
        run("ppm", T("ppm.in"), img.shape[1], img.shape[0], T("ref.ppm"))
        out["ppm_in_sha"] = RI.sha(img)
        out["ppm_bytes"] = np.frombuffer(open(T("ref.ppm"), "rb").read(), np.uint8).copy()
    dst = os.path.join(ROOT, "tests", "golden", "ref_pins.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes;", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})
    sc.close()


if __name__ == "__main__":
    main()
