"""Deterministic inputs of the stub-free reference pins (tests/golden/ref_pins.npz).

oracle/make_ref_fixtures.py feeds exactly these arrays to oracle/_ref/ref_probe2 (the reference's own boundingBox.cpp / boxTree.cpp /
camera.hpp / ppmIO.hpp compiled in place) and stores the OUTPUTS; tests/test_ref_pins.py regenerates the inputs from here and checks
the oracle (CPU) and the HIP path (GPU) against the stored outputs.  A sha256 of every input array is stored beside the outputs, so a
drift of this generator is detected instead of silently comparing different cases.
No RNG library: a xorshift32 stream, so the arrays do not depend on a numpy version.
"""
import hashlib

import numpy as np


class XorShift:
    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFF or 0x9E3779B9

    def u32(self):
        s = self.s
        s ^= (s << 13) & 0xFFFFFFFF
        s ^= s >> 17
        s ^= (s << 5) & 0xFFFFFFFF
        self.s = s
        return s

    def f(self):                      # float32 in [-1, 1)
        return np.float32(((self.u32() >> 8) / 16777216.0) * 2.0 - 1.0)

    def u(self):                      # float32 in [0, 1)
        return np.float32((self.u32() >> 8) / 16777216.0)

    def below(self, n):
        return self.u32() % n


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def box_cases(n=12288, seed=0xB0C5):
    """[n, 12] float32: bmin, bmax, origin, dest -- generic rays, zero direction components (origin inside / outside / exactly ON a
    slab plane: 0/0), origins inside the box, boxes behind the origin, grazing rays through corners, 1e3 / 1e-3 scales, dest == origin."""
    r = XorShift(seed)
    out = np.zeros((n, 12), np.float32)
    for i in range(n):
        c = np.array([r.f(), r.f(), r.f()], np.float32) * np.float32(0.8)
        e = np.array([r.u(), r.u(), r.u()], np.float32) * np.float32(0.58) + np.float32(0.02)
        bmin, bmax = c - e, c + e
        o = np.array([r.f(), r.f(), r.f()], np.float32) * np.float32(2.0)
        d = np.array([r.f(), r.f(), r.f()], np.float32) * np.float32(2.0)
        kind = i % 12
        if kind == 1:                                   # one zero direction component
            k = r.below(3); d[k] = o[k]
        elif kind == 2:                                 # zero component, origin exactly on the slab plane (0/0 = NaN)
            k = r.below(3); o[k] = bmin[k] if r.below(2) else bmax[k]; d[k] = o[k]
        elif kind == 3:                                 # origin inside the box
            o = (bmin + (bmax - bmin) * np.array([r.u(), r.u(), r.u()], np.float32)).astype(np.float32)
        elif kind == 4:                                 # box behind the origin
            d = (o + (o - c) * np.float32(0.5)).astype(np.float32)
        elif kind == 5:                                 # two zero components
            k = r.below(3)
            for j in range(3):
                if j != k:
                    d[j] = o[j]
        elif kind == 6:                                 # aimed exactly at a corner of the box (ties tin == tout)
            corner = np.array([bmin[0] if r.below(2) else bmax[0], bmin[1] if r.below(2) else bmax[1], bmin[2] if r.below(2) else bmax[2]], np.float32)
            d = corner
        elif kind == 7:                                 # origin on a face plane, generic direction
            k = r.below(3); o[k] = bmin[k] if r.below(2) else bmax[k]
        elif kind == 8:                                 # large scale
            bmin, bmax, o, d = (x * np.float32(1000.0) for x in (bmin, bmax, o, d))
        elif kind == 9:                                 # small scale
            bmin, bmax, o, d = (x * np.float32(0.001) for x in (bmin, bmax, o, d))
        elif kind == 10 and i % 24 == 10:               # degenerate: dest == origin
            d = o.copy()
        out[i, 0:3], out[i, 3:6], out[i, 6:9], out[i, 9:12] = bmin, bmax, o, d
    return out


def tree_rays(wverts, screen_points, n=3072, seed=0x7EE5):
    """[n, 6] float32 segments (origin, dest) for BoxTree::intersect on the dodge tree: primary-like rays from the camera centre through
    screen points, shadow-like segments from area-light samples to mesh vertices, and random segments."""
    r = XorShift(seed)
    out = np.zeros((n, 6), np.float32)
    nv = wverts.shape[0]
    ns = screen_points.shape[0]
    for i in range(n):
        kind = i % 3
        if kind == 0:
            o = np.array([0.0, 0.0, 2.0], np.float32)
            d = screen_points[r.below(ns)]
        elif kind == 1:
            # arealight samples of the light at (-1, 1, 1): x = (a + 0.5) * (-0.7 / 5), y = (b + 0.5) * (1.15 / 5), z = 1  (arealight.hpp:15-25)
            a, b = r.below(5), r.below(5)
            o = np.array([np.float32(a + 0.5) * (np.float32(-1.0 + 0.3) / np.float32(5)), np.float32(b + 0.5) * (np.float32(1.0 + 0.15) / np.float32(5)), 1.0], np.float32)
            d = wverts[r.below(nv)]
        else:
            o = np.array([r.f(), r.f(), r.f()], np.float32)
            d = np.array([r.f(), r.f(), r.f()], np.float32)
        out[i, 0:3], out[i, 3:6] = o, d
    return out


def sat_pairs(nodes, wverts, face_vid):
    """[m, 15] float32 (bmin, bmax, A, B, C): the children of the root against EVERY face, every other node against every 7th face."""
    nf = face_vid.shape[0]
    tri = wverts[face_vid.reshape(-1)].reshape(nf, 9).astype(np.float32)
    rows = []
    root_children = set(c for c in nodes[0]["children"] if c >= 0)
    for i, nd in enumerate(nodes):
        if i == 0:
            continue
        faces = np.arange(nf) if i in root_children else np.arange(i % 7, nf, 7)
        blk = np.empty((faces.size, 15), np.float32)
        blk[:, 0:6] = nd["box"]
        blk[:, 6:15] = tri[faces]
        rows.append(blk)
    return np.concatenate(rows, axis=0)


def prim_cases(n=4096, seed=0x5A7):
    """[n, 16] float32: a b fa fb | v0 | v1 | boxhalfsize | three scalars for findMinMax.  Unit-ish vectors as clasifyFace produces them,
    plus exact ties (p == rad) every 16th case."""
    r = XorShift(seed)
    out = np.zeros((n, 16), np.float32)
    for i in range(n):
        row = np.array([r.f() for _ in range(16)], np.float32)
        row[2], row[3] = abs(row[0]), abs(row[1])                    # fa = |a|, fb = |b| as the callers pass them
        row[10:13] = np.abs(row[10:13])                              # box half size is positive
        if i % 16 == 0:                                              # exact tie on the X axis test: p0 == rad
            row[4:7] = np.array([0.0, 1.0, 0.0], np.float32); row[7:10] = row[4:7]
            row[0], row[1] = np.float32(0.5), np.float32(0.0); row[2], row[3] = np.float32(0.5), np.float32(0.0)
            row[10:13] = np.array([0.3, 1.0, 0.2], np.float32)
        if i % 16 == 8:                                              # zero normal component for planeBoxOverlap
            row[4] = np.float32(0.0)
        out[i] = row
    return out


def ppm_image(n=32, seed=0x99):
    """[n, n, 3] float32 with the values writePPMImage is sensitive to: k/255 and its neighbours, 0.999999, 1, > 1, negatives, tiny."""
    r = XorShift(seed)
    img = np.zeros((n, n, 3), np.float32)
    flat = img.reshape(-1)
    for i in range(flat.size):
        kind = i % 8
        k = r.below(256)
        base = np.float32(k) / np.float32(255.0)
        if kind == 0:
            v = base
        elif kind == 1:
            v = np.nextafter(base, np.float32(2.0))
        elif kind == 2:
            v = np.nextafter(base, np.float32(-2.0))
        elif kind == 3:
            v = np.float32(0.999999)
        elif kind == 4:
            v = np.float32(1.0) + r.u() * np.float32(3.0)
        elif kind == 5:
            v = -r.u() * np.float32(0.9)                 # negatives are NOT clamped by the reference: (int)(255*c) truncates toward zero
        elif kind == 6:
            v = r.u() * np.float32(1e-3)
        else:
            v = r.u()
        flat[i] = v
    return img


CAMERAS = [(256, 256, 0.0), (1920, 1080, 0.0), (1920, 1080, float(np.float32(2.0 * np.pi * 7.0 / 120.0))), (160, 90, 1.0)]
