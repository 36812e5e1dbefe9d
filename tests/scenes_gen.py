"""Deterministic authored scenes (test data generators; no RNG).

mixed_materials: small OBJ+MTL that reaches the illum 3/4/5/6/7/9 branches of traceRay (flyscene.cpp:712-760) that
cube.obj (illum 4) and dodgeColorTest.obj (illum 2) never take.  No reference output exists for these scenes:
the GPU path is compared with the CPU oracle only ("parity unpinned" branches).
wavy_grid: the cfg4-style displaced grid z = 0.1 sin(9x) cos(7y) over an illum-4 floor.
random_soup: seeded triangle soups for the fuzz parity test (clusters of small triangles, long slivers, zero-area and
duplicate triangles, shared vertices, all material kinds) -- ill-conditioned input for the culling code.
"""
import math
import os


def _quad(v, a, b, c, d):
    return [(a, b, c), (a, c, d)]


def mixed_materials(dirpath, name="mixed"):
    mtl = """newmtl floor_mirror
Ns 50.0
Kd 0.6 0.6 0.6
Ks 0.5 0.5 0.5
Ni 1.45
illum 3
newmtl box_fresnel
Ns 30.0
Kd 0.8 0.1 0.1
Ks 0.9 0.9 0.9
Ni 1.5
illum 5
newmtl pane_pass
Ns 10.0
Kd 0.1 0.7 0.2
Ks 0.3 0.3 0.3
illum 9
newmtl slab_glass
Ns 120.0
Kd 0.2 0.3 0.9
Ks 0.8 0.8 0.8
Ni 1.77
illum 6
newmtl plain
Ns 8.0
Kd 0.7 0.7 0.1
Ks 0.2 0.2 0.2
illum 2
newmtl seven
Ns 12.0
Kd 0.3 0.6 0.6
Ks 0.4 0.4 0.4
Ni 1.3
illum 7
"""
    verts, faces = [], []  # faces: (material, (i,j,k)) 1-based

    def add_quad(mat, p0, p1, p2, p3):
        base = len(verts)
        verts.extend([p0, p1, p2, p3])
        faces.append((mat, (base + 1, base + 2, base + 3)))
        faces.append((mat, (base + 1, base + 3, base + 4)))

    def add_box(mat, lo, hi):
        x0, y0, z0 = lo
        x1, y1, z1 = hi
        add_quad(mat, (x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1))
        add_quad(mat, (x1, y0, z0), (x0, y0, z0), (x0, y1, z0), (x1, y1, z0))
        add_quad(mat, (x0, y1, z0), (x0, y1, z1), (x1, y1, z1), (x1, y1, z0))
        add_quad(mat, (x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1))
        add_quad(mat, (x1, y0, z0), (x1, y1, z0), (x1, y1, z1), (x1, y0, z1))
        add_quad(mat, (x0, y0, z0), (x0, y0, z1), (x0, y1, z1), (x0, y1, z0))

    add_quad("floor_mirror", (-3, -1, -3), (3, -1, -3), (3, -1, 3), (-3, -1, 3))
    add_box("box_fresnel", (-1.6, -1, -0.6), (-0.6, 0.2, 0.4))
    add_box("plain", (0.4, -1, -1.2), (1.4, 0.0, -0.2))
    add_quad("pane_pass", (-0.4, -1, 1.0), (0.9, -1, 1.0), (0.9, 0.6, 1.0), (-0.4, 0.6, 1.0))
    add_box("slab_glass", (1.5, -1, 0.3), (2.2, 0.5, 0.9))
    add_box("seven", (-2.6, -1, 0.8), (-2.0, -0.2, 1.4))
    obj = [f"mtllib {name}.mtl"]
    obj += ["v %.6f %.6f %.6f" % p for p in verts]
    cur = None
    for mat, (a, b, c) in faces:
        if mat != cur:
            obj.append(f"usemtl {mat}")
            cur = mat
        obj.append(f"f {a} {b} {c}")
    os.makedirs(dirpath, exist_ok=True)
    with open(os.path.join(dirpath, name + ".mtl"), "w") as f:
        f.write(mtl)
    path = os.path.join(dirpath, name + ".obj")
    with open(path, "w") as f:
        f.write("\n".join(obj) + "\n")
    return path


def wavy_grid(dirpath, n=64, name="wavy"):
    """(n x n)-quad displaced grid (2*n*n triangles, illum 2) above a 2-triangle illum-4 floor."""
    mtl = "newmtl wave\nNs 20.0\nKd 0.2 0.5 0.8\nKs 0.6 0.6 0.6\nillum 2\nnewmtl floor\nNs 10.0\nKd 0.5 0.5 0.5\nKs 1.0 1.0 1.0\nillum 4\n"
    lines = [f"mtllib {name}.mtl"]
    for j in range(n + 1):
        for i in range(n + 1):
            x, y = -1.0 + 2.0 * i / n, -1.0 + 2.0 * j / n
            lines.append("v %.6f %.6f %.6f" % (x, y, 0.1 * math.sin(9 * x) * math.cos(7 * y)))
    base = (n + 1) * (n + 1)
    for p in [(-1.4, -1.4, -0.4), (1.4, -1.4, -0.4), (1.4, 1.4, -0.4), (-1.4, 1.4, -0.4)]:
        lines.append("v %.6f %.6f %.6f" % p)
    lines.append("usemtl wave")
    for j in range(n):
        for i in range(n):
            a = j * (n + 1) + i + 1
            b, c, d = a + 1, a + n + 2, a + n + 1
            lines.append(f"f {a} {b} {c}")
            lines.append(f"f {a} {c} {d}")
    lines.append("usemtl floor")
    lines.append(f"f {base + 1} {base + 2} {base + 3}")
    lines.append(f"f {base + 1} {base + 3} {base + 4}")
    os.makedirs(dirpath, exist_ok=True)
    with open(os.path.join(dirpath, name + ".mtl"), "w") as f:
        f.write(mtl)
    path = os.path.join(dirpath, name + ".obj")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return path


def random_soup(dirpath, seed, n_tri, name=None):
    """Seeded soup of n_tri triangles: 70 % small triangles in a few clusters, 2 % scene-sized and 6 % medium ones, 17 % slivers (aspect 1e3-1e6),
    a few zero-area / repeated-vertex / duplicated triangles; 6 materials incl. mirror, Fresnel, glass and pass-through."""
    import random
    rnd = random.Random(seed)
    name = name or f"soup_{seed}_{n_tri}"
    mats = [("m_plain", 2, 12.0, 1.0), ("m_hi", 4, 40.0, 1.0), ("m_mirror", 3, 60.0, 1.45), ("m_fresnel", 5, 25.0, 1.5),
            ("m_glass", 6, 90.0, 1.6), ("m_pass", 9, 10.0, 1.0)]
    mtl = []
    for mname, illum, ns, ni in mats:
        kd = [round(rnd.uniform(0.1, 0.9), 3) for _ in range(3)]
        ks = [round(rnd.uniform(0.1, 0.9), 3) for _ in range(3)]
        mtl += [f"newmtl {mname}", f"Ns {ns}", "Kd %.3f %.3f %.3f" % tuple(kd), "Ks %.3f %.3f %.3f" % tuple(ks), f"Ni {ni}", f"illum {illum}"]
    verts, faces = [], []
    centres = [(rnd.uniform(-1, 1), rnd.uniform(-1, 1), rnd.uniform(-1, 1)) for _ in range(5)]

    def vtx(p):
        verts.append(p)
        return len(verts)

    for t in range(n_tri):
        kind = rnd.random()
        mat = rnd.choices(range(len(mats)), weights=[40, 25, 10, 8, 7, 10])[0]
        if kind < 0.70:
            c = rnd.choice(centres)
            o = tuple(c[k] + rnd.gauss(0, 0.25) for k in range(3))
            sz = rnd.uniform(0.02, 0.15)
            tri = [tuple(o[k] + rnd.uniform(-sz, sz) for k in range(3)) for _ in range(3)]
        elif kind < 0.72:
            tri = [tuple(rnd.uniform(-1.5, 1.5) for _ in range(3)) for _ in range(3)]
        elif kind < 0.78:                       # medium triangles
            o = tuple(rnd.uniform(-1, 1) for _ in range(3))
            tri = [tuple(o[k] + rnd.uniform(-0.4, 0.4) for k in range(3)) for _ in range(3)]
        elif kind < 0.95:                       # sliver: two vertices far apart, the third almost on their line
            a = tuple(rnd.uniform(-1.2, 1.2) for _ in range(3))
            ln = rnd.uniform(0.05, 0.6)
            b = tuple(a[k] + rnd.uniform(-ln, ln) for k in range(3))
            w = rnd.random()
            eps = 10.0 ** rnd.uniform(-6, -3)
            c = tuple(a[k] + w * (b[k] - a[k]) + rnd.uniform(-eps, eps) for k in range(3))
            tri = [a, b, c]
        elif kind < 0.97:                       # zero area: all three on a line / two equal
            a = tuple(rnd.uniform(-1, 1) for _ in range(3))
            tri = [a, a, tuple(rnd.uniform(-1, 1) for _ in range(3))]
        elif faces:                             # exact duplicate of an earlier triangle (t ties: lowest face id must win)
            faces.append((mat, faces[rnd.randrange(len(faces))][1]))
            continue
        else:
            tri = [tuple(rnd.uniform(-1, 1) for _ in range(3)) for _ in range(3)]
        if faces and rnd.random() < 0.3:        # share a vertex with the previous triangle
            ids = (faces[-1][1][rnd.randrange(3)], vtx(tri[1]), vtx(tri[2]))
        else:
            ids = (vtx(tri[0]), vtx(tri[1]), vtx(tri[2]))
        faces.append((mat, ids))
    obj = [f"mtllib {name}.mtl"] + ["v %.7f %.7f %.7f" % p for p in verts]
    cur = None
    for mat, (a, b, c) in faces:
        if mat != cur:
            obj.append(f"usemtl {mats[mat][0]}")
            cur = mat
        obj.append(f"f {a} {b} {c}")
    os.makedirs(dirpath, exist_ok=True)
    with open(os.path.join(dirpath, name + ".mtl"), "w") as f:
        f.write("\n".join(mtl) + "\n")
    path = os.path.join(dirpath, name + ".obj")
    with open(path, "w") as f:
        f.write("\n".join(obj) + "\n")
    return path
