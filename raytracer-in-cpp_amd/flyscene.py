"""Python host mirror of the reference's scene controller (src/flyscene.hpp:28-200) for the ray-trace path.

``Flyscene`` keeps the reference's member names (initialize, raytraceScene, traceRay, lightStrikes,
createSpherePoint, addLight) over the C ABI; torch is plumbing only (device output buffers, torch.distributed).
"""
import ctypes as C
import time

import numpy as np

from . import capi


def _fptr(a):
    return a.ctypes.data_as(C.c_void_p)


class HostScene:
    """rt_host_scene: OBJ/MTL import + normalisation + bug-compatible octree + flattening (host side, GL-free)."""

    def __init__(self, obj_path, leaf_capacity=1000, max_depth=15):
        self.lib = capi.load_library()
        self.handle = C.c_void_p()
        st = self.lib.rt_host_scene_load(str(obj_path).encode(), leaf_capacity, max_depth, C.byref(self.handle))
        if st != capi.RT_OK:
            raise capi.RtError(f"rt_host_scene_load({obj_path}) failed with status {st}")
        self.view = capi.rt_scene()
        self._refresh()

    def _refresh(self):
        capi.check(self.lib, None, self.lib.rt_host_scene_view(self.handle, C.byref(self.view)), "rt_host_scene_view")

    def set_model(self, model12, rebuild_tree=True):
        m = (C.c_float * 12)(*[float(x) for x in model12])
        capi.check(self.lib, None, self.lib.rt_host_scene_set_model(self.handle, m, 1 if rebuild_tree else 0), "set_model")
        self._refresh()

    def build_gpu(self, ctx, leaf_capacity=1000, max_depth=15):
        """rt_host_scene_build_gpu: the reference's octree construction on the device of `ctx` (same result as the host build)"""
        capi.check(self.lib, ctx.handle, self.lib.rt_host_scene_build_gpu(self.handle, ctx.handle, int(leaf_capacity), int(max_depth)), "rt_host_scene_build_gpu")
        self._refresh()

    def info(self):
        out = (C.c_int32 * 8)()
        box = (C.c_float * 6)()
        self.lib.rt_host_scene_info(self.handle, out, box)
        keys = ["nodes", "leaves", "face_refs", "max_leaf", "depth", "unreachable_faces", "flat_nodes", "lost_nodes"]
        d = dict(zip(keys, [int(x) for x in out]))
        d["root_box"] = [float(x) for x in box]
        return d

    # numpy copies of the flattened arrays (tests, debugging)
    def arrays(self):
        v = self.view
        nodes = np.ctypeslib.as_array(C.cast(v.nodes, C.POINTER(C.c_uint32)), shape=(v.n_nodes, 8)).copy()
        out = {
            "node_box": nodes[:, :6].copy().view(np.float32),
            "node_first": nodes[:, 6].copy(),
            "node_count_flags": nodes[:, 7].copy(),
            "face_refs": np.ctypeslib.as_array(v.face_refs, shape=(v.n_face_refs,)).copy() if v.n_face_refs else np.zeros(0, np.uint32),
            "tri_verts": np.ctypeslib.as_array(v.tri_verts, shape=(v.n_faces, 9)).copy(),
            "face_normal": np.ctypeslib.as_array(v.face_normal, shape=(v.n_faces, 3)).copy(),
            "tri_vid": np.ctypeslib.as_array(v.tri_vid, shape=(v.n_faces, 3)).copy(),
            "mat_id": np.ctypeslib.as_array(v.mat_id, shape=(v.n_faces,)).copy(),
            "vert_normal": np.ctypeslib.as_array(v.vert_normal, shape=(v.n_vert_normals, 3)).copy(),
        }
        mats = np.ctypeslib.as_array(C.cast(v.materials, C.POINTER(C.c_uint32)), shape=(v.n_materials, 9)).copy()
        out["mat_f"] = mats[:, :8].copy().view(np.float32)
        out["mat_illum"] = mats[:, 8].copy().view(np.int32)
        return out

    def close(self):
        if self.handle:
            self.lib.rt_host_scene_free(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """rt_ctx: one per HIP device."""

    def __init__(self, device=0):
        self.lib = capi.load_library()
        self.handle = C.c_void_p()
        st = self.lib.rt_create(C.byref(self.handle), int(device))
        if st != capi.RT_OK:
            raise capi.RtError(f"rt_create(device={device}) failed with status {st} "
                               "(no HIP device? this path has no CPU fallback)")
        self.device = int(device)

    def upload(self, host_scene):
        capi.check(self.lib, self.handle, self.lib.rt_upload_scene(self.handle, C.byref(host_scene.view)), "rt_upload_scene")

    def close(self):
        if self.handle:
            self.lib.rt_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_params(width, height, max_depth=-1, row0=0, row1=None, stripe=1, rank=0, nranks=1, collect_stats=False):
    p = capi.rt_params()
    p.width, p.height, p.max_depth = int(width), int(height), int(max_depth)
    p.row0, p.row1 = int(row0), int(height if row1 is None else row1)
    p.stripe, p.rank, p.nranks = int(stripe), int(rank), int(nranks)
    p.collect_stats = 1 if collect_stats else 0
    return p


def make_lights(points=((-1.0, 1.0, 1.0),), area=True, usteps=5, vsteps=5):
    lib = capi.load_library()
    l = capi.rt_lights()
    lib.rt_default_lights(C.byref(l), 1 if area else 0)
    l.n_lights = len(points)
    for i, p in enumerate(points):
        for k in range(3):
            l.pos[i][k] = float(p[k])
    l.usteps, l.vsteps = int(usteps), int(vsteps)
    return l


def sphere_offsets(seed, radius=1.0, n=25):
    """rt_sphere_offsets: createSpherePoint's sphere loop (flyscene.cpp:976-993) with std::random_device replaced by mt19937(seed + i)"""
    lib = capi.load_library()
    out = np.zeros((int(n), 3), np.float32)
    lib.rt_sphere_offsets(int(seed) & 0xFFFFFFFF, float(radius), int(n), out.ctypes.data_as(C.c_void_p))
    return out


def set_sphere(lights, offsets):
    """switch an rt_lights to RT_LIGHT_SPHERE with the given [n, 3] float32 offsets (kept alive on the struct)"""
    off = np.ascontiguousarray(offsets, np.float32)
    lights._offsets_keepalive = off
    lights.mode = capi.RT_LIGHT_SPHERE
    lights.n_offsets = off.shape[0]
    lights.offsets = off.ctypes.data_as(C.POINTER(C.c_float))
    return lights


def default_camera(width, height, yaw=0.0):
    lib = capi.load_library()
    cam = capi.rt_camera()
    if yaw:
        lib.rt_yaw_camera(C.byref(cam), int(width), int(height), float(yaw))
    else:
        lib.rt_default_camera(C.byref(cam), int(width), int(height))
    return cam


class Flyscene:
    """Drop-in shaped like the reference's Flyscene for initialize() -> raytraceScene() -> result.ppm."""

    def __init__(self, scene_path="resources/models/cube.obj", device=0):
        self.scene_path = scene_path
        self.device = device
        self.areaLight, self.pointLight = True, False
        self.sphere_seed, self.sphere_offsets = 65, None
        self.usteps = self.vsteps = 5
        self.max_depth = -1
        self.lights = [(-1.0, 1.0, 1.0)]
        self.output_path = "result.ppm"
        self.ctx = None
        self.scene = None
        self.stats = capi.rt_stats()
        self.image = None

    # reference: flyscene.cpp:29-126 (stdin switches become arguments)
    def initialize(self, width, height, areaLight=True, pointLight=False):
        self.areaLight, self.pointLight = bool(areaLight), bool(pointLight)
        if not self.areaLight and not self.pointLight:
            # spherical mode (flyscene.cpp:974-995): 25 offsets drawn once from the seeded restatement of the reference's loop
            self.sphere_offsets = sphere_offsets(self.sphere_seed, 1.0, 25)
        self.width, self.height = int(width), int(height)
        self.camera = default_camera(width, height)
        self.scene = HostScene(self.scene_path, 1000, 15)
        self.ctx = Context(self.device)
        self.ctx.upload(self.scene)

    def _lights(self, points=None):
        l = make_lights(points if points is not None else self.lights, area=(self.areaLight and not self.pointLight),
                        usteps=self.usteps, vsteps=self.vsteps)
        if not self.areaLight and not self.pointLight:
            set_sphere(l, self.sphere_offsets)
        return l

    # reference: flyscene.cpp:519-648
    def raytraceScene(self, width=0, height=0, write_ppm=True, want_hits=False, collect_stats=False):
        t0 = time.time()
        if width == 0 or height == 0:
            width, height = self.width, self.height
        cam = self.camera
        if (width, height) != (self.width, self.height):
            cam = default_camera(width, height)
            cam.center, cam.inv_view = self.camera.center, self.camera.inv_view
        p = make_params(width, height, self.max_depth, collect_stats=collect_stats)
        L = self._lights()
        rgb = np.empty((height, width, 3), np.float32)
        hits = np.empty((height, width), np.int32) if want_hits else None
        lib = self.ctx.lib
        capi.check(lib, self.ctx.handle,
                   lib.rt_render(self.ctx.handle, C.byref(cam), C.byref(L), C.byref(p), _fptr(rgb),
                                 _fptr(hits) if want_hits else None, C.byref(self.stats)), "rt_render")
        self.image, self.hits = rgb, hits
        if write_ppm:
            capi.check(lib, None, lib.rt_write_ppm(self.output_path.encode(), _fptr(rgb), width, height), "rt_write_ppm")
        self.elapsed = time.time() - t0
        return rgb

    # reference: flyscene.cpp:651-771, batched (origins/directions [n,3])
    def traceRay(self, origin, direction, level=0, lights=None, countRay=False):
        o = np.ascontiguousarray(np.atleast_2d(np.asarray(origin, np.float32)))
        d = np.ascontiguousarray(np.atleast_2d(np.asarray(direction, np.float32)))
        n = o.shape[0]
        out = np.empty((n, 3), np.float32)
        face = np.empty(n, np.int32)
        t = np.empty(n, np.float32)
        L = self._lights(lights)
        budget = -1 if self.max_depth < 0 else max(0, self.max_depth - level)
        lib = self.ctx.lib
        capi.check(lib, self.ctx.handle,
                   lib.rt_trace_rays(self.ctx.handle, C.byref(L), budget, n, _fptr(o), _fptr(d), _fptr(out), _fptr(face), _fptr(t)),
                   "rt_trace_rays")
        self.last_face, self.last_t = face, t
        return out if np.ndim(origin) > 1 else out[0]

    # reference: flyscene.cpp:912-954
    def lightStrikes(self, hitPoint, lights):
        pts = np.ascontiguousarray(np.atleast_2d(np.asarray(lights, np.float32)))
        hit = np.ascontiguousarray(np.broadcast_to(np.asarray(hitPoint, np.float32), pts.shape).copy())
        vis = np.empty(pts.shape[0], np.uint8)
        lib = self.ctx.lib
        capi.check(lib, self.ctx.handle, lib.rt_light_strikes(self.ctx.handle, pts.shape[0], _fptr(hit), _fptr(pts), _fptr(vis)),
                   "rt_light_strikes")
        return bool(vis.any()), vis.astype(bool)

    # reference: flyscene.cpp:962-972, arealight.hpp:15-25 (float32 arithmetic in the reference's order)
    def createSpherePoint(self, lightPoint):
        p = np.asarray(lightPoint, np.float32)
        if self.pointLight:
            return p.reshape(1, 3).copy()
        if not self.areaLight:
            return (self.sphere_offsets + p).astype(np.float32)
        f = np.float32
        ux, uz, vy = f(p[0] + f(0.3) * f(1)), f(p[2] + f(0.3) * f(0)), f(p[1] + f(0.15) * f(1))
        out = np.empty((self.usteps * self.vsteps, 3), np.float32)
        k = 0
        for i in range(self.usteps):
            for j in range(self.vsteps):
                out[k] = (f(i + 0.5) * f(ux / f(self.usteps)), f(j + 0.5) * f(vy / f(self.vsteps)), uz)
                k += 1
        return out

    def addLight(self):
        if len(self.lights) < capi.RT_MAX_LIGHTS:
            self.lights.append(tuple(float(x) for x in self.camera.center))


class FrameGraph:
    """rt_graph: one frame's launch sequence captured into a hipGraph, replayed with a new camera per frame.
    Output buffers are caller-owned device memory (e.g. torch tensors); this class only keeps their addresses."""

    def __init__(self, ctx, lights, params, d_rgb_ptr=None, d_u8_ptr=None):
        self.ctx, self.lib = ctx, ctx.lib
        self.handle = C.c_void_p()
        st = self.lib.rt_graph_create(ctx.handle, C.byref(lights), C.byref(params),
                                      C.c_void_p(d_rgb_ptr) if d_rgb_ptr else None,
                                      C.c_void_p(d_u8_ptr) if d_u8_ptr else None, C.byref(self.handle))
        capi.check(self.lib, ctx.handle, st, "rt_graph_create")

    def launch(self, camera, stream_ptr=None):
        st = self.lib.rt_graph_launch(self.handle, C.byref(camera), C.c_void_p(stream_ptr) if stream_ptr else None)
        capi.check(self.lib, self.ctx.handle, st, "rt_graph_launch")

    def stats(self):
        out = capi.rt_stats()
        capi.check(self.lib, self.ctx.handle, self.lib.rt_graph_stats(self.handle, C.byref(out)), "rt_graph_stats")
        return out

    def close(self):
        if self.handle:
            self.lib.rt_graph_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
