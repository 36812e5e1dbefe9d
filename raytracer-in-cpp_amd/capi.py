"""ctypes binding of the C ABI in include/rt_mi355x.h (librt_mi355x.so).

This is the only way Python reaches the ray-trace path: there is no Python or CPU fallback.  If the shared
library has not been built (``__graft_entry__.build()`` / ``make -C raytracer-in-cpp_amd/csrc``) loading fails
loudly, and if no HIP device is visible ``rt_create`` returns RT_ERR_NO_DEVICE.
"""
import ctypes as C
import os

RT_OK = 0
RT_ERR_INVALID, RT_ERR_NO_DEVICE, RT_ERR_HIP, RT_ERR_IO, RT_ERR_UNSUPPORTED, RT_ERR_NO_SCENE = -1, -2, -3, -4, -5, -6
RT_MAX_LIGHTS = 25
RT_COMM_ID_BYTES = 128
RT_LIGHT_POINT, RT_LIGHT_AREA, RT_LIGHT_SPHERE = 0, 1, 2
RT_NODE_LEAF = 0x80000000

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_LIB") or os.path.join(_HERE, "lib", "librt_mi355x.so")


class rt_node(C.Structure):
    _fields_ = [("bmin", C.c_float * 3), ("bmax", C.c_float * 3), ("first", C.c_uint32), ("count_flags", C.c_uint32)]


class rt_material(C.Structure):
    _fields_ = [("kd", C.c_float * 3), ("ks", C.c_float * 3), ("shininess", C.c_float),
                ("optical_density", C.c_float), ("illum", C.c_int32)]


class rt_scene(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("nodes", C.POINTER(rt_node)),
                ("n_face_refs", C.c_uint32), ("face_refs", C.POINTER(C.c_uint32)),
                ("n_faces", C.c_uint32),
                ("tri_verts", C.POINTER(C.c_float)), ("face_normal", C.POINTER(C.c_float)),
                ("tri_vid", C.POINTER(C.c_uint32)), ("mat_id", C.POINTER(C.c_int32)),
                ("n_vert_normals", C.c_uint32), ("vert_normal", C.POINTER(C.c_float)),
                ("n_materials", C.c_uint32), ("materials", C.POINTER(rt_material)),
                ("model", C.c_float * 12)]


class rt_camera(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("inv_view", C.c_float * 12), ("fovy", C.c_float),
                ("aspect", C.c_float), ("viewport", C.c_float * 4)]


class rt_lights(C.Structure):
    _fields_ = [("n_lights", C.c_int32), ("pos", (C.c_float * 3) * RT_MAX_LIGHTS), ("color", C.c_float * 3),
                ("mode", C.c_int32), ("usteps", C.c_int32), ("vsteps", C.c_int32),
                ("len_x", C.c_float), ("len_y", C.c_float), ("n_offsets", C.c_int32), ("offsets", C.POINTER(C.c_float))]


class rt_params(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("max_depth", C.c_int32),
                ("row0", C.c_int32), ("row1", C.c_int32), ("stripe", C.c_int32), ("rank", C.c_int32),
                ("nranks", C.c_int32), ("collect_stats", C.c_int32)]


class rt_debug_hit(C.Structure):
    _fields_ = [("level", C.c_int32), ("status", C.c_int32), ("face", C.c_int32), ("t", C.c_float), ("pos", C.c_float * 3), ("dir", C.c_float * 3),
                ("hit_point", C.c_float * 3), ("normal", C.c_float * 3), ("reflected", C.c_float * 3), ("color", C.c_float * 3),
                ("light_visible", C.c_uint8 * RT_MAX_LIGHTS), ("pad", C.c_uint8 * 3)]


class rt_stats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_bounce", C.c_uint64), ("rays_centre", C.c_uint64),
                ("rays_sample", C.c_uint64), ("pixels", C.c_uint64), ("pixels_culled", C.c_uint64),
                ("shaded_hits", C.c_uint64), ("box_tests", C.c_uint64), ("leaf_tri_refs", C.c_uint64),
                ("box_tests_shadow", C.c_uint64), ("leaf_tri_refs_shadow", C.c_uint64),
                ("ms_trace", C.c_float), ("ms_shadow", C.c_float), ("ms_shade", C.c_float),
                ("ms_resolve", C.c_float), ("ms_total", C.c_float),
                ("launches_trace", C.c_uint32), ("launches_shadow", C.c_uint32), ("launches_shade", C.c_uint32),
                ("launches_total", C.c_uint32), ("rays_sample_walked", C.c_uint64)]

    def total_rays(self):
        """Rays as SURVEY.md §8(d) counts them: every traversal query, incl. the root-AABB-only culled pixels."""
        return (self.rays_primary + self.rays_bounce + self.rays_centre + self.rays_sample + self.pixels_culled)


# every symbol include/rt_mi355x.h declares: (name, restype, argtypes)
_P = C.POINTER
_SIGNATURES = [
    ("rt_create", C.c_int, [_P(C.c_void_p), C.c_int]),
    ("rt_destroy", None, [C.c_void_p]),
    ("rt_last_error", C.c_char_p, [C.c_void_p]),
    ("rt_version", C.c_char_p, []),
    ("rt_stream", C.c_void_p, [C.c_void_p]),
    ("rt_upload_scene", C.c_int, [C.c_void_p, _P(rt_scene)]),
    ("rt_render", C.c_int, [C.c_void_p, _P(rt_camera), _P(rt_lights), _P(rt_params), C.c_void_p, C.c_void_p, _P(rt_stats)]),
    ("rt_synchronize", C.c_int, [C.c_void_p]),
    ("rt_render_device", C.c_int, [C.c_void_p, _P(rt_camera), _P(rt_lights), _P(rt_params), C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, _P(rt_stats)]),
    ("rt_timing_collect", C.c_int, [C.c_void_p, _P(rt_stats)]),
    ("rt_graph_create", C.c_int, [C.c_void_p, _P(rt_lights), _P(rt_params), C.c_void_p, C.c_void_p, _P(C.c_void_p)]),
    ("rt_graph_launch", C.c_int, [C.c_void_p, _P(rt_camera), C.c_void_p]),
    ("rt_graph_stats", C.c_int, [C.c_void_p, _P(rt_stats)]),
    ("rt_graph_destroy", None, [C.c_void_p]),
    ("rt_comm_unique_id", C.c_int, [C.c_void_p]),
    ("rt_comm_create", C.c_int, [_P(C.c_void_p), C.c_int, C.c_void_p, C.c_int32, C.c_int32]),
    ("rt_comm_destroy", None, [C.c_void_p]),
    ("rt_comm_last_error", C.c_char_p, [C.c_void_p]),
    ("rt_comm_gather_rows", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int32, C.c_void_p]),
    ("rt_render_gather", C.c_int, [C.c_void_p, C.c_void_p, _P(rt_camera), _P(rt_lights), _P(rt_params), C.c_void_p, C.c_size_t, C.c_void_p,
                                   C.c_int32, C.c_void_p]),
    ("rt_stitch_rows", C.c_int, [C.c_void_p, C.c_size_t, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    ("rt_local_rows", C.c_int32, [_P(rt_params)]),
    ("rt_trace_rays", C.c_int, [C.c_void_p, _P(rt_lights), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p]),
    ("rt_debug_ray", C.c_int, [C.c_void_p, _P(rt_camera), _P(rt_lights), C.c_float, C.c_float, C.c_int32, _P(rt_debug_hit), _P(C.c_int32)]),
    ("rt_light_strikes", C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("rt_box_intersect", C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("rt_tree_probe", C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("rt_primary_points", C.c_int, [C.c_void_p, _P(rt_camera), C.c_int32, C.c_int32, C.c_void_p]),
    ("rt_host_scene_load", C.c_int, [C.c_char_p, C.c_int32, C.c_int32, _P(C.c_void_p)]),
    ("rt_host_scene_free", None, [C.c_void_p]),
    ("rt_host_scene_view", C.c_int, [C.c_void_p, _P(rt_scene)]),
    ("rt_host_scene_set_model", C.c_int, [C.c_void_p, _P(C.c_float), C.c_int32]),
    ("rt_host_scene_build_gpu", C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    ("rt_host_scene_info", C.c_int, [C.c_void_p, _P(C.c_int32), _P(C.c_float)]),
    ("rt_debug_chunk_stats", C.c_int, [_P(rt_scene), _P(C.c_int32)]),
    ("rt_debug_chunk_bounds", C.c_int, [_P(rt_scene), _P(C.c_float), C.c_int32, _P(C.c_int32), _P(C.c_uint32), _P(C.c_uint32)]),
    ("rt_debug_work_counters", C.c_int, [C.c_void_p, _P(C.c_uint64), C.c_int32]),
    ("rt_default_camera", None, [_P(rt_camera), C.c_int32, C.c_int32]),
    ("rt_yaw_camera", None, [_P(rt_camera), C.c_int32, C.c_int32, C.c_float]),
    ("rt_screen_to_world", None, [_P(rt_camera), C.c_float, C.c_float, _P(C.c_float)]),
    ("rt_default_lights", None, [_P(rt_lights), C.c_int32]),
    ("rt_sphere_offsets", None, [C.c_uint32, C.c_float, C.c_int32, C.c_void_p]),
    ("rt_write_ppm", C.c_int, [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32]),
    ("rt_write_ppm_u8", C.c_int, [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32]),
    ("rt_write_pfm", C.c_int, [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32]),
]
EXPORTED_SYMBOLS = [s[0] for s in _SIGNATURES]

_lib = None


def load_library(path=None):
    """Load librt_mi355x.so and bind every declared symbol.  Raises (never falls back) if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    # PyTorch-ROCm bundles its own libamdhip64 under the same soname as /opt/rocm's.  Whichever is mapped first serves BOTH this library
    # and torch; when this library came first torch later found "No HIP GPUs".  So: if torch is installed, let it map its runtime first.
    try:
        import torch  # noqa: F401
    except Exception:  # noqa: BLE001 -- torch is plumbing, not a requirement of the C ABI
        pass
    if not os.path.exists(p):
        raise RuntimeError(
            f"{p} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C raytracer-in-cpp_amd/csrc`). There is no CPU fallback for the ray-trace path.")
    lib = C.CDLL(p)
    for name, res, args in _SIGNATURES:
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


class RtError(RuntimeError):
    pass


def check(lib, ctx, status, what):
    if status != RT_OK:
        msg = lib.rt_last_error(ctx) if ctx else b""
        raise RtError(f"{what} failed with status {status}: {msg.decode() if msg else ''}")
