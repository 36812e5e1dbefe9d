# executed-work counters of the TRACE kernels (counting build): python tools/work_trace.py [dodge|wavy|cube] W H grid depth
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
pkg = bench.rtpkg.load() if hasattr(bench, "rtpkg") else __import__("rtpkg").load()
scene = sys.argv[1] if len(sys.argv) > 1 else "dodge"
W, H, G, D = (int(x) for x in (sys.argv[2:6] if len(sys.argv) > 5 else (1920, 1080, 8, 4)))
name, path = bench.scene_of(scene)
hs = pkg.HostScene(path, 1000, 15)
capi = pkg.capi
lib = capi.load_library(os.path.join(ROOT, "raytracer-in-cpp_amd", "lib", "librt_mi355x_work.so"))
ctx = C.c_void_p()
assert lib.rt_create(C.byref(ctx), 0) == 0
capi.check(lib, ctx, lib.rt_upload_scene(ctx, C.byref(hs.view)), "upload")
cam = pkg.default_camera(W, H); L = pkg.make_lights(area=True, usteps=G, vsteps=G); p = pkg.make_params(W, H, D)
rgb = np.zeros(W * H * 3, np.float32)
capi.check(lib, ctx, lib.rt_render(ctx, C.byref(cam), C.byref(L), C.byref(p), rgb.ctypes.data_as(C.c_void_p), None, None), "render")
buf = (C.c_uint64 * 768)()
capi.check(lib, ctx, lib.rt_debug_work_counters(ctx, buf, 768), "counters")
t = [int(x) for x in buf][0:96]
names = {0: "ray-mode triangle steps", 1: "ray-mode useful lanes", 2: "tri-mode (ray,chunk) steps", 3: "tri-mode useful lanes", 4: "per-ray node box steps", 6: "leaves ray-mode", 7: "leaves tri-mode",
         8: "live rays at tri-mode leaves", 12: "chunk tests (per ray)", 13: "units (all trace launches)", 14: "chunk-culled (ray,chunk)", 66: "content-culled (ray,node)",
         70: "cone tri tests (chunks)", 71: "cone tri survivors", 72: "cone rays at test", 73: "cone empty chunks", 74: "lanes at per-ray node tests", 75: "lanes hit",
         88: "groups popped", 89: "children in groups", 90: "children after cone", 91: "children hit by some ray"}
for k, v in names.items():
    print(f"  [{k:2d}] {v:32s} {t[k]:12d}")
