# one-off exactness check at full size: culled == RT_NO_CULL=1 bit for bit (GPU vs GPU): python tools/cull_check.py <dodge|wavy> W H grid depth
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench, rtpkg
pkg = rtpkg.load()
scene = sys.argv[1]
W, H, G, D = (int(x) for x in sys.argv[2:6])
name, path = bench.scene_of(scene)
hs = pkg.HostScene(path, 1000, 15)
frames = []
for no_cull in (False, True):
    if no_cull: os.environ["RT_NO_CULL"] = "1"
    ctx = pkg.Context(0); ctx.upload(hs)
    cam = pkg.default_camera(W, H); L = pkg.make_lights(area=True, usteps=G, vsteps=G); p = pkg.make_params(W, H, D)
    rgb = np.zeros((H, W, 3), np.float32); hits = np.zeros((H, W), np.int32)
    t0 = time.time()
    pkg.capi.check(ctx.lib, ctx.handle, ctx.lib.rt_render(ctx.handle, C.byref(cam), C.byref(L), C.byref(p), rgb.ctypes.data_as(C.c_void_p), hits.ctypes.data_as(C.c_void_p), None), "render")
    print("no_cull" if no_cull else "culled", "%.2f s" % (time.time() - t0), flush=True)
    frames.append((rgb, hits)); ctx.close()
same = np.array_equal(frames[0][1], frames[1][1]) and np.array_equal(frames[0][0].view(np.uint32), frames[1][0].view(np.uint32))
print(scene, W, H, G, D, "IDENTICAL" if same else "DIFFERENT", "hit pixels", int((frames[0][1] >= 0).sum()))
sys.exit(0 if same else 1)
