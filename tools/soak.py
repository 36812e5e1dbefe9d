# soak: N frames of one scene, eager, every frame's 8-bit output hashed -- all hashes must be equal (no run-to-run difference, no hang): python tools/soak.py <cube|dodge|wavy> N [W H grid depth]
import ctypes as C, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench, rtpkg
pkg = rtpkg.load()
scene, n = sys.argv[1], int(sys.argv[2])
W, H, G, D = (int(x) for x in (sys.argv[3:7] if len(sys.argv) > 6 else (1920, 1080, 8, 4)))
name, path = bench.scene_of(scene)
hs = pkg.HostScene(path, 1000, 15)
ctx = pkg.Context(0); ctx.upload(hs)
cam = pkg.default_camera(W, H); L = pkg.make_lights(area=True, usteps=G, vsteps=G); p = pkg.make_params(W, H, D)
out = pkg.hipmem.DeviceBuffer(W * H * 3 * 4); out8 = pkg.hipmem.DeviceBuffer(W * H * 3)
seen = {}
t0 = time.time()
for i in range(n):
    pkg.capi.check(ctx.lib, ctx.handle, ctx.lib.rt_render_device(ctx.handle, C.byref(cam), C.byref(L), C.byref(p), C.c_void_p(out.address), C.c_void_p(out8.address), None, None, None), "render")
    if i % 50 == 0 or i == n - 1:
        pkg.capi.check(ctx.lib, ctx.handle, ctx.lib.rt_synchronize(ctx.handle), "rt_synchronize")         # (rt_render_device is asynchronous)
        h = hashlib.sha256(out8.to_numpy(np.uint8, (H, W, 3)).tobytes()).hexdigest()
        seen[h] = seen.get(h, 0) + 1
print(scene, n, "frames in %.1f s;" % (time.time() - t0), "distinct frame hashes:", len(seen))
sys.exit(0 if len(seen) == 1 else 1)
