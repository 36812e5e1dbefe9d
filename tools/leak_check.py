import ctypes as C, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench, rtpkg
pkg = rtpkg.load()
name, path = bench.scene_of("dodge")
hs = pkg.HostScene(path, 1000, 15)
def free_mb():
    torch.cuda.synchronize(); f, t = torch.cuda.mem_get_info(); return f / 2**20
base = None
for it in range(60):
    ctx = pkg.Context(0); ctx.upload(hs)
    w, h = 640, 400
    cam = pkg.default_camera(w, h); L = pkg.make_lights(area=True, usteps=16, vsteps=16); p = pkg.make_params(w, h, 3)
    rgb = np.zeros((h, w, 3), np.float32)
    pkg.capi.check(ctx.lib, ctx.handle, ctx.lib.rt_render(ctx.handle, C.byref(cam), C.byref(L), C.byref(p), rgb.ctypes.data_as(C.c_void_p), None, None), "render")
    out = pkg.hipmem.DeviceBuffer(h * w * 3 * 4)
    g = pkg.FrameGraph(ctx, L, p, out.address, 0)
    g.launch(cam); g.stats(); g.close(); out.free()
    ctx.close()
    if it == 4: base = free_mb()
    if it in (4, 30, 59): print("iteration", it, "free MiB %.0f" % free_mb(), flush=True)
print("leak MiB over 55 create/render/graph/destroy cycles: %.1f" % (base - free_mb()))
