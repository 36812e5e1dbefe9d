# Unit-duration histograms of the trace stages at benchmark clocks: 60 frames back to back, then one frame whose fill_stats prints (RT_DEBUG).
# Needs the diagnostic build:  make -C raytracer-in-cpp_amd/csrc ab AB_FLAGS=-DRT_UNIT_HIST      usage: python tools/unit_hist.py [dodge|wavy] W H grid depth
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if not os.environ.get("RT_PRODUCT"): os.environ["RT_LIB"] = os.path.join(ROOT, "raytracer-in-cpp_amd", "lib", "librt_mi355x_ab.so")
os.environ["RT_UNIT_DUMP"] = os.path.join(ROOT, "gpurun_out", "unit_dump_%s.bin" % (sys.argv[1] if len(sys.argv) > 1 else "dodge"))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import rtpkg
pkg = rtpkg.load()
scene = sys.argv[1] if len(sys.argv) > 1 else "dodge"
W, H, G, D = (int(x) for x in (sys.argv[2:6] if len(sys.argv) > 5 else (1920, 1080, 8, 4)))
name, path = bench.scene_of(scene)
hs = pkg.HostScene(path, 1000, 15)
ctx = pkg.Context(0)
ctx.upload(hs)
lib, capi = ctx.lib, pkg.capi
cam = pkg.default_camera(W, H); L = pkg.make_lights(area=True, usteps=G, vsteps=G)
out = pkg.hipmem.DeviceBuffer(W * H * 3 * 4)
p = pkg.make_params(W, H, D)
for i in range(60 if scene != "wavy" else 4):
    capi.check(lib, ctx.handle, lib.rt_render_device(ctx.handle, C.byref(cam), C.byref(L), C.byref(p), C.c_void_p(out.address), None, None, None, None), "render")
st = capi.rt_stats()
capi.check(lib, ctx.handle, lib.rt_render_device(ctx.handle, C.byref(cam), C.byref(L), C.byref(p), C.c_void_p(out.address), None, None, None, C.byref(st)), "render")
print("ms trace %.4f shadow %.4f shade %.4f total %.4f" % (st.ms_trace, st.ms_shadow, st.ms_shade, st.ms_total))

import numpy as np
raw = np.fromfile(os.environ["RT_UNIT_DUMP"], dtype=np.uint32)
U = raw[:4 * 65536 * 8].reshape(4, 65536, 8)
Wv = raw[4 * 65536 * 8:4 * 65536 * 8 + 4 * 16384 * 4].reshape(4, 16384, 4)
names = ["stage0 walk", "stage0 tasks", "stage1 walk", "stage1 tasks"]
for k in range(4):
    u = U[k]; u = u[u[:, 0] > 0]
    w = Wv[k]; w = w[w[:, 0] > 0]
    if len(u) == 0: continue
    cyc = u[:, 0].astype(np.float64) * 16
    print(f"== {names[k]}: units {len(u)}, waves {len(w)}; unit kcycles: sum {cyc.sum()/1e3:.0f} mean {cyc.mean()/1e3:.1f} p50 {np.percentile(cyc,50)/1e3:.1f} p90 {np.percentile(cyc,90)/1e3:.1f} p99 {np.percentile(cyc,99)/1e3:.1f} max {cyc.max()/1e3:.1f}")
    wl = w[:, 0].astype(np.float64) * 16
    print(f"   wave lifetime kcycles: mean {wl.mean()/1e3:.1f} p50 {np.percentile(wl,50)/1e3:.1f} p90 {np.percentile(wl,90)/1e3:.1f} p99 {np.percentile(wl,99)/1e3:.1f} max {wl.max()/1e3:.1f}; units/wave max {w[:,1].max()}")
    order = np.argsort(-cyc)[:12]
    print("   longest units (kcycles, groups, leaves, task-atomics, ray-mode tris, tri-mode steps, chunk tests, wave):")
    for i in order:
        print("     ", f"{cyc[i]/1e3:8.1f}", u[i, 1:8].tolist())
    # cost model: cycles ~ a*groups + b*leaves + c*atomics + d*raytris + e*tristeps + f*chunktests + g
    A = np.column_stack([u[:, 1:7].astype(np.float64), np.ones(len(u))])
    coef, *_ = np.linalg.lstsq(A, cyc, rcond=None)
    print("   least-squares cycles per: group %.0f leaf %.0f task-atomic %.0f ray-mode-tri %.0f tri-mode-step %.0f chunk-test %.0f unit %.0f" % tuple(coef))

# ---- the level-0 k_shadow_shaft launch (tree scenes, > 32 samples)
off = 4 * 65536 * 8 + 4 * 16384 * 4
if raw.size >= off + 65536 * 16:
    Sh = raw[off:off + 65536 * 16].reshape(65536, 16)
    u = Sh[Sh[:, 0] > 0]
    if len(u):
        cyc = u[:, 0].astype(np.float64) * 16
        names = ["groups", "shaft-survivors", "children-hit", "chunk-batches", "chunks-with-work", "tri-shaft-tests", "ray-mode-tri-steps", "(ray,chunk)-steps"]
        print(f"== k_shadow_shaft level 0: sampled units {len(u)}; kcycles mean {cyc.mean()/1e3:.1f} p50 {np.percentile(cyc,50)/1e3:.1f} p90 {np.percentile(cyc,90)/1e3:.1f} p99 {np.percentile(cyc,99)/1e3:.1f} max {cyc.max()/1e3:.1f}")
        print("   mean steps per unit:", {n: round(float(u[:, 1 + i].mean()), 2) for i, n in enumerate(names)})
        A = np.column_stack([u[:, 1:9].astype(np.float64), np.ones(len(u))])
        coef, *_ = np.linalg.lstsq(A, cyc, rcond=None)
        print("   least-squares cycles per step:", {n: int(c) for n, c in zip(names + ["unit"], coef)})
        share = {n: round(float(coef[i] * u[:, 1 + i].sum() / cyc.sum()), 3) for i, n in enumerate(names)}
        share["unit"] = round(float(coef[8] * len(u) / cyc.sum()), 3)
        print("   share of the unit time:", share)
        if u[:, 10].max() > 0:
            nv = u[:, 10].astype(np.float64); no = u[:, 9].astype(np.float64)
            for nm, m in (("all rays occluded", no == nv), ("no ray occluded", no == 0), ("mixed", (no > 0) & (no < nv))):
                if m.any():
                    print(f"   units with {nm}: {m.mean():.3f} of the units, {cyc[m].sum()/cyc.sum():.3f} of the time, mean kcycles {cyc[m].mean()/1e3:.1f}, groups {u[m,1].mean():.1f} "
                          f"chunk-batches {u[m,4].mean():.1f} tri-shaft {u[m,6].mean():.1f} steps {u[m,7].mean()+u[m,8].mean():.1f}")
        for lo, hi in ((0, 50), (50, 90), (90, 99), (99, 100)):
            a, b = np.percentile(cyc, lo), np.percentile(cyc, hi)
            m = (cyc >= a) & (cyc <= b)
            print(f"   units p{lo}-p{hi}: share of time {cyc[m].sum()/cyc.sum():.2f}, mean groups {u[m,1].mean():.1f} children-hit {u[m,3].mean():.1f} chunk-batches {u[m,4].mean():.1f} tri-shaft {u[m,6].mean():.1f} steps {u[m,7].mean()+u[m,8].mean():.1f}")
        order = np.argsort(-cyc)[:12]
        print("   longest shaft units (kcycles | groups, shaft-survivors, children-hit, chunk-batches, chunks-with-work, tri-shaft-tests, ray-mode-tri, (ray,chunk)-steps | occluded, valid):")
        for i in order:
            print("     ", f"{cyc[i]/1e3:8.1f}", u[i, 1:9].tolist(), u[i, 9:11].tolist())
