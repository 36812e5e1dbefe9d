# throughput of K frames issued alternately on TWO contexts / streams (frame i + 1's latency-bound trace kernels beside frame i's shading) against the
# same K frames on one stream: python tools/pipelined.py [cube|dodge|wavy] [W H grid depth]
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench, rtpkg
pkg = rtpkg.load()
scene = sys.argv[1] if len(sys.argv) > 1 else "cube"
W, H, G, D = (int(x) for x in (sys.argv[2:6] if len(sys.argv) > 5 else (1920, 1080, 8, 4)))
name, path = bench.scene_of(scene)
hs = pkg.HostScene(path, 1000, 15)
dev = torch.device("cuda", 0)
K = 200 if scene != "wavy" else 8
cam = pkg.default_camera(W, H); L = pkg.make_lights(area=True, usteps=G, vsteps=G); p = pkg.make_params(W, H, D)
ctxs, outs, streams = [], [], []
for i in range(2):
    c = pkg.Context(0); c.upload(hs); ctxs.append(c)
    outs.append((torch.zeros(H * W * 3, dtype=torch.float32, device=dev), torch.zeros(H * W * 3, dtype=torch.uint8, device=dev)))
    streams.append(torch.cuda.Stream(device=dev))
def frame(i, which):
    c = ctxs[which]
    st = c.lib.rt_render_device(c.handle, C.byref(cam), C.byref(L), C.byref(p), C.c_void_p(outs[which][0].data_ptr()), C.c_void_p(outs[which][1].data_ptr()), None,
                                C.c_void_p(streams[which].cuda_stream), None)
    pkg.capi.check(c.lib, c.handle, st, "render")
for mode in ("one stream", "two streams"):
    for i in range(10): frame(i, 0 if mode == "one stream" else i & 1)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(K): frame(i, 0 if mode == "one stream" else i & 1)
    torch.cuda.synchronize(dev)
    print(f"{scene} {mode}: {(time.perf_counter() - t0) / K * 1e3:.4f} ms per frame")
assert torch.equal(outs[0][1], outs[1][1])
