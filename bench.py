#!/usr/bin/env python3
"""bench.py -- measures BASELINE.json's metric: Mrays/s (+ ms/frame) of the ray-trace path at 1920x1080, depth 4,
64-sample (8x8) area light, default scene (resources/models/cube.obj -> tests/golden/scenes/cube.obj), N GPUs.

A "step" is one whole frame: primary generation + closest hit + light-centre visibility + area-light sample
shadow rays + shading + bounces + resolve/quantise (+ the RCCL row gather when N > 1).  Inputs (flattened octree,
triangle records, materials) are resident in HBM before the timed region; output stays on the device.
`value` = rays of the whole frame (all ranks) per second, a ray being one traversal query as SURVEY.md §8(d) defines it.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--scene cube|dodge] [--width 1920 --height 1080 --grid 8 --depth 4]
N > 1 is launched by torch.distributed.run (one rank per GPU, RCCL).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VALU_PEAK_PER_SIMD_NS = 0.967   # measured: tools/micro/valu_rate.hip (plain FP32 wave64 ops, 8 waves/SIMD), 256 CUs x 4 SIMDs
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
BOX_BYTES, TRI_REF_BYTES = 24, 52   # algorithmic bytes per box test / per leaf triangle reference


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--scene", default="cube", choices=["cube", "dodge", "wavy"],
                    help="wavy = the ~1M-triangle synthetic mesh of BASELINE cfg4 (use with --width 3840 --height 2160 --grid 16 --depth 8)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--grid", type=int, default=8, help="area-light grid side (8 -> 64 samples)")
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--stripe", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="N > 1: eager launches + synchronous gather instead of graph replay + pipelined gather")
    ap.add_argument("--cpu-stride", type=int, default=0, help="oracle pixel stride for the CPU baseline (0 = auto)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import rtpkg
    pkg = rtpkg.load()
    capi = pkg.capi

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    dist = None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the ray-trace path has no CPU fallback")
    # RT_DIST_BACKEND=gloo rehearses the N > 1 path on a ONE-GPU box: every rank renders its stripes on GPU 0 and the
    # gather runs over gloo on host copies.  The real path (default) is one rank per GPU with RCCL ("nccl").
    backend = os.environ.get("RT_DIST_BACKEND", "nccl")
    gpu_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(gpu_index)
    dev = torch.device("cuda", gpu_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    on_host = world > 1 and backend != "nccl"       # collectives on CPU tensors (rehearsal only)

    if args.scene == "wavy":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import scenes_gen
        import tempfile
        scene_file = "wavy708.obj (synthetic 708x708 displaced grid + floor, 1,002,530 triangles)"
        scene_path = scenes_gen.wavy_grid(os.path.join(tempfile.gettempdir(), f"rt_wavy_{os.getpid()}"), n=708)
    else:
        scene_file = {"cube": "cube.obj", "dodge": "dodgeColorTest.obj"}[args.scene]
        scene_path = os.path.join(ROOT, "tests", "golden", "scenes", scene_file)
    W, H, G, D, S = args.width, args.height, args.grid, args.depth, args.stripe

    hs = pkg.HostScene(scene_path, 1000, 15)
    ctx = pkg.Context(gpu_index)
    ctx.upload(hs)
    lib = ctx.lib
    cam = pkg.default_camera(W, H)
    L = pkg.make_lights(area=True, usteps=G, vsteps=G)
    max_rows = pkg.shard.max_local_rows(H, S, world)
    my_rows = pkg.shard.rows_of_rank(H, S, rank, world)
    out_rgb = torch.zeros(max_rows * W * 3, dtype=torch.float32, device=dev)
    out_u8 = torch.zeros(max_rows * W * 3, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev)

    def params(collect):
        return pkg.make_params(W, H, D, 0, H, S, rank, world, collect_stats=False) if collect == 0 else _p(collect)

    def _p(collect):
        p = pkg.make_params(W, H, D, 0, H, S, rank, world)
        p.collect_stats = collect
        return p

    def render(p, stats=None):
        st = lib.rt_render_device(ctx.handle, C.byref(cam), C.byref(L), C.byref(p), C.c_void_p(out_rgb.data_ptr()),
                                  C.c_void_p(out_u8.data_ptr()), None, C.c_void_p(stream.cuda_stream),
                                  C.byref(stats) if stats is not None else None)
        capi.check(lib, ctx.handle, st, "rt_render_device")

    def gather_rows():
        src = out_u8.cpu() if on_host else out_u8
        bufs = [torch.empty_like(src) for _ in range(world)] if rank == 0 else None
        dist.gather(src, gather_list=bufs, dst=0)         # the single RCCL exchange of the frame
        return bufs

    def reduce_(t, op):
        if on_host:
            c = t.cpu()
            dist.all_reduce(c, op=op)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=op)

    def step(p):
        render(p)
        if world > 1:
            return gather_rows()
        return None

    # ---- untimed: algorithmic counters of this rank's rows (no-early-out counting variants) ----------------
    cnt = capi.rt_stats()
    render(_p(1), cnt)
    torch.cuda.synchronize(dev)
    rays_local = cnt.total_rays()
    counters = torch.tensor([rays_local, cnt.box_tests, cnt.leaf_tri_refs, cnt.box_tests_shadow, cnt.leaf_tri_refs_shadow,
                             cnt.rays_sample, cnt.rays_primary, cnt.rays_centre, cnt.rays_bounce, cnt.pixels_culled],
                            dtype=torch.float64, device=dev)
    if world > 1:
        reduce_(counters, dist.ReduceOp.SUM)
    tot = [int(x) for x in counters.tolist()]
    rays_frame = tot[0]

    # ---- N > 1: the per-rank frame is ~1/N of the work, so launch overhead and the gather dominate.  The step is then
    # a hipGraph REPLAY of the frame (one host call instead of 17 launches) and the gather is asynchronous and
    # double-buffered: render i+1 overlaps gather i (all K gathers complete inside the timed region).  N = 1 keeps eager
    # launches with per-kernel HIP events, as the roofline contract asks.
    pipe = None
    if world > 1 and not on_host and not args.eager:
        try:
            u8 = [out_u8, torch.zeros_like(out_u8)]
            graphs = [pkg.FrameGraph(ctx, L, _p(0), out_rgb.data_ptr(), u8[i].data_ptr()) for i in range(2)]
            recv = [[torch.empty_like(out_u8) for _ in range(world)] for _ in range(2)] if rank == 0 else [None, None]
            pipe = {"u8": u8, "graphs": graphs, "recv": recv, "pending": []}
        except Exception as e:        # fall back to the synchronous eager step
            print(f"[bench] graph/pipeline setup failed on rank {rank}: {e}; using the eager step", file=sys.stderr)
            pipe = None

    def step_pipe(i):
        b = i & 1
        pend = pipe["pending"]
        if len(pend) >= 2:
            pend[-2].wait()                       # stream-level wait: buffer b was read by gather i-2
        pipe["graphs"][b].launch(cam, stream.cuda_stream)
        pend.append(dist.gather(pipe["u8"][b], gather_list=pipe["recv"][b], dst=0, async_op=True))

    def drain_pipe():
        for h in pipe["pending"]:
            h.wait()
        pipe["pending"].clear()

    # ---- warmup ----------------------------------------------------------------------------------------------
    p_timed = _p(2)           # deferred per-kernel HIP events on the launch stream, no host sync inside the step
    p_plain = _p(0)
    if pipe:
        try:                                      # exercise every pipeline call (incl. the i-2 wait) before the timed region
            for i in range(max(3, args.warmup)):
                step_pipe(i)
            drain_pipe()
            torch.cuda.synchronize(dev)
        except Exception as e:
            print(f"[bench] pipelined step failed on rank {rank}: {e}; using the eager step", file=sys.stderr)
            pipe = None
    # every rank must take the same path (the collectives differ): agree on the minimum
    if world > 1:
        flag = torch.tensor([1.0 if pipe else 0.0], dtype=torch.float64, device=dev)
        reduce_(flag, dist.ReduceOp.MIN)
        if flag.item() < 0.5:
            pipe = None
    if not pipe:
        for i in range(args.warmup):
            step(p_timed)
    torch.cuda.synchronize(dev)
    warm = capi.rt_stats()
    lib.rt_timing_collect(ctx.handle, C.byref(warm))

    # ---- timed region: EXACTLY K steps between barrier + synchronize ---------------------------------------
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(args.steps):
        if pipe:
            step_pipe(i)
        else:
            # per-kernel HIP events on every 4th step only: the ~17 event records of a frame cost ~60 us (6 % of the cube frame)
            step(p_timed if i % 4 == 0 else p_plain)
    if pipe:
        drain_pipe()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if world > 1:
        reduce_(elapsed, dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    tim = capi.rt_stats()
    capi.check(lib, ctx.handle, lib.rt_timing_collect(ctx.handle, C.byref(tim)), "rt_timing_collect")
    # untimed: one fully instrumented frame (an event between every pair of launches) for the per-kernel breakdown
    brk = capi.rt_stats()
    render(_p(0), brk)
    torch.cuda.synchronize(dev)

    # ---- extra (untimed for `value`): the same K frames replayed from ONE captured hipGraph, camera yaw stepping
    # 2*pi/120 per frame (BASELINE cfg5's animation path); reported as graph_replay_ms_per_frame
    graph_ms = None
    graph_same_ms = None
    try:
        g = pkg.FrameGraph(ctx, L, _p(0), out_rgb.data_ptr(), out_u8.data_ptr())
        cams = [pkg.default_camera(W, H, float(np.float32(2.0 * np.pi * f / 120.0))) for f in range(args.steps)]
        g.launch(cams[0], stream.cuda_stream)
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        tg0 = time.perf_counter()
        for f in range(args.steps):
            g.launch(cams[f], stream.cuda_stream)
            if world > 1:
                gather_rows()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        graph_ms = (time.perf_counter() - tg0) / args.steps * 1e3
        tg0 = time.perf_counter()
        for f in range(args.steps):               # same camera as the timed region: the launch-overhead comparison
            g.launch(cam, stream.cuda_stream)
            if world > 1:
                gather_rows()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        graph_same_ms = (time.perf_counter() - tg0) / args.steps * 1e3
        g.close()
    except Exception as e:            # the graph path is an extra; never let it take the bench line down
        graph_ms = f"failed: {e}"

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    K = args.steps
    ms_step = elapsed / K * 1e3
    value = rays_frame * K / elapsed / 1e6
    # roofline of the dominant kernel (k_shadow: area-light sample shadow rays), rank 0's launches.
    # achieved = algorithmic bytes per launch / average launch duration (HIP events on the launch stream).
    timing_source = "HIP events on the launch stream inside the timed region"
    if tim.launches_shadow == 0:          # graph replay (N > 1): no per-kernel events inside the timed region
        tim = brk
        timing_source = "one instrumented eager frame outside the timed region (the timed loop replays a hipGraph)"
        K_t = 1
    else:
        K_t = (K + 3) // 4                # frames of the timed region that carried events (every 4th)
        timing_source += f" (every 4th step: {K_t} of {K} frames)"
    launches = max(1, tim.launches_shadow)
    avg_ms_shadow = tim.ms_shadow / launches
    alg_shadow_frame = BOX_BYTES * cnt.box_tests_shadow + TRI_REF_BYTES * cnt.leaf_tri_refs_shadow   # rank 0's rows
    launches_per_frame = launches / K_t
    alg_per_launch = alg_shadow_frame / launches_per_frame
    achieved = alg_per_launch / (avg_ms_shadow * 1e-3) / 1e9 if avg_ms_shadow > 0 else 0.0
    trace_alg = BOX_BYTES * (cnt.box_tests - cnt.box_tests_shadow) + TRI_REF_BYTES * (cnt.leaf_tri_refs - cnt.leaf_tri_refs_shadow)
    trace_gbs = trace_alg / (brk.ms_trace * 1e-3) / 1e9 if brk.ms_trace > 0 else 0.0
    # HBM traffic of the dominant kernel from the PMC counters (separate rocprofv3 --pmc passes, profiles/traffic.json);
    # only quoted when the committed measurement is for this very workload
    traffic = None
    traffic_note = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        ent = tj.get(args.scene)
        if ent and ent["config"] == f"{W}x{H} depth {D} {G * G} samples" and world == 1:
            traffic = ent["hbm_bytes_per_launch"]
            traffic_note = ent["how"]
    except Exception:
        pass
    # What actually bounds the kernel: VALU issue.  Wave-instructions of the level-0 launch from the committed PMC pass
    # (profiles/valu.json, tools/valu.sh) over this run's live launch time, against the plain-FP32 issue rate measured on this
    # chip by tools/micro/valu_rate.hip (0.967 wave-instructions per SIMD per ns = one wave64 v_mul/v_add/v_fma every 2 cycles;
    # v_cmp ~3, v_readlane/v_div_* ~4, v_rcp ~8 cycles: the kernel's own mix cannot reach 1.0).
    valu = None
    try:
        vj = json.load(open(os.path.join(ROOT, "profiles", "valu.json")))
        ent = vj.get(args.scene)
        if ent and ent["config"] == f"{W}x{H} depth {D} {G * G} samples" and world == 1:
            ks = {k: v for k, v in ent["kernels"].items() if "k_shadow" in k}
            insts = sum(v["valu_wave_instructions"] for v in ks.values())
            t_s = (tim.ms_shadow / K_t) * 1e-3        # all k_shadow launches of a frame (level 0 dominates; leaf-task launch included)
            rate = insts / t_s / 1e9 / 1024.0 if t_s > 0 else 0.0
            valu = {"wave_instructions_per_frame_level0": int(insts), "achieved_per_simd_per_ns": round(rate, 4),
                    "peak_per_simd_per_ns": VALU_PEAK_PER_SIMD_NS, "frac": round(rate / VALU_PEAK_PER_SIMD_NS, 4),
                    "source": ent["how"]}
    except Exception:
        pass
    roofline = {
        "bound": "hbm", "kernel": "k_shadow", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_note,
        "algorithmic_bytes_per_launch": int(alg_per_launch), "avg_launch_ms": round(avg_ms_shadow, 5),
        "launches_per_frame": launches_per_frame,
        "note": "algorithmic bytes = 24 B x box tests + 52 B x leaf triangle refs in reference semantics (every triangle of every "
                "intersected leaf, no early-out: SURVEY 8d); the kernel skips most of that work exactly (culling) and the scene is "
                "cache resident, so this is neither HBM traffic nor bounded by the HBM peak -- valu_issue is the real bound (DESIGN.md 5)",
        "valu_issue": valu,
        "k_trace": {"achieved": round(trace_gbs, 1), "ms_per_frame": round(brk.ms_trace, 4)},
        "timing_source": timing_source,
        "ms_per_frame": {"shadow": round(tim.ms_shadow / K_t, 4), "device_total": round(tim.ms_total / K_t, 4),
                         "instrumented_frame": {"trace": round(brk.ms_trace, 4), "shadow": round(brk.ms_shadow, 4),
                                                "shade": round(brk.ms_shade, 4), "resolve": round(brk.ms_resolve, 4),
                                                "total": round(brk.ms_total, 4)}},
    }

    out = {
        "metric": "Mrays/s", "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
        "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{W}x{H} depth {D} {G * G}-sample area light, {scene_file}, "
                               f"1 light, row stripes of {S} over {world} GPU(s)",
                   "width": W, "height": H, "max_depth": D, "samples": G * G, "scene": scene_file,
                   "parallelism": f"rows{world}", "step": ("hipGraph replay + pipelined RCCL gather" if pipe else ("eager launches" + (" + gather" if world > 1 else "")))},
        "rays_per_frame": rays_frame,
        "graph_replay_ms_per_frame_120_frame_yaw_path": (round(graph_ms, 4) if isinstance(graph_ms, float) else graph_ms),
        "graph_replay_ms_per_frame_same_camera": (round(graph_same_ms, 4) if isinstance(graph_same_ms, float) else graph_same_ms),
        "rays": {"primary": tot[6], "centre": tot[7], "sample": tot[5], "bounce": tot[8], "culled_pixels": tot[9]},
        "roofline": roofline,
    }

    # ---- CPU baseline beside it: the oracle (a port of the reference's algorithm), rank 0, N = 1 only ------
    if world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        orc = oracle_lib.load()
        osc = orc.load_scene(scene_path)
        # the reference uses hardware_concurrency()-1 threads (flyscene.cpp:558); a 1-GPU box's CPU share is 16 cores
        threads = max(1, min(16, os.cpu_count() or 2) - 1)
        stride = args.cpu_stride or (8 if args.scene == "wavy" else 1)     # cfg4's whole 4K frame takes the oracle minutes: 1/64 of the pixels
        # whole frames of the same workload, repeated until ~10 s of CPU work (threads x wall) have been timed
        n = 0; sec = 0.0; cpu_rays = 0; reps = 0
        while reps == 0 or (sec * threads < 10.0 and reps < 64):
            n1, s1, ost = osc.render_subsample(orc.camera(W, H), orc.lights(area=True, usteps=G, vsteps=G), W, H, stride, max_depth=D, threads=threads)
            n += n1; sec += s1; cpu_rays += ost.total_rays(); reps += 1
        out["cpu_baseline"] = {
            "value": round(cpu_rays / sec / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{reps} x every {stride}th pixel in x and y of the same {W}x{H} frame ({n} pixels, {cpu_rays} rays) in {sec:.2f} s "
                      f"on {threads} threads = {sec * threads:.0f} s of CPU work; "
                      "oracle/rt_oracle.c (C restatement, no per-node deep copies) -- NOT the unmodified reference, which "
                      "measured 0.165-2.6 Mrays/s on 7 threads (BASELINE.md)",
            "seconds": round(sec, 3),
        }
        out["speedup_vs_cpu_port"] = round(value / (cpu_rays / sec / 1e6), 1)
        osc.close()
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
