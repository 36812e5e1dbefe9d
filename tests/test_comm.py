"""Multi-GPU behind the C ABI (rt_comm_*, rt_render_gather, rt_stitch_rows): the stripe arithmetic on the CPU, and on the GPU the
whole call chain with a one-rank RCCL communicator (the only size a one-GPU box can form; the 8-GPU curve is the driver's to run)."""
import ctypes as C
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("h,stripe,n", [(1080, 8, 8), (37, 8, 3), (64, 1, 4), (10, 16, 2)])
def test_stitch_rows_matches_the_python_shard_arithmetic(rt, h, stripe, n):
    w = 13
    rng = np.random.default_rng(h * 31 + n)
    frame = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    rows = [rt.shard.rows_of_rank(h, stripe, r, n) for r in range(n)]
    block_rows = max(len(x) for x in rows)
    blocks = np.zeros((n, block_rows, w, 3), np.uint8)
    for r in range(n):
        blocks[r, :len(rows[r])] = frame[rows[r]]
        assert rt.load_library().rt_local_rows(C.byref(rt.make_params(w, h, 0, 0, h, stripe, r, n))) == len(rows[r])
    out = rt.shard.stitch_u8(blocks.reshape(-1), block_rows * w * 3, w, h, stripe, n)
    assert np.array_equal(out, frame)
    assert np.array_equal(out, rt.shard.stitch([blocks[r] for r in range(n)], h, w, 3, stripe, n))


@pytest.mark.gpu
def test_render_gather_one_rank_equals_plain_render(rt):
    import torch
    scene = os.path.join(HERE, "golden", "scenes", "dodgeColorTest.obj")
    hs = rt.HostScene(scene, 1000, 15)
    ctx = rt.Context(0)
    ctx.upload(hs)
    w, h, stripe = 160, 96, 8
    cam, L = rt.default_camera(w, h), rt.make_lights(area=True, usteps=8, vsteps=8)
    p = rt.make_params(w, h, 2, 0, h, stripe, 0, 1)
    ref = torch.zeros(h * w * 3, dtype=torch.uint8, device="cuda")
    st = ctx.lib.rt_render_device(ctx.handle, C.byref(cam), C.byref(L), C.byref(p), None, C.c_void_p(ref.data_ptr()), None, None, None)
    rt.capi.check(ctx.lib, ctx.handle, st, "rt_render_device")
    comm = rt.shard.Comm(0, rt.shard.Comm.unique_id(), 1, 0)
    local = torch.zeros_like(ref)
    gathered = torch.zeros_like(ref)
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()                      # an explicit stream; NULL would mean the context's own stream for both halves
    st = ctx.lib.rt_render_gather(ctx.handle, comm.handle, C.byref(cam), C.byref(L), C.byref(p), C.c_void_p(local.data_ptr()), local.numel(),
                                  C.c_void_p(gathered.data_ptr()), 0, C.c_void_p(stream.cuda_stream))
    rt.capi.check(ctx.lib, ctx.handle, st, "rt_render_gather")
    stream.synchronize()
    assert torch.equal(gathered, ref) and ref.any()
    gathered.zero_()
    torch.cuda.synchronize()                          # the clear runs on torch's stream, the gather on the context's
    st = ctx.lib.rt_render_gather(ctx.handle, comm.handle, C.byref(cam), C.byref(L), C.byref(p), C.c_void_p(local.data_ptr()), local.numel(),
                                  C.c_void_p(gathered.data_ptr()), 0, None)
    rt.capi.check(ctx.lib, ctx.handle, st, "rt_render_gather")
    torch.cuda.synchronize()
    assert torch.equal(gathered, ref) and ref.any()
    full = rt.shard.stitch_u8(gathered.cpu().numpy(), local.numel(), w, h, stripe, 1)
    assert np.array_equal(full.reshape(-1), ref.cpu().numpy())
    comm.close(); ctx.close(); hs.close()


def _run_bench(args, env_extra=None, launcher=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    env = dict(os.environ)
    env.update(env_extra or {})
    cmd = (launcher or [sys.executable]) + [os.path.join(root, "bench.py")] + args
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
def test_bench_multi_gpu_step_runs_with_world_size_one_over_rccl():
    """The N > 1 step of bench.py -- hipGraph replay on one stream, rt_comm_gather_rows (RCCL) on a second, two buffers, four events -- with a
    ONE-rank communicator on one GPU (bench.py --force-multi): >= 8 timed steps, and the frame stitched from the gathered rows of the last
    step equals the plain rt_render_device frame byte for byte (bench.py itself exits non-zero on a mismatch)."""
    rec = _run_bench(["--gpus", "1", "--force-multi", "--steps", "9", "--warmup", "4", "--scene", "dodge", "--width", "640", "--height", "360"],
                     {"MASTER_PORT": "29541"})
    assert "rt_comm_gather_rows (RCCL via the C ABI)" in rec["config"]["step"] and "hipGraph replay" in rec["config"]["step"]
    gf = rec["gathered_frame"]
    assert gf["match"] and gf["nonzero"] and gf["assembled_frame_sha256"] == gf["ungathered_frame_sha256"]
    assert rec["steps"] == 9 and rec["n_gpus"] == 1
    pr = rec["per_rank"]
    assert len(pr) == 1 and pr[0]["device_ms_per_frame"] > 0 and pr[0]["launches_per_frame"] >= 6


@pytest.mark.gpu
def test_bench_two_rank_gloo_rehearsal_takes_the_pipelined_step():
    """Two ranks on ONE GPU (RT_DIST_BACKEND=gloo): the same double-buffered step with a host-side stand-in for the exchange only; the
    stitched frame of the two ranks' interleaved stripes equals the plain render of the whole frame."""
    import sys
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29543"]
    rec = _run_bench(["--gpus", "2", "--steps", "8", "--warmup", "4", "--scene", "cube", "--width", "480", "--height", "270"],
                     {"RT_DIST_BACKEND": "gloo"}, launcher)
    assert "stand-in" in rec["config"]["step"] and "double buffered" in rec["config"]["step"]
    assert rec["gathered_frame"]["match"] and rec["n_gpus"] == 2
    assert [p["rank"] for p in rec["per_rank"]] == [0, 1]
