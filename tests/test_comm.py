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
