"""Multi-GPU path, rehearsed on CPU: two processes over gloo render interleaved row stripes, ONE gather assembles the
frame on rank 0.  The renderer here is the CPU oracle (tests may use it); the sharding/stitching/gather code is the
product's (raytracer-in-cpp_amd/shard.py), identical to what bench.py runs over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, stripe, out_path):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import rtpkg
    import oracle_lib
    pkg = rtpkg.load()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = oracle_lib.load()
    osc = orc.load_scene(os.path.join(ROOT, "tests", "golden", "scenes", "cube.obj"))
    cam, lights = orc.camera(w, h), orc.lights(area=True, usteps=3, vsteps=3)
    rows = pkg.shard.rows_of_rank(h, stripe, rank, world)
    maxr = pkg.shard.max_local_rows(h, stripe, world)
    local = np.zeros((maxr, w, 3), np.uint8)
    for k, y in enumerate(rows):                      # this rank renders only its own rows
        rgb, _, _ = osc.render(cam, lights, w, h, max_depth=2, threads=1, row0=y, row1=y + 1)
        local[k] = np.clip(orc.quantise(rgb[0]), 0, 255).astype(np.uint8)
    frame = pkg.shard.gather_frame(torch.from_numpy(local.reshape(-1)), h, w, 3, stripe, rank, world)
    if rank == 0:
        np.save(out_path, frame)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("stripe", [8, 5])
def test_two_rank_row_shard_gather_matches_single_rank(oracle, scenes, tmp_path, stripe):
    import torch.multiprocessing as mp
    w, h, world = 40, 37, 2
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), w, h, stripe, out), nprocs=world, join=True)
    got = np.load(out)
    osc = oracle.load_scene(os.path.join(scenes, "cube.obj"))
    rgb, _, _ = osc.render(oracle.camera(w, h), oracle.lights(area=True, usteps=3, vsteps=3), w, h, max_depth=2, threads=2)
    want = np.clip(oracle.quantise(rgb), 0, 255).astype(np.uint8)
    assert np.array_equal(got, want)
    osc.close()


def test_row_partition_is_a_partition(rt):
    sh = rt.shard
    for (h, s, r) in [(1080, 8, 8), (1080, 8, 3), (37, 5, 2), (7, 8, 4), (2160, 16, 8), (1, 1, 1)]:
        seen = []
        for k in range(r):
            rows = sh.rows_of_rank(h, s, k, r)
            assert rows == sorted(rows)
            seen += rows
        assert sorted(seen) == list(range(h))
        assert sh.max_local_rows(h, s, r) >= (h + r - 1) // r - s
    parts = [np.full((sh.max_local_rows(20, 4, 3), 6, 1), k, np.uint8) for k in range(3)]
    full = sh.stitch(parts, 20, 6, 1, 4, 3)
    assert [int(full[y, 0, 0]) for y in range(20)] == [(y // 4) % 3 for y in range(20)]
