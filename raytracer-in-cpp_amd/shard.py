"""Row sharding of one frame across the GPUs of a node (one process per GPU).

Rank r of R renders the rows y with (y // stripe) % R == r (interleaved stripes: background rows cost one box
test, hit rows 66+ rays, so contiguous blocks would be badly unbalanced for a centred object).  The scene is
replicated (read-only, MBs); the only exchange is ONE gather of the finished rows to rank 0 over RCCL/xGMI
(`torch.distributed.gather`, backend "nccl" on GPUs, "gloo" in the CPU tests), then rank 0 de-interleaves.
The reference has no counterpart (single process, std::thread pool: src/flyscene.cpp:558-629).
"""
import numpy as np


def rows_of_rank(height, stripe, rank, nranks, row0=0, row1=None):
    row1 = height if row1 is None else row1
    return [y for y in range(row0, row1) if ((y - row0) // stripe) % nranks == rank]


def max_local_rows(height, stripe, nranks):
    return max(len(rows_of_rank(height, stripe, r, nranks)) for r in range(nranks))


def stitch(parts, height, width, channels, stripe, nranks, dtype=None):
    """parts[r]: array [>= len(rows_r), width, channels] of rank r's rows in increasing y -> full [height,width,channels]."""
    first = np.asarray(parts[0])
    out = np.zeros((height, width, channels), dtype or first.dtype)
    for r in range(nranks):
        rows = rows_of_rank(height, stripe, r, nranks)
        if rows:
            out[rows] = np.asarray(parts[r])[: len(rows)].reshape(len(rows), width, channels)
    return out


def gather_frame(local, height, width, channels, stripe, rank, nranks, dst=0, group=None):
    """local: torch tensor [max_local_rows*width*channels] (this rank's rows, zero padded to the common size).
    Returns the stitched numpy frame on rank `dst`, None elsewhere.  One collective, no ring."""
    import torch
    import torch.distributed as dist
    if nranks == 1:
        rows = len(rows_of_rank(height, stripe, 0, 1))
        return stitch([local.detach().cpu().numpy().reshape(-1, width, channels)[:rows]], height, width, channels, stripe, 1)
    bufs = [torch.empty_like(local) for _ in range(nranks)] if rank == dst else None
    dist.gather(local, gather_list=bufs, dst=dst, group=group)
    if rank != dst:
        return None
    parts = [b.detach().cpu().numpy().reshape(-1, width, channels) for b in bufs]
    return stitch(parts, height, width, channels, stripe, nranks)
