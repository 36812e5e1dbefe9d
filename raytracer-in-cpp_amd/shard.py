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


# ---- the same exchange behind the C ABI (include/rt_mi355x.h: rt_comm_*): RCCL bound by the library itself, no torch collective ----
class Comm:
    """rt_comm: one RCCL communicator per process/GPU.  `id_bytes` comes from Comm.unique_id() on rank 0 and is handed to the other
    ranks out of band (bench.py broadcasts it with torch.distributed)."""

    def __init__(self, device, id_bytes, nranks, rank):
        import ctypes as C
        from . import capi
        self.lib = capi.load_library()
        self.handle = C.c_void_p()
        buf = (C.c_uint8 * capi.RT_COMM_ID_BYTES).from_buffer_copy(bytes(id_bytes))
        st = self.lib.rt_comm_create(C.byref(self.handle), int(device), buf, int(nranks), int(rank))
        if st != capi.RT_OK:
            raise capi.RtError(f"rt_comm_create(nranks={nranks}, rank={rank}) failed with status {st}")
        self.nranks, self.rank = int(nranks), int(rank)

    @staticmethod
    def unique_id():
        import ctypes as C
        from . import capi
        lib = capi.load_library()
        buf = (C.c_uint8 * capi.RT_COMM_ID_BYTES)()
        st = lib.rt_comm_unique_id(buf)
        if st != capi.RT_OK:
            raise capi.RtError(f"rt_comm_unique_id failed with status {st} (librccl not loadable?)")
        return bytes(buf)

    def gather_rows(self, d_local_ptr, nbytes, d_gathered_ptr, root=0, stream_ptr=None):
        import ctypes as C
        from . import capi
        st = self.lib.rt_comm_gather_rows(self.handle, C.c_void_p(d_local_ptr), int(nbytes), C.c_void_p(d_gathered_ptr) if d_gathered_ptr else None,
                                          int(root), C.c_void_p(stream_ptr) if stream_ptr else None)
        if st != capi.RT_OK:
            raise capi.RtError(f"rt_comm_gather_rows failed: {self.lib.rt_comm_last_error(self.handle).decode()}")

    def close(self):
        if self.handle:
            self.lib.rt_comm_destroy(self.handle)
            import ctypes as C
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def stitch_u8(gathered, block_bytes, width, height, stripe, nranks):
    """rt_stitch_rows: the root's de-interleave of the gathered blocks, host side (numpy uint8 in, [height, width, 3] out)."""
    import ctypes as C
    from . import capi
    lib = capi.load_library()
    g = np.ascontiguousarray(gathered, np.uint8)
    out = np.zeros((height, width, 3), np.uint8)
    st = lib.rt_stitch_rows(g.ctypes.data_as(C.c_void_p), int(block_bytes), int(width), int(height), int(stripe), int(nranks), out.ctypes.data_as(C.c_void_p))
    if st != capi.RT_OK:
        raise capi.RtError(f"rt_stitch_rows failed with status {st}")
    return out
