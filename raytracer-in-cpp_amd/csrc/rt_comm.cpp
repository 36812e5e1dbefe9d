// rt_comm.cpp -- multi-GPU assembly of a row-sharded frame BEHIND the C ABI: one process per GPU, the scene replicated, every rank
// renders its interleaved row stripes (rt_params.stripe / rank / nranks) and ONE RCCL gather over xGMI brings the quantised rows to the
// root, which de-interleaves them for the P3 writer.  The reference has no counterpart (single process, std::thread pool:
// src/flyscene.cpp:558-629); this is what lets a C++ host -- which is what the reference is (src/main.cpp:70) -- use the 8 GPUs of a
// node through include/rt_mi355x.h alone.
//
// RCCL is bound at run time (dlopen "librccl.so.1", falling back to "librccl.so"): the library has no link-time dependency on it, a
// single-GPU user never loads it, and inside a process that already carries an RCCL (PyTorch-ROCm ships its own copy) the same
// soname resolves to the copy that is already mapped instead of a second one.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <string>

#include "rt_mi355x.h"

namespace {
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Gather)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

RcclApi &api() {
    static RcclApi a;
    if (a.handle || !a.err.empty()) return a;
    // RTLD_NOLOAD first: reuse an RCCL the process already has (same soname), then load the system one
    const char *names[] = {"librccl.so.1", "librccl.so"};
    for (const char *n : names) { a.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (a.handle) break; }
    for (const char *n : names) { if (a.handle) break; a.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL); }
    if (!a.handle) { a.err = std::string("cannot load librccl: ") + dlerror(); return a; }
    auto sym = [&](const char *name) { void *p = dlsym(a.handle, name); if (!p && a.err.empty()) a.err = std::string("librccl lacks ") + name; return p; };
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
    a.Send = reinterpret_cast<decltype(a.Send)>(sym("ncclSend"));
    a.Recv = reinterpret_cast<decltype(a.Recv)>(sym("ncclRecv"));
    a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
    a.Gather = reinterpret_cast<decltype(a.Gather)>(dlsym(a.handle, "ncclGather"));      // RCCL extension (rccl.h:745); optional
    return a;
}
}  // namespace

struct rt_comm {
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0, device = 0;
    std::string err;
};

static_assert(sizeof(ncclUniqueId) == RT_COMM_ID_BYTES, "RT_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");

extern "C" rt_status rt_comm_unique_id(uint8_t id[RT_COMM_ID_BYTES]) {
    if (!id) return RT_ERR_INVALID;
    RcclApi &a = api();
    if (!a.err.empty()) { std::fprintf(stderr, "rt_mi355x: %s\n", a.err.c_str()); return RT_ERR_UNSUPPORTED; }
    ncclUniqueId u;
    if (a.GetUniqueId(&u) != ncclSuccess) return RT_ERR_HIP;
    std::memcpy(id, &u, sizeof u);
    return RT_OK;
}

extern "C" rt_status rt_comm_create(rt_comm **out, int device, const uint8_t id[RT_COMM_ID_BYTES], int32_t nranks, int32_t rank) {
    if (!out || !id || nranks < 1 || rank < 0 || rank >= nranks) return RT_ERR_INVALID;
    *out = nullptr;
    RcclApi &a = api();
    if (!a.err.empty()) { std::fprintf(stderr, "rt_mi355x: %s\n", a.err.c_str()); return RT_ERR_UNSUPPORTED; }
    if (hipSetDevice(device) != hipSuccess) return RT_ERR_NO_DEVICE;
    rt_comm *c = new rt_comm();
    c->nranks = nranks; c->rank = rank; c->device = device;
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    const ncclResult_t r = a.CommInitRank(&c->comm, nranks, u, rank);
    if (r != ncclSuccess) {
        std::fprintf(stderr, "rt_mi355x: ncclCommInitRank failed: %s\n", a.GetErrorString ? a.GetErrorString(r) : "?");
        delete c;
        return RT_ERR_HIP;
    }
    *out = c;
    return RT_OK;
}

extern "C" void rt_comm_destroy(rt_comm *c) {
    if (!c) return;
    if (c->comm) { (void)hipSetDevice(c->device); (void)api().CommDestroy(c->comm); }
    delete c;
}

extern "C" const char *rt_comm_last_error(const rt_comm *c) { return c ? c->err.c_str() : "rt_mi355x: null communicator"; }

// The single exchange of a frame: every rank contributes `bytes` bytes (its rows, zero padded to the common size), the root receives
// nranks * bytes, rank r's block at offset r * bytes.  Asynchronous on `stream` (capturable: no host synchronisation, no allocation).
extern "C" rt_status rt_comm_gather_rows(rt_comm *c, const void *d_local, size_t bytes, void *d_gathered, int32_t root, void *stream) {
    if (!c || !d_local || root < 0 || root >= c->nranks || (c->rank == root && !d_gathered)) return RT_ERR_INVALID;
    RcclApi &a = api();
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipSetDevice(c->device) != hipSuccess) return RT_ERR_NO_DEVICE;
    ncclResult_t r = ncclSuccess;
    if (a.Gather) {
        r = a.Gather(d_local, d_gathered, bytes, ncclUint8, root, c->comm, st);
    } else {
        // the same exchange spelt with point-to-point calls (one group): no ring, every link carries one block to the root
        r = a.GroupStart();
        if (r == ncclSuccess) r = a.Send(d_local, bytes, ncclUint8, root, c->comm, st);
        if (r == ncclSuccess && c->rank == root)
            for (int p = 0; p < c->nranks && r == ncclSuccess; ++p)
                r = a.Recv(static_cast<uint8_t *>(d_gathered) + static_cast<size_t>(p) * bytes, bytes, ncclUint8, p, c->comm, st);
        const ncclResult_t e = a.GroupEnd();
        if (r == ncclSuccess) r = e;
    }
    if (r != ncclSuccess) { c->err = std::string("RCCL gather failed: ") + (a.GetErrorString ? a.GetErrorString(r) : "?"); return RT_ERR_HIP; }
    return RT_OK;
}

// renders this rank's stripes (params carry stripe / rank / nranks) into d_local_u8 and enqueues the gather behind it on the same stream
extern "C" rt_status rt_render_gather(rt_ctx *ctx, rt_comm *c, const rt_camera *cam, const rt_lights *lights, const rt_params *p, uint8_t *d_local_u8,
                                      size_t local_bytes, uint8_t *d_gathered_u8, int32_t root, void *stream) {
    if (!ctx || !c || !p || !d_local_u8) return RT_ERR_INVALID;
    if (p->nranks != c->nranks || p->rank != c->rank) return RT_ERR_INVALID;
    const size_t need = static_cast<size_t>(rt_local_rows(p)) * static_cast<size_t>(p->width) * 3;
    if (need > local_bytes) return RT_ERR_INVALID;
    if (!stream) stream = rt_stream(ctx);          // NULL = the context's own stream for BOTH halves (never the legacy null stream for the gather)
    const rt_status s = rt_render_device(ctx, cam, lights, p, nullptr, d_local_u8, nullptr, stream, nullptr);
    if (s != RT_OK) return s;
    return rt_comm_gather_rows(c, d_local_u8, local_bytes, d_gathered_u8, root, stream);
}

// root, host side: gathered[r * block_bytes ...] holds rank r's rows in increasing y -> full frame [H][W][3]
extern "C" rt_status rt_stitch_rows(const uint8_t *gathered, size_t block_bytes, int32_t width, int32_t height, int32_t stripe, int32_t nranks, uint8_t *frame) {
    if (!gathered || !frame || width <= 0 || height <= 0 || stripe <= 0 || nranks <= 0) return RT_ERR_INVALID;
    const size_t row = static_cast<size_t>(width) * 3;
    for (int32_t r = 0; r < nranks; ++r) {
        size_t k = 0;
        for (int32_t y = 0; y < height; ++y) {
            if ((y / stripe) % nranks != r) continue;
            if ((k + 1) * row > block_bytes) return RT_ERR_INVALID;
            std::memcpy(frame + static_cast<size_t>(y) * row, gathered + static_cast<size_t>(r) * block_bytes + k * row, row);
            ++k;
        }
    }
    return RT_OK;
}
