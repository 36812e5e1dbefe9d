// rt_build.hip -- the reference's octree construction ON THE GPU: BoxTree(mesh, capacity) / BoxTree::split /
// BoxTree::clasifyFace (src/boxTree.cpp:11-31, 88-147, 203-336) reproduced bit for bit -- the SAT on NORMALISED centre-relative
// vectors, the octant boxes with their float expressions, "fewer than capacity" leaves and the lost "exactly capacity" nodes --
// so that the tree equals the host build (HostScene::build_octree) node for node and face list for face list.
//
// Level-synchronous: every node that has to split contributes 8 (node, child) PAIRS; one launch classifies every face of every
// parent against its 8 child boxes (lane = face, one byte per (pair, face)), the per-pair counts go back to the host (one small
// read per level: the statuses leaf / empty / split / lost and the output offsets are decided there, as the recursion of the
// reference decides them), and a second launch compacts the accepted faces STABLY (the reference appends in parent order) into
// the next level's face lists.  ~6 levels x 2 launches for the 1M-triangle scene.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "host_scene.hpp"

namespace rtamd {

namespace {

struct BPair {                 // one (open node, child octant)
    float lo[3], hi[3];        // the child's box
    uint32_t src_off, src_cnt; // the parent's face list in the current refs buffer
    uint32_t flag_off;         // this pair's bytes in the flag buffer
    uint32_t out_off;          // where the child's face list goes in the next refs buffer
};

__device__ __forceinline__ float bmin(float a, float b) { return (b < a) ? b : a; }     // std::min
__device__ __forceinline__ float bmax(float a, float b) { return (a < b) ? b : a; }     // std::max
__device__ __forceinline__ float bdot(const float *a, const float *b) { return a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]); }
__device__ __forceinline__ void bunit(float *v) {          // Vector3f::normalized() (Eigen/src/Core/Dot.h:124-134)
    const float z = v[0] * v[0] + (v[1] * v[1] + v[2] * v[2]);
    if (z > 0.f) { const float s = sqrtf(z); v[0] = v[0] / s; v[1] = v[1] / s; v[2] = v[2] / s; }
}
__device__ __forceinline__ bool bsep(float p_first, float p_second, float rad) {      // axisTestX01 / Y02 / Z0 / X02 / Y1
    const float hi = bmax(p_second, p_first), lo = bmin(p_second, p_first);
    return lo > rad || hi < -rad;
}
__device__ __forceinline__ bool bsep_z12(float p1, float p2, float rad) {             // axisTestZ12: std::max(p1, p2)
    const float hi = bmax(p1, p2), lo = bmin(p1, p2);
    return lo > rad || hi < -rad;
}

// BoxTree::clasifyFace (boxTree.cpp:203-336) -- the same statements as HostScene::face_touches
__device__ bool classify(const float *lo, const float *hi, const float *tv) {
    for (int k = 0; k < 3; ++k) {
        const float *v = tv + k * 3;
        if (lo[0] <= v[0] && hi[0] >= v[0] && lo[1] <= v[1] && hi[1] >= v[1] && lo[2] <= v[2] && hi[2] >= v[2]) return true;
    }
    float mid[3], h[3], A[3], B[3], C[3];
    for (int k = 0; k < 3; ++k) mid[k] = lo[k] + (hi[k] - lo[k]) / 2.f;
    for (int k = 0; k < 3; ++k) { h[k] = hi[k] - mid[k]; A[k] = tv[k] - mid[k]; B[k] = tv[3 + k] - mid[k]; C[k] = tv[6 + k] - mid[k]; }
    bunit(h); bunit(A); bunit(B); bunit(C);
    float e0[3], e1[3], e2[3];
    for (int k = 0; k < 3; ++k) { e0[k] = B[k] - A[k]; e1[k] = C[k] - B[k]; e2[k] = A[k] - C[k]; }
    {
        const float fx = fabsf(e0[0]), fy = fabsf(e0[1]), fz = fabsf(e0[2]);
        if (bsep(e0[2] * A[1] - e0[1] * A[2], e0[2] * C[1] - e0[1] * C[2], fz * h[1] + fy * h[2])) return false;
        if (bsep(-e0[2] * A[0] + e0[0] * A[2], -e0[2] * C[0] + e0[0] * C[2], fz * h[0] + fx * h[2])) return false;
        if (bsep_z12(e0[1] * B[0] - e0[0] * B[1], e0[1] * C[0] - e0[0] * C[1], fy * h[0] + fx * h[1])) return false;
    }
    {
        const float fx = fabsf(e1[0]), fy = fabsf(e1[1]), fz = fabsf(e1[2]);
        if (bsep(e1[2] * A[1] - e1[1] * A[2], e1[2] * C[1] - e1[1] * C[2], fz * h[1] + fy * h[2])) return false;
        if (bsep(-e1[2] * A[0] + e1[0] * A[2], -e1[2] * C[0] + e1[0] * C[2], fz * h[0] + fx * h[2])) return false;
        if (bsep(e1[1] * A[0] - e1[0] * A[1], e1[1] * B[0] - e1[0] * B[1], fy * h[0] + fx * h[1])) return false;
    }
    {
        const float fx = fabsf(e2[0]), fy = fabsf(e2[1]), fz = fabsf(e2[2]);
        if (bsep(e2[2] * A[1] - e2[1] * A[2], e2[2] * B[1] - e2[1] * B[2], fz * h[1] + fy * h[2])) return false;
        if (bsep(-e2[2] * A[0] + e2[0] * A[2], -e2[2] * B[0] + e2[0] * B[2], fz * h[0] + fx * h[2])) return false;
        if (bsep_z12(e2[1] * B[0] - e2[0] * B[1], e2[1] * C[0] - e2[0] * C[1], fy * h[0] + fx * h[1])) return false;
    }
    for (int k = 0; k < 3; ++k) {                        // findMinMax + slab per axis
        const float mn = bmin(bmin(A[k], B[k]), C[k]);
        const float mx = bmax(bmax(A[k], B[k]), C[k]);
        if (mn > h[k] || mx < -h[k]) return false;
    }
    float d1[3], d2[3], n[3];
    for (int k = 0; k < 3; ++k) { d1[k] = A[k] - B[k]; d2[k] = A[k] - C[k]; }
    n[0] = d1[1] * d2[2] - d1[2] * d2[1]; n[1] = d1[2] * d2[0] - d1[0] * d2[2]; n[2] = d1[0] * d2[1] - d1[1] * d2[0];
    bunit(n);
    // BoxTree::planeBoxOverlap (boxTree.cpp:345-366)
    float vlo[3], vhi[3];
    for (int i = 0; i < 3; ++i) {
        const float v = A[i];
        if (n[i] > 0.0f) { vlo[i] = -h[i] - v; vhi[i] = h[i] - v; }
        else { vlo[i] = h[i] - v; vhi[i] = -h[i] - v; }
    }
    if (bdot(n, vlo) > 0.0f) return false;
    return bdot(n, vhi) >= 0.0f;
}

// lane = (pair, face of the pair's parent): one byte per decision, the pair's count via one atomic per block
__global__ __launch_bounds__(256) void k_classify(const BPair *__restrict__ pairs, const uint32_t *__restrict__ blk_pair, const uint32_t *__restrict__ blk_first,
                                                  const uint32_t *__restrict__ src, const float *__restrict__ tri_verts, uint8_t *__restrict__ flags,
                                                  uint32_t *__restrict__ counts) {
    const uint32_t pi = blk_pair[blockIdx.x];
    const BPair P = pairs[pi];
    const uint32_t i = blk_first[blockIdx.x] + threadIdx.x;
    bool in = false;
    if (i < P.src_cnt) {
        const uint32_t f = src[P.src_off + i];
        in = classify(P.lo, P.hi, tri_verts + static_cast<size_t>(f) * 9);
        flags[P.flag_off + i] = in ? 1 : 0;
    }
    const int c = __syncthreads_count(in ? 1 : 0);
    if (threadIdx.x == 0 && c) atomicAdd(&counts[pi], static_cast<uint32_t>(c));
}

// one block per pair: STABLE compaction of the parent's list (the reference appends accepted faces in the parent's order)
__global__ __launch_bounds__(256) void k_scatter(const BPair *__restrict__ pairs, const uint32_t *__restrict__ src, const uint8_t *__restrict__ flags,
                                                 uint32_t *__restrict__ dst) {
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_base;
    const BPair P = pairs[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_base = 0u;
    __syncthreads();
    for (uint32_t c0 = 0; c0 < P.src_cnt; c0 += 256u) {
        const uint32_t i = c0 + threadIdx.x;
        const bool in = i < P.src_cnt && flags[P.flag_off + i] != 0;
        const unsigned long long m = __ballot(in);
        const uint32_t below = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
        if (lane == 0) s_wave[wave] = static_cast<uint32_t>(__popcll(m));
        __syncthreads();
        uint32_t off = s_base;
        for (int w = 0; w < wave; ++w) off += s_wave[w];
        if (in) dst[P.out_off + off + below] = src[P.src_off + i];
        __syncthreads();
        if (threadIdx.x == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
}

__global__ void k_iota(uint32_t *p, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = i;
}

struct Dev {
    void *p = nullptr;
    ~Dev() { if (p) (void)hipFree(p); }
    bool alloc(size_t bytes) { if (p) { (void)hipFree(p); p = nullptr; } return hipMalloc(&p, bytes ? bytes : 4) == hipSuccess; }
};

}  // namespace

#define BCHK(call)                                                                                   \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) { if (err) *err = std::string(#call) + ": " + hipGetErrorString(e_); return false; } \
    } while (0)

// Builds hs.pool on the GPU (world vertices and capacity / depth as HostScene::build_octree); the caller flattens.
bool gpu_build_octree(HostScene &hs, int cap, int depth, hipStream_t st, std::string *err) {
    hs.capacity = cap; hs.max_depth = depth;
    hs.pool.clear(); hs.overflow = false; hs.total_refs = 0;
    const size_t F = hs.tris.size();
    // BoundingBox(Mesh&): boundingBox.cpp:14-43 (the running maximum starts at FLT_MIN) -- on the host, O(F)
    V3 lo{3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, hi{1.175494351e-38f, 1.175494351e-38f, 1.175494351e-38f};
    std::vector<float> tv(F * 9);
    for (size_t i = 0; i < F; ++i)
        for (int k = 0; k < 3; ++k) {
            const V3 &v = hs.world[hs.tris[i].vid[k]];
            lo = {min_std(lo.x, v.x), min_std(lo.y, v.y), min_std(lo.z, v.z)};
            hi = {max_std(hi.x, v.x), max_std(hi.y, v.y), max_std(hi.z, v.z)};
            tv[i * 9 + k * 3] = v.x; tv[i * 9 + k * 3 + 1] = v.y; tv[i * 9 + k * 3 + 2] = v.z;
        }
    OctNode root;
    root.box = {lo, hi};
    hs.pool.push_back(root);
    if (F <= static_cast<size_t>(cap)) {
        if (F == 0) hs.pool[0].empty = true;
        else { hs.pool[0].leaf = true; hs.pool[0].faces.resize(F); for (size_t i = 0; i < F; ++i) hs.pool[0].faces[i] = static_cast<int>(i); }
        return true;
    }
    Dev d_tv, d_cur, d_next, d_pairs, d_blkp, d_blkf, d_flags, d_counts;
    if (!d_tv.alloc(F * 9 * sizeof(float)) || !d_cur.alloc(F * sizeof(uint32_t))) { if (err) *err = "hipMalloc failed"; return false; }
    BCHK(hipMemcpyAsync(d_tv.p, tv.data(), F * 9 * sizeof(float), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_iota, dim3(1024), dim3(256), 0, st, static_cast<uint32_t *>(d_cur.p), static_cast<uint32_t>(F));

    struct Open { int node; int depth; uint32_t off, cnt; };
    std::vector<Open> open{{0, depth, 0u, static_cast<uint32_t>(F)}};
    auto twice = [](const V3 &v) { return V3{2.f * v.x, 2.f * v.y, 2.f * v.z}; };
    while (!open.empty()) {
        // ---- the 8 octants of every open node (BoxTree::split, boxTree.cpp:93-120: the float expressions of HostScene::subdivide)
        std::vector<BPair> pairs;
        std::vector<uint32_t> blk_pair, blk_first;
        uint32_t flag_total = 0;
        for (const Open &o : open) {
            hs.pool[o.node].leaf = false;
            const V3 blo = hs.pool[o.node].box.lo, bhi = hs.pool[o.node].box.hi;
            const float dx = (bhi.x - blo.x) / 2, dy = (bhi.y - blo.y) / 2, dz = (bhi.z - blo.z) / 2;
            const V3 ex{dx, 0, 0}, ey{0, dy, 0}, ez{0, 0, dz};
            const AABB oct[8] = {
                {blo, ((blo + ex) + ey) + ez},
                {blo + ez, ((blo + ex) + ey) + twice(ez)},
                {blo + ey, ((blo + ex) + twice(ey)) + ez},
                {(blo + ey) + ez, ((blo + ex) + twice(ey)) + twice(ez)},
                {blo + ex, ((blo + twice(ex)) + ey) + ez},
                {(blo + ex) + ez, bhi - ey},
                {(blo + ex) + ey, bhi - ez},
                {((blo + ex) + ey) + ez, bhi},
            };
            const int level = hs.pool[o.node].level + 1;
            std::vector<int> kids(8);
            for (int k = 0; k < 8; ++k) {
                OctNode c;
                c.box = oct[k];
                c.level = level;
                kids[k] = static_cast<int>(hs.pool.size());
                hs.pool.push_back(c);
                BPair p{};
                p.lo[0] = oct[k].lo.x; p.lo[1] = oct[k].lo.y; p.lo[2] = oct[k].lo.z;
                p.hi[0] = oct[k].hi.x; p.hi[1] = oct[k].hi.y; p.hi[2] = oct[k].hi.z;
                p.src_off = o.off; p.src_cnt = o.cnt; p.flag_off = flag_total; p.out_off = 0;
                flag_total += o.cnt;
                for (uint32_t b = 0; b < o.cnt; b += 256u) { blk_pair.push_back(static_cast<uint32_t>(pairs.size())); blk_first.push_back(b); }
                pairs.push_back(p);
            }
            hs.pool[o.node].kids = kids;
            hs.pool[o.node].faces.clear();
        }
        if (hs.pool.size() > HostScene::kMaxNodes) { hs.overflow = true; return true; }
        const size_t np = pairs.size();
        if (!d_pairs.alloc(np * sizeof(BPair)) || !d_blkp.alloc(blk_pair.size() * 4) || !d_blkf.alloc(blk_first.size() * 4) || !d_flags.alloc(flag_total) ||
            !d_counts.alloc(np * 4)) { if (err) *err = "hipMalloc failed"; return false; }
        BCHK(hipMemcpyAsync(d_pairs.p, pairs.data(), np * sizeof(BPair), hipMemcpyHostToDevice, st));
        BCHK(hipMemcpyAsync(d_blkp.p, blk_pair.data(), blk_pair.size() * 4, hipMemcpyHostToDevice, st));
        BCHK(hipMemcpyAsync(d_blkf.p, blk_first.data(), blk_first.size() * 4, hipMemcpyHostToDevice, st));
        BCHK(hipMemsetAsync(d_counts.p, 0, np * 4, st));
        hipLaunchKernelGGL(k_classify, dim3(static_cast<uint32_t>(blk_pair.size())), dim3(256), 0, st, static_cast<const BPair *>(d_pairs.p),
                           static_cast<const uint32_t *>(d_blkp.p), static_cast<const uint32_t *>(d_blkf.p), static_cast<const uint32_t *>(d_cur.p),
                           static_cast<const float *>(d_tv.p), static_cast<uint8_t *>(d_flags.p), static_cast<uint32_t *>(d_counts.p));
        std::vector<uint32_t> counts(np);
        BCHK(hipMemcpyAsync(counts.data(), d_counts.p, np * 4, hipMemcpyDeviceToHost, st));
        BCHK(hipStreamSynchronize(st));
        // ---- statuses and output offsets, decided as the recursion of the reference decides them (boxTree.cpp:131-146)
        uint32_t out_total = 0;
        for (size_t i = 0; i < np; ++i) { pairs[i].out_off = out_total; out_total += counts[i]; hs.total_refs += counts[i]; }
        if (hs.total_refs > HostScene::kMaxRefs) { hs.overflow = true; return true; }
        if (!d_next.alloc(static_cast<size_t>(out_total) * 4)) { if (err) *err = "hipMalloc failed"; return false; }
        BCHK(hipMemcpyAsync(d_pairs.p, pairs.data(), np * sizeof(BPair), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_scatter, dim3(static_cast<uint32_t>(np)), dim3(256), 0, st, static_cast<const BPair *>(d_pairs.p), static_cast<const uint32_t *>(d_cur.p),
                           static_cast<const uint8_t *>(d_flags.p), static_cast<uint32_t *>(d_next.p));
        std::vector<uint32_t> lists(out_total);
        BCHK(hipMemcpyAsync(lists.data(), d_next.p, static_cast<size_t>(out_total) * 4, hipMemcpyDeviceToHost, st));
        BCHK(hipStreamSynchronize(st));
        std::vector<Open> next;
        size_t pi = 0;
        for (const Open &o : open)
            for (int k = 0; k < 8; ++k, ++pi) {
                const int ci = hs.pool[o.node].kids[k];
                const uint32_t nf = counts[pi];
                OctNode &c = hs.pool[ci];
                if (nf == 0) c.empty = true;
                if (nf < static_cast<uint32_t>(cap) || o.depth <= 0) c.leaf = true;
                const bool split = nf > static_cast<uint32_t>(cap) && o.depth > 0;
                if (split) next.push_back({ci, o.depth - 1, pairs[pi].out_off, nf});
                else c.faces.assign(lists.begin() + pairs[pi].out_off, lists.begin() + pairs[pi].out_off + nf);    // leaves, and the lost "== capacity" nodes
            }
        std::swap(d_cur.p, d_next.p);
        open.swap(next);
    }
    return true;
}

}  // namespace rtamd
