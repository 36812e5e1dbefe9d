// rt_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4) for the primary/shadow ray-trace path.
//
// Execution model (designed for 64-wide wavefronts + the scalar unit, not a port of a lane-per-ray CPU loop):
//   * PACKET TRAVERSAL.  One wavefront owns 64 coherent rays (an 8x8 pixel tile, 64 compacted bounce rays, or the
//     N area-light samples of one hit point).  The octree walk is WAVE-UNIFORM: node ids and per-node lane masks
//     live in a small per-wave LDS stack, node boxes and leaf triangle records are fetched with scalar loads
//     (s_load_dwordx8/x16 through the scalar cache -- no VGPR gather, no per-lane addressing), and each lane only
//     runs the arithmetic for its own ray under an exec mask.  `__ballot` decides whether any lane still needs a
//     child / a leaf.  This reproduces BoxTree::intersect's result exactly (src/boxTree.cpp:150-173: a lane's
//     candidate set = faces of every non-empty leaf whose whole ancestor chain its ray's box tests accept) while
//     touching each node/triangle once per wave instead of once per ray.
//   * PERSISTENT WAVES.  Every kernel is launched with a fixed grid sized to its residency (CUs x blocks/CU from
//     the occupancy query); waves pull CHUNKS of units from a sharded device-side queue (8 XCD-keyed heads on separate
//     cache lines, interleaved chunk ownership, next index prefetched, stealing when drained).  Measured on the way here: a
//     single atomic head caps a kernel at ~88 units/us; static striding leaves waves idle 58 % of the time.  Work sizes that depend on earlier kernels (lit
//     hits, bounce rays) are read from device memory, so a whole frame is a fixed launch sequence with no host
//     round trip and can be captured in a hipGraph.
//   * Shadow samples are a SEPARATE kernel (k_shadow) so rocprof attributes traversal time to
//     closest-hit (k_trace) vs area-light shadow rays (k_shadow) separately.
//   * Ray compaction between bounces uses wave-wide ballot + mbcnt prefix and one atomicAdd per wave.
//
// Numerics: compiled with -ffp-contract=off and correctly rounded fp32 divide/sqrt; every expression keeps the
// reference's operation order (Eigen 3.3.7: a.dot(b) = a0*b0 + (a1*b1 + a2*b2)), std::min/std::max are spelt as
// the ternaries libstdc++ uses, so integer results (face ids, 8-bit pixels) are bit-exact and float RGB differs
// from the CPU path nowhere: powf is glibc's published algorithm evaluated bit for bit (pow_shininess).
#include <hip/hip_runtime.h>

#include "rt_device.hpp"

namespace rtamd {

#define RT_WAVES 4          // waves per workgroup (256 threads)
#define RT_STACK 128        // >= 7*16 + 8: DFS stack bound for MAX_DEPTH 15 octrees (checked on upload)

// ---- libstdc++ std::min / std::max (NaN behaviour is part of the reference's slab test) -----------------------
__device__ __forceinline__ float smin(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float smax(float a, float b) { return (a < b) ? b : a; }
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return ax * bx + (ay * by + az * bz);
}
__device__ __forceinline__ void normalize3(float &x, float &y, float &z) {
    const float q = x * x + (y * y + z * z);
    if (q > 0.0f) {
        const float s = sqrtf(q);
        x = x / s; y = y / s; z = z / s;
    }
}

// Eigen's normalized() (Dot.h:124-134: v / sqrt(squaredNorm) when that is > 0) with the work the three IEEE divisions share done once.
// hipcc expands a correctly rounded x / s into  d = div_scale(s), n = div_scale(x), r = rcp(d), e = fma(-d, r, 1), r = fma(e, r, r),
// q = n r, q = fma(fma(-d, q, n), r, q), div_fmas(fma(-d, q, n), r, q), div_fixup  -- 11 instructions, of which the reciprocal and its
// refinement depend on the divisor alone.  div_scale leaves both operands as they are and div_fmas / div_fixup pass the quotient through
// whenever divisor and quotient are far from the ends of the exponent range, so for 2^-40 <= s <= 2^40 and 2^-60 <= |x| (<= s) the plain
// FMA chain below IS that sequence, bit for bit, and the three quotients share r: 18 instead of 33 instructions.  Anything else -- a zero
// or tiny component (the sign of a zero quotient comes from div_fixup), a huge or non-finite norm -- takes the three full divisions; the
// test is wave-uniform, so the common case has no divergent branch.  (k_shade runs two of these per sample.)
__device__ __forceinline__ void normalize3_shared(float &x, float &y, float &z) {
    const float q = x * x + (y * y + z * z);
    const bool ok = (q >= 0x1p-80f) && (q <= 0x1p80f) && (fminf(fminf(fabsf(x), fabsf(y)), fabsf(z)) >= 0x1p-60f);     // NaN: false
    if (__ballot(!ok) == 0ull) {
        // the correctly rounded square root as hipcc expands sqrtf -- v_sqrt_f32 (1 ulp), then the neighbour whose residual says so -- without the
        // scaling of denormal arguments and the 0 / inf pass-through, neither of which this range can need: 9 instead of 16 instructions
        float s = __builtin_amdgcn_sqrtf(q);
        const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
        const float rm = __builtin_fmaf(-sm, s, q), rp = __builtin_fmaf(-sp, s, q);
        s = rp > 0.0f ? sp : (rm <= 0.0f ? sm : s);
        float r = __builtin_amdgcn_rcpf(s);
        const float e = __builtin_fmaf(-s, r, 1.0f);
        r = __builtin_fmaf(e, r, r);
        float qx = x * r, qy = y * r, qz = z * r;
        qx = __builtin_fmaf(__builtin_fmaf(-s, qx, x), r, qx);
        qy = __builtin_fmaf(__builtin_fmaf(-s, qy, y), r, qy);
        qz = __builtin_fmaf(__builtin_fmaf(-s, qz, z), r, qz);
        x = __builtin_fmaf(__builtin_fmaf(-s, qx, x), r, qx);
        y = __builtin_fmaf(__builtin_fmaf(-s, qy, y), r, qy);
        z = __builtin_fmaf(__builtin_fmaf(-s, qz, z), r, qz);
    } else if (q > 0.0f) {
        const float s = sqrtf(q);
        x = x / s; y = y / s; z = z / s;
    }
}

__device__ __forceinline__ uint32_t uniform_u32(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
    const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
    return (static_cast<unsigned long long>(hi) << 32) | lo;
}
__device__ __forceinline__ uint32_t lanes_below(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
}

// BoundingBox::boxIntersect (src/boundingBox.cpp:48-83) with dir = dest - origin already formed by the caller.
__device__ __forceinline__ bool box_hit(const float *__restrict__ b, float ox, float oy, float oz, float dx, float dy, float dz) {
    const float tx0 = (b[0] - ox) / dx, tx1 = (b[3] - ox) / dx;
    const float ty0 = (b[1] - oy) / dy, ty1 = (b[4] - oy) / dy;
    const float tz0 = (b[2] - oz) / dz, tz1 = (b[5] - oz) / dz;
    const float tin = smax(smax(smin(tx0, tx1), smin(ty0, ty1)), smin(tz0, tz1));
    const float tout = smin(smin(smax(tx0, tx1), smax(ty0, ty1)), smax(tz0, tz1));
    return !((tin > tout) || (tout < 0));
}

// The node and leaf-triangle arrays reach the kernels as separate `const T *__restrict__` parameters (noalias +
// readonly): only then can the compiler prove that no store in the kernel clobbers them and select the
// wave-uniform loads as s_load_dwordx8/x16 (scalar cache, SGPR destination) instead of 64-lane global_loads.

// Dynamic work distribution without a single hot atomic (one queue head saturates at ~88 dequeues/us on MI355X;
// plain static striding left waves resident only ~42 % of the kernel on the dodge scene).
//   * units are handed out in CHUNKS of `chunk` consecutive units (chunk ~ units / (6 x waves), computed in-kernel
//     because the unit count itself is produced on the device), so a wave performs only ~6 atomics per launch;
//   * chunk c belongs to queue head c % 8; a workgroup uses head blockIdx % 8 -- blocks b and b+8 share an XCD, so
//     each head is hit from one XCD's L2 -- and the interleaving keeps all 8 heads equally loaded, so stealing
//     (next head, checked with a plain load first) only happens in the last few chunks;
//   * the next chunk index is fetched BEFORE the current chunk is processed: the atomic's latency hides behind work.
// Approximate-then-verify form of the slab test.  The six quotients are first formed with ONE v_rcp_f32 per axis
// (relative error of x * rcp(d) vs the correctly rounded x / d: <= ~2.4e-7).  If the resulting tin/tout are separated --
// and tout is away from zero -- by more than 1e-6 * sum|t_i| (4x that bound; a perturbation of max/min compositions never
// exceeds the largest perturbation of their inputs), the reference's decision `!((tin > tout) || (tout < 0))` is already
// determined and no IEEE division is needed.  Anything closer, and anything non-finite (zero direction components make
// inf/NaN terms whose handling depends on std::min/std::max argument order), takes the exact path.  ~30 VALU instead of
// ~100 per box test; the decision is bit-identical by construction (checked by the counter-equality tests).
__device__ __forceinline__ bool box_hit_verified(const float *__restrict__ b, float ox, float oy, float oz, float dx, float dy, float dz,
                                                 float rx, float ry, float rz) {
    const float ax0 = (b[0] - ox) * rx, ax1 = (b[3] - ox) * rx;
    const float ay0 = (b[1] - oy) * ry, ay1 = (b[4] - oy) * ry;
    const float az0 = (b[2] - oz) * rz, az1 = (b[5] - oz) * rz;
    const float tin = fmaxf(fmaxf(fminf(ax0, ax1), fminf(ay0, ay1)), fminf(az0, az1));
    const float tout = fminf(fminf(fmaxf(ax0, ax1), fmaxf(ay0, ay1)), fmaxf(az0, az1));
    const float mag = ((fabsf(ax0) + fabsf(ax1)) + (fabsf(ay0) + fabsf(ay1))) + (fabsf(az0) + fabsf(az1));
    const float e = 1e-6f * mag;
    const bool finite = mag < 3.0e38f;                       // false for inf and NaN
    const bool sure_hit = finite && (tout - tin > e) && (tout > e);
    const bool sure_miss = finite && ((tin - tout > e) || (tout < -e));
    if (sure_hit) return true;
    if (sure_miss) return false;
    return box_hit(b, ox, oy, oz, dx, dy, dz);
}

struct ShardedQueue {
    uint32_t *ctr;
    uint32_t total, chunk, nchunks, shard, tries, fetched, cur, cur_end;
    uint32_t local;          // 0: strided chunks, interleaved heads (balance first); else: chunks of `local` CONSECUTIVE units and head x
                             // owns the x-th eighth of the unit range (locality first: scenes that do not fit the 4 MB L2 of an XCD)
    int lane;
    __device__ __forceinline__ uint32_t chunks_of(uint32_t s) const {         // index range of head s (some ids may fall past nchunks)
        if (local) { const uint32_t per = (nchunks + RT_QUEUE_SHARDS - 1u) / RT_QUEUE_SHARDS; return per; }
        const uint32_t groups = (nchunks + 31u) / 32u;
        return (groups / RT_QUEUE_SHARDS + ((groups % RT_QUEUE_SHARDS) > s ? 1u : 0u)) * 32u;
    }
    __device__ __forceinline__ uint32_t grab() {
        uint32_t v = 0;
        if (lane == 0) v = atomicAdd(&ctr[shard * 16u], 1u);
        return v;
    }
    // static variant (no atomics at all): wave w owns units w, w + W, w + 2W, ...  Best when units are cheap and
    // uniform (primary tiles: measured 0.25 ms static vs 0.39 ms dynamic on the cube frame).
    __device__ __forceinline__ void init_static(uint32_t total_units, uint32_t total_waves, uint32_t wave_id, int ln) {
        local = 0u;
        ctr = nullptr; total = 0u; lane = ln; chunk = 1u; nchunks = total_waves; shard = 0; tries = 0; fetched = 0;
        cur = wave_id; cur_end = total_units;
    }
    __device__ __forceinline__ void init(uint32_t *heads, uint32_t total_units, uint32_t total_waves, uint32_t home, int ln, uint32_t local_chunk = 0u, uint32_t div = 6u) {
        ctr = heads; total = total_units; lane = ln; local = local_chunk;
        chunk = total_units / (total_waves * div);
        if (chunk < 1u) chunk = 1u;
        if (local) chunk = local;
        nchunks = (total_units + chunk - 1u) / chunk;          // chunk c = { c + j * nchunks }   (local: { c * chunk + j })
        shard = home % RT_QUEUE_SHARDS; tries = 0; cur = 0; cur_end = 0;
        fetched = total ? grab() : 0u;
    }
    __device__ __forceinline__ bool next(uint32_t &unit) {
        // a chunk is NOT a run of consecutive units: chunk c owns units c, c + nchunks, c + 2*nchunks, ...  Expensive
        // units cluster (neighbouring hit points cross the same 979-triangle leaves); consecutive membership made
        // single chunks 5x heavier than average and doubled the kernel's tail.
        if (cur < cur_end) { unit = cur; cur += local ? 1u : nchunks; return true; }
        if (total == 0u) return false;
        for (;;) {
            const uint32_t idx = uniform_u32(fetched);
            if (idx < chunks_of(shard)) {
                // head `shard` owns GROUPS of 32 consecutive chunk ids, groups dealt round-robin over the 8 heads: inside a
                // group the waves of one XCD work on neighbouring hit points (same leaves -> that XCD's 4 MB L2) while the
                // fine interleave keeps the heads equally loaded (fully contiguous ownership measured +17 % on dodge,
                // chunk-granular interleave +5 % on the 1M-triangle scene)
                const uint32_t c = local ? shard * chunks_of(shard) + idx : ((idx / 32u) * RT_QUEUE_SHARDS + shard) * 32u + (idx % 32u);
                fetched = grab();                     // prefetch the following chunk index
                if (c >= nchunks) { cur = 0; cur_end = 0; continue; }     // id past the end of the last partial group
                if (local) {
                    unit = c * chunk;
                    cur = unit + 1u;
                    cur_end = unit + chunk < total ? unit + chunk : total;
                    return true;
                }
                unit = c;
                cur = c + nchunks;
                cur_end = total;
                return true;
            }
            for (;;) {
                if (++tries >= RT_QUEUE_SHARDS) return false;
                shard = (shard + 1u) % RT_QUEUE_SHARDS;
                const uint32_t seen = uniform_u32(__hip_atomic_load(&ctr[shard * 16u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if (seen < chunks_of(shard)) break;
            }
            fetched = grab();
        }
    }
};

struct WaveStack {
    uint32_t *node;
    unsigned long long *mask;
    uint4 *stage;            // per-wave LDS staging buffer: RT_STAGE_TRIS leaf-triangle records (80 B each)
};
#ifdef RT_PROFILE
__device__ uint32_t g_wave_steps[4];     // diagnostic: written by the wave that holds the slowest tile (racy by design)
#define RT_TILE_COUNT(stk, lane, idx, val) do { if ((lane) == 0) (stk).node[RT_STACK - 4 + (idx)] += (val); } while (0)
#else
#define RT_TILE_COUNT(stk, lane, idx, val) do { } while (0)
#endif
#define RT_STAGE_TRIS 64
#ifndef RT_LDS_NODES
#define RT_LDS_NODES 256        // DNodes of the top of the tree kept in LDS by k_shadow (16 KB per block)
#endif
#define RT_SCALAR_LEAF_MAX 16   // leaves up to this size are walked with scalar loads straight from the scalar cache

// Wave-uniform octree walk for 64 rays.
//   ANY   = false: closest hit (flyscene.cpp:675-683): min t over candidates with t > 1e-5, ties to the lowest
//                  face id.  No t-ordered pruning: the reference tests every triangle of every intersected leaf,
//                  and because its tree loses/misfiles triangles a pruned walk would not be equivalent.
//   ANY   = true : lightStrikes (flyscene.cpp:912-954): visible iff min t >= 0.98, i.e. occluded iff SOME candidate
//                  (not illum 9) has 1e-5 < t < 0.98 -- so a lane may stop at its first occluder (exact).
//   COUNT = true : no early-out; counts boxIntersect calls / leaf face references with the reference's semantics
//                  (every pushed node is box-tested again when popped, boxTree.cpp:158,164).
// ---- explicit scalar load of a wave-uniform TriRec ----------------------------------------------------------------
// hipcc only selects s_load for a uniform global load when it can prove the memory unclobbered, which depends on code
// shape (it fell back to 64-lane global_loads in the flat kernels, and split the record into three dependent waits
// elsewhere).  The records are read-only for the kernel's lifetime, so the load is spelt out: s_load_dwordx16 +
// s_load_dwordx4 + ONE s_waitcnt inside a single asm statement (no output is visible before its wait; an asynchronous
// issue/wait split was tried and is unsafe: the compiler may copy the destination SGPRs before the data lands).
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ TriRec tri_load_uniform(const TriRec *p) {
    u32x16 lo;
    u32x4 hi;
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx4 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(lo), "=&s"(hi) : "s"(p) : "memory");
    TriRec t;
    t.ax = __uint_as_float(lo[0]); t.ay = __uint_as_float(lo[1]); t.az = __uint_as_float(lo[2]);
    t.e0x = __uint_as_float(lo[3]); t.e0y = __uint_as_float(lo[4]); t.e0z = __uint_as_float(lo[5]);
    t.e1x = __uint_as_float(lo[6]); t.e1y = __uint_as_float(lo[7]); t.e1z = __uint_as_float(lo[8]);
    t.nx = __uint_as_float(lo[9]); t.ny = __uint_as_float(lo[10]); t.nz = __uint_as_float(lo[11]);
    t.nA = __uint_as_float(lo[12]); t.d00 = __uint_as_float(lo[13]); t.d01 = __uint_as_float(lo[14]); t.d11 = __uint_as_float(lo[15]);
    t.inv_denom = __uint_as_float(hi[0]); t.face = hi[1]; t.flags = hi[2]; t.pad = 0u;
    return t;
}

// Plane culling for shadow units.  All 64 segments of a k_shadow unit end at the same point h (the shaded hit) and start
// at light samples inside the box [slo, shi].  Flyscene::rayTriangleIntersection (flyscene.cpp:787-819) accepts a triangle
// only for 0.00001 < t, and lightStrikes (flyscene.cpp:874-899) only counts t < 0.98, with
//      t = num / dn,   num = n.A - s.n,   dn = (h - s).n          (float, Eigen order)
// so a triangle is irrelevant for EVERY segment of the unit when interval arithmetic over the sample box proves
//   (A) num and dn have opposite signs (t <= 0: the plane lies behind the light sample), or
//   (B) |num| >= 0.981 |dn| (|t| >= 0.98: the plane is not crossed before the hit point -- this covers the face h lies on).
// Only signs and magnitudes of the reference's own two dot products are used, each bounded with a margin M = 2e-5 x the
// operand magnitudes (>= 20x the worst-case float rounding of the reference's evaluation and of the interval arithmetic),
// so the decision never depends on how accurate t is; a NaN anywhere makes every comparison false (= keep).
struct SegPacket {
    bool on;
    bool prepared;                 // the sample box is the one the lanes' LanePlane.sn_lo/sn_hi were prepared for
    float hx, hy, hz;              // common end point
    float slx, sly, slz, shx, shy, shz;   // box of the start points (exact extreme samples)
    float m0;                      // 2e-5 * (|slo|_1 + |shi|_1 + |h|_1)
};
__device__ __forceinline__ SegPacket seg_off() { return SegPacket{false, false, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; }

__device__ __forceinline__ bool plane_rules_out(const SegPacket &g, const float nx, const float ny, const float nz, const float nA) {
    const float ax = nx * g.slx, bx = nx * g.shx, ay = ny * g.sly, by = ny * g.shy, az = nz * g.slz, bz = nz * g.shz;
    const float sn_lo = fminf(ax, bx) + (fminf(ay, by) + fminf(az, bz));
    const float sn_hi = fmaxf(ax, bx) + (fmaxf(ay, by) + fmaxf(az, bz));
    const float num_lo = nA - sn_hi, num_hi = nA - sn_lo;
    const float cx = nx * (g.hx - g.shx), dx = nx * (g.hx - g.slx), cy = ny * (g.hy - g.shy), dy = ny * (g.hy - g.sly),
                cz = nz * (g.hz - g.shz), dz = nz * (g.hz - g.slz);
    const float dn_lo = fminf(cx, dx) + (fminf(cy, dy) + fminf(cz, dz));
    const float dn_hi = fmaxf(cx, dx) + (fmaxf(cy, dy) + fmaxf(cz, dz));
    const float M = g.m0 + 2e-5f * fabsf(nA);
    const bool sane = (fabsf(nx) + fabsf(ny) + fabsf(nz) <= 4.0f) && (fabsf(nA) <= 1e30f);
    const bool opposite = (num_lo > M && dn_hi < -M) || (num_hi < -M && dn_lo > M);
    const float min_abs_num = fmaxf(num_lo, -num_hi);              // <= 0 when the interval straddles zero
    const float max_abs_dn = fmaxf(fabsf(dn_lo), fabsf(dn_hi));
    const bool beyond = (min_abs_num - M) >= 0.981f * (max_abs_dn + M);
    return sane && (num_lo <= num_hi) && (dn_lo <= dn_hi) && (opposite || beyond);
}

// The same decision when the sample box is the one the lane prepared its plane for before the unit loop (scene light 0): the
// interval of s.n over the box (sn_lo, sn_hi) is a per-triangle constant, and (h - s).n = h.n - s.n turns the second interval
// into one dot product.  ~20 VALU instructions instead of ~45 per unit.
struct LanePlane { float nx, ny, nz, nA, sn_lo, sn_hi; };
__device__ __forceinline__ void plane_prepare(LanePlane &pl, const float slx, const float sly, const float slz, const float shx, const float shy, const float shz) {
    const float ax = pl.nx * slx, bx = pl.nx * shx, ay = pl.ny * sly, by = pl.ny * shy, az = pl.nz * slz, bz = pl.nz * shz;
    pl.sn_lo = fminf(ax, bx) + (fminf(ay, by) + fminf(az, bz));
    pl.sn_hi = fmaxf(ax, bx) + (fmaxf(ay, by) + fmaxf(az, bz));
}
__device__ __forceinline__ bool plane_rules_out_prepared(const SegPacket &g, const LanePlane &pl) {
    const float num_lo = pl.nA - pl.sn_hi, num_hi = pl.nA - pl.sn_lo;
    const float hn = pl.nx * g.hx + (pl.ny * g.hy + pl.nz * g.hz);
    const float dn_lo = hn - pl.sn_hi, dn_hi = hn - pl.sn_lo;
    const float M = g.m0 + 2e-5f * fabsf(pl.nA);
    const bool sane = (fabsf(pl.nx) + fabsf(pl.ny) + fabsf(pl.nz) <= 4.0f) && (fabsf(pl.nA) <= 1e30f);
    const bool opposite = (num_lo > M && dn_hi < -M) || (num_hi < -M && dn_lo > M);
    const float min_abs_num = fmaxf(num_lo, -num_hi);
    const float max_abs_dn = fmaxf(fabsf(dn_lo), fabsf(dn_hi));
    const bool beyond = (min_abs_num - M) >= 0.981f * (max_abs_dn + M);
    return sane && (num_lo <= num_hi) && (dn_lo <= dn_hi) && (opposite || beyond);
}

// How a packet walk starts and when it gives work away.
#ifdef RT_PROFILE
struct PhaseClock;
#endif
struct WalkCtl {
    bool resume;                      // leaf task: process chunks [c_begin, c_end) of leaf `start_node` for `start_mask`, nothing else
    uint32_t start_node;
    unsigned long long start_mask;
    uint32_t c_begin, c_end;
    uint32_t budget;                  // 0 = process every leaf inline; otherwise leaves whose estimated cost exceeds it become tasks
    uint32_t unit;                    // unit id stored in the emitted tasks
    ContTask *tasks;                  // output queue (capacity task_cap) and its counter
    uint32_t *task_count;
    uint32_t target;                  // estimated cost of one task piece
    uint32_t task_cap;
    SegPacket seg;                    // shadow units: plane culling (off for every other kind of packet)
    unsigned long long skip;          // flat scenes: triangles of the root leaf that no ray of the unit can hit (bit = position in the leaf)
    const float4 *cone;               // per-wave LDS record of the packet's cone (common origin, box of the ray targets): the lane = triangle test of the leaves; nullptr = none
    bool cone_box;                    // counted hits lie before the targets (t < 0.98: light-centre segments): the AABB of hull(origin, targets) bounds them too
#ifdef RT_PROFILE
    PhaseClock *pc;
#endif
#ifdef RT_UNIT_HIST
    uint32_t *dbg;                    // per-wave LDS: [0] groups popped, [1] leaf visits, [2] task emissions (returning atomics), [3] ray-mode triangles, [4] tri-mode (ray, chunk) steps, [5] chunk tests
#endif
};
#ifdef RT_UNIT_HIST
#define RT_DBG(wc, lane, i, v) do { if ((lane) == 0 && (wc).dbg) (wc).dbg[i] += (v); } while (0)
#else
#define RT_DBG(wc, lane, i, v) do { } while (0)
#endif
// (defined with the shaft code below)
__device__ __forceinline__ bool tri_outside_cone(const float4 *rec, const TriRec &tr, const float m, const bool use_box);
#ifndef RT_TRI_SHAFT_MIN
#define RT_TRI_SHAFT_MIN 8            // live rays on a chunk from which the per-triangle test (~3 triangle steps) is run
#endif
#ifndef RT_RAYMODE_EXTRA
#define RT_RAYMODE_EXTRA 8u          // scalar loads of the survivors' records
#endif
__device__ __forceinline__ WalkCtl walk_plain() {
    WalkCtl w{};
    w.target = 1u; w.seg = seg_off();
    return w;
}

__device__ __forceinline__ TriRec tri_from_regs(const u32x16 &lo, const u32x4 &hi) {
    TriRec t;
    t.ax = __uint_as_float(lo[0]); t.ay = __uint_as_float(lo[1]); t.az = __uint_as_float(lo[2]);
    t.e0x = __uint_as_float(lo[3]); t.e0y = __uint_as_float(lo[4]); t.e0z = __uint_as_float(lo[5]);
    t.e1x = __uint_as_float(lo[6]); t.e1y = __uint_as_float(lo[7]); t.e1z = __uint_as_float(lo[8]);
    t.nx = __uint_as_float(lo[9]); t.ny = __uint_as_float(lo[10]); t.nz = __uint_as_float(lo[11]);
    t.nA = __uint_as_float(lo[12]); t.d00 = __uint_as_float(lo[13]); t.d01 = __uint_as_float(lo[14]); t.d11 = __uint_as_float(lo[15]);
    t.inv_denom = __uint_as_float(hi[0]); t.face = hi[1]; t.flags = hi[2]; t.pad = 0u;
    return t;
}
// two consecutive records (p[0], p[1]) with ONE wait: the caller tests both in straight-line code, which puts two
// independent division/dot-product chains in flight per wave
__device__ __forceinline__ void tri_load_uniform2(const TriRec *p, TriRec &a, TriRec &b) {
    u32x16 lo0, lo1;
    u32x4 hi0, hi1;
    asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx4 %1, %4, 0x40\n\ts_load_dwordx16 %2, %4, 0x50\n\ts_load_dwordx4 %3, %4, 0x90\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(lo0), "=&s"(hi0), "=&s"(lo1), "=&s"(hi1) : "s"(p) : "memory");
    a = tri_from_regs(lo0, hi0);
    b = tri_from_regs(lo1, hi1);
}

// same, for two records that are not neighbours
__device__ __forceinline__ const TriRec *uniform_ptr(const TriRec *p) {
    // (folds away when the compiler already knows the value is wave-uniform; under SGPR pressure it may have parked the pointer in a VGPR)
    return reinterpret_cast<const TriRec *>(uniform_u64(reinterpret_cast<unsigned long long>(p)));
}
__device__ __forceinline__ void tri_load_uniform_pair(const TriRec *pa_, const TriRec *pb_, TriRec &a, TriRec &b) {
    const TriRec *pa = uniform_ptr(pa_), *pb = uniform_ptr(pb_);
    u32x16 lo0, lo1;
    u32x4 hi0, hi1;
    asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx4 %1, %4, 0x40\n\ts_load_dwordx16 %2, %5, 0x0\n\ts_load_dwordx4 %3, %5, 0x40\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(lo0), "=&s"(hi0), "=&s"(lo1), "=&s"(hi1) : "s"(pa), "s"(pb) : "memory");
    a = tri_from_regs(lo0, hi0);
    b = tri_from_regs(lo1, hi1);
}

__device__ __forceinline__ float lane_f(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

// Per-leaf choice between the two lane mappings (wave-uniform): issue-cost estimates in VALU instructions.
#ifdef RT_PROFILE
// profiling build: wave-level step counters, flushed with one atomic per leaf/inner visit (slow, diagnostic only)
__device__ unsigned long long *g_prof = nullptr;
__device__ uint32_t g_prof_base = 0u;       // step counters of the launch in flight go to prof[g_prof_base + idx]: 0 = trace kernels, RT_WORK_SHADOW = shadow kernels
#ifdef RT_PROFILE_STEPS
#define RT_PROF_ADD(lane, idx, val) do { const unsigned long long pv_ = static_cast<unsigned long long>(val); if ((lane) == 0 && g_prof) atomicAdd(&g_prof[g_prof_base + (idx)], pv_); } while (0)
#else
#define RT_PROF_ADD(lane, idx, val) do { } while (0)   // RT_PROFILE alone: only the per-unit cycle histogram (undistorted)
#endif
#elif defined(RT_UNIT_HIST)
// RT_UNIT_HIST build: the step counters of the unit in flight, per wave in LDS (plain adds by lane 0); k_shadow_shaft copies them into its unit record.
// slots: 0 groups (88) | 1 children surviving the shaft test (90) | 2 children hit by some ray (91) | 3 chunk batches (92) | 4 chunks with work (95) |
//        5 per-triangle shaft tests (70) | 6 ray-mode triangle steps (0) | 7 (ray, chunk) triangle steps (2)
__shared__ uint32_t g_uh[RT_WAVES * 8];
#define RT_UH_SLOT(idx) ((idx) == 88 ? 0 : (idx) == 90 ? 1 : (idx) == 91 ? 2 : (idx) == 92 ? 3 : (idx) == 95 ? 4 : (idx) == 70 ? 5 : (idx) == 0 ? 6 : (idx) == 2 ? 7 : -1)
#define RT_PROF_ADD(lane, idx, val) do { if (RT_UH_SLOT(idx) >= 0 && (lane) == 0) g_uh[(threadIdx.x >> 6) * 8 + (RT_UH_SLOT(idx) >= 0 ? RT_UH_SLOT(idx) : 0)] += static_cast<uint32_t>(val); } while (0)
#else
#define RT_PROF_ADD(lane, idx, val) do { } while (0)
#endif
#ifdef RT_PROFILE
// per-wave phase clock (diagnostic build): cycles spent in each phase of a unit, accumulated in registers and flushed with one
// atomic per phase at kernel end -> prof[70 + phase] (undistorted by per-step atomics; build WITHOUT RT_PROFILE_STEPS)
struct PhaseClock {
    long long last; int cur; long long acc[8];
    __device__ __forceinline__ void start() { last = clock64(); cur = 0; for (int i = 0; i < 8; ++i) acc[i] = 0; }
    __device__ __forceinline__ void to(int p) { const long long t = clock64(); acc[cur] += t - last; last = t; cur = p; }
    __device__ __forceinline__ void flush(int lane, int base) { to(0); if (lane == 0 && g_prof) for (int i = 0; i < 8; ++i) atomicAdd(&g_prof[base + i], static_cast<unsigned long long>(acc[i])); }
};
#ifdef RT_NO_PHASE_CLOCK
#define RT_PH(wc, p) do { } while (0)              // (make prof PROF_EXTRA=-DRT_NO_PHASE_CLOCK: unit-duration histograms without the clock reads)
#else
#define RT_PH(wc, p) do { if ((wc).pc) (wc).pc->to(p); } while (0)
#endif
#else
#define RT_PH(wc, p) do { } while (0)
#endif
// prof[0] ray-mode triangle steps   prof[1] ray-mode useful lane tests
// prof[2] tri-mode (ray,chunk) steps prof[3] tri-mode useful lane tests
// prof[4] box-test steps            prof[5] box-test useful lanes
// prof[6] leaves in ray mode        prof[7] leaves in tri mode   prof[8] sum of live rays at tri-mode leaves

#define RT_COST_RAY_MODE 45u      // per triangle, lanes = rays (all 64 lanes step through every triangle)
#define RT_COST_TRI_MODE 58u      // per (active ray, 64-triangle chunk), lanes = triangles
#define RT_COST_CHUNK_TEST 30u    // per chunk: conservative bound test for all 64 rays at once

// One leaf of a packet walk: every triangle of the leaf is a candidate for the rays in `live` (BoxTree::intersect inserts all faces of an
// intersected leaf, boxTree.cpp:158-160).  Shared by the stack walk (packet_walk) and the shaft-culled breadth-first walk of the shadow units.
struct RayLane {
    float ox, oy, oz;        // origin
    float dx, dy, dz;        // triangle-test direction
    float idx, idy, idz;     // v_rcp_f32 of the box-test direction (conservative tests only)
    float slab_pad;
};
template <bool ANY, bool COUNT, bool STAGED = true>
__device__ __forceinline__ void leaf_visit(const DNode &nd, const uint32_t ni, const TriRec *__restrict__ tris, const ChunkBound *__restrict__ chunks,
                                           const uint32_t *__restrict__ leaf_chunk0, const WaveStack stk, const int lane, const WalkCtl &wc,
                                           const RayLane &R, unsigned long long live, bool mine,
                                           float &best_t, int &best_f, bool &occluded, uint32_t &cnt_ref, uint32_t &cnt_sig) {
    const float ox = R.ox, oy = R.oy, oz = R.oz, dx = R.dx, dy = R.dy, dz = R.dz;
    const float idx_ = R.idx, idy_ = R.idy, idz_ = R.idz, slab_pad = R.slab_pad;
    const uint32_t cnt = nd.count_flags & 0x7fffffffu;
    if (COUNT && mine) { cnt_ref += cnt; cnt_sig += ni * 2654435761u; }      // cnt_sig: signature of the ray's set of intersected leaves (rt_tree_probe)
    const TriRec *__restrict__ T = tris + nd.first;
    const uint32_t nchunk = (cnt + 63u) >> 6;
    // lanes=triangles estimate: one bound test per chunk + live rays x the share of chunks a ray cannot skip
    // STAGED = false (kernels without the LDS staging buffer): leaves too large for the scalar-load walk always take lanes = triangles
    const bool tri_mode = (!STAGED && cnt > RT_SCALAR_LEAF_MAX) ||
                          nchunk * RT_COST_CHUNK_TEST + static_cast<uint32_t>(__popcll(live)) * ((nchunk + 2u) / 3u) * RT_COST_TRI_MODE
                          < cnt * RT_COST_RAY_MODE;
    const uint32_t est = tri_mode ? nchunk * RT_COST_CHUNK_TEST + static_cast<uint32_t>(__popcll(live)) * ((nchunk + 2u) / 3u) * RT_COST_TRI_MODE
                                  : cnt * RT_COST_RAY_MODE;
    uint32_t cb = 0u, ce = nchunk;                       // chunk range processed here
    if (wc.resume) {
        cb = wc.c_begin; ce = wc.c_end < nchunk ? wc.c_end : nchunk;
    } else if (wc.budget != 0u && est > wc.budget) {
        // hand the leaf away as ~target-instruction pieces (whole chunks); one atomic reserves the slots
        uint32_t ntask = (est + wc.target - 1u) / wc.target;
        if (ntask > nchunk) ntask = nchunk;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(wc.task_count, ntask);
        base = uniform_u32(base);
        RT_DBG(wc, lane, 2, 1u);
        // The counter only ever grows (the consumer clamps it to the capacity): pieces that fall past the end of the
        // queue are simply processed here.  (Giving a failed reservation back with an atomicSub is unsound: a later
        // reservation can land inside the window and end up beyond the final count.)
        const uint32_t fit = base >= wc.task_cap ? 0u : (ntask < wc.task_cap - base ? ntask : wc.task_cap - base);
        for (uint32_t i = static_cast<uint32_t>(lane); i < fit; i += 64u) {
            ContTask t;
            t.unit = wc.unit; t.node = ni; t.mask = live;
            t.c_begin = static_cast<uint32_t>(static_cast<unsigned long long>(nchunk) * i / ntask);
            t.c_end = static_cast<uint32_t>(static_cast<unsigned long long>(nchunk) * (i + 1u) / ntask);
            t.pad0 = t.pad1 = 0u;
            wc.tasks[base + i] = t;
        }
        if (fit == ntask) return;
        cb = static_cast<uint32_t>(static_cast<unsigned long long>(nchunk) * fit / ntask);       // the rest of the leaf, inline
    }
    RT_PROF_ADD(lane, tri_mode ? 7 : 6, 1);
    RT_DBG(wc, lane, 1, 1u);
    if (!tri_mode) RT_DBG(wc, lane, 3, cnt);
    RT_TILE_COUNT(stk, lane, tri_mode ? 1 : 0, cnt);
    RT_PH(wc, tri_mode ? 3 : (cnt <= RT_SCALAR_LEAF_MAX ? 4 : 5));
    if (tri_mode) {
        RT_PROF_ADD(lane, 8, __popcll(live));
        // ---- lanes = triangles.  Each lane keeps ONE leaf triangle in registers (coalesced 80-B records, next
        // chunk prefetched); the active rays are broadcast one at a time with v_readlane and every lane tests
        // its triangle against that ray.  A ballot reports the hits: exact early-out per ray for shadow rays,
        // and a scalar pick of the (t, face) minimum for closest hit.  No per-triangle memory round trip.
        unsigned long long occ_new = 0ull;
        const ChunkBound *__restrict__ cbounds = chunks + nd.pad[0];      // first chunk bound of the leaf (rt_capi.cpp: DNode.pad[0])
        // per-ray quantities of the conservative chunk test (approximate arithmetic is fine: they only ever SKIP
        // work); computed per leaf visit so that scenes that never take this mode (the cube) pay nothing
        const uint32_t t_end = ce * 64u < cnt ? ce * 64u : cnt;
        TriRec tr = T[cb * 64u + static_cast<uint32_t>(lane) < cnt ? cb * 64u + static_cast<uint32_t>(lane) : 0u];
        for (uint32_t c0 = cb * 64u; c0 < t_end; c0 += 64u) {
            const uint32_t n = cnt - c0 < 64u ? cnt - c0 : 64u;
            const bool has = static_cast<uint32_t>(lane) < n && !(ANY && (tr.flags & 1u));
            // conservative chunk test (lanes = rays): a live ray skips the chunk when no point of its line that a hit could
            // count at lies in the chunk's inflated box (rt_capi.cpp, build_chunk_bounds, has the error analysis)
            unsigned long long todo = live;
            const ChunkBound bd = cbounds[c0 >> 6];
            {
                if (bd.never < 1.5f) {
                    RT_PROF_ADD(lane, 12, 1);
                    RT_DBG(wc, lane, 5, 1u);
                    const float t0x = (bd.lo[0] - slab_pad - ox) * idx_, t1x = (bd.hi[0] + slab_pad - ox) * idx_;
                    const float t0y = (bd.lo[1] - slab_pad - oy) * idy_, t1y = (bd.hi[1] + slab_pad - oy) * idy_;
                    const float t0z = (bd.lo[2] - slab_pad - oz) * idz_, t1z = (bd.hi[2] + slab_pad - oz) * idz_;
                    const float tin = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z));
                    const float tout = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
                    // the line misses the box, or only meets it where no hit can count: behind the origin (t <= 0.00001 is
                    // never accepted), and beyond 0.98 for a shadow ray / beyond the current closest hit otherwise
                    // (t of an accepted hit lies in [tin, tout] up to ~1e-6 relative: its point is inside the padded box)
                    const bool miss = (tin > tout) || (tout < -1e-3f) ||
                                      (ANY ? (tin > 0.981f) : (tin > best_t + 1e-3f * (1.0f + fabsf(best_t))));
                    const unsigned long long culled = __ballot(miss) & live;
                    todo = live & ~culled;
                    RT_PROF_ADD(lane, 14, __popcll(culled));
                }
            }
            // lane = triangle cone test (packets with a common origin: primary tiles, light-centre segments): the triangles of the chunk that
            // ANY ray of the packet can hit (tri_outside_cone); few survivors and many rays -> lanes = rays over the survivors
            bool has_c = has;
            if (!COUNT && wc.cone != nullptr && static_cast<uint32_t>(__popcll(todo)) >= RT_TRI_SHAFT_MIN && bd.never < 1.5f) {
                __builtin_amdgcn_wave_barrier();
                has_c = has && !tri_outside_cone(wc.cone, tr, bd.infl * 1.0625f, wc.cone_box);
                unsigned long long tmask = __ballot(has_c);
                RT_PROF_ADD(lane, 70, 1); RT_PROF_ADD(lane, 71, __popcll(tmask)); RT_PROF_ADD(lane, 72, __popcll(todo)); RT_PROF_ADD(lane, 73, tmask == 0ull ? 1 : 0);
                if (tmask == 0ull) {
                    todo = 0ull;
                } else if (static_cast<uint32_t>(__popcll(tmask)) * (RT_COST_RAY_MODE + RT_RAYMODE_EXTRA) < static_cast<uint32_t>(__popcll(todo)) * RT_COST_TRI_MODE) {
                    const bool my = ((todo >> lane) & 1ull) != 0ull;
                    bool occ_r = false;
                    const TriRec *__restrict__ Tc = T + c0;
                    auto ray_lane = [&](const TriRec &ta) {
                        // Flyscene::rayTriangleIntersection, flyscene.cpp:787-819 (lanes = rays)
                        const float dn = dot3(dx, dy, dz, ta.nx, ta.ny, ta.nz);
                        const float t = (ta.nA - dot3(ox, oy, oz, ta.nx, ta.ny, ta.nz)) / dn;
                        const float v2x = (ox + t * dx) - ta.ax, v2y = (oy + t * dy) - ta.ay, v2z = (oz + t * dz) - ta.az;
                        const float d02 = dot3(ta.e0x, ta.e0y, ta.e0z, v2x, v2y, v2z);
                        const float d12 = dot3(ta.e1x, ta.e1y, ta.e1z, v2x, v2y, v2z);
                        const float u = (ta.d11 * d02 - ta.d01 * d12) * ta.inv_denom;
                        const float v = (ta.d00 * d12 - ta.d01 * d02) * ta.inv_denom;
                        const bool ok = my && !(ANY && (ta.flags & 1u)) && (dn != 0) && (u >= 0) && (v >= 0) && (u + v < 1) && (t > 0.00001f);
                        if (ANY) {
                            occ_r = occ_r || (ok && t < 0.98f);
                        } else {
                            const bool better = ok && (t < best_t || (t == best_t && static_cast<int>(ta.face) < best_f));
                            best_t = better ? t : best_t;
                            best_f = better ? static_cast<int>(ta.face) : best_f;
                        }
                    };
                    while (tmask != 0ull) {
                        const uint32_t j0 = static_cast<uint32_t>(__builtin_ctzll(tmask));
                        tmask &= tmask - 1ull;
                        const bool two = tmask != 0ull;
                        const uint32_t j1 = two ? static_cast<uint32_t>(__builtin_ctzll(tmask)) : j0;
                        if (two) tmask &= tmask - 1ull;
                        RT_PROF_ADD(lane, 0, two ? 2 : 1);
                        TriRec ta, tb;
                        tri_load_uniform_pair(Tc + j0, Tc + j1, ta, tb);
                        ray_lane(ta);
                        ray_lane(tb);                 // (j1 == j0 for an odd tail: the same triangle twice changes nothing)
                        if (ANY && __ballot(my && !occ_r) == 0ull) break;
                    }
                    if (ANY) {
                        const unsigned long long ob = __ballot(occ_r);
                        occ_new |= ob;
                        live &= ~ob;
                    }
                    todo = 0ull;
                }
            }
            // step 3 (lanes = triangles): full test of the surviving rays, two rays per step for ILP (two independent
            // division chains in flight)
            while (todo != 0ull) {
                const int r0 = static_cast<int>(__builtin_ctzll(todo));
                todo &= todo - 1ull;
                const bool two = todo != 0ull;
                const int r1 = two ? static_cast<int>(__builtin_ctzll(todo)) : r0;
                if (two) todo &= todo - 1ull;
                RT_PROF_ADD(lane, 2, two ? 2 : 1); RT_PROF_ADD(lane, 3, (two ? 2 : 1) * __popcll(__ballot(has_c)));
                RT_DBG(wc, lane, 4, two ? 2u : 1u);
                RT_TILE_COUNT(stk, lane, 2, two ? 2 : 1);
                // Flyscene::rayTriangleIntersection, flyscene.cpp:787-819 (same operations as the ray-lane form)
                float tq[2]; bool inq[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int r = q == 0 ? r0 : r1;
                    const float rdx = lane_f(dx, r), rdy = lane_f(dy, r), rdz = lane_f(dz, r);
                    const float rox = lane_f(ox, r), roy = lane_f(oy, r), roz = lane_f(oz, r);
                    const float dn = dot3(rdx, rdy, rdz, tr.nx, tr.ny, tr.nz);
                    const float t = (tr.nA - dot3(rox, roy, roz, tr.nx, tr.ny, tr.nz)) / dn;
                    const float v2x = (rox + t * rdx) - tr.ax, v2y = (roy + t * rdy) - tr.ay, v2z = (roz + t * rdz) - tr.az;
                    const float d02 = dot3(tr.e0x, tr.e0y, tr.e0z, v2x, v2y, v2z);
                    const float d12 = dot3(tr.e1x, tr.e1y, tr.e1z, v2x, v2y, v2z);
                    const float u = (tr.d11 * d02 - tr.d01 * d12) * tr.inv_denom;
                    const float v = (tr.d00 * d12 - tr.d01 * d02) * tr.inv_denom;
                    tq[q] = t;
                    inq[q] = has_c && (dn != 0) && (u >= 0) && (v >= 0) && (u + v < 1) && (t > 0.00001f);
                }
                if (!two) inq[1] = false;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int r = q == 0 ? r0 : r1;
                    if (ANY) {
                        if (__ballot(inq[q] && tq[q] < 0.98f) != 0ull) {
                            occ_new |= 1ull << r;
                            if (!COUNT) live &= ~(1ull << r);
                        }
                    } else {
                        unsigned long long hm = __ballot(inq[q]);
                        if (hm != 0ull) {
                            float bt = lane_f(best_t, r);
                            int bf = __builtin_amdgcn_readlane(best_f, r);
                            do {
                                const int l = static_cast<int>(__builtin_ctzll(hm));
                                hm &= hm - 1ull;
                                const float tl = lane_f(tq[q], l);
                                const int fl = __builtin_amdgcn_readlane(static_cast<int>(tr.face), l);
                                if (tl < bt || (tl == bt && fl < bf)) { bt = tl; bf = fl; }
                            } while (hm != 0ull);
                            if (lane == r) { best_t = bt; best_f = bf; }
                        }
                    }
                }
            }
            if (ANY && !COUNT && live == 0ull) break;
            { const uint32_t nx = c0 + 64u + static_cast<uint32_t>(lane); if (c0 + 64u < t_end) tr = T[nx < cnt ? nx : 0u]; }
        }
        if (ANY) occluded = occluded || (((occ_new >> lane) & 1ull) != 0ull);
    } else {
        // ---- lanes = rays: every lane steps through the leaf's triangles for its own ray.
        // Small leaves: the records are wave-uniform scalar loads (s_load_dwordx16 + x4, scalar cache).
        // Larger leaves: 64-triangle chunks are staged into this wave's LDS buffer with coalesced 16-byte
        // loads (ONE memory round trip per chunk instead of one per triangle), then read back as LDS
        // broadcasts (all lanes the same address), software-pipelined one record ahead.
        auto test_lane = [&](const TriRec &tr) {
            // Flyscene::rayTriangleIntersection, flyscene.cpp:787-819 (straight-line form, see flat_walk)
            const float dn = dot3(dx, dy, dz, tr.nx, tr.ny, tr.nz);
            const float t = (tr.nA - dot3(ox, oy, oz, tr.nx, tr.ny, tr.nz)) / dn;
            const float v2x = (ox + t * dx) - tr.ax, v2y = (oy + t * dy) - tr.ay, v2z = (oz + t * dz) - tr.az;
            const float d02 = dot3(tr.e0x, tr.e0y, tr.e0z, v2x, v2y, v2z);
            const float d12 = dot3(tr.e1x, tr.e1y, tr.e1z, v2x, v2y, v2z);
            const float u = (tr.d11 * d02 - tr.d01 * d12) * tr.inv_denom;
            const float v = (tr.d00 * d12 - tr.d01 * d02) * tr.inv_denom;
            const bool ok = mine && !(ANY && (tr.flags & 1u)) && (dn != 0) && (u >= 0) && (v >= 0) && (u + v < 1) && (t > 0.00001f);
            if (ANY) {
                occluded = occluded || (ok && t < 0.98f);
            } else {
                const bool better = ok && (t < best_t || (t == best_t && static_cast<int>(tr.face) < best_f));
                best_t = better ? t : best_t;
                best_f = better ? static_cast<int>(tr.face) : best_f;
            }
        };
        if (cnt <= RT_SCALAR_LEAF_MAX) {
            uint32_t k = 0;
            for (; k + 1u < cnt; k += 2u) {
                TriRec ta, tb;
                tri_load_uniform2(T + k, ta, tb);
                RT_PROF_ADD(lane, 0, 2); RT_PROF_ADD(lane, 1, 2 * __popcll(__ballot(mine)));
                test_lane(ta);
                test_lane(tb);
            }
            if (k < cnt) {
                RT_PROF_ADD(lane, 0, 1); RT_PROF_ADD(lane, 1, __popcll(__ballot(mine)));
                test_lane(tri_load_uniform(T + k));
            }
        } else if (STAGED) {
            const uint32_t r_end = ce * 64u < cnt ? ce * 64u : cnt;
            for (uint32_t c0 = cb * 64u; c0 < r_end; c0 += RT_STAGE_TRIS) {
                const uint32_t n = cnt - c0 < RT_STAGE_TRIS ? cnt - c0 : RT_STAGE_TRIS;
                const uint4 *__restrict__ src = reinterpret_cast<const uint4 *>(T + c0);
                __builtin_amdgcn_wave_barrier();
                for (uint32_t q = static_cast<uint32_t>(lane); q < n * 5u; q += 64u) stk.stage[q] = src[q];
                __builtin_amdgcn_wave_barrier();
                const TriRec *staged = reinterpret_cast<const TriRec *>(stk.stage);
                uint32_t k = 0;
                for (; k + 1u < n; k += 2u) {       // two records per step: two independent chains in flight
                    const TriRec ta = staged[k], tb = staged[k + 1u];
                    RT_PROF_ADD(lane, 0, 2); RT_PROF_ADD(lane, 1, 2 * __popcll(__ballot(mine)));
                    test_lane(ta);
                    test_lane(tb);
                }
                if (k < n) {
                    RT_PROF_ADD(lane, 0, 1); RT_PROF_ADD(lane, 1, __popcll(__ballot(mine)));
                    test_lane(staged[k]);
                }
                if (ANY && !COUNT) {
                    mine = mine && !occluded;
                    if (__ballot(mine) == 0ull) break;
                }
            }
        }
    }
}

template <bool ANY, bool COUNT, bool STAGED = true>
__device__ __forceinline__ void packet_walk(const DNode *__restrict__ nodes, const TriRec *__restrict__ tris,
                                            const ChunkBound *__restrict__ chunks, const uint32_t *__restrict__ leaf_chunk0,
                                            const float extent, const WaveStack stk, const int lane, const WalkCtl wc, bool in_root,
                                            const float ox, const float oy, const float oz,      // ray origin
                                            const float dx, const float dy, const float dz,      // triangle-test direction
                                            const float bx, const float by, const float bz,      // box-test direction (dest - origin)
                                            const float brx, const float bry, const float brz,   // v_rcp_f32 of it (approximate)
                                            float &best_t, int &best_f, bool &occluded,
                                            uint32_t &cnt_box, uint32_t &cnt_ref, uint32_t &cnt_sig) {
    // The conservative box tests (content boxes of nodes, chunk boxes of big leaves) only ever SKIP work, so approximate
    // arithmetic is fine: they use the v_rcp_f32 reciprocals of the box-test direction that the caller already holds (it is the
    // triangle-test direction up to one rounding).  A zero component gives inf: an origin inside that slab yields (-inf, +inf) = no
    // constraint, one outside yields tin = +inf = miss (both correct for a line parallel to the slab), 0 * inf = NaN keeps the box.
    const float idx_ = brx, idy_ = bry, idz_ = brz;
    const float slab_pad = 4e-4f * (fabsf(ox) + fabsf(oy) + fabsf(oz) + extent);
    int sp = 0;
    {
        unsigned long long m0 = __ballot(in_root);
        if (wc.resume) m0 &= wc.start_mask;
        if (m0 == 0ull) return;
        if (COUNT && in_root && !wc.resume) cnt_box += 1;      // BoxTree::intersect re-tests the root it was just given
        stk.node[0] = wc.resume ? wc.start_node : 0u;
        stk.mask[0] = m0;
        sp = 1;
    }
    while (sp > 0) {
        --sp;
        RT_PH(wc, 1);
        __builtin_amdgcn_wave_barrier();
        const uint32_t ni = uniform_u32(stk.node[sp]);
        const unsigned long long m = uniform_u64(stk.mask[sp]);
        bool mine = ((m >> lane) & 1ull) != 0ull;
        if (ANY && !COUNT) mine = mine && !occluded;
        unsigned long long live = __ballot(mine);          // rays that still need this node
        if (live == 0ull) continue;
        const DNode nd = nodes[ni];
        const uint32_t cnt = nd.count_flags & 0x7fffffffu;
        if (nd.count_flags & RT_NODE_LEAF) {
            leaf_visit<ANY, COUNT, STAGED>(nd, ni, tris, chunks, leaf_chunk0, stk, lane, wc, RayLane{ox, oy, oz, dx, dy, dz, idx_, idy_, idz_, slab_pad}, live, mine,
                                   best_t, best_f, occluded, cnt_ref, cnt_sig);
        } else {
            RT_TILE_COUNT(stk, lane, 3, cnt);
            RT_PH(wc, 2);
            for (uint32_t c = 0; c < cnt; ++c) {
                const uint32_t ci = nd.first + c;
                const DNode ch = nodes[ci];
                bool h = mine;
                if (!COUNT && ch.pad[1] == 0u) {         // pad[1]: a never-cullable chunk below -> the content box only bounds the rest
                    // content test first (it is the cheaper one and rules out most children): no countable point of the line
                    // inside the subtree's content box -> nothing below can be hit, whatever the reference's box test says
                    const float t0x = (ch.clo[0] - slab_pad - ox) * idx_, t1x = (ch.chi[0] + slab_pad - ox) * idx_;
                    const float t0y = (ch.clo[1] - slab_pad - oy) * idy_, t1y = (ch.chi[1] + slab_pad - oy) * idy_;
                    const float t0z = (ch.clo[2] - slab_pad - oz) * idz_, t1z = (ch.chi[2] + slab_pad - oz) * idz_;
                    const float tin = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z));
                    const float tout = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
                    const bool miss = (tin > tout) || (tout < -1e-3f) ||
                                      (ANY ? (tin > 0.981f) : (tin > best_t + 1e-3f * (1.0f + fabsf(best_t))));
                    RT_PROF_ADD(lane, 66, __popcll(__ballot(h && miss)));
                    h = h && !miss;
                    if (__ballot(h) == 0ull) continue;
                }
                h = h && box_hit_verified(ch.bmin, ox, oy, oz, bx, by, bz, brx, bry, brz);   // BoundingBox::boxIntersect, exact
                RT_PROF_ADD(lane, 4, 1); RT_PROF_ADD(lane, 5, __popcll(__ballot(mine)));
                if (COUNT && mine) cnt_box += h ? 2u : 1u;
                const unsigned long long hm = __ballot(h);
                if (hm != 0ull) {
                    stk.node[sp] = ci;           // same value from every lane
                    stk.mask[sp] = hm;
                    ++sp;
                }
            }
        }
    }
    RT_PH(wc, 0);
}

// Flat scenes (the root is itself a small leaf -- cube.obj: 1 node, 12 triangles): no stack, no LDS, no mode choice;
// every ray that passes the root test steps through the same wave-uniform triangle list (scalar loads).
// Lane k of the wave keeps the plane (n, n.A) of root triangle k for the whole kernel (k_shadow): one plane_rules_out per
// unit then tells which of the root's triangles any of the unit's 64 segments can still be blocked by.

template <bool ANY, bool COUNT>
__device__ __forceinline__ void flat_walk(const DNode &root, const TriRec *__restrict__ tris, bool in_root, const SegPacket &seg, const unsigned long long skip, const LanePlane &pl,
                                          const float ox, const float oy, const float oz,
                                          const float dx, const float dy, const float dz,
                                          float &best_t, int &best_f, bool &occluded, uint32_t &cnt_box, uint32_t &cnt_ref) {
    if (__ballot(in_root) == 0ull) return;
    const uint32_t cnt = root.count_flags & 0x7fffffffu;
    if (COUNT && in_root) { cnt_box += 1; cnt_ref += cnt; }
    const TriRec *__restrict__ T = tris + root.first;
    bool mine = in_root;
    auto test_one = [&](const TriRec &tr) {
        // Flyscene::rayTriangleIntersection, flyscene.cpp:787-819 -- straight-line form: every lane evaluates the same
        // operations (a zero d.n just produces inf/NaN that the final predicate rejects, exactly like the reference's early
        // `return -72`), which removes the exec-mask juggling of nested branches.
        const float dn = dot3(dx, dy, dz, tr.nx, tr.ny, tr.nz);
        const float t = (tr.nA - dot3(ox, oy, oz, tr.nx, tr.ny, tr.nz)) / dn;
        const float v2x = (ox + t * dx) - tr.ax, v2y = (oy + t * dy) - tr.ay, v2z = (oz + t * dz) - tr.az;
        const float d02 = dot3(tr.e0x, tr.e0y, tr.e0z, v2x, v2y, v2z);
        const float d12 = dot3(tr.e1x, tr.e1y, tr.e1z, v2x, v2y, v2z);
        const float u = (tr.d11 * d02 - tr.d01 * d12) * tr.inv_denom;
        const float v = (tr.d00 * d12 - tr.d01 * d02) * tr.inv_denom;
        const bool ok = mine && !(ANY && (tr.flags & 1u)) && (dn != 0) && (u >= 0) && (v >= 0) && (u + v < 1) && (t > 0.00001f);
        if (ANY) {
            occluded = occluded || (ok && t < 0.98f);
        } else {
            const bool better = ok && (t < best_t || (t == best_t && static_cast<int>(tr.face) < best_f));
            best_t = better ? t : best_t;
            best_f = better ? static_cast<int>(tr.face) : best_f;
        }
    };
    if (ANY && !COUNT && seg.on) {
        // shadow unit: only the triangles whose plane is crossed between a light sample and the hit point
        unsigned long long keep = (cnt >= 64u ? ~0ull : ((1ull << cnt) - 1ull)) & ~skip;
        keep &= ~__ballot(seg.prepared ? plane_rules_out_prepared(seg, pl) : plane_rules_out(seg, pl.nx, pl.ny, pl.nz, pl.nA));
        RT_PROF_ADD(threadIdx.x & 63, 6, 1); RT_PROF_ADD(threadIdx.x & 63, 1, __popcll(keep));
        while (keep != 0ull) {
            const uint32_t k0 = static_cast<uint32_t>(__builtin_ctzll(keep));
            keep &= keep - 1ull;
            if (keep != 0ull) {
                const uint32_t k1 = static_cast<uint32_t>(__builtin_ctzll(keep));
                keep &= keep - 1ull;
                TriRec ta, tb;
                tri_load_uniform_pair(T + k0, T + k1, ta, tb);
                RT_PROF_ADD(threadIdx.x & 63, 0, 2);
                test_one(ta);
                test_one(tb);
            } else {
                RT_PROF_ADD(threadIdx.x & 63, 0, 1);
                test_one(tri_load_uniform(T + k0));
            }
            mine = mine && !occluded;
            if (__ballot(mine) == 0ull) return;
        }
        return;
    }
    uint32_t k = 0;
    RT_PROF_ADD(threadIdx.x & 63, 6, 1);
    for (; k + 1u < cnt; k += 2u) {          // two records per step: two independent chains in flight
        TriRec ta, tb;
        tri_load_uniform2(T + k, ta, tb);
        RT_PROF_ADD(threadIdx.x & 63, 0, 2);
        test_one(ta);
        test_one(tb);
        if (ANY && !COUNT && (k & 6u) == 6u) {
            mine = mine && !occluded;
            if (__ballot(mine) == 0ull) return;
        }
    }
    if (k < cnt) { RT_PROF_ADD(threadIdx.x & 63, 0, 1); test_one(tri_load_uniform(T + k)); }
}

template <bool ANY, bool COUNT, bool FLAT, bool STAGED = true>
__device__ __forceinline__ void walk(const DNode &root, const DNode *__restrict__ nodes, const TriRec *__restrict__ tris,
                                     const ChunkBound *__restrict__ chunks, const uint32_t *__restrict__ leaf_chunk0,
                                     const float extent, const WaveStack stk, const int lane, const WalkCtl wc, const LanePlane &pl, bool in_root,
                                     const float ox, const float oy, const float oz, const float dx, const float dy, const float dz,
                                     const float bx, const float by, const float bz,
                                     const float brx, const float bry, const float brz,
                                     float &best_t, int &best_f, bool &occluded, uint32_t &cnt_box, uint32_t &cnt_ref) {
    uint32_t sig_unused = 0u;
    if (FLAT) flat_walk<ANY, COUNT>(root, tris, in_root, wc.seg, wc.skip, pl, ox, oy, oz, dx, dy, dz, best_t, best_f, occluded, cnt_box, cnt_ref);
    else packet_walk<ANY, COUNT, STAGED>(nodes, tris, chunks, leaf_chunk0, extent, stk, lane, wc, in_root, ox, oy, oz, dx, dy, dz, bx, by, bz,
                                 brx, bry, brz, best_t, best_f, occluded, cnt_box, cnt_ref, sig_unused);
}

// ======================================================================================================
// SHAFT WALK -- the traversal of a shadow unit (k_shadow_shaft: one (hit, light) pair -- or one 64-sample pass of it -- per wave).
//
// All 64 segments of the unit end at the shaded hit point h and start inside the box S of the unit's light samples: they lie in the
// SHAFT hull(S, h).  A counted hit (0.00001 < t < 0.98) has its computed point inside the inflated chunk box it belongs to
// (rt_capi.cpp, build_chunk_bounds) and on its segment up to rounding, so a node whose CONTENT box (padded like the per-ray
// content test) does not meet the shaft cannot contribute to any ray of the unit -- whatever the reference's own box tests say.
// And a node whose OWN box (padded) misses every point a ray of the unit can reach -- the shaft and the FAR cone behind h, the
// rays continue beyond the hit point: boxIntersect has no upper bound -- fails the reference's boxIntersect for every ray of the
// unit by a margin far above the rounding of its float slab test, so no ray enters the subtree.
//
// The shaft test runs with LANE = (CHILD, TEST): the 8 children of an inner node x 8 separating tests (six tangent planes through
// h -- two per axis-aligned projection; the x-z and y-z pairs are the side faces of the pyramid over an axis-aligned light
// rectangle -- plus the near and the far bounding box) take ONE wave step; each lane keeps its test's coefficients in registers for
// the whole unit.  Only the surviving children are tested per ray (lane = ray) with the reference's exact boxIntersect; their
// records are broadcast with v_readlane from the lanes that loaded them (no second memory round trip).  The top of the tree is
// read from LDS (the first RT_LDS_NODES DNodes of the breadth-first array), the rest from global memory.  Leaves found by the
// walk are processed afterwards, chunk bounds again with lane = (chunk, test).
// The candidate set of every ray is unchanged: leaves are still entered only through the per-ray exact tests of the whole
// ancestor chain (BoxTree::intersect, boxTree.cpp:150-173).
// ======================================================================================================
#ifndef RT_SHAFT_TRUNC
#define RT_SHAFT_TRUNC 0.0185f        // the tip of the shaft that holds no counted point (t > 0.9815; 0: keep it)
#endif
struct ShaftLanes {
    float r[8];          // this lane's test (lane & 7): planes 0-5 (ax+, ax-, ay+, ay-, az+, az-, c, 2*margin); 6: near box (lo, hi); 7: far box
    float pad;           // padding of the tested boxes (>= the per-ray slab_pad of every segment of the unit); wave-uniform
    bool node_ok;        // the node-box test is valid: no ray of the unit has a zero / non-finite direction component (0/0 = NaN makes
                         // the reference's min/max chain ACCEPT whatever the geometry says)
};

// lane & 7 = test index.  The tangent lines from h to the rectangle S are computed in the projection the lane's plane belongs to:
// planes 0,1 (x,z), 2,3 (y,z), 4,5 (x,y); a line a*u + b*v + c = 0 through h with S on its non-positive side, given as the plane
// (ax, ay, az, c) with the third coefficient zero and split into positive / negative parts, so that
//      min over the corners of a box = ax+ * lx + ax- * hx + ay+ * ly + ay- * hy + az+ * lz + az- * hz + c      (max: lo <-> hi)
__device__ __forceinline__ ShaftLanes make_shaft_lanes(const int lane, const float hx, const float hy, const float hz, const float slx, const float sly,
                                                       const float slz, const float shx, const float shy, const float shz, const float extent,
                                                       const float trunc = 0.0f) {
    ShaftLanes SL;
    const int tk = lane & 7, proj = tk >> 1;
    const bool q1 = (tk & 1) != 0;
    const float big = fmaxf(fmaxf(fabsf(slx), fabsf(shx)), fmaxf(fabsf(sly), fabsf(shy))) + fmaxf(fabsf(slz), fabsf(shz));
    const float scale = extent + big + (fabsf(hx) + fabsf(hy) + fabsf(hz));
    // per-ray slab_pad = 4e-4 * (|o|_1 + extent) with o inside S
    SL.pad = 4e-4f * ((fmaxf(fabsf(slx), fabsf(shx)) + fmaxf(fabsf(sly), fabsf(shy)) + fmaxf(fabsf(slz), fabsf(shz))) + extent) * 1.001f;
    SL.node_ok = true;
    const float hu = proj == 1 ? hy : hx, hv = proj == 2 ? hy : hz;
    const float u0 = proj == 1 ? sly : slx, u1 = proj == 1 ? shy : shx;
    const float v0 = proj == 2 ? sly : slz, v1 = proj == 2 ? shy : shz;
    const bool ul = hu < u0, ur = hu > u1, vl = hv < v0, vr = hv > v1;
    const bool su0 = !(ul || ur), sv0 = !(vl || vr);
    const float near_u = ul ? u0 : u1, far_u = ul ? u1 : u0, near_v = vl ? v0 : v1, far_v = vl ? v1 : v0;
    // the two tangent corners of the rectangle seen from h (q = 0 / 1)
    const float cu = q1 ? (su0 ? u1 : (sv0 ? near_u : far_u)) : (su0 ? u0 : near_u);
    const float cv = q1 ? (sv0 ? v1 : near_v) : (sv0 ? v0 : (su0 ? near_v : far_v));
    const float mu = 0.5f * (u0 + u1) - hu, mv = 0.5f * (v0 + v1) - hv;        // rectangle centre relative to h
    const float nu = -(cv - hv), nv = cu - hu;
    const float fm = nu * mu + nv * mv;
    const float mag = fabsf(nu) + fabsf(nv);
    // usable: h outside the rectangle and the centre strictly on one side (a degenerate rectangle in line with h has no tangent)
    const bool ok = !(su0 && sv0) && (fabsf(fm) > 1e-5f * mag * scale);
    const float sgn = fm > 0.0f ? -1.0f : 1.0f;
    const float a = ok ? sgn * nu : 0.0f, b = ok ? sgn * nv : 0.0f;
    const float margin = 2e-5f * mag * scale;                                   // >> the rounding of the plane evaluations
    const float c = ok ? -(a * hu + b * hv) - margin : -1.0f;                   // unusable: never separates (value -1, far margin 3e38)
    const float ax = proj == 1 ? 0.0f : a, ay = proj == 0 ? 0.0f : (proj == 1 ? a : b), az = proj == 2 ? 0.0f : b;
    SL.r[0] = fmaxf(ax, 0.0f); SL.r[1] = fminf(ax, 0.0f); SL.r[2] = fmaxf(ay, 0.0f); SL.r[3] = fminf(ay, 0.0f);
    SL.r[4] = fmaxf(az, 0.0f); SL.r[5] = fminf(az, 0.0f); SL.r[6] = c; SL.r[7] = ok ? 2.0f * margin : 3e38f;
    if (tk == 6) {          // near box: AABB of hull(S, h)
        SL.r[0] = fminf(slx, hx); SL.r[1] = fminf(sly, hy); SL.r[2] = fminf(slz, hz);
        SL.r[3] = fmaxf(shx, hx); SL.r[4] = fmaxf(shy, hy); SL.r[5] = fmaxf(shz, hz);
        if (trunc > 0.0f) {
            // ... without its TIP: where h lies beyond S along an axis (h_k < sl_k), a point p of a segment s -> h with p_k < h_k + trunc (sl_k - h_k)
            // has t > 1 - trunc = 0.9815 on that segment -- lightStrikes counts t < 0.98 only.  The box then bounds every COUNTED point (content
            // boxes, chunk bounds, the triangles' vertex boxes); it says nothing about where the rays run, so own boxes are not tested against it
            // (shaft_lane_test: near_box = false).  m: far above the rounding of a computed point o + t d (~1e-7 |o|).
            const float m = 1e-5f * scale;
            SL.r[0] += hx < slx ? fmaxf(trunc * (slx - hx) - m, 0.0f) : 0.0f; SL.r[3] -= hx > shx ? fmaxf(trunc * (hx - shx) - m, 0.0f) : 0.0f;
            SL.r[1] += hy < sly ? fmaxf(trunc * (sly - hy) - m, 0.0f) : 0.0f; SL.r[4] -= hy > shy ? fmaxf(trunc * (hy - shy) - m, 0.0f) : 0.0f;
            SL.r[2] += hz < slz ? fmaxf(trunc * (slz - hz) - m, 0.0f) : 0.0f; SL.r[5] -= hz > shz ? fmaxf(trunc * (hz - shz) - m, 0.0f) : 0.0f;
        }
    }
    if (tk == 7) {          // far box: AABB of the far cone { h + tau * (h - s) : s in S, tau >= 0 }
        SL.r[0] = hx >= shx ? hx : -3e38f; SL.r[1] = hy >= shy ? hy : -3e38f; SL.r[2] = hz >= shz ? hz : -3e38f;
        SL.r[3] = hx <= slx ? hx : 3e38f; SL.r[4] = hy <= sly ? hy : 3e38f; SL.r[5] = hz <= slz ? hz : 3e38f;
    }
    return SL;
}

// this lane's test on a padded box: near = the box is outside the near shaft by this test, far = outside the far cone by this test.
// NaN anywhere compares false = keep.
__device__ __forceinline__ void shaft_lane_test(const ShaftLanes &SL, const int tk, const float lx, const float ly, const float lz, const float hx, const float hy,
                                                const float hz, bool &near_out, bool &far_out, const bool near_box = true) {
    const float mn = __builtin_fmaf(SL.r[0], lx, __builtin_fmaf(SL.r[1], hx, __builtin_fmaf(SL.r[2], ly, __builtin_fmaf(SL.r[3], hy,
                     __builtin_fmaf(SL.r[4], lz, __builtin_fmaf(SL.r[5], hz, SL.r[6]))))));
    const float mx = __builtin_fmaf(SL.r[0], hx, __builtin_fmaf(SL.r[1], lx, __builtin_fmaf(SL.r[2], hy, __builtin_fmaf(SL.r[3], ly,
                     __builtin_fmaf(SL.r[4], hz, __builtin_fmaf(SL.r[5], lz, SL.r[6]))))));
    const bool box_out = (lx > SL.r[3]) || (hx < SL.r[0]) || (ly > SL.r[4]) || (hy < SL.r[1]) || (lz > SL.r[5]) || (hz < SL.r[2]);
    near_out = tk < 6 ? (mn > 0.0f) : (tk == 6 && near_box && box_out);
    far_out = tk < 6 ? (mx + SL.r[7] < 0.0f) : (tk == 7 && box_out);
}
// byte j of a 64-bit ballot, for lane j < 8: does any of the 8 tests of child / chunk j say "outside"?
__device__ __forceinline__ bool ballot_byte_any(const unsigned long long b, const int lane) {
    return ((b >> ((lane & 7) * 8)) & 0xffull) != 0ull;
}

// The lanes' coefficients are parked in LDS between uses (8 records of 8 floats per wave, written by lanes 0-7 once per unit): they are
// needed once per group and once per 8 chunk bounds, and eight registers held across the whole walk were eight registers spilled.
struct ShaftCtl { float pad; bool node_ok; };
__device__ __forceinline__ void shaft_lanes_store(float4 *lds, const int lane, const ShaftLanes &SL) {
    if (lane < 8) { lds[2 * lane] = make_float4(SL.r[0], SL.r[1], SL.r[2], SL.r[3]); lds[2 * lane + 1] = make_float4(SL.r[4], SL.r[5], SL.r[6], SL.r[7]); }
}
__device__ __forceinline__ ShaftLanes shaft_lanes_load(const float4 *lds, const int tk, const ShaftCtl &SC) {
    const float4 a = lds[2 * tk], b = lds[2 * tk + 1];
    ShaftLanes SL;
    SL.r[0] = a.x; SL.r[1] = a.y; SL.r[2] = a.z; SL.r[3] = a.w; SL.r[4] = b.x; SL.r[5] = b.y; SL.r[6] = b.z; SL.r[7] = b.w;
    SL.pad = SC.pad; SL.node_ok = SC.node_ok;
    return SL;
}

// ---- lane = triangle: one 64-triangle chunk against the unit's shaft --------------------------------------------------------------
// A counted hit of triangle T (0.00001 < t < 0.98 and the barycentric test of rayTriangleIntersection passed IN FLOAT) has its computed
// point within ChunkBound::infl (per axis) of T itself (rt_capi.cpp: build_chunk_bounds -- the chunk box is the union of exactly these
// neighbourhoods), and that point lies on its segment up to rounding, i.e. inside hull(S, h).  So T cannot be hit by ANY ray of the unit
// when the three boxes [v - m, v + m] around its vertices (m = 1.0625 * infl: B and C are re-derived as A + edge, one rounding each)
// lie strictly outside one of the shaft's tangent planes (their convex hull contains T's neighbourhood) or outside the near box; the
// planes carry the same margins as in the node / chunk tests.  Independently, plane_rules_out (round 1, flat scenes) proves from the
// reference's own two dot products that t <= 0 or |t| >= 0.98 for every sample of S: this is what removes the face h lies on and its
// coplanar neighbours, which no geometric test can separate from h.
// The unit's planes are kept as a per-wave LDS record (written once per unit by the lanes that own them) so that the test costs no
// register for the rest of the walk: 11 broadcast ds_read_b128.
#define RT_SHAFT_TRI_REC 11           // float4 per wave: 6 planes (a_u, a_v, c, |a_u| + |a_v|), near box lo / hi, (h, m0), S lo, S hi
__device__ __forceinline__ void shaft_tri_store(float4 *rec, const int lane, const ShaftLanes &SL, const float hx, const float hy, const float hz,
                                                const float slx, const float sly, const float slz, const float shx, const float shy, const float shz) {
    const int proj = (lane & 7) >> 1;
    const float ax = SL.r[0] + SL.r[1], ay = SL.r[2] + SL.r[3], az = SL.r[4] + SL.r[5];        // (one addend is zero: exact)
    const float au = proj == 1 ? ay : ax, av = proj == 2 ? ay : az;
    if (lane < 6) rec[lane] = make_float4(au, av, SL.r[6], fabsf(au) + fabsf(av));
    if (lane == 6) { rec[6] = make_float4(SL.r[0], SL.r[1], SL.r[2], 0.f); rec[7] = make_float4(SL.r[3], SL.r[4], SL.r[5], 0.f); }
    if (lane == 7) {
        const float m0 = 2e-5f * ((fabsf(slx) + fabsf(shx)) + (fabsf(sly) + fabsf(shy)) + (fabsf(slz) + fabsf(shz)) + (fabsf(hx) + fabsf(hy) + fabsf(hz)));
        rec[8] = make_float4(hx, hy, hz, m0); rec[9] = make_float4(slx, sly, slz, 0.f); rec[10] = make_float4(shx, shy, shz, 0.f);
    }
}
__device__ __forceinline__ bool tri_outside_shaft(const float4 *rec, const TriRec &tr, const float m) {
    const float bx = tr.ax + tr.e1x, by = tr.ay + tr.e1y, bz = tr.az + tr.e1z;
    const float cx = tr.ax + tr.e0x, cy = tr.ay + tr.e0y, cz = tr.az + tr.e0z;
    bool out = false;
#pragma unroll
    for (int p = 0; p < 6; ++p) {
        const float4 pl = rec[p];
        const int proj = p >> 1;
        const float ua = proj == 1 ? tr.ay : tr.ax, ub = proj == 1 ? by : bx, uc = proj == 1 ? cy : cx;
        const float va = proj == 2 ? tr.ay : tr.az, vb = proj == 2 ? by : bz, vc = proj == 2 ? cy : cz;
        const float fa = __builtin_fmaf(pl.x, ua, pl.y * va), fb = __builtin_fmaf(pl.x, ub, pl.y * vb), fc = __builtin_fmaf(pl.x, uc, pl.y * vc);
        out = out || (fminf(fminf(fa, fb), fc) + (pl.z - m * pl.w) > 0.0f);
    }
    const float4 lo = rec[6], hi = rec[7];
    out = out || (fminf(fminf(tr.ax, bx), cx) - m > hi.x) || (fmaxf(fmaxf(tr.ax, bx), cx) + m < lo.x)
              || (fminf(fminf(tr.ay, by), cy) - m > hi.y) || (fmaxf(fmaxf(tr.ay, by), cy) + m < lo.y)
              || (fminf(fminf(tr.az, bz), cz) - m > hi.z) || (fmaxf(fmaxf(tr.az, bz), cz) + m < lo.z);
    const float4 h = rec[8], s0 = rec[9], s1 = rec[10];
    const SegPacket g{true, false, h.x, h.y, h.z, s0.x, s0.y, s0.z, s1.x, s1.y, s1.z, h.w};
    return out || plane_rules_out(g, tr.nx, tr.ny, tr.nz, tr.nA);
}

// The same test for a packet whose rays share their ORIGIN o (a primary tile: the camera centre; light-centre segments: the light) and
// run through targets inside a box: every point a ray can reach (any t >= 0) lies in the cone from o through that box, all of it on the
// non-positive side of the six tangent planes through o -- so a triangle whose vertex boxes are strictly outside one plane is missed by
// every ray, whatever t.  The near box (AABB of hull(o, targets)) only bounds hits BEFORE the targets: use_box is set for the
// light-centre segments (counted hits have t < 0.98) and not for closest-hit rays.  The record is the one shaft_tri_store writes with
// h := o and S := the target box.
__device__ __forceinline__ bool tri_outside_cone(const float4 *rec, const TriRec &tr, const float m, const bool use_box) {
    const float bx = tr.ax + tr.e1x, by = tr.ay + tr.e1y, bz = tr.az + tr.e1z;
    const float cx = tr.ax + tr.e0x, cy = tr.ay + tr.e0y, cz = tr.az + tr.e0z;
    bool out = false;
#pragma unroll
    for (int p = 0; p < 6; ++p) {
        const float4 pl = rec[p];
        const int proj = p >> 1;
        const float ua = proj == 1 ? tr.ay : tr.ax, ub = proj == 1 ? by : bx, uc = proj == 1 ? cy : cx;
        const float va = proj == 2 ? tr.ay : tr.az, vb = proj == 2 ? by : bz, vc = proj == 2 ? cy : cz;
        const float fa = __builtin_fmaf(pl.x, ua, pl.y * va), fb = __builtin_fmaf(pl.x, ub, pl.y * vb), fc = __builtin_fmaf(pl.x, uc, pl.y * vc);
        out = out || (fminf(fminf(fa, fb), fc) + (pl.z - m * pl.w) > 0.0f);
    }
    if (use_box) {
        const float4 lo = rec[6], hi = rec[7];
        out = out || (fminf(fminf(tr.ax, bx), cx) - m > hi.x) || (fmaxf(fmaxf(tr.ax, bx), cx) + m < lo.x)
                  || (fminf(fminf(tr.ay, by), cy) - m > hi.y) || (fmaxf(fmaxf(tr.ay, by), cy) + m < lo.y)
                  || (fminf(fminf(tr.az, bz), cz) - m > hi.z) || (fmaxf(fmaxf(tr.az, bz), cz) + m < lo.z);
    }
    return out;
}

#define RT_LEAF_SLOTS 16
#define RT_COST_TRI_STEP 58u

struct ShaftLds {
    const DNode *nodes;            // LDS copy of the first n_lds nodes (the top of the breadth-first array)
    uint32_t n_lds;
    uint32_t *lnode;               // per-wave leaf list (RT_LEAF_SLOTS)
    unsigned long long *lmask;
    float4 *tri;                   // per-wave shaft record of the lane = triangle test (RT_SHAFT_TRI_REC)
    float4 *shaft;                 // per-wave copy of the lanes' coefficients (16 float4: ShaftLanes::r of test tk at [2 tk], [2 tk + 1])
#ifdef RT_PROFILE
    PhaseClock *pc;
#endif
};

// One leaf of the shaft walk.  Chunk bounds are shaft-tested 8 at a time (lane = (chunk, test)); what survives goes through the per-ray
// chunk test (lane = ray) and the triangle tests (lane = triangle).
// where a shaft walk hands big leaves to (the leaf-task queue of the k_shadow<.., CONT> launch that follows): budget 0 = never
struct ShaftTasks {
    ContTask *tasks;
    uint32_t *count;
    uint32_t cap, budget, target, unit;
};

template <bool TASKS>
__device__ __forceinline__ void shaft_leaf(const uint32_t ni, const uint32_t first, const uint32_t cnt, const uint32_t chunk0, const TriRec *__restrict__ tris,
                                           const ChunkBound *__restrict__ chunks, const int lane, const RayLane &R, const ShaftCtl &SC, const ShaftTasks &TQ,
                                           const ShaftLds &sl, const uint32_t c_begin, const uint32_t c_end, unsigned long long live, bool &occluded) {
    const float4 *__restrict__ srec = sl.tri;
    const float ox = R.ox, oy = R.oy, oz = R.oz, dx = R.dx, dy = R.dy, dz = R.dz;
    const TriRec *__restrict__ T = tris + first;
    bool mine = ((live >> lane) & 1ull) != 0ull;
    auto test_lane = [&](const TriRec &tr) {
        // Flyscene::rayTriangleIntersection, flyscene.cpp:787-819 (lanes = rays)
        const float dn = dot3(dx, dy, dz, tr.nx, tr.ny, tr.nz);
        const float t = (tr.nA - dot3(ox, oy, oz, tr.nx, tr.ny, tr.nz)) / dn;
        const float v2x = (ox + t * dx) - tr.ax, v2y = (oy + t * dy) - tr.ay, v2z = (oz + t * dz) - tr.az;
        const float d02 = dot3(tr.e0x, tr.e0y, tr.e0z, v2x, v2y, v2z);
        const float d12 = dot3(tr.e1x, tr.e1y, tr.e1z, v2x, v2y, v2z);
        const float u = (tr.d11 * d02 - tr.d01 * d12) * tr.inv_denom;
        const float v = (tr.d00 * d12 - tr.d01 * d02) * tr.inv_denom;
        const bool ok = mine && !(tr.flags & 1u) && (dn != 0) && (u >= 0) && (v >= 0) && (u + v < 1) && (t > 0.00001f);
        occluded = occluded || (ok && t < 0.98f);
    };
    if (c_begin == 0u && cnt <= RT_SCALAR_LEAF_MAX && static_cast<uint32_t>(__popcll(live)) * RT_COST_TRI_STEP > cnt * RT_COST_RAY_MODE) {
        // small leaf, many rays: every lane steps through the wave-uniform records (scalar loads)
        uint32_t k = 0;
        for (; k + 1u < cnt; k += 2u) {
            TriRec ta, tb;
            tri_load_uniform2(T + k, ta, tb);
            test_lane(ta);
            test_lane(tb);
        }
        if (k < cnt) test_lane(tri_load_uniform(T + k));
        return;
    }
    const uint32_t nchunk_all = (cnt + 63u) >> 6;
    const uint32_t nchunk = c_end < nchunk_all ? c_end : nchunk_all;                  // chunks [c_begin, nchunk) are processed here
    const ChunkBound *__restrict__ cbounds = chunks + chunk0;
    const int tk = lane & 7, tc = lane >> 3;
    unsigned long long occ_new = 0ull;
    uint32_t c_first = c_begin;
    // A big leaf with many live rays is not ground through by this wave (one unit crossing a 979-triangle leaf with 64 rays is ~60k
    // instructions: the tail of the whole launch): it becomes chunk-range tasks for the leaf-task launch, which merges its occluded
    // bits into `vis` with atomicAnd.  Estimate as leaf_visit: a third of the chunks survive the per-ray test.
    if (TASKS && TQ.budget != 0u) {
        const uint32_t est = nchunk * RT_COST_CHUNK_TEST + static_cast<uint32_t>(__popcll(live)) * ((nchunk + 2u) / 3u) * RT_COST_TRI_STEP;
        if (est > TQ.budget) {
            uint32_t ntask = (est + TQ.target - 1u) / TQ.target;
            if (ntask > nchunk) ntask = nchunk;
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(TQ.count, ntask);
            base = uniform_u32(base);
            // the counter only grows (the consumer clamps it): pieces past the end of the queue are processed here
            const uint32_t fit = base >= TQ.cap ? 0u : (ntask < TQ.cap - base ? ntask : TQ.cap - base);
            for (uint32_t i = static_cast<uint32_t>(lane); i < fit; i += 64u) {
                ContTask t;
                t.unit = TQ.unit; t.node = ni; t.mask = live;
                t.c_begin = static_cast<uint32_t>(static_cast<unsigned long long>(nchunk) * i / ntask);
                t.c_end = static_cast<uint32_t>(static_cast<unsigned long long>(nchunk) * (i + 1u) / ntask);
                t.pad0 = t.pad1 = 0u;
                TQ.tasks[base + i] = t;
            }
            if (fit == ntask) return;
            c_first = static_cast<uint32_t>(static_cast<unsigned long long>(nchunk) * fit / ntask);     // (queue full) the rest of the leaf, inline
        }
    }
    for (uint32_t cb0 = c_first & ~7u; cb0 < nchunk && live != 0ull; cb0 += 8u) {
        const uint32_t myc = cb0 + static_cast<uint32_t>(tc);
        const ChunkBound bd = cbounds[myc < nchunk ? myc : cb0];
        bool near_out, far_out;
        __builtin_amdgcn_wave_barrier();
        const ShaftLanes SL = shaft_lanes_load(sl.shaft, tk, SC);
        shaft_lane_test(SL, tk, bd.lo[0] - SL.pad, bd.lo[1] - SL.pad, bd.lo[2] - SL.pad, bd.hi[0] + SL.pad, bd.hi[1] + SL.pad, bd.hi[2] + SL.pad, near_out, far_out);
        const unsigned long long b_out = __ballot(near_out && bd.never < 1.5f);
        const uint32_t nhere = nchunk - cb0 < 8u ? nchunk - cb0 : 8u;
        unsigned long long cm = __ballot(static_cast<uint32_t>(lane) < nhere && cb0 + static_cast<uint32_t>(lane) >= c_first && !ballot_byte_any(b_out, lane));      // bit j: chunk cb0 + j survives
        RT_PROF_ADD(lane, 92, 1); RT_PROF_ADD(lane, 93, nhere); RT_PROF_ADD(lane, 94, __popcll(cm));
        // next chunk of `cm` that some live ray cannot skip (per-ray conservative test, lanes = rays), or -1
        auto find_next = [&](unsigned long long &todo_out) -> int {
            while (cm != 0ull) {
                const int j = static_cast<int>(__builtin_ctzll(cm));
                cm &= cm - 1ull;
                unsigned long long todo = live;
                if (lane_f(bd.never, 8 * j) < 1.5f) {
                    const float l0 = lane_f(bd.lo[0], 8 * j), l1 = lane_f(bd.lo[1], 8 * j), l2 = lane_f(bd.lo[2], 8 * j);
                    const float h0 = lane_f(bd.hi[0], 8 * j), h1 = lane_f(bd.hi[1], 8 * j), h2 = lane_f(bd.hi[2], 8 * j);
                    const float t0x = (l0 - R.slab_pad - ox) * R.idx, t1x = (h0 + R.slab_pad - ox) * R.idx;
                    const float t0y = (l1 - R.slab_pad - oy) * R.idy, t1y = (h1 + R.slab_pad - oy) * R.idy;
                    const float t0z = (l2 - R.slab_pad - oz) * R.idz, t1z = (h2 + R.slab_pad - oz) * R.idz;
                    const float tin = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z));
                    const float tout = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
                    bool miss = (tin > tout) || (tout < -1e-3f) || (tin > 0.981f);
#ifndef RT_NO_SLAB
                    // ... and the chunk's slab (ChunkBound::sn): f(t) = sn . (o + t d) over the part of the line inside the box and the counted range;
                    // a counted point has f in [slo, shi].  m: far above the rounding of the two dot products (~4e-7 (|o| + |d|) |sn|_1)
                    const float n0 = lane_f(bd.sn[0], 8 * j), n1 = lane_f(bd.sn[1], 8 * j), n2 = lane_f(bd.sn[2], 8 * j);
                    const float fa = __builtin_fmaf(n0, ox, __builtin_fmaf(n1, oy, n2 * oz)), fb = __builtin_fmaf(n0, dx, __builtin_fmaf(n1, dy, n2 * dz));
                    const float f0 = __builtin_fmaf(fmaxf(tin, -1e-3f), fb, fa), f1 = __builtin_fmaf(fminf(tout, 0.981f), fb, fa);
                    const float m = 0.04f * R.slab_pad;
                    miss = miss || (fmaxf(f0, f1) + m < lane_f(bd.slo, 8 * j)) || (fminf(f0, f1) - m > lane_f(bd.shi, 8 * j));
#endif
                    todo = live & ~__ballot(miss);
                }
                if (todo != 0ull) { RT_PROF_ADD(lane, 95, 1); todo_out = todo; return j; }
            }
            return -1;
        };
        auto load_chunk = [&](const int j) -> TriRec {
            const uint32_t k = (cb0 + static_cast<uint32_t>(j)) * 64u + static_cast<uint32_t>(lane);
            return T[k < cnt ? k : 0u];
        };
        unsigned long long todo_cur = 0ull, todo_nxt = 0ull;
        int cur = find_next(todo_cur);
        if (cur < 0) continue;
        TriRec tr = load_chunk(cur);
        while (cur >= 0) {
            const uint32_t c0 = (cb0 + static_cast<uint32_t>(cur)) * 64u;
            const uint32_t n = cnt - c0 < 64u ? cnt - c0 : 64u;
            bool hast = static_cast<uint32_t>(lane) < n && !(tr.flags & 1u);
            unsigned long long todo = todo_cur & live;
            const uint32_t m_rays = static_cast<uint32_t>(__popcll(todo));
            unsigned long long tmask = ~0ull;
            // (a chunk that is never culled as a whole still carries the inflation of its well-conditioned triangles -- 0: it has none --; the
            //  ill-conditioned ones, TriRec::flags bit 1, are kept whatever the test says: dodge's collinear slivers sit one or two to a chunk and
            //  used to cost 64 rays x 64 triangles wherever a unit entered their leaf)
            if (m_rays >= RT_TRI_SHAFT_MIN && (lane_f(bd.never, 8 * cur) < 1.5f || lane_f(bd.infl, 8 * cur) > 0.0f)) {
                // lane = triangle: which triangles of the chunk can be hit by ANY ray of the unit
                RT_PH(sl, 4);
                __builtin_amdgcn_wave_barrier();
                hast = hast && ((tr.flags & 2u) != 0u || !tri_outside_shaft(srec, tr, lane_f(bd.infl, 8 * cur) * 1.0625f));
                tmask = __ballot(hast);
                RT_PROF_ADD(lane, 70, 1); RT_PROF_ADD(lane, 71, __popcll(tmask)); RT_PROF_ADD(lane, 72, m_rays); RT_PROF_ADD(lane, 73, tmask == 0ull ? 1 : 0);
            }
            RT_PH(sl, 5);
            if (tmask == 0ull) {
                todo = 0ull;
            } else if (tmask != ~0ull && static_cast<uint32_t>(__popcll(tmask)) * (RT_COST_RAY_MODE + RT_RAYMODE_EXTRA) < m_rays * RT_COST_TRI_STEP) {
                // few triangles left, many rays: lanes = rays step through the survivors (wave-uniform records, scalar loads)
                const bool my = ((todo >> lane) & 1ull) != 0ull;
                bool occ_r = false;
                const TriRec *__restrict__ Tc = T + c0;
                auto hit_lane = [&](const TriRec &ta) -> bool {
                    // Flyscene::rayTriangleIntersection, flyscene.cpp:787-819 (lanes = rays)
                    const float dn = dot3(dx, dy, dz, ta.nx, ta.ny, ta.nz);
                    const float t = (ta.nA - dot3(ox, oy, oz, ta.nx, ta.ny, ta.nz)) / dn;
                    const float v2x = (ox + t * dx) - ta.ax, v2y = (oy + t * dy) - ta.ay, v2z = (oz + t * dz) - ta.az;
                    const float d02 = dot3(ta.e0x, ta.e0y, ta.e0z, v2x, v2y, v2z);
                    const float d12 = dot3(ta.e1x, ta.e1y, ta.e1z, v2x, v2y, v2z);
                    const float u = (ta.d11 * d02 - ta.d01 * d12) * ta.inv_denom;
                    const float v = (ta.d00 * d12 - ta.d01 * d02) * ta.inv_denom;
                    return my && (dn != 0) && (u >= 0) && (v >= 0) && (u + v < 1) && (t > 0.00001f) && (t < 0.98f);
                };
                while (tmask != 0ull) {
                    const uint32_t j0 = static_cast<uint32_t>(__builtin_ctzll(tmask));
                    tmask &= tmask - 1ull;
                    const bool two = tmask != 0ull;
                    const uint32_t j1 = two ? static_cast<uint32_t>(__builtin_ctzll(tmask)) : j0;
                    if (two) tmask &= tmask - 1ull;
                    RT_PROF_ADD(lane, 0, two ? 2 : 1);
                    TriRec ta, tb;
                    tri_load_uniform_pair(Tc + j0, Tc + j1, ta, tb);
                    const bool ha = hit_lane(ta), hb = hit_lane(tb);
                    occ_r = occ_r || ha || hb;
                    if (__ballot(my && !occ_r) == 0ull) break;
                }
                const unsigned long long ob = __ballot(occ_r);
                occ_new |= ob; live &= ~ob;
                todo = 0ull;
            }
            while (todo != 0ull) {
                const int r0 = static_cast<int>(__builtin_ctzll(todo));
                todo &= todo - 1ull;
                const bool two = todo != 0ull;
                const int r1 = two ? static_cast<int>(__builtin_ctzll(todo)) : r0;
                if (two) todo &= todo - 1ull;
                float tq[2]; bool inq[2];
                RT_PROF_ADD(lane, 2, two ? 2 : 1);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    // Flyscene::rayTriangleIntersection, flyscene.cpp:787-819 (lanes = triangles, the ray broadcast)
                    const int r = q == 0 ? r0 : r1;
                    const float rdx = lane_f(dx, r), rdy = lane_f(dy, r), rdz = lane_f(dz, r);
                    const float rox = lane_f(ox, r), roy = lane_f(oy, r), roz = lane_f(oz, r);
                    const float dn = dot3(rdx, rdy, rdz, tr.nx, tr.ny, tr.nz);
                    const float t = (tr.nA - dot3(rox, roy, roz, tr.nx, tr.ny, tr.nz)) / dn;
                    const float v2x = (rox + t * rdx) - tr.ax, v2y = (roy + t * rdy) - tr.ay, v2z = (roz + t * rdz) - tr.az;
                    const float d02 = dot3(tr.e0x, tr.e0y, tr.e0z, v2x, v2y, v2z);
                    const float d12 = dot3(tr.e1x, tr.e1y, tr.e1z, v2x, v2y, v2z);
                    const float u = (tr.d11 * d02 - tr.d01 * d12) * tr.inv_denom;
                    const float v = (tr.d00 * d12 - tr.d01 * d02) * tr.inv_denom;
                    tq[q] = t;
                    inq[q] = hast && (dn != 0) && (u >= 0) && (v >= 0) && (u + v < 1) && (t > 0.00001f);
                }
                if (!two) inq[1] = false;
                if (__ballot(inq[0] && tq[0] < 0.98f) != 0ull) { occ_new |= 1ull << r0; live &= ~(1ull << r0); }
                if (__ballot(inq[1] && tq[1] < 0.98f) != 0ull) { occ_new |= 1ull << r1; live &= ~(1ull << r1); }
            }
            RT_PH(sl, 3);
            if (live == 0ull) break;
            // (the next chunk's records used to be requested before this one was worked on: twenty registers for one hidden round trip.
            // Without them the kernel fits 80 registers -- six waves per SIMD cover the latency better: cfg4 -6 %, dodge -3 %)
            cur = find_next(todo_nxt); todo_cur = todo_nxt;
            if (cur >= 0) tr = load_chunk(cur);
        }
    }
    occluded = occluded || (((occ_new >> lane) & 1ull) != 0ull);
}


__device__ __forceinline__ DNode node_from_lane(const DNode &mine, const int j) {
    DNode o;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o.bmin[k] = lane_f(mine.bmin[k], j); o.bmax[k] = lane_f(mine.bmax[k], j);
        o.clo[k] = lane_f(mine.clo[k], j); o.chi[k] = lane_f(mine.chi[k], j);
    }
    o.first = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mine.first), j));
    o.count_flags = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mine.count_flags), j));
    o.pad[0] = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mine.pad[0]), j));
    o.pad[1] = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mine.pad[1]), j));
    return o;
}

template <bool TASKS>
__device__ __forceinline__ void shaft_walk(const DNode *__restrict__ nodes, const TriRec *__restrict__ tris, const ChunkBound *__restrict__ chunks,
                                           const WaveStack stk, const ShaftLds sl, const int lane, const DNode &root, const bool in_root,
                                           const RayLane &R, const float brx, const float bry, const float brz, const ShaftCtl &SC, const ShaftTasks &TQ, bool &occluded) {
    const float ox = R.ox, oy = R.oy, oz = R.oz;
    const unsigned long long m0 = __ballot(in_root);
    if (m0 == 0ull) return;
    const int tk = lane & 7, tc = lane >> 3;
    int sp = 0, nleaf = 0;
    if (root.count_flags & RT_NODE_LEAF) {
        if (lane == 0) { sl.lnode[0] = 0u; sl.lmask[0] = m0; }
        nleaf = 1;
    } else {
        // stack entry = one GROUP: the children [first, first + cnt) of an inner node whose box the rays in `mask` hit
        if (lane == 0) { stk.node[0] = root.first | ((root.count_flags & 0xfu) << 28); stk.mask[0] = m0; }
        sp = 1;
    }
    while (sp > 0 || nleaf > 0) {
        if (sp == 0 || nleaf > RT_LEAF_SLOTS - 8) {
            // the leaves found so far (ONE inlined copy of the leaf code: all of them are processed here)
            RT_PH(sl, 3);
            for (int k = 0; k < nleaf; ++k) {
                __builtin_amdgcn_wave_barrier();
                const uint32_t li = uniform_u32(sl.lnode[k]);
                unsigned long long lm = uniform_u64(sl.lmask[k]);
                lm &= ~__ballot(occluded);
                if (lm == 0ull) continue;
                uint32_t lf, lc, l0;
                if (li < sl.n_lds) { lf = sl.nodes[li].first; lc = sl.nodes[li].count_flags; l0 = sl.nodes[li].pad[0]; }
                else { lf = nodes[li].first; lc = nodes[li].count_flags; l0 = nodes[li].pad[0]; }
                shaft_leaf<TASKS>(li, uniform_u32(lf), uniform_u32(lc) & 0x7fffffffu, uniform_u32(l0), tris, chunks, lane, R, SC, TQ, sl, 0u, 0xffffffffu, lm, occluded);
            }
            nleaf = 0;
            continue;
        }
        --sp;
        RT_PH(sl, 1);
        __builtin_amdgcn_wave_barrier();
        const uint32_t ent = uniform_u32(stk.node[sp]);
        unsigned long long gm = uniform_u64(stk.mask[sp]);
        gm &= ~__ballot(occluded);
        if (gm == 0ull) continue;
        const uint32_t base = ent & 0x0fffffffu, gcnt = ent >> 28;
        // lane = (child tc, test tk): the child's record, then this lane's separating test on its content box and on its own box
        const uint32_t ci = base + (static_cast<uint32_t>(tc) < gcnt ? static_cast<uint32_t>(tc) : 0u);
        DNode ch;
        const bool resident = base + gcnt <= sl.n_lds;
        if (resident) ch = sl.nodes[ci];
        else ch = nodes[ci];
        bool c_near, c_far, n_near, n_far;
        const ShaftLanes SL = shaft_lanes_load(sl.shaft, tk, SC);
        shaft_lane_test(SL, tk, ch.clo[0] - SL.pad, ch.clo[1] - SL.pad, ch.clo[2] - SL.pad, ch.chi[0] + SL.pad, ch.chi[1] + SL.pad, ch.chi[2] + SL.pad, c_near, c_far);
        shaft_lane_test(SL, tk, ch.bmin[0] - SL.pad, ch.bmin[1] - SL.pad, ch.bmin[2] - SL.pad, ch.bmax[0] + SL.pad, ch.bmax[1] + SL.pad, ch.bmax[2] + SL.pad, n_near, n_far, false);
        const unsigned long long b_c = __ballot(c_near && ch.pad[1] == 0u), b_nn = __ballot(n_near), b_nf = __ballot(n_far);
        const bool culled = (SL.node_ok && ballot_byte_any(b_nn, lane) && ballot_byte_any(b_nf, lane)) || ballot_byte_any(b_c, lane);
        unsigned long long surv = __ballot(static_cast<uint32_t>(lane) < gcnt && !culled);       // bit j: child j survives
        RT_PROF_ADD(lane, 88, 1); RT_PROF_ADD(lane, 89, gcnt); RT_PROF_ADD(lane, 90, __popcll(surv));
        RT_PH(sl, 2);
        while (surv != 0ull) {
            const int j = static_cast<int>(__builtin_ctzll(surv));
            surv &= surv - 1ull;
            // the survivor's record as wave-uniform values: an LDS broadcast read where the group sits in the LDS copy of the top of the
            // tree (4 ds_read_b128), else 16 v_readlane from the lane that loaded it (cfg4: -2.5 % on k_shadow_shaft).  Measured and
            // rejected here: skipping the per-ray content test on inner nodes (+8 % on cfg4), prefetching the next unit's item with a
            // scalar load (+4 %: 16 more live SGPRs -> spills) or through one VGPR with the queue looking one unit ahead (+6 %: the
            // other waves of the SIMD already cover that latency), the records of non-resident groups through an LDS slot instead of
            // v_readlane (+1.5 % on cfg4), 2x / 4x / 8x larger k_stage grids (0 %), two survivors per step as independent chains (round 3:
            // +3 % dodge, +7 % cfg4 -- the second record's registers are spilled), the survivors in descending order (0.0 %: the any-hit walk
            // does not care which child comes first).
            const DNode nd = resident ? sl.nodes[base + static_cast<uint32_t>(j)] : node_from_lane(ch, 8 * j);
            bool h = ((gm >> lane) & 1ull) != 0ull && !occluded;
            RT_PROF_ADD(lane, 74, __popcll(__ballot(h)));
            if (nd.pad[1] == 0u) {   // per-ray content test (as packet_walk): no countable point of the segment inside the subtree's content box
                const float t0x = (nd.clo[0] - R.slab_pad - ox) * R.idx, t1x = (nd.chi[0] + R.slab_pad - ox) * R.idx;
                const float t0y = (nd.clo[1] - R.slab_pad - oy) * R.idy, t1y = (nd.chi[1] + R.slab_pad - oy) * R.idy;
                const float t0z = (nd.clo[2] - R.slab_pad - oz) * R.idz, t1z = (nd.chi[2] + R.slab_pad - oz) * R.idz;
                const float tin = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z));
                const float tout = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
                const bool miss = (tin > tout) || (tout < -1e-3f) || (tin > 0.981f);
                h = h && !miss;
                if (__ballot(h) == 0ull) continue;
            }
            h = h && box_hit_verified(nd.bmin, ox, oy, oz, R.dx, R.dy, R.dz, brx, bry, brz);     // BoundingBox::boxIntersect, exact
            const unsigned long long hm = __ballot(h);
            if (hm == 0ull) continue;
            RT_PROF_ADD(lane, 91, 1); RT_PROF_ADD(lane, 75, __popcll(hm));
            const uint32_t cj = base + static_cast<uint32_t>(j);
            if (nd.count_flags & RT_NODE_LEAF) {
                if ((nd.count_flags & 0x7fffffffu) == 0u) continue;
                // (a group adds at most 8 leaves and the list is emptied when it reaches RT_LEAF_SLOTS - 8 entries before the next group)
                if (lane == 0) { sl.lnode[nleaf] = cj; sl.lmask[nleaf] = hm; }
                ++nleaf;
            } else {
                if ((nd.count_flags & 0xfu) == 0u) continue;              // a "lost" node: no children
                if (lane == 0) { stk.node[sp] = nd.first | ((nd.count_flags & 0xfu) << 28); stk.mask[sp] = hm; }
                ++sp;
            }
        }
    }
    RT_PH(sl, 0);
}

// ======================================================================================================
// CONE WALK -- BoxTree::intersect (boxTree.cpp:150-173) for the 64 rays of a closest-hit or light-centre packet, GROUP by group.
//
// The stack walk (packet_walk) pops ONE node and then fetches and tests its children one at a time, each behind a dependent 64-byte
// scalar load: the walking launches of the trace stages issued at 4-13 % of the VALU rate, waiting on memory.  Here a stack entry is a
// GROUP -- the children of one inner node -- and a popped group takes ONE round trip: lane = (child, test), every lane reads its
// child's record (from the LDS copy of the top of the tree when the group lies there, else one vector load per lane, eight lanes
// per 64-byte record), exactly the layout of the shaft walk.
//
// Packets whose rays share their ORIGIN o (a primary tile: the camera centre; the light-centre segments of a tile: the light) get the
// shaft walk's separating tests with o as the apex: every point a ray reaches at any t >= 0 lies in the cone from o through the box
// of the packet's targets, on the non-positive side of the six tangent planes through o (make_shaft_lanes with h := o, S := target
// box).  A child whose CONTENT box lies strictly outside one plane holds no point a counted hit can have; so does one outside the AABB
// of hull(o, targets) when counted hits lie before the targets (light-centre segments: t < 0.98).  A child whose OWN (padded) box lies
// outside one plane is missed by every ray at every t >= 0: boxIntersect fails for all 64 by a margin far above the rounding of its
// slab test (not claimed when a ray has a zero / non-finite direction component: 0/0 = NaN makes the reference's min/max chain
// accept -- node_ok).  Only the surviving children are tested per ray (lane = ray) with the per-ray content test and the reference's
// exact boxIntersect, their records broadcast from LDS or with v_readlane.  Packets without a common origin (bounce rays, lanes with
// light lists of their own) skip the cone tests and keep the one-round-trip group fetch.
// Leaves are visited as soon as the group that found them has been handled (closest hit: best_t then prunes what is still on the
// stack), through leaf_visit with the cone record for its lane = triangle test.  The candidate set of every ray is unchanged: a leaf
// is entered only through the exact per-ray tests of its whole ancestor chain.
// ======================================================================================================
struct ConeCtl {
    bool have;           // the packet has a common origin and its lanes' coefficients sit in ShaftLds::shaft
    bool node_ok;        // no ray of the packet has a zero / non-finite box-test direction component
    bool box;            // counted hits lie before the targets: the near box bounds content too
    float pad;           // padding of the tested boxes (>= the per-ray slab_pad: all rays start at the apex)
};

// What a cone walk hands to the task launch that follows, and where it starts.
//   * Work is handed away in ONE reservation per unit: the leaves whose estimated cost exceeds `budget` and -- once the unit has popped
//     `group_budget` groups -- the groups still on its stack are collected in a per-wave pending list and written as tasks with a single
//     returning atomicAdd when the walk runs dry.  (Measured on dodgeColorTest.obj at 1080p, per-unit records of the diagnostic build: the
//     walking launches of both stages last exactly as long as their longest unit -- 150 k / 260 k cycles against a mean wave lifetime of
//     28 k / 16 k -- and such a unit is 10-16 groups at ~4 k cycles, 6-23 leaf visits at 2-6 k and 7-11 task reservations at 2-3 k
//     each: a serial chain, not a throughput problem.  The task launches were bound the same way by single (chunk, 64 rays) pieces.)
//   * A leaf piece is a chunk range AND a subset of the rays: a piece of one 64-triangle chunk for 64 rays is ~3,700 instructions, so
//     pieces that cannot be cut by chunks any more are cut by rays (up to four parts of the lane mask).
//   * The task launch runs the same walk from a task: kind 1 = a group entry (the children of an inner node) with its ray mask,
//     kind 2 = chunks [c_begin, c_end) of a leaf.  It hands nothing on.
#define RT_PEND_SLOTS 32
struct ConeTasks {
    ContTask *tasks;               // nullptr / budget 0: keep everything inline
    uint32_t *count;
    uint32_t cap, budget, target, group_budget, unit;
    uint32_t start_kind;           // 0: the root; 1: group entry `start_node`; 2: leaf `start_node`, chunks [start_cb, start_ce)
    uint32_t start_node, start_cb, start_ce;
    unsigned long long start_mask;
};
struct ConeLds {
    uint32_t *pnode;               // per-wave pending list (RT_PEND_SLOTS): node / group entry ...
    unsigned long long *pmask;     // ... ray mask ...
    uint32_t *pinfo;               // ... bit 31: group entry; bit 30: to be processed inline (queue full); low bits: first chunk of the inline part
    uint32_t *lcb, *lce;           // per-wave leaf list: chunk range of the visit (lcb = 0xffffffff: a fresh leaf, all chunks, may still be handed away)
};

// estimated cost of one leaf visit in VALU instructions and the lane mapping it would take (leaf_visit decides the same way)
__device__ __forceinline__ uint32_t leaf_cost(const uint32_t cnt, const unsigned long long live, const bool staged) {
    const uint32_t nchunk = (cnt + 63u) >> 6;
    const uint32_t tri_cost = nchunk * RT_COST_CHUNK_TEST + static_cast<uint32_t>(__popcll(live)) * ((nchunk + 2u) / 3u) * RT_COST_TRI_MODE;
    const bool tri_mode = (!staged && cnt > RT_SCALAR_LEAF_MAX) || tri_cost < cnt * RT_COST_RAY_MODE;
    return tri_mode ? tri_cost : cnt * RT_COST_RAY_MODE;
}

template <bool ANY>
__device__ __forceinline__ void cone_walk(const DNode *__restrict__ nodes, const TriRec *__restrict__ tris, const ChunkBound *__restrict__ chunks,
                                          const uint32_t *__restrict__ leaf_chunk0, const WaveStack stk, const ShaftLds sl, const ConeLds cl, const int lane,
                                          const WalkCtl &wc, const ConeTasks &TQ, const DNode &root, const bool in_root, const RayLane &R, const float bx, const float by,
                                          const float bz, const float brx, const float bry, const float brz, const ConeCtl CC, float &best_t, int &best_f, bool &occluded) {
    const float ox = R.ox, oy = R.oy, oz = R.oz;
    unsigned long long m0 = __ballot(in_root);
    if (TQ.start_kind != 0u) m0 &= TQ.start_mask;
    if (m0 == 0ull) return;
    const int tk = lane & 7, tc = lane >> 3;
    int sp = 0, nleaf = 0, npend = 0;
    uint32_t groups_done = 0u;
    bool hand_away = TQ.tasks != nullptr && (TQ.budget != 0u || TQ.group_budget != 0u);
    uint32_t cnt_unused = 0u, sig_unused = 0u;
    if (TQ.start_kind == 2u || (TQ.start_kind == 0u && (root.count_flags & RT_NODE_LEAF))) {
        if (lane == 0) {
            cl.lcb[0] = TQ.start_kind == 2u ? TQ.start_cb : 0xffffffffu; cl.lce[0] = TQ.start_kind == 2u ? TQ.start_ce : 0xffffffffu;
            sl.lnode[0] = TQ.start_kind == 2u ? TQ.start_node : 0u; sl.lmask[0] = m0;
        }
        nleaf = 1;
    } else {
        if (lane == 0) { stk.node[0] = TQ.start_kind == 1u ? TQ.start_node : (root.first | ((root.count_flags & 0xfu) << 28)); stk.mask[0] = m0; }
        sp = 1;
    }
    for (;;) {
        if (nleaf > 0) {
            // the leaves the last group found (ONE inlined copy of the leaf code)
            RT_PH(wc, 3);
            for (int k = 0; k < nleaf; ++k) {
                __builtin_amdgcn_wave_barrier();
                const uint32_t li = uniform_u32(sl.lnode[k]);
                const unsigned long long lm = uniform_u64(sl.lmask[k]);
                const uint32_t cb = uniform_u32(cl.lcb[k]), ce = uniform_u32(cl.lce[k]);
                bool mine = ((lm >> lane) & 1ull) != 0ull;
                if (ANY) mine = mine && !occluded;
                const unsigned long long live = __ballot(mine);
                if (live == 0ull) continue;
                DNode nd;
                if (li < sl.n_lds) { nd.first = sl.nodes[li].first; nd.count_flags = sl.nodes[li].count_flags; nd.pad[0] = sl.nodes[li].pad[0]; }
                else { nd.first = nodes[li].first; nd.count_flags = nodes[li].count_flags; nd.pad[0] = nodes[li].pad[0]; }
                nd.first = uniform_u32(nd.first); nd.count_flags = uniform_u32(nd.count_flags); nd.pad[0] = uniform_u32(nd.pad[0]);
                const bool fresh = cb == 0xffffffffu;
                if (fresh && hand_away && TQ.budget != 0u && npend < RT_PEND_SLOTS && leaf_cost(nd.count_flags & 0x7fffffffu, live, false) > TQ.budget) {
                    if (lane == 0) { cl.pnode[npend] = li; cl.pmask[npend] = live; cl.pinfo[npend] = 0u; }
                    ++npend;
                    continue;
                }
                WalkCtl wl = wc;
                wl.budget = 0u; wl.tasks = nullptr;
                if (!fresh) { wl.resume = true; wl.c_begin = cb; wl.c_end = ce; }
                leaf_visit<ANY, false, false>(nd, li, tris, chunks, leaf_chunk0, stk, lane, wl, R, live, mine, best_t, best_f, occluded, cnt_unused, sig_unused);
            }
            nleaf = 0;
            continue;
        }
        if (npend > 0 && !hand_away) {
            // (the task queue was full: the pending work comes back, as many leaves at a time as the leaf list holds)
            __builtin_amdgcn_wave_barrier();
            while (npend > 0 && nleaf < RT_LEAF_SLOTS) {
                --npend;
                const uint32_t pn = uniform_u32(cl.pnode[npend]), pi = uniform_u32(cl.pinfo[npend]);
                const unsigned long long pm = uniform_u64(cl.pmask[npend]);
                if (pi & 0x80000000u) { if (lane == 0) { stk.node[sp] = pn; stk.mask[sp] = pm; } ++sp; }
                else { if (lane == 0) { sl.lnode[nleaf] = pn; sl.lmask[nleaf] = pm; cl.lcb[nleaf] = pi & 0x3fffffffu; cl.lce[nleaf] = 0xffffffffu; } ++nleaf; }
            }
            continue;
        }
        // a unit that has popped its share of groups hands the rest of its stack away
        if (hand_away && TQ.group_budget != 0u && sp > 0 && groups_done >= TQ.group_budget && npend + sp <= RT_PEND_SLOTS) {
            __builtin_amdgcn_wave_barrier();
            for (int k = 0; k < sp; ++k) {
                const uint32_t ent = uniform_u32(stk.node[k]);
                const unsigned long long em = uniform_u64(stk.mask[k]);
                if (lane == 0) { cl.pnode[npend] = ent; cl.pmask[npend] = em; cl.pinfo[npend] = 0x80000000u; }
                ++npend;
            }
            sp = 0;
        }
        if (sp == 0) {
            if (npend == 0) break;
            // ---- ONE reservation for everything this unit hands away.  Lane e prices pending entry e.
            __builtin_amdgcn_wave_barrier();
            const bool mine_e = lane < npend;
            const uint32_t e_node = mine_e ? cl.pnode[lane] : 0u, e_info = mine_e ? cl.pinfo[lane] : 0u;
            const unsigned long long e_mask = mine_e ? cl.pmask[lane] : 0ull;
            const bool e_group = (e_info & 0x80000000u) != 0u;
            uint32_t e_cnt = 0u;
            if (mine_e && !e_group) e_cnt = (e_node < sl.n_lds ? sl.nodes[e_node].count_flags : nodes[e_node].count_flags) & 0x7fffffffu;
            const uint32_t e_nchunk = (e_cnt + 63u) >> 6;
            const uint32_t e_rays = static_cast<uint32_t>(__popcll(e_mask));
            uint32_t e_npc = 1u, e_nrs = 1u;                      // chunk pieces x ray parts
            if (mine_e && !e_group) {
                const uint32_t est = leaf_cost(e_cnt, e_mask, false);
                uint32_t ntask = (est + TQ.target - 1u) / TQ.target;
                if (ntask < 1u) ntask = 1u;
                e_npc = ntask < e_nchunk ? ntask : e_nchunk;
                e_nrs = (ntask + e_npc - 1u) / e_npc;
                if (e_nrs > 4u) e_nrs = 4u;
                if (e_nrs > e_rays) e_nrs = e_rays;
                if (e_nrs < 1u) e_nrs = 1u;
            }
            const uint32_t e_pieces = mine_e ? e_npc * e_nrs : 0u;
            uint32_t incl = e_pieces;
            for (int d = 1; d < RT_PEND_SLOTS; d <<= 1) {
                const uint32_t t = __shfl_up(incl, d, 64);
                if (lane >= d) incl += t;
            }
            const uint32_t total = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), RT_PEND_SLOTS - 1));
            uint32_t base = 0u;
            if (lane == 0) base = atomicAdd(TQ.count, total);
            base = uniform_u32(base);
            RT_DBG(wc, lane, 2, 1u);
            // (the counter only grows, the consumer clamps it: what falls past the end of the queue is processed here, and the unit stops handing work away)
            const uint32_t room = base >= TQ.cap ? 0u : TQ.cap - base;
            int kept = 0;
            for (int e = 0; e < npend; ++e) {
                const uint32_t off = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl - e_pieces), e));
                const uint32_t pieces = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(e_pieces), e));
                const uint32_t npc = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(e_npc), e));
                const uint32_t nrs = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(e_nrs), e));
                const uint32_t nchunk = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(e_nchunk), e));
                const uint32_t node = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(e_node), e));
                const bool group = __builtin_amdgcn_readlane(static_cast<int>(e_group ? 1 : 0), e) != 0;
                const unsigned long long M = uniform_u64(cl.pmask[e]);
                const uint32_t fit = off >= room ? 0u : (pieces < room - off ? pieces : room - off);
                // the ray parts of this entry's mask: lanes of equal rank share a part
                const bool in_m = ((M >> lane) & 1ull) != 0ull;
                const uint32_t rank = lanes_below(M), pc = static_cast<uint32_t>(__popcll(M));
                const uint32_t part = (rank * nrs) / (pc ? pc : 1u);
                unsigned long long sub[4];
#pragma unroll
                for (uint32_t r = 0; r < 4u; ++r) sub[r] = __ballot(in_m && part == r);
                for (uint32_t i = static_cast<uint32_t>(lane); i < fit; i += 64u) {
                    const uint32_t ci = i / nrs, rp = i - ci * nrs;
                    ContTask t;
                    t.unit = TQ.unit; t.node = node;
                    t.mask = nrs == 1u ? M : (rp == 0u ? sub[0] : (rp == 1u ? sub[1] : (rp == 2u ? sub[2] : sub[3])));
                    t.c_begin = group ? 0u : static_cast<uint32_t>(static_cast<unsigned long long>(nchunk) * ci / npc);
                    t.c_end = group ? 0u : static_cast<uint32_t>(static_cast<unsigned long long>(nchunk) * (ci + 1u) / npc);
                    t.pad0 = group ? 1u : 2u; t.pad1 = 0u;
                    TQ.tasks[base + off + i] = t;
                }
                if (fit < pieces) {
                    // inline from the first chunk piece that did not go out whole (its earlier ray parts are simply done twice: the results merge)
                    const uint32_t cb0 = group ? 0u : static_cast<uint32_t>(static_cast<unsigned long long>(nchunk) * (fit / nrs) / npc);
                    if (lane == 0) { cl.pnode[kept] = node; cl.pmask[kept] = M; cl.pinfo[kept] = (group ? 0x80000000u : 0u) | 0x40000000u | cb0; }
                    ++kept;
                }
            }
            npend = kept;
            if (kept > 0) hand_away = false;
            continue;
        }
        --sp;
        ++groups_done;
        RT_PH(wc, 1);
        __builtin_amdgcn_wave_barrier();
        const uint32_t ent = uniform_u32(stk.node[sp]);
        unsigned long long gm = uniform_u64(stk.mask[sp]);
        if (ANY) gm &= ~__ballot(occluded);
        if (gm == 0ull) continue;
        const uint32_t base = ent & 0x0fffffffu, gcnt = ent >> 28;
        // lane = (child tc, test tk): the child's record, then this lane's separating test on its content box and on its own box
        const uint32_t ci = base + (static_cast<uint32_t>(tc) < gcnt ? static_cast<uint32_t>(tc) : 0u);
        DNode ch;
        const bool resident = base + gcnt <= sl.n_lds;
        if (resident) ch = sl.nodes[ci];
        else ch = nodes[ci];
        unsigned long long surv;
        if (CC.have) {
            ShaftCtl SC{CC.pad, CC.node_ok};
            const ShaftLanes SL = shaft_lanes_load(sl.shaft, tk, SC);
            bool c_near, c_far, n_near, n_far;
            shaft_lane_test(SL, tk, ch.clo[0] - SL.pad, ch.clo[1] - SL.pad, ch.clo[2] - SL.pad, ch.chi[0] + SL.pad, ch.chi[1] + SL.pad, ch.chi[2] + SL.pad, c_near, c_far);
            shaft_lane_test(SL, tk, ch.bmin[0] - SL.pad, ch.bmin[1] - SL.pad, ch.bmin[2] - SL.pad, ch.bmax[0] + SL.pad, ch.bmax[1] + SL.pad, ch.bmax[2] + SL.pad, n_near, n_far);
            // content: the planes, and the near box when the counted hits lie before the targets; own box: the planes only (the rays run on
            // behind their targets, boxIntersect has no upper bound)
            const unsigned long long b_c = __ballot(c_near && (tk < 6 || CC.box) && ch.pad[1] == 0u), b_n = __ballot(n_near && tk < 6);
            const bool culled = (CC.node_ok && ballot_byte_any(b_n, lane)) || ballot_byte_any(b_c, lane);
            surv = __ballot(static_cast<uint32_t>(lane) < gcnt && !culled);       // bit j: child j survives
        } else {
            surv = gcnt >= 64u ? ~0ull : ((1ull << gcnt) - 1ull);
        }
        RT_PROF_ADD(lane, 88, 1); RT_PROF_ADD(lane, 89, gcnt); RT_PROF_ADD(lane, 90, __popcll(surv));
        RT_DBG(wc, lane, 0, 1u);
        RT_PH(wc, 2);
        // the survivors, two at a time: the per-ray tests of two children are independent chains (content slab test, verified slab test of the
        // reference's box) that the wave runs interleaved -- a lone wave waits out every dependent instruction of a single chain
        const bool gl = ((gm >> lane) & 1ull) != 0ull && !(ANY && occluded);
        auto ray_test = [&](const DNode &nd) -> bool {
            bool h = gl;
            if (nd.pad[1] == 0u) {   // per-ray content test (as packet_walk): no countable point of the ray inside the subtree's content box
                const float t0x = (nd.clo[0] - R.slab_pad - ox) * R.idx, t1x = (nd.chi[0] + R.slab_pad - ox) * R.idx;
                const float t0y = (nd.clo[1] - R.slab_pad - oy) * R.idy, t1y = (nd.chi[1] + R.slab_pad - oy) * R.idy;
                const float t0z = (nd.clo[2] - R.slab_pad - oz) * R.idz, t1z = (nd.chi[2] + R.slab_pad - oz) * R.idz;
                const float tin = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z));
                const float tout = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
                const bool miss = (tin > tout) || (tout < -1e-3f) ||
                                  (ANY ? (tin > 0.981f) : (tin > best_t + 1e-3f * (1.0f + fabsf(best_t))));
                h = h && !miss;
            }
            return h && box_hit_verified(nd.bmin, ox, oy, oz, bx, by, bz, brx, bry, brz);     // BoundingBox::boxIntersect, exact
        };
        auto enter = [&](const DNode &nd, const uint32_t cj, const unsigned long long hm) {
            if (hm == 0ull) return;
            RT_PROF_ADD(lane, 91, 1); RT_PROF_ADD(lane, 75, __popcll(hm));
            if (nd.count_flags & RT_NODE_LEAF) {
                if ((nd.count_flags & 0x7fffffffu) == 0u) return;
                if (lane == 0) { sl.lnode[nleaf] = cj; sl.lmask[nleaf] = hm; cl.lcb[nleaf] = 0xffffffffu; cl.lce[nleaf] = 0xffffffffu; }      // (at most 8 per group; the list is emptied before the next group)
                ++nleaf;
            } else {
                if ((nd.count_flags & 0xfu) == 0u) return;               // a "lost" node: no children
                if (lane == 0) { stk.node[sp] = nd.first | ((nd.count_flags & 0xfu) << 28); stk.mask[sp] = hm; }
                ++sp;
            }
        };
        while (surv != 0ull) {
            const int j0 = static_cast<int>(__builtin_ctzll(surv));
            surv &= surv - 1ull;
            const bool two = surv != 0ull;
            const int j1 = two ? static_cast<int>(__builtin_ctzll(surv)) : j0;
            if (two) surv &= surv - 1ull;
            const DNode na = resident ? sl.nodes[base + static_cast<uint32_t>(j0)] : node_from_lane(ch, 8 * j0);
            const DNode nb = resident ? sl.nodes[base + static_cast<uint32_t>(j1)] : node_from_lane(ch, 8 * j1);
            RT_PROF_ADD(lane, 74, (two ? 2 : 1) * __popcll(__ballot(gl))); RT_PROF_ADD(lane, 4, two ? 2 : 1);
            const bool ha = ray_test(na), hb = ray_test(nb);
            const unsigned long long hma = __ballot(ha), hmb = two ? __ballot(hb) : 0ull;
            enter(na, base + static_cast<uint32_t>(j0), hma);
            enter(nb, base + static_cast<uint32_t>(j1), hmb);
        }
    }
    RT_PH(wc, 0);
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// ---- sharded compaction lists (rt_device.hpp, RT_LIST_SHARDS) ----
// Lane s (< RT_LIST_SHARDS) of a ShardMap holds shard s: its element count and where its work units start in the dense
// numbering [0, total) of the consumer's work (tiles of 64 elements, or shadow units).
struct ShardMap {
    uint32_t start, next, cnt;   // per lane
    uint32_t total;              // uniform
};
__device__ __forceinline__ ShardMap shard_map(const uint32_t *counters, const int lane, const uint32_t cap, const uint32_t mult, const uint32_t gran) {
    uint32_t c = lane < RT_LIST_SHARDS ? counters[lane * 16] : 0u;
    if (c > cap) c = cap;
    const uint32_t w = static_cast<uint32_t>((static_cast<unsigned long long>(c) * mult + gran - 1u) / gran);
    uint32_t incl = w;
    for (int d = 1; d < RT_LIST_SHARDS; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    ShardMap m;
    m.cnt = c;
    m.start = incl - w;
    m.next = lane < RT_LIST_SHARDS ? incl : 0xffffffffu;
    m.total = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), RT_LIST_SHARDS - 1));
    return m;
}
// dense work number w (wave-uniform, < total) -> shard, work number inside the shard, the shard's element count
__device__ __forceinline__ void shard_find(const ShardMap &m, const uint32_t w, uint32_t &shard, uint32_t &local, uint32_t &cnt) {
    shard = static_cast<uint32_t>(__popcll(__ballot(m.next <= w)));
    if (shard >= RT_LIST_SHARDS) shard = RT_LIST_SHARDS - 1;     // never taken for w < total
    local = w - static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(m.start), static_cast<int>(shard)));
    cnt = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(m.cnt), static_cast<int>(shard)));
}
// one returning atomic per wave on the counter of the list shard picked by the producing tile / group number
__device__ __forceinline__ uint32_t shard_reserve(uint32_t *counters, uint32_t *overflow, const uint32_t producer, const uint32_t n, const uint32_t cap,
                                                  const int lane, bool &fits) {
    const uint32_t sh = producer & (RT_LIST_SHARDS - 1u);
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&counters[sh * 16u], n);
    base = uniform_u32(base);
    fits = base + n <= cap;       // always true: the capacity is derived from the tile count (rt_capi.cpp, list_cap)
    if (!fits && lane == 0) atomicOr(overflow, 1u);      // ... and if it ever is not, the frame is reported as failed, not silently short
    return sh * cap + base;
}

// arealight::getPointLights / createSpherePoint (arealight.hpp:15-25, flyscene.cpp:956-972): sample s of light p
// N > 64 samples: a k_shadow pass holds 64 of them.  When both grid sides are multiples of 8 a pass is an 8x8 BLOCK of the grid
// (pass p = block (p / (vsteps/8), p % (vsteps/8)), lane l = cell (l / 8, l % 8) of it) instead of 64 consecutive samples (a
// 4 x 16 strip of a 16 x 16 light): the 64 segments of a unit form a tighter bundle (-6 % on cfg4's k_shadow).  The visibility
// word of pass p keeps bit l for lane l, so k_shade looks sample (i, j) up in word (i/8)*(vsteps/8) + j/8, bit (i%8)*8 + j%8.
__device__ __forceinline__ bool sample_blocks(const DLights &L) {
    return L.mode == RT_LIGHT_AREA && L.n_samples > 64 && (L.usteps & 7) == 0 && (L.vsteps & 7) == 0;
}

// The sample grid of one light: sample (i, j) = ((i + 0.5) * cx, (j + 0.5) * cy, z).  cx and cy each hold a float division, so a
// caller that needs several samples of the same light (k_shadow: the lane's sample + the two corners of the sample box) builds
// the grid once; (i, j) = (s / vsteps, s % vsteps) are passed as the floats i + 0.5 and j + 0.5, which the callers keep out of
// their hot loops (an integer division by a run-time value costs ~25 VALU instructions).
struct LightGrid { float cx, cy, z, px, py; bool point; };
__device__ __forceinline__ LightGrid light_grid(const DLights &L, const float px, const float py, const float pz) {
    LightGrid g;
    g.point = L.mode == RT_LIGHT_POINT;
    g.px = px; g.py = py;
    const float ux = px + L.len_x * 1.0f;     // uvec = corner + lengthX * (1,0,0)
    const float uz = pz + L.len_x * 0.0f;
    const float vy = py + L.len_y * 1.0f;     // vvec = corner + lengthY * (0,1,0)
    g.cx = ux / static_cast<float>(L.usteps);
    g.cy = vy / static_cast<float>(L.vsteps);
    g.z = g.point ? pz : uz;
    return g;
}
__device__ __forceinline__ void grid_sample(const LightGrid &g, const float fi, const float fj, float &sx, float &sy, float &sz) {
    sx = g.point ? g.px : fi * g.cx;
    sy = g.point ? g.py : fj * g.cy;
    sz = g.z;
}
__device__ __forceinline__ void light_sample_ij(const DLights &L, const float px, const float py, const float pz, const float fi, const float fj,
                                                float &sx, float &sy, float &sz) {
    grid_sample(light_grid(L, px, py, pz), fi, fj, sx, sy, sz);
}
// RT_LIGHT_SPHERE (createSpherePoint's third branch, flyscene.cpp:974-995): sample s of a light at p = offsets[s] + p, its samples lie in
// p + obox (float addition is monotone, so the box of the offsets bounds the samples exactly)
__device__ __forceinline__ void sphere_sample(const DLights &L, const uint32_t s, const float px, const float py, const float pz, float &sx, float &sy, float &sz) {
    const uint32_t k = s < static_cast<uint32_t>(L.n_samples) ? s : 0u;
    sx = L.offsets[k * 3u] + px; sy = L.offsets[k * 3u + 1u] + py; sz = L.offsets[k * 3u + 2u] + pz;
}
__device__ __forceinline__ void sphere_box(const DLights &L, const float px, const float py, const float pz, float &x0, float &y0, float &z0, float &x1, float &y1,
                                           float &z1) {
    x0 = L.obox[0] + px; y0 = L.obox[1] + py; z0 = L.obox[2] + pz; x1 = L.obox[3] + px; y1 = L.obox[4] + py; z1 = L.obox[5] + pz;
}
__device__ __forceinline__ void light_sample(const DLights &L, const float px, const float py, const float pz, const int s,
                                             float &sx, float &sy, float &sz) {
    const int i = s / L.vsteps, j = s - i * L.vsteps;
    light_sample_ij(L, px, py, pz, static_cast<float>(i) + 0.5f, static_cast<float>(j) + 0.5f, sx, sy, sz);
}

// Camera::screenToWorld (camera.hpp:155-173): raster -> [-1,1] in double, cast, perspective scale, inverse view.  One definition for
// the fused k_trace, the staged k_stage and the probe kernel (rt_primary_points).
// the same point for the pixel (x0 + (lane & 7), y0 + (lane >> 3)) of an 8 x 8 tile, with the tile's sixteen double divisions done ONCE: lane c < 8
// evaluates the column term of x0 + c, lane 8 + r the row term of row ys[r] (the caller passes each lane ITS candidate: the row of lane 8 + r is
// the y of the lanes r * 8 .. r * 8 + 7), every lane then fetches its two terms.  Same double operations on the same operands as screen_point --
// 2 divisions per lane become 1 (a double division is ~30 half-rate instructions: a third of what a sky tile costs).
__device__ __forceinline__ void screen_point_tile(const DCam &cam, const int lane, const int x0, const int y_of_row_lane, float &sx, float &sy, float &sz) {
    const bool col = lane < 8;
    const float f = col ? static_cast<float>(x0 + lane) : static_cast<float>(y_of_row_lane);
    const double q = 2.0 * static_cast<double>(f - (col ? cam.vp[0] : cam.vp[1])) / static_cast<double>(col ? cam.vp[2] : cam.vp[3]);
    const float term = col ? static_cast<float>(q - 1.0) : static_cast<float>(1.0 - q);
    float n0 = __shfl(term, lane & 7, 64);
    float n1 = __shfl(term, 8 + (lane >> 3), 64);
    const float n2 = -1.0f;
    n0 = n0 * cam.k0;
    n1 = n1 * cam.k1;
    const float *m = cam.inv_view;
    sx = ((m[0] * n0 + m[1] * n1) + m[2] * n2) + m[3] * 1.0f;
    sy = ((m[4] * n0 + m[5] * n1) + m[6] * n2) + m[7] * 1.0f;
    sz = ((m[8] * n0 + m[9] * n1) + m[10] * n2) + m[11] * 1.0f;
}
__device__ __forceinline__ void screen_point(const DCam &cam, const int x, const int y, float &sx, float &sy, float &sz) {
    const float fi = static_cast<float>(x), fj = static_cast<float>(y);
    float n0 = static_cast<float>(2.0 * static_cast<double>(fi - cam.vp[0]) / static_cast<double>(cam.vp[2]) - 1.0);
    float n1 = static_cast<float>(1.0 - 2.0 * static_cast<double>(fj - cam.vp[1]) / static_cast<double>(cam.vp[3]));
    const float n2 = -1.0f;
    n0 = n0 * cam.k0;
    n1 = n1 * cam.k1;
    const float *m = cam.inv_view;
    sx = ((m[0] * n0 + m[1] * n1) + m[2] * n2) + m[3] * 1.0f;
    sy = ((m[4] * n0 + m[5] * n1) + m[6] * n2) + m[7] * 1.0f;
    sz = ((m[8] * n0 + m[9] * n1) + m[10] * n2) + m[11] * 1.0f;
}

// ======================================================================================================
// K1: closest hit + light-centre visibility.  PRIMARY: fused primary-ray generation (Camera::screenToWorld)
// and root-AABB cull of raytraceScene's serial loop (flyscene.cpp:573-598); otherwise reads compacted rays.
// ======================================================================================================
template <bool PRIMARY, bool COUNT, bool FLAT>
__global__ __launch_bounds__(RT_WAVES * 64) void k_trace(const DNode *__restrict__ nodes, const TriRec *__restrict__ tris,
                                                          const ChunkBound *__restrict__ chunks, const uint32_t *__restrict__ leaf_chunk0,
                                                          const DScene S, const DCam *__restrict__ camp, const DLights L, const DFrame F,
                                                          const int level, const int ctr_slot,
                                                          const RayItem *__restrict__ rays_in, ShadeItem *__restrict__ items,
                                                          Control *__restrict__ ctl, float4 *__restrict__ rec,
                                                          int32_t *__restrict__ out_hit, float *__restrict__ out_t) {
    __shared__ uint4 s_stage[FLAT ? 1 : RT_WAVES * RT_STAGE_TRIS * 5];        // flat scenes need neither staging buffer nor stack
    __shared__ unsigned long long s_mask[FLAT ? 1 : RT_WAVES * RT_STACK];
    __shared__ uint32_t s_node[FLAT ? 1 : RT_WAVES * RT_STACK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const WaveStack stk{s_node + (FLAT ? 0 : wave * RT_STACK), s_mask + (FLAT ? 0 : wave * RT_STACK), s_stage + (FLAT ? 0 : wave * RT_STAGE_TRIS * 5)};
    ShardMap rmap{0u, 0u, 0u, 0u};
    if (!PRIMARY) rmap = shard_map(ctl->n_rays[level], lane, 0xffffffffu, 1u, 64u);
    const uint32_t ntiles = PRIMARY ? static_cast<uint32_t>(F.tiles_x) * static_cast<uint32_t>(F.tiles_y) : rmap.total;
    const DNode root = nodes[0];
    // the camera lives in device memory so that a captured hipGraph of the frame can be replayed with a new camera
    DCam cam;
    if (PRIMARY) cam = *camp;

    uint32_t c_rays = 0, c_cull = 0, c_centre = 0, c_box = 0, c_ref = 0;
    ShardedQueue q;
    if (F.dyn_trace) q.init(ctl->queue[ctr_slot], ntiles, gridDim.x * RT_WAVES, blockIdx.x, lane);
    else q.init_static(ntiles, gridDim.x * RT_WAVES, uniform_u32(blockIdx.x * RT_WAVES + static_cast<uint32_t>(wave)), lane);
    for (uint32_t tile = 0; q.next(tile);) {
        RT_PROF_ADD(lane, 13, 1);
#ifdef RT_PROFILE_HIST
        const long long prof_t0 = clock64();
        if (lane == 0) { stk.node[RT_STACK - 4] = 0; stk.node[RT_STACK - 3] = 0; stk.node[RT_STACK - 2] = 0; stk.node[RT_STACK - 1] = 0; }
#endif
        bool valid;
        uint32_t pix = 0, lmode = 0;
        float ox, oy, oz, dx, dy, dz, lx = 0.f, ly = 0.f, lz = 0.f;
        bool in_root;
        if (PRIMARY) {
            const int tx = static_cast<int>(tile % static_cast<uint32_t>(F.tiles_x)), ty = static_cast<int>(tile / static_cast<uint32_t>(F.tiles_x));
            const int x = tx * 8 + (lane & 7), lr = ty * 8 + (lane >> 3);
            valid = (x < F.width) && (lr < F.local_rows);
            pix = static_cast<uint32_t>(lr) * static_cast<uint32_t>(F.width) + static_cast<uint32_t>(x);
            float sx, sy, sz;
            {   // (lane 8 + r evaluates the row term of tile row r: the frame row of local row ty * 8 + r)
                const int lr_r = ty * 8 + ((lane - 8) & 7);
                const int y_r = F.row0 + ((lr_r / F.stripe) * F.nranks + F.rank) * F.stripe + (lr_r % F.stripe);
                screen_point_tile(cam, lane, tx * 8, y_r, sx, sy, sz);
            }
            ox = cam.center[0]; oy = cam.center[1]; oz = cam.center[2];
            dx = sx - ox; dy = sy - oy; dz = sz - oz;          // direction = screen - origin (UNNORMALISED), flyscene.cpp:619
            const bool pre = valid && box_hit_verified(root.bmin, ox, oy, oz, dx, dy, dz, __builtin_amdgcn_rcpf(dx), __builtin_amdgcn_rcpf(dy), __builtin_amdgcn_rcpf(dz));   // flyscene.cpp:576
            c_cull += (valid && !pre) ? 1u : 0u;
            in_root = pre;
        } else {
            uint32_t sh, tj, n_in;
            shard_find(rmap, tile, sh, tj, n_in);
            const uint32_t r = tj * 64u + static_cast<uint32_t>(lane);
            valid = r < n_in;
            const RayItem it = rays_in[valid ? sh * F.ray_cap + r : 0u];
            ox = it.ox; oy = it.oy; oz = it.oz; dx = it.dx; dy = it.dy; dz = it.dz;
            lx = it.lx; ly = it.ly; lz = it.lz; lmode = it.lmode; pix = it.pix;
            in_root = valid;
        }
        c_rays += in_root ? 1u : 0u;
        // traceRay: boxIntersect(origin, origin + direction) -- the box-test direction is (o + d) - o (flyscene.cpp:655)
        const float bx = (ox + dx) - ox, by = (oy + dy) - oy, bz = (oz + dz) - oz;
        if (COUNT && in_root) c_box += 1;
        const float brx = __builtin_amdgcn_rcpf(bx), bry = __builtin_amdgcn_rcpf(by), brz = __builtin_amdgcn_rcpf(bz);
        in_root = in_root && box_hit_verified(root.bmin, ox, oy, oz, bx, by, bz, brx, bry, brz);

        float best_t = 3.402823466e+38f;
        int best_f = -1;
        bool dummy = false;
        walk<false, COUNT, FLAT>(root, nodes, tris, chunks, leaf_chunk0, S.extent, stk, lane, walk_plain(), LanePlane{0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, in_root, ox, oy, oz, dx, dy, dz, bx, by, bz, brx, bry, brz, best_t, best_f, dummy, c_box, c_ref);
        const bool hit = valid && (best_f >= 0) && (static_cast<uint32_t>(best_f) < S.n_faces);
        const float hx = ox + best_t * dx, hy = oy + best_t * dy, hz = oz + best_t * dz;   // flyscene.cpp:695

        // lightStrikes(hitPoint, lights): one segment per light CENTRE (flyscene.cpp:700)
        bool lit = false;
        const int nl_lane = lmode ? 1 : L.n_lights;
        const int nl_wave = (__ballot(hit && lmode == 0u) != 0ull) ? L.n_lights : 1;
        if (__ballot(hit) != 0ull) {
            for (int l = 0; l < nl_wave; ++l) {
                const bool act = hit && (l < nl_lane);
                const float px = lmode ? lx : L.pos[l][0], py = lmode ? ly : L.pos[l][1], pz = lmode ? lz : L.pos[l][2];
                const float sdx = hx - px, sdy = hy - py, sdz = hz - pz;      // direction = hitPoint - origin
                c_centre += act ? 1u : 0u;
                if (COUNT && act) c_box += 1;
                const float srx = __builtin_amdgcn_rcpf(sdx), sry = __builtin_amdgcn_rcpf(sdy), srz = __builtin_amdgcn_rcpf(sdz);
                const bool sroot = act && box_hit_verified(root.bmin, px, py, pz, sdx, sdy, sdz, srx, sry, srz);
                float t_unused = 0.f; int f_unused = -1;
                bool occ = false;
                walk<true, COUNT, FLAT>(root, nodes, tris, chunks, leaf_chunk0, S.extent, stk, lane, walk_plain(), LanePlane{0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sroot, px, py, pz, sdx, sdy, sdz, sdx, sdy, sdz, srx, sry, srz, t_unused, f_unused, occ, c_box, c_ref);
                lit = lit || (act && !occ);
            }
        }

        if (valid) {
            if (!hit) rec[pix] = make_float4(1.f, 1.f, 1.f, __uint_as_float(KIND_CONST));          // BACKGROUND
            else if (!lit) rec[pix] = make_float4(0.f, 0.f, 0.f, __uint_as_float(KIND_CONST));     // SHADOW
            if (out_hit) out_hit[pix] = hit ? best_f : -1;
            if (out_t) out_t[pix] = hit ? best_t : -1.0f;
        }
        // compaction: lit hits -> shade list (wave ballot + prefix, one atomic per wave)
        const unsigned long long lm = __ballot(lit);
        if (lm != 0ull) {
            bool fits;
            const uint32_t base = shard_reserve(ctl->n_items[level], &ctl->overflow, tile, static_cast<uint32_t>(__popcll(lm)), F.item_cap, lane, fits);
            if (lit && fits) {
                ShadeItem o;
                o.ox = ox; o.oy = oy; o.oz = oz; o.dx = dx; o.dy = dy; o.dz = dz;
                o.lx = lx; o.ly = ly; o.lz = lz; o.lmode = lmode; o.pix = pix; o.face = best_f; o.t = best_t;
                o.pad0 = o.pad1 = o.pad2 = 0u;
                items[base + lanes_below(lm)] = o;
            }
        }
#ifdef RT_PROFILE_HIST
        {   // per-tile cycle histogram of k_trace: prof[40 + log2(cycles)] (capped at 2^23), max prof[38], sum prof[39]
            const unsigned long long dt = static_cast<unsigned long long>(clock64() - prof_t0);
            if (lane == 0 && g_prof) {
                if (atomicMax(&g_prof[38], dt) < dt) {      // new slowest tile: remember what it did
                    g_prof[56] = tile; g_prof[57] = stk.node[RT_STACK - 4]; g_prof[58] = stk.node[RT_STACK - 3];
                    g_prof[59] = stk.node[RT_STACK - 2]; g_prof[60] = stk.node[RT_STACK - 1];
                }
                atomicAdd(&g_prof[39], dt);
                int b = 63 - __builtin_clzll(dt | 1ull); if (b > 23) b = 23;
                atomicAdd(&g_prof[40 + b], 1ull);
            }
        }
#endif
    }
    // per-wave counters -> control block
    c_rays = wave_sum(c_rays); c_cull = wave_sum(c_cull); c_centre = wave_sum(c_centre);
    if (COUNT) { c_box = wave_sum(c_box); c_ref = wave_sum(c_ref); }
    if (lane == 0) {
        if (c_rays) atomicAdd((PRIMARY || level == 0) ? &ctl->stat[blockIdx.x & (RT_STAT_SHARDS - 1)][ST_RAYS_PRIMARY] : &ctl->stat[blockIdx.x & (RT_STAT_SHARDS - 1)][ST_RAYS_BOUNCE], static_cast<unsigned long long>(c_rays));
        if (c_cull) atomicAdd(&ctl->stat[blockIdx.x & (RT_STAT_SHARDS - 1)][ST_PIXELS_CULLED], static_cast<unsigned long long>(c_cull));
        if (c_centre) atomicAdd(&ctl->stat[blockIdx.x & (RT_STAT_SHARDS - 1)][ST_RAYS_CENTRE], static_cast<unsigned long long>(c_centre));
        if (COUNT) {
            if (c_box) atomicAdd(&ctl->stat[blockIdx.x & (RT_STAT_SHARDS - 1)][ST_BOX_TESTS], static_cast<unsigned long long>(c_box));
            if (c_ref) atomicAdd(&ctl->stat[blockIdx.x & (RT_STAT_SHARDS - 1)][ST_LEAF_TRI_REFS], static_cast<unsigned long long>(c_ref));
        }
    }
}

// ======================================================================================================
// K1 for TREE scenes, split in three so that every traversal can hand work away (measured on dodgeColorTest: ONE 8x8 tile
// cost 2.45 M cycles -- the whole fused kernel -- while the average wave had 0.18 M cycles of work):
//   STAGE 0  closest hit        -> best[tile*64+lane] = (t bits << 32 | face), merged by continuations with atomicMin
//                                  (minimum t, ties to the lowest face id: exactly the reference's strict '<' over the
//                                  ascending std::set, flyscene.cpp:675-683)
//   STAGE 1  light-centre rays  -> lit[tile*lslots+l] = lane mask of visible centres, continuations clear bits (atomicAnd)
//   STAGE 2  finish             -> BACKGROUND / SHADOW records, out_hit/out_t, compaction of lit hits
// The fused k_trace above remains the path of flat scenes (cube.obj).
// ======================================================================================================

struct TileRay {
    bool valid, pre;
    uint32_t pix, lmode;
    float ox, oy, oz, dx, dy, dz, lx, ly, lz;
};

template <bool PRIMARY>
__device__ __forceinline__ TileRay tile_ray(const uint32_t tile, const int lane, const DFrame &F, const DCam &cam, const DNode &root,
                                            const RayItem *__restrict__ rays_in, const ShardMap &rmap) {
    TileRay r;
    r.lx = r.ly = r.lz = 0.f; r.lmode = 0u;
    if (PRIMARY) {
        const int tx = static_cast<int>(tile % static_cast<uint32_t>(F.tiles_x)), ty = static_cast<int>(tile / static_cast<uint32_t>(F.tiles_x));
        const int x = tx * 8 + (lane & 7), lr = ty * 8 + (lane >> 3);
        r.valid = (x < F.width) && (lr < F.local_rows);
        r.pix = static_cast<uint32_t>(lr) * static_cast<uint32_t>(F.width) + static_cast<uint32_t>(x);
        float sx, sy, sz;
        {   // (lane 8 + r evaluates the row term of tile row r: the frame row of local row ty * 8 + r)
            const int lr_r = ty * 8 + ((lane - 8) & 7);
            const int y_r = F.row0 + ((lr_r / F.stripe) * F.nranks + F.rank) * F.stripe + (lr_r % F.stripe);
            screen_point_tile(cam, lane, tx * 8, y_r, sx, sy, sz);
        }
        r.ox = cam.center[0]; r.oy = cam.center[1]; r.oz = cam.center[2];
        r.dx = sx - r.ox; r.dy = sy - r.oy; r.dz = sz - r.oz;          // flyscene.cpp:619
        r.pre = r.valid && box_hit_verified(root.bmin, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, __builtin_amdgcn_rcpf(r.dx),
                                            __builtin_amdgcn_rcpf(r.dy), __builtin_amdgcn_rcpf(r.dz));                 // flyscene.cpp:576
    } else {
        uint32_t sh, tj, n_in;
        shard_find(rmap, tile, sh, tj, n_in);
        const uint32_t k = tj * 64u + static_cast<uint32_t>(lane);
        r.valid = k < n_in;
        const RayItem it = rays_in[r.valid ? sh * F.ray_cap + k : 0u];
        r.ox = it.ox; r.oy = it.oy; r.oz = it.oz; r.dx = it.dx; r.dy = it.dy; r.dz = it.dz;
        r.lx = it.lx; r.ly = it.ly; r.lz = it.lz; r.lmode = it.lmode; r.pix = it.pix;
        r.pre = r.valid;
    }
    return r;
}

#define RT_NO_HIT_KEY 0xffffffffffffffffull

template <bool PRIMARY, bool COUNT, int STAGE, bool CONT>
__global__ __launch_bounds__(RT_WAVES * 64) void k_stage(const DNode *__restrict__ nodes, const TriRec *__restrict__ tris,
                                                          const ChunkBound *__restrict__ chunks, const uint32_t *__restrict__ leaf_chunk0,
                                                          const DScene S, const DCam *__restrict__ camp, const DLights L, const DFrame F,
                                                          const int level, const int lslots,
                                                          const RayItem *__restrict__ rays_in, ShadeItem *__restrict__ items,
                                                          Control *__restrict__ ctl, float4 *__restrict__ rec,
                                                          int32_t *__restrict__ out_hit, float *__restrict__ out_t,
                                                          unsigned long long *best, unsigned long long *lit, const TaskQueues Q) {
    // GROUP (build flag -DRT_GROUP_WALK, default OFF): the two traversal stages of the fast variants take the cone walk above -- groups of
    // children in one round trip, the top of the tree in LDS, common-origin cone tests, one task reservation per unit -- instead of the
    // stack walk.  Measured A/B on one box (round 3, tools/r3_env.sh; trace group per frame): dodgeColorTest.obj 1080p 0.305-0.316 ms against
    // 0.259-0.268 ms for the stack walk, cfg4 3.92 ms (2.45 ms with RT_GROUP_BUDGET=4) against 2.18 ms.  The cone test removes 75 % of the
    // per-ray child tests (29 k of 117 k children survive on dodge) but the launches are not bound by those: their duration IS their longest
    // unit (per-unit records of the RT_UNIT_HIST build: 150 k / 260 k cycles for the two stages against a mean wave lifetime of 28 k / 16 k),
    // a serial chain of ~10-16 node visits and 6-23 leaf visits that runs at 15-20 cycles per instruction in a wave of its own, and the cone
    // walk's extra state (138 VGPRs: 3 waves per SIMD) makes that chain longer, not shorter.  What did help both walks is in Control::n_task_tr.
#ifdef RT_GROUP_WALK
    constexpr bool GROUP = !COUNT && STAGE < 2;
#else
    constexpr bool GROUP = false;
#endif
    constexpr bool CONE = GROUP || (CONT && STAGE < 2 && !COUNT);
    __shared__ uint4 s_stage[GROUP ? 1 : RT_WAVES * RT_STAGE_TRIS * 5];
    __shared__ unsigned long long s_mask[RT_WAVES * RT_STACK];
    __shared__ uint32_t s_node[RT_WAVES * RT_STACK];
    __shared__ float4 s_cone[CONE ? RT_WAVES * RT_SHAFT_TRI_REC : 1];
    __shared__ uint4 s_top[GROUP ? RT_LDS_NODES * 4 : 1];           // the top of the octree: first RT_LDS_NODES DNodes (breadth-first order)
    __shared__ unsigned long long s_lmask[GROUP ? RT_WAVES * RT_LEAF_SLOTS : 1];
    __shared__ uint32_t s_lnode[GROUP ? RT_WAVES * RT_LEAF_SLOTS : 1];
    __shared__ uint32_t s_lcb[GROUP ? RT_WAVES * RT_LEAF_SLOTS : 1], s_lce[GROUP ? RT_WAVES * RT_LEAF_SLOTS : 1];
    __shared__ unsigned long long s_pmask[GROUP ? RT_WAVES * RT_PEND_SLOTS : 1];
    __shared__ uint32_t s_pnode[GROUP ? RT_WAVES * RT_PEND_SLOTS : 1], s_pinfo[GROUP ? RT_WAVES * RT_PEND_SLOTS : 1];
    __shared__ float4 s_shaft[GROUP ? RT_WAVES * 16 : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const WaveStack stk{s_node + wave * RT_STACK, s_mask + wave * RT_STACK, s_stage + (GROUP ? 0 : wave * RT_STAGE_TRIS * 5)};
    float4 *const cone_rec = s_cone + (CONE ? wave * RT_SHAFT_TRI_REC : 0);
    ShardMap rmap{0u, 0u, 0u, 0u};
    if (!PRIMARY) rmap = shard_map(ctl->n_rays[level], lane, 0xffffffffu, 1u, 64u);
    const uint32_t ntiles = PRIMARY ? static_cast<uint32_t>(F.tiles_x) * static_cast<uint32_t>(F.tiles_y) : rmap.total;

    uint32_t n_units = STAGE == 1 ? ntiles * static_cast<uint32_t>(lslots) : ntiles;
    ShardMap tmap{0u, 0u, 0u, 0u};
    if (CONT) {
        tmap = shard_map(ctl->n_task_tr[level][Q.q_in & 1u], lane, Q.cap / RT_LIST_SHARDS, 1u, 1u);      // producers clamp to the per-shard capacity too
        n_units = tmap.total;
    }
    if (n_units == 0u) return;                    // an empty bounce level / no leaf tasks: leave before any set-up (every wave takes this branch)
    const uint32_t n_lds = GROUP ? (S.n_nodes < RT_LDS_NODES ? S.n_nodes : RT_LDS_NODES) : 0u;
    if (GROUP) {
        const uint4 *__restrict__ src = reinterpret_cast<const uint4 *>(nodes);
        for (uint32_t i = threadIdx.x; i < n_lds * 4u; i += blockDim.x) s_top[i] = src[i];
        __syncthreads();
    }
    const ShaftLds sl{reinterpret_cast<const DNode *>(s_top), n_lds, s_lnode + (GROUP ? wave * RT_LEAF_SLOTS : 0), s_lmask + (GROUP ? wave * RT_LEAF_SLOTS : 0), cone_rec,
                      s_shaft + (GROUP ? wave * 16 : 0)
#ifdef RT_PROFILE
                      , nullptr
#endif
    };
    const ConeLds cl{s_pnode + (GROUP ? wave * RT_PEND_SLOTS : 0), s_pmask + (GROUP ? wave * RT_PEND_SLOTS : 0), s_pinfo + (GROUP ? wave * RT_PEND_SLOTS : 0),
                     s_lcb + (GROUP ? wave * RT_LEAF_SLOTS : 0), s_lce + (GROUP ? wave * RT_LEAF_SLOTS : 0)};
    (void)sl; (void)cl;
    const DNode root = nodes[0];
    DCam cam;
    if (PRIMARY) cam = *camp;

    uint32_t c_rays = 0, c_cull = 0, c_centre = 0, c_box = 0, c_ref = 0;
    ShardedQueue q;
    q.init_static(n_units, gridDim.x * RT_WAVES, uniform_u32(blockIdx.x * RT_WAVES + static_cast<uint32_t>(wave)), lane);
#ifdef RT_PROFILE
    PhaseClock pclk; pclk.start();
    const unsigned long long wave_t0 = __builtin_amdgcn_s_memrealtime();        // 100 MHz
#endif
#ifdef RT_UNIT_HIST          // (make ab AB_FLAGS=-DRT_UNIT_HIST: the product kernels + two clock reads and one 32-byte record per unit, plain stores; RT_UNIT_DUMP=file)
    __shared__ uint32_t s_dbg[RT_WAVES * 8];
    const unsigned long long hist_t0 = static_cast<unsigned long long>(clock64());      // s_memtime (shader cycles); s_memrealtime serialises chip-wide
    uint32_t hist_units = 0u;
#endif
    for (uint32_t work = 0; q.next(work);) {
        uint32_t unit = work;
        WalkCtl wc = walk_plain();
        RT_PROF_ADD(lane, 13, 1);
#ifdef RT_PROFILE
        if (STAGE < 2 && !COUNT) { wc.pc = &pclk; pclk.to(6); }
#endif
#ifdef RT_UNIT_HIST
        const unsigned long long unit_t0 = static_cast<unsigned long long>(clock64());
        wc.dbg = s_dbg + wave * 8;
        if (lane == 0) for (int k = 0; k < 6; ++k) wc.dbg[k] = 0u;
#endif
        ConeTasks TQ{nullptr, nullptr, 0u, 0u, 1u, 0u, 0u, 0u, 0u, 0u, 0u, 0ull};
        if (CONT) {
            uint32_t tsh, tloc, tn;
            shard_find(tmap, work, tsh, tloc, tn);
            const ContTask task = Q.tasks_in[tsh * (Q.cap / RT_LIST_SHARDS) + tloc];
            unit = uniform_u32(task.unit);
            wc.resume = true;
            wc.start_node = uniform_u32(task.node);
            wc.start_mask = uniform_u64(task.mask);
            wc.c_begin = uniform_u32(task.c_begin);
            wc.c_end = uniform_u32(task.c_end);
            // (cone walk: a group entry or a chunk range of a leaf, for the rays of the task's mask; a task hands nothing on)
            TQ.start_kind = uniform_u32(task.pad0) == 1u ? 1u : 2u;
            TQ.start_node = wc.start_node; TQ.start_mask = wc.start_mask; TQ.start_cb = wc.c_begin; TQ.start_ce = wc.c_end;
            if (GROUP) wc.resume = false;
        }
        if (STAGE < 2 && Q.tasks_out != nullptr && (Q.budget != 0u || Q.group_budget != 0u)) {
            const uint32_t tsh = blockIdx.x & (RT_LIST_SHARDS - 1u), tcap = Q.cap / RT_LIST_SHARDS;      // sharded task queue (Control::n_task_tr)
            if (GROUP) {
                TQ.tasks = Q.tasks_out + tsh * tcap; TQ.count = &ctl->n_task_tr[level][Q.q_out & 1u][tsh * 16u]; TQ.cap = tcap;
                TQ.budget = Q.budget; TQ.target = Q.target ? Q.target : (Q.budget ? Q.budget : 1000u); TQ.group_budget = Q.group_budget; TQ.unit = unit;
            } else {
                wc.budget = Q.budget; wc.unit = unit; wc.tasks = Q.tasks_out + tsh * tcap; wc.task_count = &ctl->n_task_tr[level][Q.q_out & 1u][tsh * 16u]; wc.task_cap = tcap;
                wc.target = Q.target ? Q.target : Q.budget;
            }
        }
        const uint32_t tile = STAGE == 1 ? unit / static_cast<uint32_t>(lslots) : unit;
        const int l = STAGE == 1 ? static_cast<int>(unit - tile * static_cast<uint32_t>(lslots)) : 0;
        const TileRay r = tile_ray<PRIMARY>(tile, lane, F, cam, root, rays_in, rmap);
        const size_t ray_slot = static_cast<size_t>(tile) * 64u + static_cast<size_t>(lane);
        // the packet's cone for the lane = triangle test of its leaves (leaf_visit): common origin (ax, ay, az), box of the targets of the
        // lanes in `on` (exact wave min / max).  Only the leaf-task launches build it: there every unit is a run of 64-triangle chunks of a big
        // leaf (cfg4: closest-hit tasks 0.92 -> 0.68 ms, light-centre tasks 0.55 -> 0.48 ms), while on the walking launches the ~200
        // instructions per tile cost more than the few big leaves they keep inline return (dodge: +7 us on both)
        ConeCtl CC{false, false, false, 0.f};
        auto set_cone = [&](const bool on, const float ax, const float ay, const float az, const float tx, const float ty, const float tz, const bool box) {
            float lx = on ? tx : 3e38f, ly = on ? ty : 3e38f, lz = on ? tz : 3e38f, hx_ = on ? tx : -3e38f, hy_ = on ? ty : -3e38f, hz_ = on ? tz : -3e38f;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                lx = fminf(lx, __shfl_xor(lx, o, 64)); ly = fminf(ly, __shfl_xor(ly, o, 64)); lz = fminf(lz, __shfl_xor(lz, o, 64));
                hx_ = fmaxf(hx_, __shfl_xor(hx_, o, 64)); hy_ = fmaxf(hy_, __shfl_xor(hy_, o, 64)); hz_ = fmaxf(hz_, __shfl_xor(hz_, o, 64));
            }
            const ShaftLanes SLc = make_shaft_lanes(lane, ax, ay, az, lx, ly, lz, hx_, hy_, hz_, S.extent);
            __builtin_amdgcn_wave_barrier();
            shaft_tri_store(cone_rec, lane, SLc, ax, ay, az, lx, ly, lz, hx_, hy_, hz_);
            if (GROUP) shaft_lanes_store(sl.shaft, lane, SLc);
            __builtin_amdgcn_wave_barrier();
            wc.cone = cone_rec; wc.cone_box = box;
            // (every ray of the packet starts at the apex: the group tests pad their boxes by the per-ray slab_pad of that origin)
            CC.have = true; CC.box = box; CC.pad = 4e-4f * (fabsf(ax) + fabsf(ay) + fabsf(az) + S.extent) * 1.001f;
        };
        // no ray of the packet with a zero / non-finite component of its box-test direction (see cone_walk)
        auto dirs_ok = [&](const bool on, const float vx, const float vy, const float vz) -> bool {
            return __ballot(on && !(fabsf(vx) > 0.0f && fabsf(vy) > 0.0f && fabsf(vz) > 0.0f && fabsf(vx) + fabsf(vy) + fabsf(vz) < 3e38f)) == 0ull;
        };

        if (STAGE == 0) {
            // ---- closest hit (flyscene.cpp:655-691)
            bool in_root = r.pre;
            const float bx = (r.ox + r.dx) - r.ox, by = (r.oy + r.dy) - r.oy, bz = (r.oz + r.dz) - r.oz;   // (o + d) - o, flyscene.cpp:655
            const float brx = __builtin_amdgcn_rcpf(bx), bry = __builtin_amdgcn_rcpf(by), brz = __builtin_amdgcn_rcpf(bz);
            if (!CONT) {
                c_cull += (PRIMARY && r.valid && !r.pre) ? 1u : 0u;
                c_rays += in_root ? 1u : 0u;
                if (COUNT && in_root) c_box += 1;
                in_root = in_root && box_hit_verified(root.bmin, r.ox, r.oy, r.oz, bx, by, bz, brx, bry, brz);
            }
            // (primary tiles: every ray starts at the camera centre and runs through its screen point o + d)
            if ((CONT || GROUP) && PRIMARY && !COUNT && __ballot(in_root) != 0ull) set_cone(r.pre, r.ox, r.oy, r.oz, r.ox + r.dx, r.oy + r.dy, r.oz + r.dz, false);
            float best_t = 3.402823466e+38f;
            int best_f = -1;
            bool dummy = false;
            uint32_t sig_unused = 0u;
            if (GROUP) {
                CC.node_ok = dirs_ok(in_root, bx, by, bz);
                const RayLane R{r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, brx, bry, brz, 4e-4f * (fabsf(r.ox) + fabsf(r.oy) + fabsf(r.oz) + S.extent)};
                cone_walk<false>(nodes, tris, chunks, leaf_chunk0, stk, sl, cl, lane, wc, TQ, root, in_root, R, bx, by, bz, brx, bry, brz, CC, best_t, best_f, dummy);
            } else {
                packet_walk<false, COUNT>(nodes, tris, chunks, leaf_chunk0, S.extent, stk, lane, wc, in_root, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz,
                                          bx, by, bz, brx, bry, brz, best_t, best_f, dummy, c_box, c_ref, sig_unused);
            }
            const bool found = best_f >= 0 && static_cast<uint32_t>(best_f) < S.n_faces;
            const unsigned long long key = found ? ((static_cast<unsigned long long>(__float_as_uint(best_t)) << 32) | static_cast<uint32_t>(best_f))
                                                 : RT_NO_HIT_KEY;
            if (CONT) { if (found) atomicMin(&best[ray_slot], key); }
            else best[ray_slot] = key;
        } else {
            const unsigned long long key = best[ray_slot];
            const bool hit = r.valid && key != RT_NO_HIT_KEY;
            const float best_t = __uint_as_float(static_cast<uint32_t>(key >> 32));
            const int best_f = static_cast<int>(static_cast<uint32_t>(key));
            const float hx = r.ox + best_t * r.dx, hy = r.oy + best_t * r.dy, hz = r.oz + best_t * r.dz;   // flyscene.cpp:695
            const int nl_lane = r.lmode ? 1 : L.n_lights;
            if (STAGE == 1) {
                // ---- lightStrikes(hitPoint, lights): the segment to light CENTRE l (flyscene.cpp:700)
                bool act = hit && (l < nl_lane);
                const unsigned long long lit_index = static_cast<unsigned long long>(tile) * static_cast<unsigned long long>(lslots) + static_cast<unsigned long long>(l);
                if (CONT) act = act && (((lit[lit_index] >> lane) & 1ull) != 0ull);     // still believed visible
                if (__ballot(act) == 0ull) {
                    if (!CONT && lane == 0) lit[lit_index] = 0ull;
                    continue;
                }
                const float px = r.lmode ? r.lx : L.pos[l][0], py = r.lmode ? r.ly : L.pos[l][1], pz = r.lmode ? r.lz : L.pos[l][2];
                const float sdx = hx - px, sdy = hy - py, sdz = hz - pz;
                const float srx = __builtin_amdgcn_rcpf(sdx), sry = __builtin_amdgcn_rcpf(sdy), srz = __builtin_amdgcn_rcpf(sdz);
                bool sroot = act;
                if (!CONT) {
                    c_centre += act ? 1u : 0u;
                    if (COUNT && act) c_box += 1;
                    sroot = act && box_hit_verified(root.bmin, px, py, pz, sdx, sdy, sdz, srx, sry, srz);
                }
                // (every segment starts at the light: one cone per (tile, light) unless the lanes carry lights of their own)
                if ((CONT || GROUP) && !COUNT && __ballot(act && r.lmode != 0u) == 0ull && __ballot(sroot) != 0ull) set_cone(act, L.pos[l][0], L.pos[l][1], L.pos[l][2], hx, hy, hz, true);
                float t_unused = 0.f; int f_unused = -1;
                bool occ = false;
                uint32_t sig_unused = 0u;
                if (GROUP) {
                    CC.node_ok = dirs_ok(sroot, sdx, sdy, sdz);
                    const RayLane R{px, py, pz, sdx, sdy, sdz, srx, sry, srz, 4e-4f * (fabsf(px) + fabsf(py) + fabsf(pz) + S.extent)};
                    cone_walk<true>(nodes, tris, chunks, leaf_chunk0, stk, sl, cl, lane, wc, TQ, root, sroot, R, sdx, sdy, sdz, srx, sry, srz, CC, t_unused, f_unused, occ);
                } else {
                    packet_walk<true, COUNT>(nodes, tris, chunks, leaf_chunk0, S.extent, stk, lane, wc, sroot, px, py, pz, sdx, sdy, sdz, sdx, sdy, sdz,
                                             srx, sry, srz, t_unused, f_unused, occ, c_box, c_ref, sig_unused);
                }
                if (CONT) {
                    const unsigned long long om = __ballot(act && occ);
                    if (lane == 0 && om != 0ull) atomicAnd(&lit[lit_index], ~om);
                } else {
                    const unsigned long long vm = __ballot(act && !occ);
                    if (lane == 0) lit[lit_index] = vm;
                }
            } else {
                // ---- finish: classification, outputs, compaction of lit hits (wave ballot + prefix, one atomic per wave)
                bool is_lit = false;
                for (int k = 0; k < lslots; ++k) {
                    const unsigned long long w = lit[static_cast<unsigned long long>(tile) * static_cast<unsigned long long>(lslots) + static_cast<unsigned long long>(k)];
                    is_lit = is_lit || (hit && k < nl_lane && ((w >> lane) & 1ull) != 0ull);
                }
                if (r.valid) {
                    if (!hit) rec[r.pix] = make_float4(1.f, 1.f, 1.f, __uint_as_float(KIND_CONST));          // BACKGROUND
                    else if (!is_lit) rec[r.pix] = make_float4(0.f, 0.f, 0.f, __uint_as_float(KIND_CONST)); // SHADOW
                    if (out_hit) out_hit[r.pix] = hit ? best_f : -1;
                    if (out_t) out_t[r.pix] = hit ? best_t : -1.0f;
                }
                const unsigned long long lm = __ballot(is_lit);
                if (lm != 0ull) {
                    bool fits;
                    const uint32_t base = shard_reserve(ctl->n_items[level], &ctl->overflow, tile, static_cast<uint32_t>(__popcll(lm)), F.item_cap, lane, fits);
                    if (is_lit && fits) {
                        ShadeItem o;
                        o.ox = r.ox; o.oy = r.oy; o.oz = r.oz; o.dx = r.dx; o.dy = r.dy; o.dz = r.dz;
                        o.lx = r.lx; o.ly = r.ly; o.lz = r.lz; o.lmode = r.lmode; o.pix = r.pix; o.face = best_f; o.t = best_t;
                        o.pad0 = o.pad1 = o.pad2 = 0u;
                        items[base + lanes_below(lm)] = o;
                    }
                }
            }
        }
#ifdef RT_UNIT_HIST
        if (STAGE < 2 && !COUNT && level == 0 && S.dbg != nullptr) {
            // record of this unit: kernel k = 2 STAGE + CONT, slot `work` (< 65536): cycles / 16, the six step counters, the wave
            const unsigned long long dtu = (static_cast<unsigned long long>(clock64()) - unit_t0) >> 4;
            __builtin_amdgcn_wave_barrier();
            const uint32_t kk = 2u * STAGE + (CONT ? 1u : 0u);
            if (work < 65536u && lane < 8) {
                uint32_t v = lane == 0 ? static_cast<uint32_t>(dtu) : (lane < 7 ? s_dbg[wave * 8 + lane - 1] : blockIdx.x * RT_WAVES + static_cast<uint32_t>(wave));
                S.dbg[(static_cast<size_t>(kk) * 65536u + work) * 8u + static_cast<uint32_t>(lane)] = v;
            }
            ++hist_units;
        }
#endif
    }
#ifdef RT_PROFILE
    if (STAGE < 2 && !COUNT && level == 0) {
        // phase clocks of the level-0 trace stages: prof[592 + 16 * STAGE + 8 * CONT + phase]; wave lifetimes (10 ns ticks): sum / waves / first start / last end
        pclk.flush(lane, 592 + 16 * STAGE + (CONT ? 8 : 0));
        if (lane == 0 && g_prof) {
            const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
            const int b = 624 + 8 * STAGE + (CONT ? 4 : 0);
            atomicAdd(&g_prof[b], t1 - wave_t0); atomicAdd(&g_prof[b + 1], 1ull);
            atomicMax(&g_prof[b + 2], ~wave_t0);
            atomicMax(&g_prof[b + 3], t1);
        }
    }
#endif
#ifdef RT_UNIT_HIST
    if (STAGE < 2 && !COUNT && level == 0 && lane == 0 && S.dbg != nullptr) {
        // record of this wave (after the unit records: 4 x 65536 x 8 words): kernel k, wave id < 16384: lifetime / 16, units, XCC id
        const unsigned long long t1 = static_cast<unsigned long long>(clock64());
        const uint32_t wid = blockIdx.x * RT_WAVES + static_cast<uint32_t>(wave), kk = 2u * STAGE + (CONT ? 1u : 0u);
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if (wid < 16384u) {
            uint32_t *wp = S.dbg + static_cast<size_t>(4u) * 65536u * 8u + (static_cast<size_t>(kk) * 16384u + wid) * 4u;
            wp[0] = static_cast<uint32_t>((t1 - hist_t0) >> 4); wp[1] = hist_units; wp[2] = xcc & 7u; wp[3] = static_cast<uint32_t>(hist_t0 >> 4);
        }
    }
#endif
    c_rays = wave_sum(c_rays); c_cull = wave_sum(c_cull); c_centre = wave_sum(c_centre);
    if (COUNT) { c_box = wave_sum(c_box); c_ref = wave_sum(c_ref); }
    if (lane == 0) {
        if (c_rays) atomicAdd((PRIMARY || level == 0) ? &ctl->stat[blockIdx.x & (RT_STAT_SHARDS - 1)][ST_RAYS_PRIMARY] : &ctl->stat[blockIdx.x & (RT_STAT_SHARDS - 1)][ST_RAYS_BOUNCE], static_cast<unsigned long long>(c_rays));
        if (c_cull) atomicAdd(&ctl->stat[blockIdx.x & (RT_STAT_SHARDS - 1)][ST_PIXELS_CULLED], static_cast<unsigned long long>(c_cull));
        if (c_centre) atomicAdd(&ctl->stat[blockIdx.x & (RT_STAT_SHARDS - 1)][ST_RAYS_CENTRE], static_cast<unsigned long long>(c_centre));
        if (COUNT) {
            if (c_box) atomicAdd(&ctl->stat[blockIdx.x & (RT_STAT_SHARDS - 1)][ST_BOX_TESTS], static_cast<unsigned long long>(c_box));
            if (c_ref) atomicAdd(&ctl->stat[blockIdx.x & (RT_STAT_SHARDS - 1)][ST_LEAF_TRI_REFS], static_cast<unsigned long long>(c_ref));
        }
    }
}

// ======================================================================================================
// K2: area-light sample shadow rays (phongShade's lightStrikes(hitPoint, samples), flyscene.cpp:834-836).
// One wave = the N samples of one (hit, light) pair (N = 64), several pairs per wave (N < 64) or
// ceil(N/64) wave passes per pair (N > 64).  Output: one visibility bit per sample.
// ======================================================================================================
// Leaf tasks (tree scenes): leaves whose estimated cost exceeds `budget` are not processed by the walking wave but written
// to `tasks_out` as chunk-range pieces; k_shadow<.., CONT=true> processes them on whichever wave is free and merges the
// occluded bits into `vis` with atomicAnd.  (A first version handed over whole sub-trees once a unit was over budget: the
// pieces were still 979-triangle leaves and the passes ran back to back -- no gain.)

// 97 VGPRs would cost the tree variants their fifth wave per SIMD; asked for 5, the allocator finds them without spilling
#ifndef RT_SHADOW_WPE
#define RT_SHADOW_WPE 5
#endif
template <bool COUNT, bool FLAT, bool CONT>
__global__ __launch_bounds__(RT_WAVES * 64) __attribute__((amdgpu_waves_per_eu(FLAT ? 6 : RT_SHADOW_WPE, 8))) void k_shadow(const DNode *__restrict__ nodes, const TriRec *__restrict__ tris,
                                                          const ChunkBound *__restrict__ chunks, const uint32_t *__restrict__ leaf_chunk0,
                                                           const DScene S, const DLights L, const int level, const int ctr_slot,
                                                           const int lslots, const uint32_t item_cap, const ShadeItem *__restrict__ items,
                                                           Control *__restrict__ ctl, unsigned long long *vis, const TaskQueues Q, const uint32_t *__restrict__ sidx) {
    __shared__ uint4 s_stage[1];                                              // k_shadow never stages leaves in LDS (leaf_visit<.., STAGED = false>)
    __shared__ float4 s_fv[(FLAT && !COUNT) ? 64 * 3 : 1];                    // flat scenes: the vertices A, B, C of the root leaf's triangles
    __shared__ unsigned long long s_mask[FLAT ? 1 : RT_WAVES * RT_STACK];     // flat scenes need no stack
    __shared__ uint32_t s_node[FLAT ? 1 : RT_WAVES * RT_STACK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const WaveStack stk{s_node + (FLAT ? 0 : wave * RT_STACK), s_mask + (FLAT ? 0 : wave * RT_STACK), s_stage};
    const uint32_t N = static_cast<uint32_t>(L.n_samples);
    const uint32_t G = N <= 64u ? 64u / N : 1u;              // (hit,light) pairs per wave
    const uint32_t P = (N + 63u) / 64u;                       // 64-sample passes (= mask words) per pair
    // units of list shard s: ceil(cnt_s * lslots / G) (N <= 64) or cnt_s * lslots * P (N > 64), numbered shard after shard
    // (after k_beam: the units are the hits of its compacted survivor list)
    const uint32_t *__restrict__ item_counts = sidx != nullptr ? ctl->n_sitems[level] : ctl->n_items[level];
    const ShardMap imap = N <= 64u ? shard_map(item_counts, lane, item_cap, static_cast<uint32_t>(lslots), G)
                                   : shard_map(item_counts, lane, item_cap, static_cast<uint32_t>(lslots) * P, 1u);
    const unsigned long long units = imap.total;
    if (!CONT && units == 0ull) return;           // an empty bounce level: leave before any set-up (every wave takes this branch)
    const DNode root = nodes[0];
    const uint32_t slot = N <= 64u ? static_cast<uint32_t>(lane) / N : 0u;
    const uint32_t s_in = N <= 64u ? static_cast<uint32_t>(lane) - slot * N : static_cast<uint32_t>(lane);
    const unsigned long long low = N >= 64u ? ~0ull : ((1ull << N) - 1ull);
    // sample grid coordinates of this lane: fixed for the whole kernel when a unit holds all N <= 64 samples
    const uint32_t vst = static_cast<uint32_t>(L.vsteps > 0 ? L.vsteps : 1);
    const float fi_lane = static_cast<float>(s_in / vst) + 0.5f, fj_lane = static_cast<float>(s_in % vst) + 0.5f;
    const float fi_last = static_cast<float>(L.usteps - 1) + 0.5f, fj_last = static_cast<float>(L.vsteps - 1) + 0.5f;
    const bool blocks = sample_blocks(L);

    LanePlane plane{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (FLAT && static_cast<uint32_t>(lane) < (root.count_flags & 0x7fffffffu)) {
        const TriRec *tp = tris + root.first + lane;
        plane = LanePlane{tp->nx, tp->ny, tp->nz, tp->nA, 0.f, 0.f};
    }
    // Flat scenes, one (hit, light) pair per wave: besides the plane rule, the geometric shaft test of the tree scenes, here with
    // lane = (triangle, test) -- 8 triangles x (6 tangent planes + near box) per wave step, ceil(count / 8) steps per unit; the vertices
    // come from LDS (unit-invariant), the margin from the leaf's chunk bound.  cube.obj at 1080p/64: 6.8 -> 0.011 triangle tests per
    // unit (99.8 % of the units keep no triangle at all and leave before a single ray is set up).
    float flat_m = -1.0f;
    if (FLAT && !COUNT) {
        const uint32_t fcnt = root.count_flags & 0x7fffffffu;
        for (uint32_t i = threadIdx.x; i < fcnt && i < 64u; i += blockDim.x) {
            const TriRec t = tris[root.first + i];
            s_fv[3u * i] = make_float4(t.ax, t.ay, t.az, 0.f);
            s_fv[3u * i + 1u] = make_float4(t.ax + t.e1x, t.ay + t.e1y, t.az + t.e1z, 0.f);
            s_fv[3u * i + 2u] = make_float4(t.ax + t.e0x, t.ay + t.e0y, t.az + t.e0z, 0.f);
        }
        __syncthreads();
        const ChunkBound cb0 = chunks[root.pad[0]];
        if (cb0.never < 1.5f && fcnt <= 64u) flat_m = cb0.infl * 1.0625f;
    }
    // (tried on octree leaves too: 25 % fewer ray-mode triangle tests on dodgeColorTest.obj but no whole 64-triangle chunk is ever
    // skipped, and the per-leaf plane pass plus its registers cost more than they saved: 2.31 ms vs 1.93 ms)
    const bool plane_cull = FLAT && !COUNT && G == 1u && S.plane_cull != 0;
    // the lane's sample and the sample box of scene light 0 (see the unit loop)
    const bool scene_light0 = lslots == 1 && N <= 64u;
    float pre_sx = 0.f, pre_sy = 0.f, pre_sz = 0.f, pre_x0 = 0.f, pre_x1 = 0.f, pre_y0 = 0.f, pre_y1 = 0.f, pre_z0 = 0.f, pre_z1 = 0.f;
    if (scene_light0) {
        const LightGrid lg = light_grid(L, L.pos[0][0], L.pos[0][1], L.pos[0][2]);
        grid_sample(lg, fi_lane, fj_lane, pre_sx, pre_sy, pre_sz);
        grid_sample(lg, 0.5f, 0.5f, pre_x0, pre_y0, pre_z0);
        grid_sample(lg, fi_last, fj_last, pre_x1, pre_y1, pre_z1);
        if (L.mode == RT_LIGHT_SPHERE) {
            sphere_sample(L, s_in, L.pos[0][0], L.pos[0][1], L.pos[0][2], pre_sx, pre_sy, pre_sz);
            sphere_box(L, L.pos[0][0], L.pos[0][1], L.pos[0][2], pre_x0, pre_y0, pre_z0, pre_x1, pre_y1, pre_z1);
        }
        if (FLAT) plane_prepare(plane, fminf(pre_x0, pre_x1), fminf(pre_y0, pre_y1), fminf(pre_z0, pre_z1), fmaxf(pre_x0, pre_x1), fmaxf(pre_y0, pre_y1),
                                fmaxf(pre_z0, pre_z1));
    }

    uint32_t c_rays = 0, c_box = 0, c_ref = 0, c_walked = 0;
    ShardedQueue q;
    uint32_t n_work = 0;
    ShardMap tmap{0u, 0u, 0u, 0u};
    if (CONT) {
        tmap = shard_map(ctl->n_task_sh[level], lane, Q.cap / RT_LIST_SHARDS, 1u, 1u);     // producers clamp to the per-shard capacity too
        n_work = tmap.total;
        q.init_static(n_work, gridDim.x * RT_WAVES, uniform_u32(blockIdx.x * RT_WAVES + static_cast<uint32_t>(wave)), lane);
    } else {
        // N > 64: the passes of one (hit, light) pair are consecutive units and walk the same part of the tree -- hand them out two
        // at a time (4K / 256 samples / 1 M triangles: k_shadow 40 -> 32 ms); otherwise balance first (neighbouring heavy units
        // pile up: dodge 1.03 -> 1.9 ms with chunks of 8)
        // (a launch with only a few units per wave -- what k_beam leaves of a flat scene -- would spend its time on the queue heads: one
        // returning atomic per unit and 88 of them per microsecond and head; such a launch is dealt out statically)
        if (FLAT && units < 16ull * gridDim.x * RT_WAVES)
            q.init_static(static_cast<uint32_t>(units), gridDim.x * RT_WAVES, uniform_u32(blockIdx.x * RT_WAVES + static_cast<uint32_t>(wave)), lane);
        else
            q.init(ctl->queue[ctr_slot], static_cast<uint32_t>(units), gridDim.x * RT_WAVES, blockIdx.x, lane,
                   S.queue_local >= 0 ? static_cast<uint32_t>(S.queue_local) : (P > 1u ? 2u : 0u));
    }
#ifdef RT_PROFILE
    PhaseClock pclk; pclk.start();
#endif
    for (uint32_t work = 0; q.next(work);) {
#ifdef RT_PROFILE
        pclk.to(6);
#endif
#ifdef RT_PROFILE_HIST
        const long long prof_t0 = clock64();
#endif
        uint32_t unit = work;
        WalkCtl wc = walk_plain();
        RT_PROF_ADD(lane, 13, 1);
#ifdef RT_PROFILE
        wc.pc = &pclk;
#endif
        if (CONT) {
            uint32_t tsh, tloc, tn;
            shard_find(tmap, work, tsh, tloc, tn);
            const ContTask task = Q.tasks_in[tsh * (Q.cap / RT_LIST_SHARDS) + tloc];
            unit = uniform_u32(task.unit);
            wc.resume = true;
            wc.start_node = uniform_u32(task.node);
            wc.start_mask = uniform_u64(task.mask);
            wc.c_begin = uniform_u32(task.c_begin);
            wc.c_end = uniform_u32(task.c_end);
        }
        // leaf tasks only pay when the launch has few units per wave (dodge at 1080p: 41, +5 %); with thousands of units per
        // wave the dynamic queue balances on its own and the second pass is pure overhead (cfg4: 3,600 per wave, 58 vs 46 ms)
        if (!FLAT && Q.tasks_out != nullptr && Q.budget != 0u && units < 256ull * gridDim.x * RT_WAVES) {
            const uint32_t tsh = blockIdx.x & (RT_LIST_SHARDS - 1u), tcap = Q.cap / RT_LIST_SHARDS;      // sharded task queue (Control::n_task_sh)
            wc.budget = Q.budget; wc.unit = unit; wc.tasks = Q.tasks_out + tsh * tcap; wc.task_count = &ctl->n_task_sh[level][tsh * 16u]; wc.task_cap = tcap;
            wc.target = Q.target ? Q.target : Q.budget;
        }
        uint32_t g, s, pass = 0;
        bool valid, slot_ok = false;
        uint32_t sh, lu, n_sh;
        shard_find(imap, unit, sh, lu, n_sh);
        const uint32_t groups = n_sh * static_cast<uint32_t>(lslots);     // (hit, light) pairs of this list shard
        if (N <= 64u) {
            g = lu * G + slot; s = s_in;
            valid = (slot < G) && (g < groups);
            slot_ok = valid;
        } else {
            g = lu / P; pass = lu - g * P; s = pass * 64u + s_in;
            if (blocks) {
                const uint32_t bpr = vst >> 3, bi = pass / bpr, bj = pass - bi * bpr;
                s = (bi * 8u + (s_in >> 3)) * vst + bj * 8u + (s_in & 7u);
            }
            valid = s < N;
        }
        uint32_t item_i = valid ? (lslots == 1 ? g : g / static_cast<uint32_t>(lslots)) : 0u;
        int l = valid ? static_cast<int>(g - item_i * static_cast<uint32_t>(lslots)) : 0;
        uint32_t item_at;
        ShadeItem it;
        if (G == 1u) {
            // one pair per wave: g is the same in every lane, and so is the item -- a wave-uniform index (scalar load).  Every lane, also
            // one past the last sample, then carries the unit's h and light: the triangle lanes of the culling tests rely on that.
            const uint32_t gu = uniform_u32(g);
            item_i = lslots == 1 ? gu : gu / static_cast<uint32_t>(lslots);
            l = static_cast<int>(gu - item_i * static_cast<uint32_t>(lslots));
            item_at = uniform_u32(sh * item_cap + item_i);
            if (sidx != nullptr) item_at = uniform_u32(sidx[item_at]);           // k_beam's survivors: position in the list -> item storage index
            it = items[item_at];
        } else {
            item_at = sh * item_cap + item_i;
            if (sidx != nullptr) item_at = sidx[item_at];
            it = items[item_at];
        }
        g = item_at * static_cast<uint32_t>(lslots) + static_cast<uint32_t>(l);     // (item storage index) * lslots + light: the vis slot
        const int nl = it.lmode ? 1 : L.n_lights;
        valid = valid && (l < nl);
        const float hx = it.ox + it.t * it.dx, hy = it.oy + it.t * it.dy, hz = it.oz + it.t * it.dz;
        const float px = it.lmode ? it.lx : L.pos[l][0], py = it.lmode ? it.ly : L.pos[l][1], pz = it.lmode ? it.lz : L.pos[l][2];
        // Everything that depends on the light alone was computed before the loop for scene light 0; it applies when every lane's
        // item sees the scene lights (level-0 items always do) and there is one of them.
        const bool own_light = scene_light0 ? (__ballot(valid && it.lmode != 0u) != 0ull) : true;
        float sx = pre_sx, sy = pre_sy, sz = pre_sz;
        float x0 = pre_x0, x1 = pre_x1, y0 = pre_y0, y1 = pre_y1, z0 = pre_z0, z1 = pre_z1;
        if (own_light) {
            const LightGrid lg = light_grid(L, px, py, pz);
            if (N <= 64u) grid_sample(lg, fi_lane, fj_lane, sx, sy, sz);
            else grid_sample(lg, static_cast<float>(s / vst) + 0.5f, static_cast<float>(s % vst) + 0.5f, sx, sy, sz);
            // box of this light's sample positions: the samples are monotone in each grid index, so the first and the last give
            // the exact extremes
            grid_sample(lg, 0.5f, 0.5f, x0, y0, z0);
            grid_sample(lg, fi_last, fj_last, x1, y1, z1);
            if (L.mode == RT_LIGHT_SPHERE) {
                sphere_sample(L, s, px, py, pz, sx, sy, sz);
                sphere_box(L, px, py, pz, x0, y0, z0, x1, y1, z1);
            }
        }
        const unsigned long long vis_index = N <= 64u ? static_cast<unsigned long long>(g) : static_cast<unsigned long long>(g) * P + pass;
        if (plane_cull) {
            wc.seg.on = true;
            wc.seg.prepared = !own_light;
            wc.seg.hx = hx; wc.seg.hy = hy; wc.seg.hz = hz;
            wc.seg.slx = fminf(x0, x1); wc.seg.shx = fmaxf(x0, x1);
            wc.seg.sly = fminf(y0, y1); wc.seg.shy = fmaxf(y0, y1);
            wc.seg.slz = fminf(z0, z1); wc.seg.shz = fmaxf(z0, z1);
            wc.seg.m0 = 2e-5f * ((fabsf(x0) + fabsf(x1)) + (fabsf(y0) + fabsf(y1)) + (fabsf(z0) + fabsf(z1)) + (fabsf(hx) + fabsf(hy) + fabsf(hz)));
        }
        if (FLAT && !COUNT && plane_cull && flat_m >= 0.0f) {
            // Which triangles of the (flat) scene can ANY ray of the unit hit?  Known before a single ray is set up: the geometric shaft
            // test (lane = (triangle, test)) and the plane rule (lane = triangle) only need h and the box of the samples.
            const ShaftLanes SLf = make_shaft_lanes(lane, wc.seg.hx, wc.seg.hy, wc.seg.hz, wc.seg.slx, wc.seg.sly, wc.seg.slz, wc.seg.shx, wc.seg.shy, wc.seg.shz, S.extent);
            const int tk = lane & 7;
            const float pax = SLf.r[0] + SLf.r[1], pay = SLf.r[2] + SLf.r[3], paz = SLf.r[4] + SLf.r[5];          // (one addend is zero: exact)
            const float pcm = SLf.r[6] - flat_m * ((fabsf(pax) + fabsf(pay)) + fabsf(paz));
            const uint32_t fcnt = root.count_flags & 0x7fffffffu;
            unsigned long long skip = 0ull;
            for (uint32_t t0 = 0u; t0 < fcnt; t0 += 8u) {
                const uint32_t t = t0 + (static_cast<uint32_t>(lane) >> 3);
                const bool tv = t < fcnt;
                const float4 A = s_fv[3u * (tv ? t : 0u)], B = s_fv[3u * (tv ? t : 0u) + 1u], C = s_fv[3u * (tv ? t : 0u) + 2u];
                const float fa = __builtin_fmaf(pax, A.x, __builtin_fmaf(pay, A.y, paz * A.z));
                const float fb = __builtin_fmaf(pax, B.x, __builtin_fmaf(pay, B.y, paz * B.z));
                const float fc = __builtin_fmaf(pax, C.x, __builtin_fmaf(pay, C.y, paz * C.z));
                const bool p_out = fminf(fminf(fa, fb), fc) + pcm > 0.0f;
                const bool b_out = (fminf(fminf(A.x, B.x), C.x) - flat_m > SLf.r[3]) || (fmaxf(fmaxf(A.x, B.x), C.x) + flat_m < SLf.r[0])
                                || (fminf(fminf(A.y, B.y), C.y) - flat_m > SLf.r[4]) || (fmaxf(fmaxf(A.y, B.y), C.y) + flat_m < SLf.r[1])
                                || (fminf(fminf(A.z, B.z), C.z) - flat_m > SLf.r[5]) || (fmaxf(fmaxf(A.z, B.z), C.z) + flat_m < SLf.r[2]);
                const unsigned long long bo = __ballot(tv && (tk < 6 ? p_out : (tk == 6 && b_out)));
                const unsigned long long m8 = __ballot(lane < 8 && ballot_byte_any(bo, lane)) & 0xffull;
                skip |= m8 << t0;
            }
            skip |= __ballot(wc.seg.prepared ? plane_rules_out_prepared(wc.seg, plane) : plane_rules_out(wc.seg, plane.nx, plane.ny, plane.nz, plane.nA));
            wc.skip = skip;
            RT_PROF_ADD(lane, 70, 1); RT_PROF_ADD(lane, 71, __popcll(skip & (fcnt >= 64u ? ~0ull : ((1ull << fcnt) - 1ull))));
            if (((fcnt >= 64u ? ~0ull : ((1ull << fcnt) - 1ull)) & ~skip) == 0ull) {
                // nothing in the scene can block any ray of this unit: every sample is visible, whatever the root test of its ray says
                c_rays += valid ? 1u : 0u;
                const unsigned long long vm = __ballot(valid);
                if (N <= 64u) {
                    if (s_in == 0u && slot_ok) vis[vis_index] = (vm >> (slot * N)) & low;
                } else if (lane == 0) {
                    vis[vis_index] = vm;
                }
                continue;
            }
        }
        const float ddx = hx - sx, ddy = hy - sy, ddz = hz - sz;
        const float srx = __builtin_amdgcn_rcpf(ddx), sry = __builtin_amdgcn_rcpf(ddy), srz = __builtin_amdgcn_rcpf(ddz);
        bool sroot;
        if (CONT) {
            // the root test was passed when the task was emitted (the mask only holds lanes that reached `node`); rays
            // another piece of this unit already found occluded are dropped
            const unsigned long long seen = valid ? vis[vis_index] : 0ull;
            sroot = valid && (((seen >> s_in) & 1ull) != 0ull);
        } else {
            c_rays += valid ? 1u : 0u;
            c_walked += valid ? 1u : 0u;                        // segments actually formed (units the culling devices left before this line formed none)
            if (COUNT && valid) c_box += 1;
            sroot = valid && box_hit_verified(root.bmin, sx, sy, sz, ddx, ddy, ddz, srx, sry, srz);
        }
        float t_unused = 0.f; int f_unused = -1;
        bool occ = false;
#ifdef RT_PROFILE
        pclk.to(0);
#endif
        walk<true, COUNT, FLAT, false>(root, nodes, tris, chunks, leaf_chunk0, S.extent, stk, lane, wc, plane, sroot, sx, sy, sz, ddx, ddy, ddz, ddx, ddy, ddz, srx, sry, srz, t_unused, f_unused, occ, c_box, c_ref);
#ifdef RT_PROFILE
        pclk.to(7);
#endif
        if (CONT) {
            const unsigned long long om = __ballot(valid && occ);
            if (N <= 64u) {
                const unsigned long long word = (om >> (slot * N)) & low;
                if (s_in == 0u && slot_ok && word != 0ull) atomicAnd(&vis[vis_index], ~word);
            } else if (lane == 0 && om != 0ull) {
                atomicAnd(&vis[vis_index], ~om);
            }
        } else {
            const unsigned long long vm = __ballot(valid && !occ);
            if (N <= 64u) {
                if (s_in == 0u && slot_ok) vis[vis_index] = (vm >> (slot * N)) & low;
            } else if (lane == 0) {
                vis[vis_index] = vm;
            }
        }
#ifdef RT_PROFILE_HIST
        {   // per-unit cycle histogram: prof[16 + log2(cycles)], max in prof[9], sum in prof[10], count in prof[11]
            const unsigned long long dt = static_cast<unsigned long long>(clock64() - prof_t0);
            if (lane == 0 && g_prof) {
                atomicMax(&g_prof[9], dt); atomicAdd(&g_prof[10], dt); atomicAdd(&g_prof[11], 1ull);
                int b = 63 - __builtin_clzll(dt | 1ull); if (b > 40) b = 40;
                atomicAdd(&g_prof[16 + b], 1ull);
            }
        }
#endif
    }
#ifdef RT_PROFILE
    pclk.flush(lane, CONT ? 488 : 480);          // (prof[480, 496): clear of the step counters of both regions)
#endif
    c_rays = wave_sum(c_rays); c_walked = wave_sum(c_walked);
    if (COUNT) { c_box = wave_sum(c_box); c_ref = wave_sum(c_ref); }
    if (lane == 0) {
        if (c_rays) atomicAdd(&ctl->stat[blockIdx.x & (RT_STAT_SHARDS - 1)][ST_RAYS_SAMPLE], static_cast<unsigned long long>(c_rays));
        if (c_walked) atomicAdd(&ctl->stat[blockIdx.x & (RT_STAT_SHARDS - 1)][ST_SAMPLE_WALKED], static_cast<unsigned long long>(c_walked));
        if (COUNT) {
            if (c_box) atomicAdd(&ctl->stat[blockIdx.x & (RT_STAT_SHARDS - 1)][ST_BOX_TESTS_SHADOW], static_cast<unsigned long long>(c_box));
            if (c_ref) atomicAdd(&ctl->stat[blockIdx.x & (RT_STAT_SHARDS - 1)][ST_LEAF_TRI_REFS_SHADOW], static_cast<unsigned long long>(c_ref));
        }
    }
}

// ======================================================================================================
// K2 on TREE scenes with one (hit, light) pair -- or one 64-sample pass of it -- per wave (N >= 33 samples): the shaft walk.
// Same units, same queue and same output words as k_shadow<false, false, false>.
// ======================================================================================================
#ifndef RT_SHAFT_WPE
#define RT_SHAFT_WPE 6
#endif
#ifndef RT_UNIT_STRIDE
#define RT_UNIT_STRIDE 0          // RT_UNIT_HIST build: every 2^RT_UNIT_STRIDE-th unit of the level-0 shaft launch is recorded
#endif
template <bool CONT, bool TASKS>
__global__ __launch_bounds__(RT_WAVES * 64) __attribute__((amdgpu_waves_per_eu(RT_SHAFT_WPE, 8)))
void k_shadow_shaft(const DNode *__restrict__ nodes, const TriRec *__restrict__ tris, const ChunkBound *__restrict__ chunks,
                    const DScene S, const DLights L, const int level, const int ctr_slot, const int lslots, const uint32_t item_cap,
                    const ShadeItem *__restrict__ items, Control *__restrict__ ctl, unsigned long long *__restrict__ vis, const TaskQueues Q,
                    const uint32_t *__restrict__ sidx) {
    __shared__ unsigned long long s_mask[RT_WAVES * RT_STACK];
    __shared__ uint32_t s_node[RT_WAVES * RT_STACK];
    __shared__ uint4 s_top[CONT ? 1 : RT_LDS_NODES * 4];           // the top of the octree: first RT_LDS_NODES DNodes (breadth-first order)
    __shared__ unsigned long long s_lmask[RT_WAVES * RT_LEAF_SLOTS];
    __shared__ uint32_t s_lnode[RT_WAVES * RT_LEAF_SLOTS];
    __shared__ float4 s_tri[RT_WAVES * RT_SHAFT_TRI_REC];
    __shared__ float4 s_shaft[RT_WAVES * 16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const WaveStack stk{s_node + wave * RT_STACK, s_mask + wave * RT_STACK, nullptr};
    const uint32_t n_lds = CONT ? 0u : (S.n_nodes < RT_LDS_NODES ? S.n_nodes : RT_LDS_NODES);
    ShaftLds sl{reinterpret_cast<const DNode *>(s_top), n_lds, s_lnode + wave * RT_LEAF_SLOTS, s_lmask + wave * RT_LEAF_SLOTS, s_tri + wave * RT_SHAFT_TRI_REC, s_shaft + wave * 16
#ifdef RT_PROFILE
                , nullptr
#endif
    };
    const uint32_t N = static_cast<uint32_t>(L.n_samples);
    const uint32_t P = (N + 63u) / 64u;                       // 64-sample passes (= mask words) per pair
    const ShardMap imap = shard_map(sidx != nullptr ? ctl->n_sitems[level] : ctl->n_items[level], lane, item_cap, static_cast<uint32_t>(lslots) * P, 1u);
    const uint32_t units = imap.total;
    if (units == 0u) return;                      // an empty bounce level: leave before the LDS copy (every wave of the block takes this branch)
    if (!CONT) {
        const uint4 *__restrict__ src = reinterpret_cast<const uint4 *>(nodes);
        for (uint32_t i = threadIdx.x; i < n_lds * 4u; i += blockDim.x) s_top[i] = src[i];
        __syncthreads();
    }
    const uint32_t vst = static_cast<uint32_t>(L.vsteps > 0 ? L.vsteps : 1);
    const bool blocks = sample_blocks(L);
    const float fi_last = static_cast<float>(L.usteps - 1) + 0.5f, fj_last = static_cast<float>(L.vsteps - 1) + 0.5f;
    // Everything that depends on the light alone is evaluated once for scene light 0 (one light, item sees the scene lights: every
    // level-0 item); the lane's grid coordinates inside a pass are lane constants; the two run-time divisions of a unit (pair / pass,
    // block row / column) go through a float reciprocal with an exact correction step instead of the ~30-instruction integer division.
    const LightGrid lg0 = light_grid(L, L.pos[0][0], L.pos[0][1], L.pos[0][2]);
    const uint32_t bpr = blocks ? (vst >> 3) : 1u;
    const float inv_P = 1.0f / static_cast<float>(P), inv_bpr = 1.0f / static_cast<float>(bpr), inv_ls = 1.0f / static_cast<float>(lslots);
    // (a power-of-two divisor -- 256 samples: 4 passes, 2 blocks per row -- is a shift)
    const int sh_P = (P & (P - 1u)) == 0u ? __builtin_ctz(P) : -1, sh_bpr = (bpr & (bpr - 1u)) == 0u ? __builtin_ctz(bpr) : -1;
    const int sh_ls = (static_cast<uint32_t>(lslots) & (static_cast<uint32_t>(lslots) - 1u)) == 0u ? __builtin_ctz(static_cast<uint32_t>(lslots)) : -1;
    auto udiv = [](const uint32_t n, const uint32_t d, const float inv, const int sh) -> uint32_t {
        if (sh >= 0) return n >> sh;
        // float(n) * inv is off by about q * 2^-23 (q = n / d): one correction step each way is exact while q < 2^23; beyond that
        // (a per-shard unit count of a very large frame with many lights and a non-power-of-two pass count) the integer division
        // itself decides -- never a wrong (item, light, pass)
        uint32_t qv = static_cast<uint32_t>(static_cast<float>(n) * inv);
        if (qv * d > n) qv -= 1u;
        if (n - qv * d >= d) qv += 1u;
        if (qv * d > n || n - qv * d >= d) qv = n / d;
        return qv;
    };
    const float fi_lane = blocks ? static_cast<float>(static_cast<uint32_t>(lane) >> 3) : static_cast<float>(static_cast<uint32_t>(lane) / vst);
    const float fj_lane = blocks ? static_cast<float>(static_cast<uint32_t>(lane) & 7u) : static_cast<float>(static_cast<uint32_t>(lane) % vst);
    uint32_t c_rays = 0;
    ShardedQueue q;
    ShardMap tmap{0u, 0u, 0u, 0u};
    if (CONT) {
        // the leaf-task launch: work = the chunk-range tasks the walking launch emitted (sharded queue, Control::n_task_sh)
        tmap = shard_map(ctl->n_task_sh[level], lane, Q.cap / RT_LIST_SHARDS, 1u, 1u);
        q.init_static(tmap.total, gridDim.x * RT_WAVES, uniform_u32(blockIdx.x * RT_WAVES + static_cast<uint32_t>(wave)), lane);
    } else {
        q.init(ctl->queue[ctr_slot], units, gridDim.x * RT_WAVES, blockIdx.x, lane, S.queue_local >= 0 ? static_cast<uint32_t>(S.queue_local) : (P > 1u ? 2u : 0u),
               static_cast<uint32_t>(S.queue_div));
    }
#ifdef RT_PROFILE
    PhaseClock pclk; pclk.start();
    sl.pc = &pclk;
    const unsigned long long wave_t0 = __builtin_amdgcn_s_memrealtime();        // 100 MHz
    if (lane == 0 && g_prof) atomicMin(&g_prof[100], wave_t0 + 1ull);           // (memset 0 = unset: see the host side)
#endif
    for (uint32_t work = 0; q.next(work);) {
#ifdef RT_PROFILE
        pclk.to(6);
        const unsigned long long unit_t0 = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef RT_UNIT_HIST
        const unsigned long long uh_t0 = static_cast<unsigned long long>(clock64());
        if (lane < 8) g_uh[wave * 8 + lane] = 0u;
#endif
        RT_PROF_ADD(lane, 13, 1);
        uint32_t unit = work;
        uint32_t t_node = 0u, t_cb = 0u, t_ce = 0u;
        unsigned long long t_mask = 0ull;
        if (CONT) {
            uint32_t tsh, tloc, tn;
            shard_find(tmap, work, tsh, tloc, tn);
            const ContTask task = Q.tasks_in[tsh * (Q.cap / RT_LIST_SHARDS) + tloc];
            unit = uniform_u32(task.unit); t_node = uniform_u32(task.node); t_mask = uniform_u64(task.mask);
            t_cb = uniform_u32(task.c_begin); t_ce = uniform_u32(task.c_end);
        }
        uint32_t sh, lu, n_sh;
        shard_find(imap, unit, sh, lu, n_sh);
        const uint32_t g = P == 1u ? lu : udiv(lu, P, inv_P, sh_P), pass = lu - g * P;                       // (hit, light) pair of the list shard, pass of it
        const uint32_t item_i = lslots == 1 ? g : udiv(g, static_cast<uint32_t>(lslots), inv_ls, sh_ls);
        const int l = static_cast<int>(g - item_i * static_cast<uint32_t>(lslots));
        uint32_t item_at = uniform_u32(sh * item_cap + item_i);
        if (sidx != nullptr) item_at = uniform_u32(sidx[item_at]);            // k_beam's survivors: position in the list -> item storage index
        if (!CONT && Q.pair_done != nullptr) {
            // several lights behind k_pair_beam: the beam of this (hit, light) pair found it unblocked and wrote its words (accounted for there)
            const size_t di = static_cast<size_t>(item_at) * static_cast<size_t>(lslots) + static_cast<size_t>(l);
            const uint32_t dw = reinterpret_cast<const uint32_t *>(Q.pair_done)[di >> 2];          // wave-uniform: a scalar load
            if (((dw >> (8u * static_cast<uint32_t>(di & 3u))) & 0xffu) != 0u) continue;
        }
        const ShadeItem it = items[item_at];                                  // wave-uniform: a scalar load
        uint32_t s = pass * 64u + static_cast<uint32_t>(lane);
        float fi = fi_lane + 0.5f, fj = fj_lane + 0.5f;                       // the lane's sample (i + 0.5, j + 0.5) -- P == 1: s = lane
        float b_i0 = 0.5f, b_j0 = 0.5f, b_i1 = fi_last, b_j1 = fj_last;      // grid corners of the unit's samples (the whole light unless in blocks)
        if (blocks) {
            const uint32_t bi = udiv(pass, bpr, inv_bpr, sh_bpr), bj = pass - bi * bpr;
            s = (bi * 8u + (static_cast<uint32_t>(lane) >> 3)) * vst + bj * 8u + (static_cast<uint32_t>(lane) & 7u);
            b_i0 = static_cast<float>(bi * 8u) + 0.5f; b_j0 = static_cast<float>(bj * 8u) + 0.5f;
            b_i1 = static_cast<float>(bi * 8u + 7u) + 0.5f; b_j1 = static_cast<float>(bj * 8u + 7u) + 0.5f;
            fi = (static_cast<float>(bi * 8u) + fi_lane) + 0.5f; fj = (static_cast<float>(bj * 8u) + fj_lane) + 0.5f;     // exact small integers
        } else if (P > 1u) {
            fi = static_cast<float>(s / vst) + 0.5f; fj = static_cast<float>(s % vst) + 0.5f;
        }
        const int nl = it.lmode ? 1 : L.n_lights;
        const bool valid = s < N && l < nl;
        const float hx = it.ox + it.t * it.dx, hy = it.oy + it.t * it.dy, hz = it.oz + it.t * it.dz;
        LightGrid lg = lg0;
        if (it.lmode != 0u || l != 0) {
            const float px = it.lmode ? it.lx : L.pos[l][0], py = it.lmode ? it.ly : L.pos[l][1], pz = it.lmode ? it.lz : L.pos[l][2];
            lg = light_grid(L, px, py, pz);
        }
        float sx, sy, sz, x0, y0, z0, x1, y1, z1;
        grid_sample(lg, fi, fj, sx, sy, sz);
        grid_sample(lg, b_i0, b_j0, x0, y0, z0);             // the samples are monotone in each grid index: two corners give the exact box
        grid_sample(lg, b_i1, b_j1, x1, y1, z1);
        if (L.mode == RT_LIGHT_SPHERE) {
            const float px = it.lmode ? it.lx : L.pos[l][0], py = it.lmode ? it.ly : L.pos[l][1], pz = it.lmode ? it.lz : L.pos[l][2];
            sphere_sample(L, s, px, py, pz, sx, sy, sz);
            sphere_box(L, px, py, pz, x0, y0, z0, x1, y1, z1);
        }
        const float ddx = hx - sx, ddy = hy - sy, ddz = hz - sz;
        const float srx = __builtin_amdgcn_rcpf(ddx), sry = __builtin_amdgcn_rcpf(ddy), srz = __builtin_amdgcn_rcpf(ddz);
        const unsigned long long vis_index = (static_cast<unsigned long long>(item_at) * static_cast<unsigned long long>(lslots) + static_cast<unsigned long long>(l)) * P + pass;
        if (!CONT) c_rays += static_cast<uint32_t>(__popcll(__ballot(valid)));      // wave-uniform: a scalar register, not a lane counter held (and spilled) across the walk
        ShaftLanes SL = make_shaft_lanes(lane, hx, hy, hz, fminf(x0, x1), fminf(y0, y1), fminf(z0, z1), fmaxf(x0, x1), fmaxf(y0, y1), fmaxf(z0, z1), S.extent, RT_SHAFT_TRUNC);
        __builtin_amdgcn_wave_barrier();
        shaft_tri_store(sl.tri, lane, SL, hx, hy, hz, fminf(x0, x1), fminf(y0, y1), fminf(z0, z1), fmaxf(x0, x1), fmaxf(y0, y1), fmaxf(z0, z1));
        shaft_lanes_store(sl.shaft, lane, SL);
        __builtin_amdgcn_wave_barrier();
        ShaftCtl SC{SL.pad, true};
        SC.node_ok = __ballot(valid && !(fabsf(ddx) > 0.0f && fabsf(ddy) > 0.0f && fabsf(ddz) > 0.0f && fabsf(ddx) + fabsf(ddy) + fabsf(ddz) < 3e38f)) == 0ull;
        const RayLane R{sx, sy, sz, ddx, ddy, ddz, srx, sry, srz, 4e-4f * (fabsf(sx) + fabsf(sy) + fabsf(sz) + S.extent)};
        bool occ = false;
#ifdef RT_PROFILE
        pclk.to(0);
#endif
        // leaf tasks only pay when the launch has few units per wave (k_shadow has the numbers)
        const bool tasks_on = TASKS && Q.tasks_out != nullptr && Q.budget != 0u && units < 256u * gridDim.x * RT_WAVES;
        const uint32_t tsh = blockIdx.x & (RT_LIST_SHARDS - 1u), tcap = Q.cap / RT_LIST_SHARDS;          // sharded task queue (Control::n_task_sh)
        if (CONT) {
            // one chunk range of one big leaf for the rays that reached it; rays another piece already found occluded are dropped
            const ShaftTasks TQ{nullptr, nullptr, 0u, 0u, 0u, unit};
            const unsigned long long live = t_mask & uniform_u64(vis[vis_index]);
            if (live != 0ull) {
                const DNode leaf = nodes[t_node];
                shaft_leaf<false>(t_node, uniform_u32(leaf.first), uniform_u32(leaf.count_flags) & 0x7fffffffu, uniform_u32(leaf.pad[0]), tris, chunks, lane, R, SC, TQ,
                           sl, t_cb, t_ce, live, occ);
            }
            const unsigned long long om = __ballot(valid && occ);
            if (lane == 0 && om != 0ull) atomicAnd(&vis[vis_index], ~om);
        } else {
            // (the root record comes from the LDS copy of the top of the tree at each use: held in 16 scalar registers for the kernel's lifetime it was
            //  spilled and restored around every unit -- dodge 1.079 -> 1.052 ms, cfg4 26.8 -> 26.2 ms.  Parking the shard map and the lane constants in
            //  LDS as well cut the spilled VGPRs from 14 to 10 and LOST 1.5 %: the unit prologue waits for every extra LDS read)
            const DNode rootl = sl.nodes[0];
            const bool sroot = valid && box_hit_verified(rootl.bmin, sx, sy, sz, ddx, ddy, ddz, srx, sry, srz);
            const ShaftTasks TQ{Q.tasks_out + tsh * tcap, &ctl->n_task_sh[level][tsh * 16u], tcap, tasks_on ? Q.budget : 0u, Q.target ? Q.target : Q.budget, unit};
            shaft_walk<TASKS>(nodes, tris, chunks, stk, sl, lane, rootl, sroot, R, srx, sry, srz, SC, TQ, occ);
            const unsigned long long vm = __ballot(valid && !occ);
            if (lane == 0) vis[vis_index] = vm;
        }
#ifdef RT_PROFILE
        pclk.to(7);
#endif
#ifdef RT_UNIT_HIST
        if (!CONT && !TASKS && level == 0 && S.dbg != nullptr && (work & ((1u << RT_UNIT_STRIDE) - 1u)) == 0u && (work >> RT_UNIT_STRIDE) < 65536u) {
            // unit record (kernel block 4 of the debug buffer): cycles / 16, the eight step counters of g_uh, occluded and valid rays -- 11 words of a 16-word slot
            const unsigned long long dtu = (static_cast<unsigned long long>(clock64()) - uh_t0) >> 4;
            const uint32_t n_occ = static_cast<uint32_t>(__popcll(__ballot(valid && occ))), n_val = static_cast<uint32_t>(__popcll(__ballot(valid)));
            __builtin_amdgcn_wave_barrier();
            if (lane < 11) S.dbg[RT_UNIT_DBG_WORDS + static_cast<size_t>(work >> RT_UNIT_STRIDE) * 16u + static_cast<uint32_t>(lane)] =
                lane == 0 ? static_cast<uint32_t>(dtu) : (lane == 9 ? n_occ : (lane == 10 ? n_val : g_uh[wave * 8 + lane - 1]));
        }
#endif
#ifdef RT_PROFILE
        if (lane == 0 && g_prof) {      // per-unit duration histogram: prof[560 + log2(10 ns ticks)], max prof[559]
            const unsigned long long tu = __builtin_amdgcn_s_memrealtime();
            const unsigned long long dtu = tu - unit_t0;
            int b = 63 - __builtin_clzll(dtu | 1ull); if (b > 30) b = 30;
            atomicAdd(&g_prof[560 + b], 1ull);
            atomicMax(&g_prof[559], dtu);
        }
#endif
    }
#ifdef RT_PROFILE
    pclk.flush(lane, 480);
    const uint32_t c_rays_dbg = c_rays;
    if (lane == 0 && g_prof) {
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        atomicMax(&g_prof[101], t1);
        atomicAdd(&g_prof[102], t1 - wave_t0);
        atomicAdd(&g_prof[103], 1ull);
        const unsigned long long ref = g_prof[104];        // launch reference written by the first wave that ends (approximate origin)
        if (ref == 0ull) atomicCAS(&g_prof[104], 0ull, wave_t0);
        const unsigned long long org = g_prof[104] ? g_prof[104] : wave_t0;
        unsigned long long bin = (t1 > org ? t1 - org : 0ull) / 5000ull;   // 50 us bins
        if (bin > 39ull) bin = 39ull;
        atomicAdd(&g_prof[110 + bin], 1ull);
        unsigned long long sbin = (wave_t0 > org ? wave_t0 - org : 0ull) / 5000ull;
        if (sbin > 39ull) sbin = 39ull;
        atomicAdd(&g_prof[520 + sbin], 1ull);
#ifdef RT_PROFILE_XCC            // (per-XCD end-time bins: [160, 480) now holds the unit-duration histograms of the trace stages)
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        atomicAdd(&g_prof[160 + (xcc & 7u) * 40u + bin], 1ull);
        atomicAdd(&g_prof[500 + (xcc & 7u)], static_cast<unsigned long long>(c_rays_dbg));
#else
        (void)c_rays_dbg;
#endif
    }
#endif
    if (lane == 0 && c_rays) {
        atomicAdd(&ctl->stat[blockIdx.x & (RT_STAT_SHARDS - 1)][ST_RAYS_SAMPLE], static_cast<unsigned long long>(c_rays));
        atomicAdd(&ctl->stat[blockIdx.x & (RT_STAT_SHARDS - 1)][ST_SAMPLE_WALKED], static_cast<unsigned long long>(c_rays));     // the shaft walk forms every segment of its units
    }
}

// ======================================================================================================
// K1.5 BEAM TEST -- one wave per tile of 64 consecutive lit hits (a k_shade tile: neighbouring pixels of a primary tile, neighbouring
// bounce rays), before any sample shadow ray is formed.
//
// The sample segments of every hit of the tile to scene light l lie in the BEAM hull(S, H): S the box of the light's samples, H the box
// of the tile's hit points.  With h_c the centre of H and e its half extent, hull(S, h') is contained in hull(S, h_c) + [-e, e] for
// every h' in H (the same parameter on the two segments s -> h' and s -> h_c gives points that differ by at most e per axis), so a box or
// triangle neighbourhood GROWN by e that lies outside the shaft of (S, h_c) is outside the shaft of every hit of the tile.  The shaft
// tests of the unit walk are therefore reused as they are: the lanes' coefficients are built for h_c and then relaxed by e
// (shaft_inflate).  The tree is walked by content box only (a counted hit lies in the content box of every ancestor of its leaf and in
// its chunk box, whatever the reference's own box tests say), chunk bounds, then triangle by triangle; the plane rules take the
// interval of h.n over H (plane_rules_out_box).  If NO triangle of the scene survives, every sample of every hit of the tile sees the
// light: the visibility words are written here and the hits never become shadow units.  One surviving triangle, an uncullable chunk or a
// hit with a light list of its own (mirror bounce) and the items go to the shadow kernels through the compacted index list `sidx`.
// ======================================================================================================
// the decision of plane_rules_out for EVERY (s, h), s in the sample box, h in [hlo, hhi]; m0 as there, over both boxes
__device__ __forceinline__ bool plane_rules_out_box(const float slx, const float sly, const float slz, const float shx, const float shy, const float shz,
                                                    const float hlx, const float hly, const float hlz, const float hhx, const float hhy, const float hhz,
                                                    const float m0, const float nx, const float ny, const float nz, const float nA) {
    const float ax = nx * slx, bx = nx * shx, ay = ny * sly, by = ny * shy, az = nz * slz, bz = nz * shz;
    const float sn_lo = fminf(ax, bx) + (fminf(ay, by) + fminf(az, bz));
    const float sn_hi = fmaxf(ax, bx) + (fmaxf(ay, by) + fmaxf(az, bz));
    const float cx = nx * hlx, dx = nx * hhx, cy = ny * hly, dy = ny * hhy, cz = nz * hlz, dz = nz * hhz;
    const float hn_lo = fminf(cx, dx) + (fminf(cy, dy) + fminf(cz, dz));
    const float hn_hi = fmaxf(cx, dx) + (fmaxf(cy, dy) + fmaxf(cz, dz));
    const float num_lo = nA - sn_hi, num_hi = nA - sn_lo;                 // num = n.A - s.n
    const float dn_lo = hn_lo - sn_hi, dn_hi = hn_hi - sn_lo;             // dn = (h - s).n, s and h independent
    const float M = m0 + 2e-5f * fabsf(nA);
    const bool sane = (fabsf(nx) + fabsf(ny) + fabsf(nz) <= 4.0f) && (fabsf(nA) <= 1e30f);
    const bool opposite = (num_lo > M && dn_hi < -M) || (num_hi < -M && dn_lo > M);
    const float min_abs_num = fmaxf(num_lo, -num_hi);              // <= 0 when the interval straddles zero
    const float max_abs_dn = fmaxf(fabsf(dn_lo), fabsf(dn_hi));
    const bool beyond = (min_abs_num - M) >= 0.981f * (max_abs_dn + M);
    // the same s in both: dn = num + e with e = h.n - n.A, so |e| <= 0.018 |num| puts t = num / dn into [0.982, 1.019] -- not counted
    // (lightStrikes wants t < 0.98).  This is the face the hits lie on and its near-coplanar neighbours, at any light orientation.
    const float e_max = fmaxf(fabsf(hn_lo - nA), fabsf(hn_hi - nA));
    const float a_min = min_abs_num - M;
    const bool near_plane = (a_min > 0.0f) && (e_max + 2.0f * M <= 0.018f * a_min);
    return sane && (num_lo <= num_hi) && (dn_lo <= dn_hi) && (hn_lo <= hn_hi) && (opposite || beyond || near_plane);
}
// relaxes this lane's test by the half extent e of the apex box (see above): a plane value over a box grown by e is lower by
// sum |a_k| e_k; the near box grows by e.  The far tests are not used by beams.
__device__ __forceinline__ void shaft_inflate(ShaftLanes &SL, const int tk, const float ex, const float ey, const float ez) {
    if (tk < 6) {
        const float ax = SL.r[0] + SL.r[1], ay = SL.r[2] + SL.r[3], az = SL.r[4] + SL.r[5];
        SL.r[6] = SL.r[6] - ((fabsf(ax) * ex + fabsf(ay) * ey) + fabsf(az) * ez) * 1.0001f;
    } else if (tk == 6) {
        SL.r[0] -= ex; SL.r[1] -= ey; SL.r[2] -= ez; SL.r[3] += ex; SL.r[4] += ey; SL.r[5] += ez;
    }
}
// ---- lane-local shaft of ONE hit: every lane builds the six tangent planes of its own (S, h) -- the construction of make_shaft_lanes with
// the projection and the tangent as compile-time constants -- and tests wave-uniform boxes against them.  Used by k_beam for the few
// leaves that hold a chunk which may never be culled: can a ray from the light's samples to this hit, or its continuation behind the hit
// (boxIntersect has no upper bound), enter the leaf's own box?  If not, the reference never looks at that leaf for this hit.
struct ItemPlane { float a, b, c, far_margin; };     // a*u + b*v + c over the projection (u, v); unusable: never separates
template <int PROJ, bool Q1>
__device__ __forceinline__ ItemPlane item_plane(const float hx, const float hy, const float hz, const float slx, const float sly, const float slz,
                                                const float shx, const float shy, const float shz, const float scale) {
    const float hu = PROJ == 1 ? hy : hx, hv = PROJ == 2 ? hy : hz;
    const float u0 = PROJ == 1 ? sly : slx, u1 = PROJ == 1 ? shy : shx;
    const float v0 = PROJ == 2 ? sly : slz, v1 = PROJ == 2 ? shy : shz;
    const bool ul = hu < u0, ur = hu > u1, vl = hv < v0, vr = hv > v1;
    const bool su0 = !(ul || ur), sv0 = !(vl || vr);
    const float near_u = ul ? u0 : u1, far_u = ul ? u1 : u0, near_v = vl ? v0 : v1, far_v = vl ? v1 : v0;
    const float cu = Q1 ? (su0 ? u1 : (sv0 ? near_u : far_u)) : (su0 ? u0 : near_u);
    const float cv = Q1 ? (sv0 ? v1 : near_v) : (sv0 ? v0 : (su0 ? near_v : far_v));
    const float mu = 0.5f * (u0 + u1) - hu, mv = 0.5f * (v0 + v1) - hv;
    const float nu = -(cv - hv), nv = cu - hu;
    const float fm = nu * mu + nv * mv;
    const float mag = fabsf(nu) + fabsf(nv);
    const bool ok = !(su0 && sv0) && (fabsf(fm) > 1e-5f * mag * scale);
    const float sgn = fm > 0.0f ? -1.0f : 1.0f;
    const float margin = 2e-5f * mag * scale;
    ItemPlane p;
    p.a = ok ? sgn * nu : 0.0f; p.b = ok ? sgn * nv : 0.0f;
    p.c = ok ? -(p.a * hu + p.b * hv) - margin : -1.0f;
    p.far_margin = ok ? 2.0f * margin : 3e38f;
    return p;
}
// box [lo, hi] (already padded) against one plane: near = outside the shaft by this plane, far = outside the far cone by it
template <int PROJ>
__device__ __forceinline__ void item_plane_test(const ItemPlane &p, const float lx, const float ly, const float lz, const float hx, const float hy, const float hz,
                                                bool &near_out, bool &far_out) {
    const float ul = PROJ == 1 ? ly : lx, uh = PROJ == 1 ? hy : hx, vl = PROJ == 2 ? ly : lz, vh = PROJ == 2 ? hy : hz;
    const float ap = fmaxf(p.a, 0.0f), an = fminf(p.a, 0.0f), bp = fmaxf(p.b, 0.0f), bn = fminf(p.b, 0.0f);
    const float mn = __builtin_fmaf(ap, ul, __builtin_fmaf(an, uh, __builtin_fmaf(bp, vl, __builtin_fmaf(bn, vh, p.c))));
    const float mx = __builtin_fmaf(ap, uh, __builtin_fmaf(an, ul, __builtin_fmaf(bp, vh, __builtin_fmaf(bn, vl, p.c))));
    near_out = near_out || (mn > 0.0f);
    far_out = far_out || (mx + p.far_margin < 0.0f);
}

#ifndef RT_BEAM_BUDGET
#define RT_BEAM_BUDGET 1024               // group steps + chunk bound batches + chunks tested triangle by triangle, per beam
#endif
#ifndef RT_PAIR_WPE
#define RT_PAIR_WPE 6
#endif
#define RT_BEAM_REC 13                 // float4 per wave: shaft_tri_store's 11 (planes, near box, (h_c, m0), S lo, S hi) + H lo, H hi

// lane = triangle: can ANY segment of the beam hit this triangle with a counted t?  (planes and near box from the relaxed record)
__device__ __forceinline__ bool tri_outside_beam(const float4 *rec, const TriRec &tr, const float m) {
    if (tri_outside_cone(rec, tr, m, true)) return true;
    const float4 hm = rec[8], s0 = rec[9], s1 = rec[10], h0 = rec[11], h1 = rec[12];
    return plane_rules_out_box(s0.x, s0.y, s0.z, s1.x, s1.y, s1.z, h0.x, h0.y, h0.z, h1.x, h1.y, h1.z, hm.w, tr.nx, tr.ny, tr.nz, tr.nA);
}

// one leaf against the beam: true = some triangle may be hit (or a chunk that may never be culled is in the way)
__device__ __forceinline__ bool beam_leaf(const uint32_t first, const uint32_t cnt, const uint32_t chunk0, const TriRec *__restrict__ tris,
                                          const ChunkBound *__restrict__ chunks, const int lane, const float4 *shaft, const float4 *rec, const ShaftCtl &SC,
                                          const bool per_item, int &budget) {
    const TriRec *__restrict__ T = tris + first;
    const ChunkBound *__restrict__ cbounds = chunks + chunk0;
    const uint32_t nchunk = (cnt + 63u) >> 6;
    const int tk = lane & 7, tc = lane >> 3;
    for (uint32_t cb0 = 0u; cb0 < nchunk; cb0 += 8u) {
        const uint32_t myc = cb0 + static_cast<uint32_t>(tc);
        const ChunkBound bd = cbounds[myc < nchunk ? myc : cb0];
        const ShaftLanes SL = shaft_lanes_load(shaft, tk, SC);
        bool near_out, far_out;
        shaft_lane_test(SL, tk, bd.lo[0], bd.lo[1], bd.lo[2], bd.hi[0], bd.hi[1], bd.hi[2], near_out, far_out);
        const unsigned long long b_out = __ballot(near_out && bd.never < 1.5f);
        const uint32_t nhere = nchunk - cb0 < 8u ? nchunk - cb0 : 8u;
        unsigned long long cm = __ballot(static_cast<uint32_t>(lane) < nhere && !ballot_byte_any(b_out, lane));      // bit j: chunk cb0 + j is in the beam
        // a chunk that may never be culled: its leaf is tested per hit afterwards (S.bad_leaves) -- or, without that list, it blocks the beam
        const unsigned long long nv = __ballot(static_cast<uint32_t>(lane) < nhere && ((cm >> lane) & 1ull) != 0ull && __shfl(bd.never, 8 * (lane & 7), 64) >= 1.5f);
        if (nv != 0ull && !per_item) return true;
        cm &= ~nv;
        budget -= 1 + static_cast<int>(__popcll(cm));
        RT_PROF_ADD(lane, 85, 1); RT_PROF_ADD(lane, 86, __popcll(cm)); if (cb0 == 0u) RT_PROF_ADD(lane, 84, 1);
        if (budget < 0) return true;                                       // too much work for one wave: let the shadow units decide
        while (cm != 0ull) {
            const int j = static_cast<int>(__builtin_ctzll(cm));
            cm &= cm - 1ull;
            const uint32_t c0 = (cb0 + static_cast<uint32_t>(j)) * 64u;
            const uint32_t k = c0 + static_cast<uint32_t>(lane);
            const TriRec tr = T[k < cnt ? k : 0u];
            const bool hast = k < cnt && !(tr.flags & 1u);                        // (illum 9 faces never occlude: lightStrikes skips them)
            if (__ballot(hast && !tri_outside_beam(rec, tr, lane_f(bd.infl, 8 * j) * 1.0625f)) != 0ull) return true;
        }
    }
    return false;
}

// no sample segment of light (px, py, pz) to the hit h has a zero direction component (0/0 = NaN makes the reference's min/max chain accept any box)
__device__ __forceinline__ bool beam_dirs_ok(const DLights &L, const LightGrid &lg, const uint32_t N, const float px, const float py, const float pz,
                                             const float hx, const float hy, const float hz) {
    bool dirs_ok = fabsf(hx) + fabsf(hy) + fabsf(hz) < 1e30f;
    if (L.mode == RT_LIGHT_SPHERE) {
        for (uint32_t k = 0u; k < N; ++k) {
            float sx, sy, sz;
            sphere_sample(L, k, px, py, pz, sx, sy, sz);
            dirs_ok = dirs_ok && (hx - sx != 0.0f) && (hy - sy != 0.0f) && (hz - sz != 0.0f);
        }
    } else {
        float sx, sy, sz;
        for (int i = 0; i < L.usteps; ++i) { grid_sample(lg, static_cast<float>(i) + 0.5f, 0.5f, sx, sy, sz); dirs_ok = dirs_ok && (hx - sx != 0.0f); }
        for (int j = 0; j < L.vsteps; ++j) { grid_sample(lg, 0.5f, static_cast<float>(j) + 0.5f, sx, sy, sz); dirs_ok = dirs_ok && (hy - sy != 0.0f); }
        dirs_ok = dirs_ok && (hz - sz != 0.0f);
    }
    return dirs_ok;
}

// the same for a wave-uniform hit, the samples spread over the lanes
__device__ __forceinline__ bool beam_dirs_ok_wave(const DLights &L, const LightGrid &lg, const uint32_t N, const float px, const float py, const float pz,
                                                  const float hx, const float hy, const float hz, const int lane) {
    bool bad = !(fabsf(hx) + fabsf(hy) + fabsf(hz) < 1e30f);
    if (L.mode == RT_LIGHT_SPHERE) {
        for (uint32_t k = static_cast<uint32_t>(lane); k < N; k += 64u) {
            float sx, sy, sz;
            sphere_sample(L, k, px, py, pz, sx, sy, sz);
            bad = bad || !((hx - sx != 0.0f) && (hy - sy != 0.0f) && (hz - sz != 0.0f));
        }
    } else {
        float sx, sy, sz;
        for (int i = lane; i < L.usteps; i += 64) { grid_sample(lg, static_cast<float>(i) + 0.5f, 0.5f, sx, sy, sz); bad = bad || !(hx - sx != 0.0f); }
        for (int j = lane; j < L.vsteps; j += 64) { grid_sample(lg, 0.5f, static_cast<float>(j) + 0.5f, sx, sy, sz); bad = bad || !(hy - sy != 0.0f); }
        grid_sample(lg, 0.5f, 0.5f, sx, sy, sz);
        bad = bad || !(hz - sz != 0.0f);
    }
    return __ballot(bad) == 0ull;
}

struct BeamCtx {
    const DNode *nodes; const TriRec *tris; const ChunkBound *chunks;
    uint32_t *stack; float4 *rec; float4 *shaft;           // per-wave LDS
    uint32_t *yield; bool brake, per_item, blocks;
    uint32_t N, P, item_cap; int lslots, level;
    float fi_last, fj_last;
    const DNode *lds_nodes; uint32_t n_lds;                // the top of the tree in LDS (k_pair_beam), or (nullptr, 0)
};
// The walk of one beam: groups of children by content box (and, for a beam with ONE hit, by own box against the shaft and the far cone behind the
// hit: a child outside both is entered by no sample ray -- shaft_walk's rule, with its condition `far_ok`: no ray with a zero / non-finite
// direction component), leaves chunk by chunk, triangle by triangle.  true = something may block a segment of the beam (or the budget ran out).
__device__ __forceinline__ bool beam_walk(const BeamCtx &B, const DScene &S, const DNode &root, const int lane, const ShaftCtl &SC, const bool far_ok, int &budget) {
    const DNode *__restrict__ nodes = B.nodes; const TriRec *__restrict__ tris = B.tris; const ChunkBound *__restrict__ chunks = B.chunks;
    uint32_t *const stack = B.stack; float4 *const rec = B.rec; float4 *const shaft = B.shaft;
    const bool per_item = B.per_item;
    const int tk = lane & 7, tc = lane >> 3;
    bool blocked = false;
    int sp = 0;
    if (root.count_flags & RT_NODE_LEAF) {
        blocked = (root.count_flags & 0x7fffffffu) != 0u &&
                  beam_leaf(uniform_u32(root.first), uniform_u32(root.count_flags) & 0x7fffffffu, uniform_u32(root.pad[0]), tris, chunks, lane, shaft, rec, SC, per_item, budget);
    } else if ((root.count_flags & 0xfu) != 0u) {
        if (lane == 0) stack[0] = root.first | ((root.count_flags & 0xfu) << 28);
        sp = 1;
    }
    while (sp > 0 && !blocked) {
        --sp;
        __builtin_amdgcn_wave_barrier();
        const uint32_t ent = uniform_u32(stack[sp]);
        const uint32_t base = ent & 0x0fffffffu, gcnt = ent >> 28;
        const uint32_t ci = base + (static_cast<uint32_t>(tc) < gcnt ? static_cast<uint32_t>(tc) : 0u);
        DNode ch;
        if (base + gcnt <= B.n_lds) ch = B.lds_nodes[ci];
        else ch = nodes[ci];
        const ShaftLanes SLg = shaft_lanes_load(shaft, tk, SC);
        bool c_near, c_far;
        shaft_lane_test(SLg, tk, ch.clo[0] - SC.pad, ch.clo[1] - SC.pad, ch.clo[2] - SC.pad, ch.chi[0] + SC.pad, ch.chi[1] + SC.pad, ch.chi[2] + SC.pad, c_near, c_far);
        // (per_item: the content box bounds the cullable chunks below; the others are tested per hit afterwards)
        const unsigned long long b_c = __ballot(c_near && (ch.pad[1] == 0u || per_item));
        bool culled = ballot_byte_any(b_c, lane);
        if (far_ok) {
            bool n_near, n_far;
            shaft_lane_test(SLg, tk, ch.bmin[0] - SC.pad, ch.bmin[1] - SC.pad, ch.bmin[2] - SC.pad, ch.bmax[0] + SC.pad, ch.bmax[1] + SC.pad, ch.bmax[2] + SC.pad, n_near, n_far, false);
            const unsigned long long b_nn = __ballot(n_near), b_nf = __ballot(n_far);
            culled = culled || (ballot_byte_any(b_nn, lane) && ballot_byte_any(b_nf, lane));
        }
        unsigned long long surv = __ballot(static_cast<uint32_t>(lane) < gcnt && !culled);
        if (--budget < 0) blocked = true;
        RT_PROF_ADD(lane, 82, 1); RT_PROF_ADD(lane, 83, __popcll(surv));
        while (surv != 0ull && !blocked) {
            const int j = static_cast<int>(__builtin_ctzll(surv));
            surv &= surv - 1ull;
            const uint32_t cf = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(ch.count_flags), 8 * j));
            const uint32_t ff = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(ch.first), 8 * j));
            if (cf & RT_NODE_LEAF) {
                const uint32_t lc = cf & 0x7fffffffu;
                if (lc != 0u) blocked = beam_leaf(ff, lc, static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(ch.pad[0]), 8 * j)), tris, chunks, lane, shaft, rec, SC, per_item, budget);
            } else if ((cf & 0xfu) != 0u) {
                if (lane == 0) stack[sp] = ff | ((cf & 0xfu) << 28);
                ++sp;
            }
        }
    }
    return blocked;
}

// (Measured and rejected: running this inside the flat k_trace on the tile's own hits -- no k_beam launch, no second pass over the items.
// The kernel grows from ~90 to 145 VGPRs, its primary launch from 57 to 84 us, and the frame stays at 0.40 ms.)
// One tile of lit hits (lane = hit: `have`, its item storage index, hit point and light mode) through the beam test: writes the visibility
// words of the hits nothing can block, appends the others to the survivor list, accounts the sample rays of the former.
__device__ __forceinline__ void beam_tile(const BeamCtx &B, const DScene &S, const DLights &L, const DNode &root, Control *__restrict__ ctl,
                                          unsigned long long *__restrict__ vis, uint32_t *__restrict__ sidx, const int lane, const uint32_t tile,
                                          const bool have, const uint32_t idx, const float hx, const float hy, const float hz, const uint32_t lmode, uint32_t &c_rays) {
    float4 *const rec = B.rec; float4 *const shaft = B.shaft; uint32_t *const yield = B.yield;
    const bool brake = B.brake, per_item = B.per_item, blocks = B.blocks;
    const uint32_t N = B.N, P = B.P, item_cap = B.item_cap;
    const int lslots = B.lslots, level = B.level;
    const float fi_last = B.fi_last, fj_last = B.fj_last;
    const int tk = lane & 7;
    const bool scene = have && lmode == 0u;              // sees the scene lights (a mirror bounce carries a light list of its own)
    bool survive = have && !scene;
    const unsigned long long sm0 = __ballot(scene);
    bool give_up = false;
    if (brake) {
        const uint32_t y_t = uniform_u32(__hip_atomic_load(&yield[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        const uint32_t y_u = uniform_u32(__hip_atomic_load(&yield[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        give_up = y_t >= 16u && y_u * 4u < y_t;
    }
    if (give_up) survive = have;
    if (sm0 != 0ull && !give_up) {
        // H: exact wave min / max of the hit points
        float lx = scene ? hx : 3e38f, ly = scene ? hy : 3e38f, lz = scene ? hz : 3e38f;
        float ux = scene ? hx : -3e38f, uy = scene ? hy : -3e38f, uz = scene ? hz : -3e38f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lx = fminf(lx, __shfl_xor(lx, o, 64)); ly = fminf(ly, __shfl_xor(ly, o, 64)); lz = fminf(lz, __shfl_xor(lz, o, 64));
            ux = fmaxf(ux, __shfl_xor(ux, o, 64)); uy = fmaxf(uy, __shfl_xor(uy, o, 64)); uz = fmaxf(uz, __shfl_xor(uz, o, 64));
        }
        const float cx = 0.5f * (lx + ux), cy = 0.5f * (ly + uy), cz = 0.5f * (lz + uz);
        // half extent around the ROUNDED centre, rounded up
        const float ex = fmaxf(ux - cx, cx - lx) * 1.0001f + 1e-7f * S.extent, ey = fmaxf(uy - cy, cy - ly) * 1.0001f + 1e-7f * S.extent,
                    ez = fmaxf(uz - cz, cz - lz) * 1.0001f + 1e-7f * S.extent;
        for (int l = 0; l < L.n_lights; ++l) {
            float x0, y0, z0, x1, y1, z1;
            const float px = L.pos[l][0], py = L.pos[l][1], pz = L.pos[l][2];
            const LightGrid lg = light_grid(L, px, py, pz);
            grid_sample(lg, 0.5f, 0.5f, x0, y0, z0);             // the samples are monotone in each grid index: two corners give the exact box
            grid_sample(lg, fi_last, fj_last, x1, y1, z1);
            if (L.mode == RT_LIGHT_SPHERE) sphere_box(L, px, py, pz, x0, y0, z0, x1, y1, z1);
            const float slx = fminf(x0, x1), sly = fminf(y0, y1), slz = fminf(z0, z1), shx = fmaxf(x0, x1), shy = fmaxf(y0, y1), shz = fmaxf(z0, z1);
            ShaftLanes SL = make_shaft_lanes(lane, cx, cy, cz, slx, sly, slz, shx, shy, shz, S.extent);
            shaft_inflate(SL, tk, ex, ey, ez);
            const ShaftCtl SC{SL.pad, false};
            __builtin_amdgcn_wave_barrier();
            shaft_tri_store(rec, lane, SL, cx, cy, cz, slx, sly, slz, shx, shy, shz);
            shaft_lanes_store(shaft, lane, SL);
            if (lane == 8) {
                const float m0 = 2e-5f * (((fabsf(slx) + fabsf(shx)) + (fabsf(sly) + fabsf(shy)) + (fabsf(slz) + fabsf(shz))) +
                                          ((fabsf(lx) + fabsf(ux)) + (fabsf(ly) + fabsf(uy)) + (fabsf(lz) + fabsf(uz))));
                rec[8] = make_float4(cx, cy, cz, m0); rec[11] = make_float4(lx, ly, lz, 0.f); rec[12] = make_float4(ux, uy, uz, 0.f);
            }
            __builtin_amdgcn_wave_barrier();
            int budget = S.beam_budget;
            const bool blocked = beam_walk(B, S, root, lane, SC, false, budget);
            RT_PROF_ADD(lane, 76, 1); RT_PROF_ADD(lane, 77, blocked ? 0 : 1); RT_PROF_ADD(lane, 80, blocked ? 0 : S.beam_budget - budget); RT_PROF_ADD(lane, 81, budget < 0 ? 1 : 0);
            if (brake && lane == 0) { atomicAdd(&yield[0], 1u); if (!blocked) atomicAdd(&yield[1], 1u); }
            // the leaves with a chunk that may never be culled (degenerate triangles whose computed barycentrics are noise): can a ray to
            // THIS hit -- or its continuation behind the hit -- enter the leaf's own box?  Lane-local shaft of (S, h): if the padded box is
            // outside the near shaft by one plane AND outside the far cone by one, the reference's boxIntersect fails for every
            // sample ray of the hit by a margin far above its rounding, so that leaf is never looked at for it.  (Not valid with a zero
            // direction component: 0/0 = NaN makes the reference's min/max chain accept -- such hits stay with the shadow units.)
            bool reach = false;
            if (!blocked && per_item && S.n_bad_leaves != 0u) {
                const bool dirs_ok = beam_dirs_ok(L, lg, N, px, py, pz, hx, hy, hz);
                const float big = fmaxf(fmaxf(fabsf(slx), fabsf(shx)), fmaxf(fabsf(sly), fabsf(shy))) + fmaxf(fabsf(slz), fabsf(shz));
                const float scale = S.extent + big + (fabsf(hx) + fabsf(hy) + fabsf(hz));
                const float pad = 4e-4f * ((fmaxf(fabsf(slx), fabsf(shx)) + fmaxf(fabsf(sly), fabsf(shy)) + fmaxf(fabsf(slz), fabsf(shz))) + S.extent) * 1.001f;
                const ItemPlane p0 = item_plane<0, false>(hx, hy, hz, slx, sly, slz, shx, shy, shz, scale), p1 = item_plane<0, true>(hx, hy, hz, slx, sly, slz, shx, shy, shz, scale);
                const ItemPlane p2 = item_plane<1, false>(hx, hy, hz, slx, sly, slz, shx, shy, shz, scale), p3 = item_plane<1, true>(hx, hy, hz, slx, sly, slz, shx, shy, shz, scale);
                const ItemPlane p4 = item_plane<2, false>(hx, hy, hz, slx, sly, slz, shx, shy, shz, scale), p5 = item_plane<2, true>(hx, hy, hz, slx, sly, slz, shx, shy, shz, scale);
                // near box: AABB of hull(S, h); far box: AABB of the far cone { h + tau (h - s) }
                const float nlx = fminf(slx, hx), nly = fminf(sly, hy), nlz = fminf(slz, hz), nhx = fmaxf(shx, hx), nhy = fmaxf(shy, hy), nhz = fmaxf(shz, hz);
                const float flx = hx >= shx ? hx : -3e38f, fly = hy >= shy ? hy : -3e38f, flz = hz >= shz ? hz : -3e38f;
                const float fhx = hx <= slx ? hx : 3e38f, fhy = hy <= sly ? hy : 3e38f, fhz = hz <= slz ? hz : 3e38f;
                for (uint32_t b = 0u; b < S.n_bad_leaves; ++b) {
                    const float *bb = S.bad_leaves + 6u * b;
                    const float lx_ = bb[0] - pad, ly_ = bb[1] - pad, lz_ = bb[2] - pad, hx_ = bb[3] + pad, hy_ = bb[4] + pad, hz_ = bb[5] + pad;
                    bool near_out = (lx_ > nhx) || (hx_ < nlx) || (ly_ > nhy) || (hy_ < nly) || (lz_ > nhz) || (hz_ < nlz);
                    bool far_out = (lx_ > fhx) || (hx_ < flx) || (ly_ > fhy) || (hy_ < fly) || (lz_ > fhz) || (hz_ < flz);
                    item_plane_test<0>(p0, lx_, ly_, lz_, hx_, hy_, hz_, near_out, far_out); item_plane_test<0>(p1, lx_, ly_, lz_, hx_, hy_, hz_, near_out, far_out);
                    item_plane_test<1>(p2, lx_, ly_, lz_, hx_, hy_, hz_, near_out, far_out); item_plane_test<1>(p3, lx_, ly_, lz_, hx_, hy_, hz_, near_out, far_out);
                    item_plane_test<2>(p4, lx_, ly_, lz_, hx_, hy_, hz_, near_out, far_out); item_plane_test<2>(p5, lx_, ly_, lz_, hx_, hy_, hz_, near_out, far_out);
                    reach = reach || !(dirs_ok && near_out && far_out);
                }
                RT_PROF_ADD(lane, 78, __popcll(__ballot(scene))); RT_PROF_ADD(lane, 79, __popcll(__ballot(scene && reach)));
            }
            if (blocked) {
                survive = survive || scene;
            } else if (scene && reach) {
                survive = true;
            } else if (scene) {
                // nothing can block any sample segment of these hits to light l: all N samples visible
                const unsigned long long slot0 = (static_cast<unsigned long long>(idx) * static_cast<unsigned long long>(lslots) + static_cast<unsigned long long>(l)) * P;
                for (uint32_t p = 0u; p < P; ++p) {
                    const uint32_t left = N - p * 64u;
                    vis[slot0 + p] = (blocks || left >= 64u) ? ~0ull : ((1ull << left) - 1ull);
                }
            }
        }
    }
    // hits that need no shadow unit at all: their sample rays are accounted for here (the shadow kernels count the others)
    c_rays += (have && !survive) ? N * static_cast<uint32_t>(L.n_lights) : 0u;
    const unsigned long long sm = __ballot(survive);
    if (sm != 0ull) {
        bool fits;
        const uint32_t at = shard_reserve(ctl->n_sitems[level], &ctl->overflow, tile, static_cast<uint32_t>(__popcll(sm)), item_cap, lane, fits);
        if (survive && fits) sidx[at + lanes_below(sm)] = idx;
    }
}

__global__ __launch_bounds__(RT_WAVES * 64) void k_beam(const DNode *__restrict__ nodes, const TriRec *__restrict__ tris, const ChunkBound *__restrict__ chunks,
                                                        const DScene S, const DLights L, const int level, const int lslots, const uint32_t item_cap,
                                                        const ShadeItem *__restrict__ items, Control *__restrict__ ctl, unsigned long long *__restrict__ vis,
                                                        uint32_t *__restrict__ sidx) {
    __shared__ uint32_t s_node[RT_WAVES * RT_STACK];
    __shared__ float4 s_rec[RT_WAVES * RT_BEAM_REC];
    __shared__ float4 s_shaft[RT_WAVES * 16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *const stack = s_node + wave * RT_STACK;
    float4 *const rec = s_rec + wave * RT_BEAM_REC;
    float4 *const shaft = s_shaft + wave * 16;
    const ShardMap imap = shard_map(ctl->n_items[level], lane, item_cap, 1u, 64u);        // tiles of 64 items, shard after shard
    const uint32_t ntiles = imap.total;
    if (ntiles == 0u) return;
    const uint32_t N = static_cast<uint32_t>(L.n_samples);
    const uint32_t P = (N + 63u) / 64u;
    const bool blocks = sample_blocks(L);
    const DNode root = nodes[0];
    const float fi_last = static_cast<float>(L.usteps - 1) + 0.5f, fj_last = static_cast<float>(L.vsteps - 1) + 0.5f;
    const bool per_item = S.n_bad_leaves != 0xffffffffu;
    const bool brake = (root.count_flags & RT_NODE_LEAF) == 0u;      // (a flat scene is one leaf: always cheap to test, nothing to watch)
    uint32_t c_rays = 0;
    const uint32_t wave_id = uniform_u32(blockIdx.x * RT_WAVES + static_cast<uint32_t>(wave)), wave_count = gridDim.x * RT_WAVES;
    // The test only pays where most beams come out unblocked (a convex object, a car body under a light: 85-95 %); under grazing light over
    // a height field 9 in 10 beams meet a triangle, after a long walk.  The launch watches its own yield: every wave reports (tested,
    // unblocked) after each beam into one of 16 counter pairs, and once a pair holds 16 beams with fewer than a quarter unblocked the waves
    // that report to it pass their remaining tiles on untested.  The results do not depend on it -- an untested tile simply goes to the shadow kernels.
    uint32_t *const yield = ctl->beam_yield[level] + (blockIdx.x & (RT_LIST_SHARDS - 1u)) * 16u;      // this workgroup's shard (one returning or
                                                                                                      // non-returning atomic word takes ~88 updates per microsecond)
    const BeamCtx B{nodes, tris, chunks, stack, rec, shaft, yield, brake, per_item, blocks, N, P, item_cap, lslots, level, fi_last, fj_last, nullptr, 0u};
    for (uint32_t tile = wave_id; tile < ntiles; tile += wave_count) {
        uint32_t sh, tj, n_sh;
        shard_find(imap, tile, sh, tj, n_sh);
        const bool have = tj * 64u + static_cast<uint32_t>(lane) < n_sh;
        const uint32_t idx = sh * item_cap + tj * 64u + static_cast<uint32_t>(lane);     // item storage index (also keys vis)
        const ShadeItem it = items[have ? idx : sh * item_cap];
        const float hx = it.ox + it.t * it.dx, hy = it.oy + it.t * it.dy, hz = it.oz + it.t * it.dz;      // as the shadow kernels form it
        beam_tile(B, S, L, root, ctl, vis, sidx, lane, tile, have, idx, hx, hy, hz, it.lmode, c_rays);
    }
    c_rays = wave_sum(c_rays);
    if (lane == 0 && c_rays) atomicAdd(&ctl->stat[blockIdx.x & (RT_STAT_SHARDS - 1)][ST_RAYS_SAMPLE], static_cast<unsigned long long>(c_rays));
}

// The beam test at the granularity of ONE lit hit -- tree scenes with more than 64 samples per light, in front of k_shadow_shaft.
// A (hit, light) pair is 2 to 16 shadow units there (one per 64-sample pass), each of which builds its own shaft and walks the top of the tree
// on its own; where nothing at all lies between the hit and the light -- nine units in ten under the 16 x 16 light of the 4K height field, and
// they were half of that kernel's time -- ONE walk with the shaft of the whole light decides all passes at once: the visibility words are
// written here and the pair never becomes a unit.  The apex of the shaft is the hit itself, so (unlike a 64-hit tile, whose apex is a box)
//   * the shaft is the exact hull of the hit and the light's samples plus the rounding margins: the surface around the hit does not block it
//     (the hit's own face and its near-coplanar neighbours are ruled out by plane_rules_out_box's t > 0.981 rule);
//   * the far cone behind the hit is known, so a child whose OWN box lies outside the shaft and outside the far cone is skipped exactly as
//     shaft_walk skips it -- in the reference's over-inclusive tree (clasifyFace tests normalised vectors) the content boxes alone cull
//     almost nothing: 7.5 of 8 children survived per group, 360 steps per beam; with the own-box rule 2.7 of 8 and 11 steps (dodge).
// One wave = one hit at a time, taken from the item list as k_shadow_shaft takes its units (wave-uniform: the item is a scalar load, the hit
// lives in SGPRs): lane = (child, test) in the walk, lane = chunk / triangle in the leaves, lane = leaf in the check against the leaves that
// hold a chunk which may never be culled, lane = pass when the visibility words are written.  A beam is a chain of dependent loads like a
// shadow unit (~2 groups, ~4 leaves with their chunk bounds and triangles), so what counts is how many run at once: everything per hit is
// scalar, the kernel fits the register budget of 6 waves per SIMD.  Hits that may be blocked go to the survivor list of their OWN list
// shard (its capacity holds every item of the shard), 16 at a time through a per-wave LDS buffer.
// The brake is k_beam's: a shard's waves stop testing once it has seen 256 beams with fewer than a quarter unblocked (counters updated
// every 16 beams per wave).
#define RT_PAIR_BUF 16
__global__ __launch_bounds__(RT_WAVES * 64) __attribute__((amdgpu_waves_per_eu(RT_PAIR_WPE, 8)))
void k_pair_beam(const DNode *__restrict__ nodes, const TriRec *__restrict__ tris, const ChunkBound *__restrict__ chunks,
                 const DScene S, const DLights L, const int level, const int lslots, const uint32_t item_cap,
                 const ShadeItem *__restrict__ items, Control *__restrict__ ctl, unsigned long long *__restrict__ vis,
                 uint32_t *__restrict__ sidx, uint8_t *__restrict__ done) {
    __shared__ uint32_t s_node[RT_WAVES * RT_STACK];
    __shared__ float4 s_rec[RT_WAVES * RT_BEAM_REC];
    __shared__ float4 s_shaft[RT_WAVES * 16];
    __shared__ uint4 s_top[RT_LDS_NODES * 4];           // the top of the octree, as in k_shadow_shaft
    __shared__ float s_bad[32 * 6];                     // boxes of the leaves with a chunk that may never be culled (at most 32: rt_capi.cpp)
    __shared__ uint32_t s_buf[RT_WAVES * RT_PAIR_BUF];  // survivors waiting for their list reservation
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *const stack = s_node + wave * RT_STACK;
    float4 *const rec = s_rec + wave * RT_BEAM_REC;
    float4 *const shaft = s_shaft + wave * 16;
    uint32_t *const buf = s_buf + wave * RT_PAIR_BUF;
    const ShardMap imap = shard_map(ctl->n_items[level], lane, item_cap, 1u, 1u);
    const uint32_t n_items = imap.total;
    if (n_items == 0u) return;
    const uint32_t n_lds = S.n_nodes < RT_LDS_NODES ? S.n_nodes : RT_LDS_NODES;
    const bool per_item = S.n_bad_leaves != 0xffffffffu;
    const uint32_t n_bad = per_item ? S.n_bad_leaves : 0u;
    {
        const uint4 *__restrict__ src = reinterpret_cast<const uint4 *>(nodes);
        for (uint32_t i = threadIdx.x; i < n_lds * 4u; i += blockDim.x) s_top[i] = src[i];
        if (threadIdx.x < n_bad * 6u) s_bad[threadIdx.x] = S.bad_leaves[threadIdx.x];
        __syncthreads();
    }
    const uint32_t N = static_cast<uint32_t>(L.n_samples);
    const uint32_t P = (N + 63u) / 64u;
    const bool blocks = sample_blocks(L);
    const float fi_last = static_cast<float>(L.usteps - 1) + 0.5f, fj_last = static_cast<float>(L.vsteps - 1) + 0.5f;
    uint32_t c_rays = 0;
    const uint32_t wave_id = uniform_u32(blockIdx.x * RT_WAVES + static_cast<uint32_t>(wave)), wave_count = gridDim.x * RT_WAVES;
    uint32_t *const yield = ctl->beam_yield[level] + (blockIdx.x & (RT_LIST_SHARDS - 1u)) * 16u;
    const BeamCtx B{nodes, tris, chunks, stack, rec, shaft, yield, false, per_item, blocks, N, P, item_cap, lslots, level, fi_last, fj_last,
                    reinterpret_cast<const DNode *>(s_top), n_lds};
    const int tk = lane & 7;
    uint32_t n_buf = 0u, buf_shard = 0u;            // survivors in the buffer, all of list shard buf_shard
    uint32_t n_tested = 0u, n_unblocked = 0u, since = 0u;
    bool give_up = false;
    auto flush = [&]() {
        if (n_buf == 0u) return;
        uint32_t base = 0u;
        if (lane == 0) base = atomicAdd(&ctl->n_sitems[level][buf_shard * 16u], n_buf);
        base = uniform_u32(base);
        __builtin_amdgcn_wave_barrier();
        if (static_cast<uint32_t>(lane) < n_buf) {
            if (base + n_buf <= item_cap) sidx[buf_shard * item_cap + base + static_cast<uint32_t>(lane)] = buf[lane];
            else if (lane == 0) atomicOr(&ctl->overflow, 1u);           // (never: the shard's survivors are a subset of its items)
        }
        __builtin_amdgcn_wave_barrier();
        n_buf = 0u;
    };
    for (uint32_t w = wave_id; w < n_items; w += wave_count) {
        uint32_t sh, li, n_sh;
        shard_find(imap, w, sh, li, n_sh);
        const uint32_t idx = uniform_u32(sh * item_cap + li);               // item storage index (also keys vis)
        const ShadeItem it = items[idx];                                    // wave-uniform: a scalar load
        const float ax = it.ox + it.t * it.dx, ay = it.oy + it.t * it.dy, az = it.oz + it.t * it.dz;      // the hit, as the shadow kernels form it
        bool survive = it.lmode != 0u || give_up;          // (a mirror bounce carries a light list of its own: straight to the shadow units)
        // Several lights (`done`): a hit blocked from one light is still decided for the others -- every (hit, light) pair gets a byte, 1 = its
        // words are written, and k_shadow_shaft skips the units of such pairs; the sample rays of a decided pair are accounted for here.
        // One light: the first blocked light ends the hit's beams (it becomes a survivor, the unit kernel does all of it).
        uint32_t n_decided = 0u;
        if (done != nullptr && survive && lane < L.n_lights) done[static_cast<size_t>(idx) * static_cast<size_t>(lslots) + static_cast<uint32_t>(lane)] = 0;
        const bool skip_all = survive;
        for (int l = 0; l < L.n_lights && !skip_all && (done != nullptr || !survive); ++l) {
            float x0, y0, z0, x1, y1, z1;
            const float px = L.pos[l][0], py = L.pos[l][1], pz = L.pos[l][2];
            const LightGrid lg = light_grid(L, px, py, pz);
            grid_sample(lg, 0.5f, 0.5f, x0, y0, z0);             // the samples are monotone in each grid index: two corners give the exact box
            grid_sample(lg, fi_last, fj_last, x1, y1, z1);
            if (L.mode == RT_LIGHT_SPHERE) sphere_box(L, px, py, pz, x0, y0, z0, x1, y1, z1);
            const float slx = fminf(x0, x1), sly = fminf(y0, y1), slz = fminf(z0, z1), shx = fmaxf(x0, x1), shy = fmaxf(y0, y1), shz = fmaxf(z0, z1);
            const float e = 1e-7f * S.extent;
            ShaftLanes SL = make_shaft_lanes(lane, ax, ay, az, slx, sly, slz, shx, shy, shz, S.extent, RT_SHAFT_TRUNC);
            shaft_inflate(SL, tk, e, e, e);
            const ShaftCtl SC{SL.pad, false};
            __builtin_amdgcn_wave_barrier();
            shaft_tri_store(rec, lane, SL, ax, ay, az, slx, sly, slz, shx, shy, shz);
            shaft_lanes_store(shaft, lane, SL);
            if (lane == 8) {
                const float m0 = 2e-5f * (((fabsf(slx) + fabsf(shx)) + (fabsf(sly) + fabsf(shy)) + (fabsf(slz) + fabsf(shz))) +
                                          ((fabsf(ax) + fabsf(ax)) + (fabsf(ay) + fabsf(ay)) + (fabsf(az) + fabsf(az))));
                rec[8] = make_float4(ax, ay, az, m0); rec[11] = make_float4(ax, ay, az, 0.f); rec[12] = make_float4(ax, ay, az, 0.f);
            }
            __builtin_amdgcn_wave_barrier();
            const bool dirs_ok = beam_dirs_ok_wave(L, lg, N, px, py, pz, ax, ay, az, lane);
            int budget = S.beam_budget;
            const DNode root = B.lds_nodes[0];
            bool blocked = beam_walk(B, S, root, lane, SC, dirs_ok, budget);
            RT_PROF_ADD(lane, 76, 1); RT_PROF_ADD(lane, 77, blocked ? 0 : 1); RT_PROF_ADD(lane, 80, blocked ? 0 : S.beam_budget - budget); RT_PROF_ADD(lane, 81, budget < 0 ? 1 : 0);
            n_tested += 1u; n_unblocked += blocked ? 0u : 1u;
            // the leaves with a chunk that may never be culled (beam_tile has the argument): lane = leaf
            if (!blocked && n_bad != 0u) {
                const float big = fmaxf(fmaxf(fabsf(slx), fabsf(shx)), fmaxf(fabsf(sly), fabsf(shy))) + fmaxf(fabsf(slz), fabsf(shz));
                const float scale = S.extent + big + (fabsf(ax) + fabsf(ay) + fabsf(az));
                const float pad = 4e-4f * ((fmaxf(fabsf(slx), fabsf(shx)) + fmaxf(fabsf(sly), fabsf(shy)) + fmaxf(fabsf(slz), fabsf(shz))) + S.extent) * 1.001f;
                const ItemPlane p0 = item_plane<0, false>(ax, ay, az, slx, sly, slz, shx, shy, shz, scale), p1 = item_plane<0, true>(ax, ay, az, slx, sly, slz, shx, shy, shz, scale);
                const ItemPlane p2 = item_plane<1, false>(ax, ay, az, slx, sly, slz, shx, shy, shz, scale), p3 = item_plane<1, true>(ax, ay, az, slx, sly, slz, shx, shy, shz, scale);
                const ItemPlane p4 = item_plane<2, false>(ax, ay, az, slx, sly, slz, shx, shy, shz, scale), p5 = item_plane<2, true>(ax, ay, az, slx, sly, slz, shx, shy, shz, scale);
                const float nlx = fminf(slx, ax), nly = fminf(sly, ay), nlz = fminf(slz, az), nhx = fmaxf(shx, ax), nhy = fmaxf(shy, ay), nhz = fmaxf(shz, az);
                const float flx = ax >= shx ? ax : -3e38f, fly = ay >= shy ? ay : -3e38f, flz = az >= shz ? az : -3e38f;
                const float fhx = ax <= slx ? ax : 3e38f, fhy = ay <= sly ? ay : 3e38f, fhz = az <= slz ? az : 3e38f;
                const float *bb = s_bad + 6u * (static_cast<uint32_t>(lane) < n_bad ? static_cast<uint32_t>(lane) : 0u);
                const float lx_ = bb[0] - pad, ly_ = bb[1] - pad, lz_ = bb[2] - pad, hx_ = bb[3] + pad, hy_ = bb[4] + pad, hz_ = bb[5] + pad;
                bool near_out = (lx_ > nhx) || (hx_ < nlx) || (ly_ > nhy) || (hy_ < nly) || (lz_ > nhz) || (hz_ < nlz);
                bool far_out = (lx_ > fhx) || (hx_ < flx) || (ly_ > fhy) || (hy_ < fly) || (lz_ > fhz) || (hz_ < flz);
                item_plane_test<0>(p0, lx_, ly_, lz_, hx_, hy_, hz_, near_out, far_out); item_plane_test<0>(p1, lx_, ly_, lz_, hx_, hy_, hz_, near_out, far_out);
                item_plane_test<1>(p2, lx_, ly_, lz_, hx_, hy_, hz_, near_out, far_out); item_plane_test<1>(p3, lx_, ly_, lz_, hx_, hy_, hz_, near_out, far_out);
                item_plane_test<2>(p4, lx_, ly_, lz_, hx_, hy_, hz_, near_out, far_out); item_plane_test<2>(p5, lx_, ly_, lz_, hx_, hy_, hz_, near_out, far_out);
                blocked = __ballot(static_cast<uint32_t>(lane) < n_bad && !(dirs_ok && near_out && far_out)) != 0ull;
                RT_PROF_ADD(lane, 78, 1); RT_PROF_ADD(lane, 79, blocked ? 1 : 0);
            }
            if (done != nullptr && lane == 0) done[static_cast<size_t>(idx) * static_cast<size_t>(lslots) + static_cast<uint32_t>(l)] = blocked ? 0 : 1;
            n_decided += blocked ? 0u : 1u;
            if (blocked) {
                survive = true;              // (one light: the words of lights already written stay, the unit kernel rewrites the same values)
            } else if (static_cast<uint32_t>(lane) < P) {
                // nothing can block any sample segment of this hit to light l: all N samples visible (lane = pass)
                const uint32_t left = N - static_cast<uint32_t>(lane) * 64u;
                vis[(static_cast<unsigned long long>(idx) * static_cast<unsigned long long>(lslots) + static_cast<unsigned long long>(l)) * P + static_cast<uint32_t>(lane)] =
                    (blocks || left >= 64u) ? ~0ull : ((1ull << left) - 1ull);
            }
        }
        if (survive) {
            if (n_buf != 0u && buf_shard != sh) flush();
            buf_shard = sh;
            if (lane == 0) buf[n_buf] = idx;
            if (++n_buf == RT_PAIR_BUF) flush();
            if (done != nullptr) c_rays += N * n_decided;            // (the units of the decided pairs will be skipped)
        } else {
            c_rays += N * static_cast<uint32_t>(L.n_lights);        // sample rays that never become a shadow unit are accounted for here (wave-uniform)
        }
        if (++since == 16u && !give_up) {
            since = 0u;
            uint32_t a = 0u, b = 0u;
            if (lane == 0) { a = atomicAdd(&yield[0], n_tested) + n_tested; b = atomicAdd(&yield[1], n_unblocked) + n_unblocked; }
            a = uniform_u32(a); b = uniform_u32(b);
            n_tested = 0u; n_unblocked = 0u;
            give_up = a >= 256u && b * 4u < a;
        }
    }
    flush();
    if (lane == 0 && c_rays) atomicAdd(&ctl->stat[blockIdx.x & (RT_STAT_SHARDS - 1)][ST_RAYS_SAMPLE], static_cast<unsigned long long>(c_rays));
}

// ======================================================================================================
// K3: Phong shading of lit hits + material dispatch + bounce-ray emission with wave-level compaction.
// phongShade flyscene.cpp:822-859, getInterpolatedNormal :864-888, fresnel :890-910, traceRay dispatch :712-760.
// ======================================================================================================
// powf(cosphi, Ns) of phongShade (flyscene.cpp:852).  The reference calls libm's powf: glibc 2.35's e_powf.c (the ARM "optimized
// routines" algorithm: log2 through a 16-entry table + degree-5 polynomial, exp2 through a 32-entry table + degree-3 polynomial, all
// in double) in the build the x86-64 ifunc selects on FMA+AVX2 CPUs.  That published algorithm is evaluated here with the same
// tables, the same operation order and fused multiply-adds at the same places, so the result is the libm result BIT FOR BIT and
// the float accumulators of the frame equal the CPU path's exactly (the tests compare with tolerance 0).
__device__ const double POWF_LOG2_INVC[16] = {0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010bp+0, 0x1.3c995b0b80385p+0, 0x1.30d190c8864a5p+0,
                                              0x1.25e227b0b8eap+0, 0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0, 0x1.0953f419900a7p+0, 0x1p+0,
                                              0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aap-1, 0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1,
                                              0x1.767dcf5534862p-1};
__device__ const double POWF_LOG2_LOGC[16] = {-0x1.efec65b963019p-2, -0x1.b0b6832d4fca4p-2, -0x1.7418b0a1fb77bp-2, -0x1.39de91a6dcf7bp-2, -0x1.01d9bf3f2b631p-2,
                                              -0x1.97c1d1b3b7afp-3, -0x1.2f9e393af3c9fp-3, -0x1.960cbbf788d5cp-4, -0x1.a6f9db6475fcep-5, 0x0p+0,
                                              0x1.338ca9f24f53dp-4, 0x1.476a9543891bap-3, 0x1.e840b4ac4e4d2p-3, 0x1.40645f0c6651cp-2, 0x1.88e9c2c1b9ff8p-2,
                                              0x1.ce0a44eb17bccp-2};
__device__ const unsigned long long POWF_EXP2_TAB[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull,
    0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull,
    0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

#define RT_POW_TAB 64
// The tables live in LDS for the kernel (64 x 8 B: INVC[16] | LOGC[16] | EXP2_TAB[32]): a lookup is a ds_read_b64, not a global load on the
// critical path of every sample.  The function has NO divergent branch: the one common special case (cosphi = +0 with a positive finite
// shininess: the answer is +0) and the three range answers of the main path are selects; everything else e_powf.c answers without
// arithmetic (NaN, inf, subnormal x, y = 0 / inf / NaN, negative y at x = 0) sits behind ONE wave-uniform test and is resolved with selects
// there.  (The straight translation -- nine early returns -- cost k_shade ~40 exec-mask branches and three dependent global loads per two
// samples: cube k_shade 0.236 ms.)
__device__ __forceinline__ float pow_shininess(const float x, const float y, const double *__restrict__ tab) {
    const uint32_t ix0 = __float_as_uint(x);
    const uint32_t iy = __float_as_uint(y);
    const bool y_special = (2u * iy - 1u) >= (2u * 0x7f800000u - 1u);                 // y is 0, inf or NaN
    const bool special = !(ix0 - 0x00800000u < 0x7f800000u - 0x00800000u) || y_special;
    const bool zero_common = (ix0 == 0u) && !y_special && (iy >> 31) == 0u;            // pow(+0, y), y > 0 finite: +0
    uint32_t ix = zero_common ? 0x3f800000u : ix0;
    bool use_spec = false;
    float spec = 0.0f;
    if (__ballot(special && !zero_common) != 0ull) {
        // e_powf.c's cases without arithmetic, first match wins (x >= 0 or NaN here: cosphi = max(0, .)), then subnormal x
        const bool odd = special && !zero_common;
        const bool c1 = 2u * iy == 0u;                                                  // pow(x, +-0) = 1
        const bool c2 = x == 1.0f;
        const bool c3 = (x != x) || (y != y);                                           // NaN
        const bool c4 = 2u * iy == 2u * 0x7f800000u;                                    // y = +-inf
        const bool c5 = 2u * ix0 == 0u;                                                 // pow(+-0, y)
        const bool c6 = ix0 == 0x7f800000u;                                             // pow(+inf, y)
        const bool c7 = (ix0 >> 31) != 0u;                                              // negative x: not reachable from phongShade
        const bool yneg = (iy >> 31) != 0u;
        const bool small = 2u * ix0 < 2u * 0x3f800000u;                                // |x| < 1
        const float inf = __uint_as_float(0x7f800000u);
        const float v4 = (small == yneg) ? inf : 0.0f;
        const float v5 = yneg ? inf : 0.0f;
        const float v6 = yneg ? 0.0f : x;
        spec = c1 ? 1.0f : (c2 ? 1.0f : (c3 ? x + y : (c4 ? v4 : (c5 ? v5 : (c6 ? v6 : __uint_as_float(0x7fc00000u))))));
        use_spec = odd && (c1 || c2 || c3 || c4 || c5 || c6 || c7);
        // subnormal x: normalise as e_powf.c does
        uint32_t isub = __float_as_uint(x * 0x1p23f);
        isub &= 0x7fffffffu;
        isub -= 23u << 23;
        ix = use_spec ? 0x3f800000u : ((odd && !use_spec) ? isub : ix);
    }
    // log2_inline
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> 19) & 15u;
    const uint32_t top = tmp & 0xff800000u;
    const int k = static_cast<int>(top) >> 23;
    const double z = static_cast<double>(__uint_as_float(ix - top));
    const double r = __builtin_fma(z, tab[i], -1.0);
    const double y0 = tab[16u + i] + static_cast<double>(k);
    const double r2 = r * r;
    const double yy = __builtin_fma(0x1.27616c9496e0bp-2, r, -0x1.71969a075c67ap-2);
    const double p = __builtin_fma(0x1.ec70a6ca7baddp-2, r, -0x1.7154748bef6c8p-1);
    const double r4 = r2 * r2;
    double q = __builtin_fma(0x1.71547652ab82bp+0, r, y0);
    q = __builtin_fma(p, r2, q);
    const double logx = __builtin_fma(yy, r4, q);
    const double ylogx = static_cast<double>(y) * logx;
    // |y * log2(x)| >= 126: __math_oflowf / __math_uflowf / __math_may_uflowf, else the ordinary path (also for -149 <= y log2 x <= -126)
    const bool big = ((static_cast<unsigned long long>(__double_as_longlong(ylogx)) >> 47) & 0xffffull) >= 0x80bfull;
    const bool r_of = big && (ylogx > 0x1.fffffffd1d571p+6);
    const bool r_uf = big && !r_of && (ylogx <= -150.0);
    const bool r_mu = big && !r_of && !r_uf && (ylogx < -149.0);
    // exp2_inline
    const double shift = 0x1.8p+47;
    double kd = ylogx + shift;
    const unsigned long long ki = static_cast<unsigned long long>(__double_as_longlong(kd));
    kd -= shift;
    const double rr = ylogx - kd;
    const unsigned long long t = static_cast<unsigned long long>(__double_as_longlong(tab[32u + static_cast<uint32_t>(ki & 31ull)])) + (ki << 47);
    const double sc = __longlong_as_double(static_cast<long long>(t));
    const double zz = __builtin_fma(0x1.c6af84b912394p-5, rr, 0x1.ebfce50fac4f3p-3);
    const double rr2 = rr * rr;
    double yv = __builtin_fma(0x1.62e42ff0c52d6p-1, rr, 1.0);
    yv = __builtin_fma(zz, rr2, yv);
    yv = yv * sc;
    float res = static_cast<float>(yv);
    res = r_of ? __uint_as_float(0x7f800000u) : (r_uf ? 0.0f : (r_mu ? __uint_as_float(1u) : res));
    res = use_spec ? spec : res;
    return zero_common ? 0.0f : res;
}

__device__ __forceinline__ float fresnel_term(float ix, float iy, float iz, float nx, float ny, float nz, float ior) {
    float cosi = dot3(ix, iy, iz, nx, ny, nz);
    float etai = 1, etat = ior;
    if (cosi > 0) { const float tmp = etai; etai = etat; etat = tmp; }
    const float sint = etai / etat * sqrtf(smax(0.f, 1 - cosi * cosi));
    if (sint >= 1) return 1;
    const float cost = sqrtf(smax(0.f, 1 - sint * sint));
    cosi = fabsf(cosi);
    const float Rs = ((etat * cosi) - (etai * cost)) / ((etat * cosi) + (etai * cost));
    const float Rp = ((etai * cosi) - (etat * cost)) / ((etai * cosi) + (etat * cost));
    return (Rs * Rs + Rp * Rp) / 2;
}

// waves per SIMD: 3 (149 VGPRs) -> 4 (128) was -7 % on the 576k-item cube frame in round 1; 4 -> 5 (96 VGPRs, 20 B of scratch) is another -4 % now that
// the sample loop is leaner (round 3, same-box A/B: k_shade 0.192 -> 0.185 ms).  (The powf coefficients stay literals: kept in LDS they are
// hoisted into 24 VGPRs, which is +1 % at four waves and 112 B of scratch at five.)
#ifndef RT_SHADE_WPE
#define RT_SHADE_WPE 5
#endif
#define RT_SHADE_ATTR __attribute__((amdgpu_waves_per_eu(RT_SHADE_WPE, 8)))
// ---- phongShade / getInterpolatedNormal / the material dispatch of traceRay, shared by k_shade (visibility words from the shadow kernels) and
// k_deep (the deep bounce levels of flat scenes, visibility computed in place).  Every function keeps the reference's operation order.
struct ShadeHit {
    float hx, hy, hz;            // hitPoint = origin + t * direction (flyscene.cpp:695)
    float nx, ny, nz;            // (modelMatrix * interpolated normal).normalized()
    float ex, ey, ez;            // eyeToHitPoint
    float fnx, fny, fnz;         // face normal
    rt_material mat;
};
__device__ __forceinline__ ShadeHit shade_hit(const DScene &S, const int face, const float ox, const float oy, const float oz, const float dx, const float dy,
                                              const float dz, const float t) {
    ShadeHit H;
    const float hx = ox + t * dx, hy = oy + t * dy, hz = oz + t * dz;
    const float *tv = S.tri_verts + static_cast<size_t>(face) * 9;
    const float Ax = tv[0], Ay = tv[1], Az = tv[2], Bx = tv[3], By = tv[4], Bz = tv[5], Cx = tv[6], Cy = tv[7], Cz = tv[8];
    const uint32_t ia = S.tri_vid[face * 3], ib = S.tri_vid[face * 3 + 1], ic = S.tri_vid[face * 3 + 2];
    const float *nA = S.vert_normal + static_cast<size_t>(ia) * 3, *nB = S.vert_normal + static_cast<size_t>(ib) * 3,
                *nC = S.vert_normal + static_cast<size_t>(ic) * 3;
    H.mat = S.mats[S.mat_id[face]];
    H.fnx = S.face_normal[face * 3]; H.fny = S.face_normal[face * 3 + 1]; H.fnz = S.face_normal[face * 3 + 2];
    // getInterpolatedNormal (flyscene.cpp:864-888)
    const float v0x = Bx - Ax, v0y = By - Ay, v0z = Bz - Az;
    const float v1x = Cx - Ax, v1y = Cy - Ay, v1z = Cz - Az;
    const float v2x = hx - Ax, v2y = hy - Ay, v2z = hz - Az;
    const float d00 = dot3(v0x, v0y, v0z, v0x, v0y, v0z), d01 = dot3(v0x, v0y, v0z, v1x, v1y, v1z);
    const float d11 = dot3(v1x, v1y, v1z, v1x, v1y, v1z);
    const float d20 = dot3(v2x, v2y, v2z, v0x, v0y, v0z), d21 = dot3(v2x, v2y, v2z, v1x, v1y, v1z);
    const float denom = d00 * d11 - d01 * d01;
    const float bv = (d11 * d20 - d01 * d21) / denom;
    const float bw = (d00 * d21 - d01 * d20) / denom;
    const float bu = 1.0f - bv - bw;
    const float inx = (bu * nA[0] + bv * nB[0]) + bw * nC[0];
    const float iny = (bu * nA[1] + bv * nB[1]) + bw * nC[1];
    const float inz = (bu * nA[2] + bv * nB[2]) + bw * nC[2];
    // mesh.getModelMatrix() * n: Affine * Vector3f adds the translation (flyscene.cpp:829)
    const float *M = S.model;
    float nx = ((M[0] * inx + M[1] * iny) + M[2] * inz) + M[3] * 1.0f;
    float ny = ((M[4] * inx + M[5] * iny) + M[6] * inz) + M[7] * 1.0f;
    float nz = ((M[8] * inx + M[9] * iny) + M[10] * inz) + M[11] * 1.0f;
    normalize3(nx, ny, nz);
    // eyeToHitPoint = (-1 * (hitPoint - origin)).normalized(): invariant over samples
    float ex = -1.0f * (hx - ox), ey = -1.0f * (hy - oy), ez = -1.0f * (hz - oz);
    normalize3(ex, ey, ez);
    H.hx = hx; H.hy = hy; H.hz = hz; H.nx = nx; H.ny = ny; H.nz = nz; H.ex = ex; H.ey = ey; H.ez = ez;
    return H;
}
// the diffuse + specular term of ONE light sample at (sx, sy, sz) (flyscene.cpp:838-853), colour x material factors passed in
__device__ __forceinline__ void phong_sample(const ShadeHit &H, const float sx, const float sy, const float sz, const float lkd0, const float lkd1, const float lkd2,
                                             const float lks0, const float lks1, const float lks2, const double *__restrict__ tab, float &tr_, float &tg_, float &tb_) {
    float ldx = sx - H.hx, ldy = sy - H.hy, ldz = sz - H.hz;
    normalize3_shared(ldx, ldy, ldz);
    const float ldn = dot3(ldx, ldy, ldz, H.nx, H.ny, H.nz);
    const float costheta = smax(0.0f, ldn);
    const float two = 2 * ldn;
    float rx = ldx - two * H.nx, ry = ldy - two * H.ny, rz = ldz - two * H.nz;
    normalize3_shared(rx, ry, rz);
    const float cosphi = smax(0.0f, dot3(H.ex, H.ey, H.ez, -1.0f * rx, -1.0f * ry, -1.0f * rz));
    // (the eye on the far side of the reflected ray for EVERY hit of the wave: powf(+0, Ns) = +0 for a positive finite Ns -- e_powf.c's zero case --
    //  without the 60-instruction double-precision path; one wave-uniform test)
    const bool zero_pow = cosphi == 0.0f && H.mat.shininess > 0.0f && H.mat.shininess < __uint_as_float(0x7f800000u);
    const float pw = (__ballot(!zero_pow) == 0ull) ? 0.0f : pow_shininess(cosphi, H.mat.shininess, tab);
    tr_ = lkd0 * costheta + lks0 * pw; tg_ = lkd1 * costheta + lks1 * pw; tb_ = lkd2 * costheta + lks2 * pw;
}
// material dispatch of traceRay (flyscene.cpp:712-760); a hit at level == max_depth is plain Phong (extension).  Returns the blend kind; for
// kinds other than KIND_CONST `child` is the spawned ray (origin, pix filled by the caller); illum 5 stores its Fresnel factor.
__device__ __forceinline__ uint32_t material_dispatch(const ShadeHit &H, const bool may_spawn, const float dx, const float dy, const float dz, const float lx, const float ly,
                                                      const float lz, const uint32_t lmode, RayItem &child, float *__restrict__ fres_pix) {
    const int imodel = H.mat.illum;
    uint32_t kind = KIND_CONST;
    if (may_spawn) {
        if (imodel == 9) {
            kind = KIND_PASS;
            child.dx = dx; child.dy = dy; child.dz = dz;
            child.lx = lx; child.ly = ly; child.lz = lz; child.lmode = lmode;
        } else if (imodel == 6) {
            kind = KIND_REFRACT;
            const float c1 = fabsf(dot3(dx, dy, dz, H.fnx, H.fny, H.fnz));
            const float inv = 1 / H.mat.optical_density;
            const double p1 = static_cast<double>(inv) * static_cast<double>(inv);     // pow((1/Ni), 2)
            const double p2 = static_cast<double>(c1) * static_cast<double>(c1);       // pow(c1, 2)
            const float c2 = static_cast<float>(sqrt(1 - p1 * (1 - p2)));
            const float k = inv * c1 - c2;
            child.dx = inv * dx + k * H.fnx; child.dy = inv * dy + k * H.fny; child.dz = inv * dz + k * H.fnz;
            child.lx = lx; child.ly = ly; child.lz = lz; child.lmode = lmode;
        } else if (imodel > 2 && imodel < 6) {
            kind = imodel == 5 ? KIND_FRESNEL : KIND_MIRROR;
            const float two = 2 * dot3(dx, dy, dz, H.fnx, H.fny, H.fnz);
            child.dx = dx - two * H.fnx; child.dy = dy - two * H.fny; child.dz = dz - two * H.fnz;
            child.lx = H.hx; child.ly = H.hy; child.lz = H.hz; child.lmode = 1u;             // reflectedLights = {hitPoint}
            if (imodel == 5) *fres_pix = fresnel_term(child.dx, child.dy, child.dz, H.fnx, H.fny, H.fnz, H.mat.optical_density);
        }
    }
    return kind;
}
__device__ __forceinline__ void pow_tables_to_lds(double *s_pow) {
    if (threadIdx.x < RT_POW_TAB) {
        const uint32_t ti = threadIdx.x;
        s_pow[ti] = ti < 16u ? POWF_LOG2_INVC[ti] : (ti < 32u ? POWF_LOG2_LOGC[ti - 16u] : __longlong_as_double(static_cast<long long>(POWF_EXP2_TAB[ti - 32u])));
    }
    __syncthreads();
}

// SIMPLE: the light is a point or a grid of at most 64 samples (one visibility word per (hit, light), sample s = bit s, no 8 x 8 blocks, no
// sphere offsets) -- the reference's own 5 x 5 and the 8 x 8 headline.  The sample loop then carries no word / block / mode branches.
// FLAT: the scene is one root leaf.  A spawned child ray is then tested against that leaf right here (the closest-hit walk k_trace would run for
// it at the next level: ~45 instructions per triangle for the whole tile); a child that hits nothing gets its BACKGROUND record now and never
// becomes a ray -- on a convex mirror object (cube.obj) that is every child: level 1 stays empty, its 576k-ray k_trace launch (24 us) is gone.
// Children that do hit something go through the wide kernels as before (their walk is repeated there: the exception, not the rule).
template <bool SIMPLE, bool FLAT>
__global__ __launch_bounds__(256) RT_SHADE_ATTR void k_shade(const DNode *__restrict__ nodes, const TriRec *__restrict__ tris, const DScene S, const DLights L, const DFrame F,
                                               const int level, const int ctr_slot,
                                               const int lslots, const ShadeItem *__restrict__ items, Control *__restrict__ ctl,
                                               const unsigned long long *__restrict__ vis, float4 *__restrict__ rec,
                                               float *__restrict__ fres, RayItem *__restrict__ rays_out) {
    __shared__ double s_pow[RT_POW_TAB];       // powf tables: INVC[16] | LOGC[16] | EXP2_TAB[32] (bit patterns)
    pow_tables_to_lds(s_pow);
    const int lane = threadIdx.x & 63;
    const ShardMap imap = shard_map(ctl->n_items[level], lane, F.item_cap, 1u, 64u);      // groups of 64 items, shard after shard
    const uint32_t ntiles = imap.total;
    const uint32_t N = static_cast<uint32_t>(L.n_samples);
    const uint32_t P = (N + 63u) / 64u;
    uint32_t c_shaded = 0, c_spawn = 0, c_resolved = 0;
    const uint32_t wave_id = uniform_u32(blockIdx.x * 4u + (threadIdx.x >> 6));
    const uint32_t wave_count = gridDim.x * 4u;
    DNode root;
    if (FLAT) root = nodes[0];
    for (uint32_t tile = wave_id; tile < ntiles; tile += wave_count) {
        uint32_t sh, tj, n_sh;
        shard_find(imap, tile, sh, tj, n_sh);
        const bool valid = tj * 64u + static_cast<uint32_t>(lane) < n_sh;
        const uint32_t idx = sh * F.item_cap + tj * 64u + static_cast<uint32_t>(lane);     // item storage index (also keys vis)
        const ShadeItem it = items[valid ? idx : sh * F.item_cap];
        bool spawn = false;
        RayItem child;
        child.pad = 0u;
        if (valid) {
            c_shaded += 1;
            const ShadeHit H = shade_hit(S, it.face, it.ox, it.oy, it.oz, it.dx, it.dy, it.dz, it.t);
            float fr = 0.f, fg = 0.f, fb = 0.f;
            const int nl = it.lmode ? 1 : L.n_lights;
            for (int l = 0; l < nl; ++l) {
                const float px = it.lmode ? it.lx : L.pos[l][0], py = it.lmode ? it.ly : L.pos[l][1], pz = it.lmode ? it.lz : L.pos[l][2];
                const unsigned long long *vw = vis + (static_cast<unsigned long long>(idx) * static_cast<unsigned long long>(lslots) + static_cast<unsigned long long>(l)) * P;
                float sum = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
                unsigned long long word = 0ull;
                // The per-sample term is evaluated for EVERY sample and added as `visible ? term : +0` in sample order --
                // identical to skipping invisible samples (x + 0 == x for the non-negative accumulators) but branch-free.
                const float lkd0 = L.color[0] * H.mat.kd[0], lkd1 = L.color[1] * H.mat.kd[1], lkd2 = L.color[2] * H.mat.kd[2];
                const float lks0 = L.color[0] * H.mat.ks[0], lks1 = L.color[1] * H.mat.ks[1], lks2 = L.color[2] * H.mat.ks[2];
                const LightGrid lg = light_grid(L, px, py, pz);
                uint32_t si = 0, sj = 0;                       // s = si * vsteps + sj, kept as counters: no division per sample
                const uint32_t vst = static_cast<uint32_t>(L.vsteps > 0 ? L.vsteps : 1);
                const bool blocks = !SIMPLE && sample_blocks(L);
                const uint32_t bpr = vst >> 3;
                if (SIMPLE) word = vw[0];
                // (unrolled by two at five waves per SIMD: 0.185 against 0.1865 ms -- noise; by two or four at four waves: +5 %.  The loop is not waiting on
                //  its own dependency chains: what it lacks is issue slots -- FP64 at half rate, v_sqrt / v_rcp at quarter rate)
                for (uint32_t s = 0; s < N; ++s) {
                    uint32_t bit = s & 63u;
                    if (!SIMPLE) {
                        if (blocks) {
                            if ((sj & 7u) == 0u) word = vw[(si >> 3) * bpr + (sj >> 3)];       // a new block every 8 samples of a grid row
                            bit = ((si & 7u) << 3) | (sj & 7u);
                        } else if (bit == 0u) {
                            word = vw[s >> 6];
                        }
                    }
                    const bool visible = ((word >> bit) & 1ull) != 0ull;
                    sum += visible ? 1.0f : 0.0f;
                    float sx, sy, sz;
                    grid_sample(lg, static_cast<float>(si) + 0.5f, static_cast<float>(sj) + 0.5f, sx, sy, sz);
                    if (!SIMPLE && L.mode == RT_LIGHT_SPHERE) sphere_sample(L, s, px, py, pz, sx, sy, sz);
                    if (++sj == vst) { sj = 0; ++si; }
                    float tr_, tg_, tb_;
                    phong_sample(H, sx, sy, sz, lkd0, lkd1, lkd2, lks0, lks1, lks2, s_pow, tr_, tg_, tb_);
                    cr = cr + (visible ? tr_ : 0.0f);
                    cg = cg + (visible ? tg_ : 0.0f);
                    cb = cb + (visible ? tb_ : 0.0f);
                }
                const float a = sum / static_cast<float>(N), b = 1.3f / static_cast<float>(N);
                fr = fr + (cr * a) * b;
                fg = fg + (cg * a) * b;
                fb = fb + (cb * a) * b;
            }
            const uint32_t kind = material_dispatch(H, level < F.max_depth, it.dx, it.dy, it.dz, it.lx, it.ly, it.lz, it.lmode, child, &fres[it.pix]);
            rec[it.pix] = make_float4(fr, fg, fb, __uint_as_float(kind));
            if (kind != KIND_CONST) {
                spawn = true;
                child.ox = H.hx; child.oy = H.hy; child.oz = H.hz; child.pix = it.pix;
            }
        }
        if (FLAT && __ballot(spawn) != 0ull) {
            // traceRay of the child, closest hit only (flyscene.cpp:655-691), exactly as k_trace runs it; no hit -> BACKGROUND at the next level
            const float bx = (child.ox + child.dx) - child.ox, by = (child.oy + child.dy) - child.oy, bz = (child.oz + child.dz) - child.oz;
            const float brx = __builtin_amdgcn_rcpf(bx), bry = __builtin_amdgcn_rcpf(by), brz = __builtin_amdgcn_rcpf(bz);
            const bool in_root = spawn && box_hit_verified(root.bmin, child.ox, child.oy, child.oz, bx, by, bz, brx, bry, brz);
            float best_t = 3.402823466e+38f;
            int best_f = -1;
            bool dummy = false;
            uint32_t cu0 = 0, cu1 = 0;
            flat_walk<false, false>(root, tris, in_root, seg_off(), 0ull, LanePlane{0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, child.ox, child.oy, child.oz, child.dx, child.dy, child.dz,
                                    best_t, best_f, dummy, cu0, cu1);
            const bool hit = (best_f >= 0) && (static_cast<uint32_t>(best_f) < S.n_faces);
            if (spawn && !hit) {
                rec[static_cast<size_t>(F.npix) + child.pix] = make_float4(1.f, 1.f, 1.f, __uint_as_float(KIND_CONST));          // BACKGROUND (level + 1)
                c_resolved += 1;                      // a bounce ray that was traced here
                spawn = false;
            }
        }
        const unsigned long long sm = __ballot(spawn);
        if (sm != 0ull) {
            bool fits;
            const uint32_t base = shard_reserve(ctl->n_rays[level + 1], &ctl->overflow, tile, static_cast<uint32_t>(__popcll(sm)), F.ray_cap, lane, fits);
            if (spawn && fits) { rays_out[base + lanes_below(sm)] = child; c_spawn += 1; }
        }
    }
    c_shaded = wave_sum(c_shaded);
    (void)c_spawn;
    if (lane == 0 && c_shaded) atomicAdd(&ctl->stat[blockIdx.x & (RT_STAT_SHARDS - 1)][ST_SHADED_HITS], static_cast<unsigned long long>(c_shaded));
    if (FLAT) {
        c_resolved = wave_sum(c_resolved);
        if (lane == 0 && c_resolved) atomicAdd(&ctl->stat[blockIdx.x & (RT_STAT_SHARDS - 1)][ST_RAYS_BOUNCE], static_cast<unsigned long long>(c_resolved));
    }
}

// ======================================================================================================
// K3b: the DEEP bounce levels of flat scenes (levels 2 .. max_depth), one launch.
// A bounce level costs four launches (trace, beam, shadow, shade) whether anything reaches it or not: on a convex mirror object -- cube.obj, the
// headline -- levels 2 .. 4 are empty and their 12 launches were 40 us of a 0.36 ms frame.  A pixel's chain of bounces depends on nothing but
// itself, so here a wave takes 64 level-2 rays and carries them through ALL remaining levels on its own: closest hit and light-centre
// visibility over the root leaf (flat_walk, as k_trace), then phongShade with the visibility of every light sample computed in place
// (lane = hit, one flat_walk over the root leaf per sample: the segment lightStrikes would test), material dispatch, next level.  Nothing is
// compacted and nothing is culled: every decision is the reference's own test on the ray itself, and the per-sample arithmetic is the code
// k_shade runs (shade_hit / phong_sample / material_dispatch).  An empty level 2 is one launch that reads one counter.  Flat scenes only (a
// root leaf of <= 64 triangles: a sample costs <= 64 triangle tests); tree scenes and levels 0 / 1 keep the wide kernels.
// ======================================================================================================
__global__ __launch_bounds__(256) RT_SHADE_ATTR void k_deep(const DNode *__restrict__ nodes, const TriRec *__restrict__ tris, const DScene S, const DLights L, const DFrame F,
                                                            const int level0, const RayItem *__restrict__ rays_in, Control *__restrict__ ctl,
                                                            float4 *__restrict__ rec0, float *__restrict__ fres0) {
    __shared__ double s_pow[RT_POW_TAB];
    const int lane = threadIdx.x & 63;
    const ShardMap rmap = shard_map(ctl->n_rays[level0], lane, 0xffffffffu, 1u, 64u);
    const uint32_t ntiles = rmap.total;
    if (ntiles == 0u) return;                     // nothing reaches the deep levels: every wave leaves here
    pow_tables_to_lds(s_pow);
    const DNode root = nodes[0];
    const LanePlane pl0{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const uint32_t N = static_cast<uint32_t>(L.n_samples);
    const uint32_t vst = static_cast<uint32_t>(L.vsteps > 0 ? L.vsteps : 1);
    uint32_t c_rays = 0, c_centre = 0, c_sample = 0, c_shaded = 0, c_unused0 = 0, c_unused1 = 0;
    const uint32_t wave_id = uniform_u32(blockIdx.x * 4u + (threadIdx.x >> 6));
    const uint32_t wave_count = gridDim.x * 4u;
    for (uint32_t tile = wave_id; tile < ntiles; tile += wave_count) {
        uint32_t sh, tj, n_in;
        shard_find(rmap, tile, sh, tj, n_in);
        const uint32_t k = tj * 64u + static_cast<uint32_t>(lane);
        bool alive = k < n_in;
        const RayItem it0 = rays_in[alive ? sh * F.ray_cap + k : 0u];
        float ox = it0.ox, oy = it0.oy, oz = it0.oz, dx = it0.dx, dy = it0.dy, dz = it0.dz, lx = it0.lx, ly = it0.ly, lz = it0.lz;
        uint32_t lmode = it0.lmode;
        const uint32_t pix = it0.pix;
        for (int level = level0; level <= F.max_depth && __ballot(alive) != 0ull; ++level) {
            float4 *__restrict__ rec = rec0 + static_cast<size_t>(level - level0) * F.npix;
            float *__restrict__ fres = fres0 + static_cast<size_t>(level - level0) * F.npix;
            // ---- traceRay: closest hit (flyscene.cpp:655-691), as k_trace
            c_rays += alive ? 1u : 0u;
            const float bx = (ox + dx) - ox, by = (oy + dy) - oy, bz = (oz + dz) - oz;
            const float brx = __builtin_amdgcn_rcpf(bx), bry = __builtin_amdgcn_rcpf(by), brz = __builtin_amdgcn_rcpf(bz);
            const bool in_root = alive && box_hit_verified(root.bmin, ox, oy, oz, bx, by, bz, brx, bry, brz);
            float best_t = 3.402823466e+38f;
            int best_f = -1;
            bool dummy = false;
            flat_walk<false, false>(root, tris, in_root, seg_off(), 0ull, pl0, ox, oy, oz, dx, dy, dz, best_t, best_f, dummy, c_unused0, c_unused1);
            const bool hit = alive && (best_f >= 0) && (static_cast<uint32_t>(best_f) < S.n_faces);
            const float hx = ox + best_t * dx, hy = oy + best_t * dy, hz = oz + best_t * dz;   // flyscene.cpp:695
            // ---- lightStrikes(hitPoint, lights): one segment per light centre (flyscene.cpp:700)
            bool lit = false;
            const int nl_lane = lmode ? 1 : L.n_lights;
            const int nl_wave = (__ballot(hit && lmode == 0u) != 0ull) ? L.n_lights : 1;
            if (__ballot(hit) != 0ull) {
                for (int l = 0; l < nl_wave; ++l) {
                    const bool act = hit && (l < nl_lane);
                    const float px = lmode ? lx : L.pos[l][0], py = lmode ? ly : L.pos[l][1], pz = lmode ? lz : L.pos[l][2];
                    const float sdx = hx - px, sdy = hy - py, sdz = hz - pz;
                    c_centre += act ? 1u : 0u;
                    const float srx = __builtin_amdgcn_rcpf(sdx), sry = __builtin_amdgcn_rcpf(sdy), srz = __builtin_amdgcn_rcpf(sdz);
                    const bool sroot = act && box_hit_verified(root.bmin, px, py, pz, sdx, sdy, sdz, srx, sry, srz);
                    float t_unused = 0.f; int f_unused = -1;
                    bool occ = false;
                    flat_walk<true, false>(root, tris, sroot, seg_off(), 0ull, pl0, px, py, pz, sdx, sdy, sdz, t_unused, f_unused, occ, c_unused0, c_unused1);
                    lit = lit || (act && !occ);
                }
            }
            if (alive) {
                if (!hit) rec[pix] = make_float4(1.f, 1.f, 1.f, __uint_as_float(KIND_CONST));          // BACKGROUND
                else if (!lit) rec[pix] = make_float4(0.f, 0.f, 0.f, __uint_as_float(KIND_CONST));     // SHADOW
            }
            alive = lit;
            if (__ballot(lit) == 0ull) break;
            // ---- phongShade (flyscene.cpp:822-859): the lanes that hold a lit hit; every sample's lightStrikes segment is walked in place
            c_shaded += lit ? 1u : 0u;
            const int face = lit ? best_f : 0;
            const ShadeHit H = shade_hit(S, face, ox, oy, oz, dx, dy, dz, lit ? best_t : 0.0f);      // (lanes without a lit hit: finite stand-in values, never stored)
            float fr = 0.f, fg = 0.f, fb = 0.f;
            const int nl_shade = (__ballot(lit && lmode == 0u) != 0ull) ? L.n_lights : 1;
            const float lkd0 = L.color[0] * H.mat.kd[0], lkd1 = L.color[1] * H.mat.kd[1], lkd2 = L.color[2] * H.mat.kd[2];
            const float lks0 = L.color[0] * H.mat.ks[0], lks1 = L.color[1] * H.mat.ks[1], lks2 = L.color[2] * H.mat.ks[2];
            for (int l = 0; l < nl_shade; ++l) {
                const bool act = lit && (l < nl_lane);
                const float px = lmode ? lx : L.pos[l][0], py = lmode ? ly : L.pos[l][1], pz = lmode ? lz : L.pos[l][2];
                const LightGrid lg = light_grid(L, px, py, pz);
                float sum = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
                uint32_t si = 0, sj = 0;
                for (uint32_t s = 0; s < N; ++s) {
                    float sx, sy, sz;
                    grid_sample(lg, static_cast<float>(si) + 0.5f, static_cast<float>(sj) + 0.5f, sx, sy, sz);
                    if (L.mode == RT_LIGHT_SPHERE) sphere_sample(L, s, px, py, pz, sx, sy, sz);
                    if (++sj == vst) { sj = 0; ++si; }
                    // lightStrikes(hitPoint, {sample}): origin = sample, direction = hitPoint - sample (flyscene.cpp:912-954), as k_shadow forms it
                    const float ddx = H.hx - sx, ddy = H.hy - sy, ddz = H.hz - sz;
                    const float srx = __builtin_amdgcn_rcpf(ddx), sry = __builtin_amdgcn_rcpf(ddy), srz = __builtin_amdgcn_rcpf(ddz);
                    c_sample += act ? 1u : 0u;
                    const bool sroot = act && box_hit_verified(root.bmin, sx, sy, sz, ddx, ddy, ddz, srx, sry, srz);
                    float t_unused = 0.f; int f_unused = -1;
                    bool occ = false;
                    flat_walk<true, false>(root, tris, sroot, seg_off(), 0ull, pl0, sx, sy, sz, ddx, ddy, ddz, t_unused, f_unused, occ, c_unused0, c_unused1);
                    const bool visible = act && !occ;
                    sum += visible ? 1.0f : 0.0f;
                    float tr_, tg_, tb_;
                    phong_sample(H, sx, sy, sz, lkd0, lkd1, lkd2, lks0, lks1, lks2, s_pow, tr_, tg_, tb_);
                    cr = cr + (visible ? tr_ : 0.0f);
                    cg = cg + (visible ? tg_ : 0.0f);
                    cb = cb + (visible ? tb_ : 0.0f);
                }
                const float a = sum / static_cast<float>(N), b = 1.3f / static_cast<float>(N);
                fr = act ? fr + (cr * a) * b : fr;
                fg = act ? fg + (cg * a) * b : fg;
                fb = act ? fb + (cb * a) * b : fb;
            }
            RayItem child;
            child.pad = 0u;
            uint32_t kind = KIND_CONST;
            if (lit) {
                kind = material_dispatch(H, level < F.max_depth, dx, dy, dz, lx, ly, lz, lmode, child, &fres[pix]);
                rec[pix] = make_float4(fr, fg, fb, __uint_as_float(kind));
            }
            alive = lit && kind != KIND_CONST;
            if (alive) {       // the child ray of this lane: origin = hitPoint
                ox = H.hx; oy = H.hy; oz = H.hz; dx = child.dx; dy = child.dy; dz = child.dz;
                lx = child.lx; ly = child.ly; lz = child.lz; lmode = child.lmode;
            }
        }
    }
    c_rays = wave_sum(c_rays); c_centre = wave_sum(c_centre); c_sample = wave_sum(c_sample); c_shaded = wave_sum(c_shaded);
    if (lane == 0) {
        unsigned long long *st = ctl->stat[blockIdx.x & (RT_STAT_SHARDS - 1)];
        if (c_rays) atomicAdd(&st[ST_RAYS_BOUNCE], static_cast<unsigned long long>(c_rays));
        if (c_centre) atomicAdd(&st[ST_RAYS_CENTRE], static_cast<unsigned long long>(c_centre));
        if (c_sample) { atomicAdd(&st[ST_RAYS_SAMPLE], static_cast<unsigned long long>(c_sample)); atomicAdd(&st[ST_SAMPLE_WALKED], static_cast<unsigned long long>(c_sample)); }
        if (c_shaded) atomicAdd(&st[ST_SHADED_HITS], static_cast<unsigned long long>(c_shaded));
    }
}

// ======================================================================================================
// K4: resolve.  Folds the per-level records from the deepest level outwards -- c_k = a*phong_k + b*c_{k+1} --
// which keeps the recursion's rounding (a running throughput product would not), then quantises like
// writePPMImage (ppmIO.hpp:145): min(255, (int)(255*c)).
// ======================================================================================================
__global__ __launch_bounds__(256) void k_resolve(const DFrame F, const float4 *__restrict__ rec, const float *__restrict__ fres,
                                                 float *__restrict__ out_rgb, uint8_t *__restrict__ out_u8) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x; pix < F.npix; pix += stride) {
        int k = 0;
        float4 r = rec[pix];
        while (__float_as_uint(r.w) != KIND_CONST && k < F.max_depth) {
            ++k;
            r = rec[static_cast<size_t>(k) * F.npix + pix];
        }
        float vr = r.x, vg = r.y, vb = r.z;
        for (int j = k - 1; j >= 0; --j) {
            const float4 p = rec[static_cast<size_t>(j) * F.npix + pix];
            const uint32_t kind = __float_as_uint(p.w);
            float a, b;
            if (kind == KIND_PASS) { a = 0.10f; b = 0.90f; }
            else if (kind == KIND_REFRACT) { a = 0.2f; b = 0.8f; }
            else { a = 0.15f; b = 0.85f; }
            vr = a * p.x + b * vr; vg = a * p.y + b * vg; vb = a * p.z + b * vb;
            if (kind == KIND_FRESNEL) {
                const float f = fres[static_cast<size_t>(j) * F.npix + pix];
                vr = f * vr; vg = f * vg; vb = f * vb;
            }
        }
        if (out_rgb) { out_rgb[pix * 3] = vr; out_rgb[pix * 3 + 1] = vg; out_rgb[pix * 3 + 2] = vb; }
        if (out_u8) {
            const float c3[3] = {vr, vg, vb};
            for (int c = 0; c < 3; ++c) {
                int q = static_cast<int>(255 * c3[c]);
                q = q < 255 ? q : 255;
                out_u8[pix * 3 + c] = static_cast<uint8_t>(q < 0 ? 0 : q);
            }
        }
    }
}

// ======================================================================================================
// lightStrikes on explicit segments (rt_light_strikes): lane = segment light[i] -> hit[i]
// ======================================================================================================
__global__ __launch_bounds__(RT_WAVES * 64) void k_segments(const DNode *__restrict__ nodes, const TriRec *__restrict__ tris,
                                                          const ChunkBound *__restrict__ chunks, const uint32_t *__restrict__ leaf_chunk0,
                                                             const DScene S, const int n, const float *__restrict__ hit,
                                                             const float *__restrict__ light, uint8_t *__restrict__ vis) {
    __shared__ uint4 s_stage[RT_WAVES * RT_STAGE_TRIS * 5];
    __shared__ unsigned long long s_mask[RT_WAVES * RT_STACK];
    __shared__ uint32_t s_node[RT_WAVES * RT_STACK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const WaveStack stk{s_node + (false ? 0 : wave * RT_STACK), s_mask + (false ? 0 : wave * RT_STACK), s_stage + (false ? 0 : wave * RT_STAGE_TRIS * 5)};
    const DNode root = nodes[0];
    const int waves_total = gridDim.x * RT_WAVES;
    for (int base = (blockIdx.x * RT_WAVES + wave) * 64; base < n; base += waves_total * 64) {
        const int i = base + lane;
        const bool valid = i < n;
        const int j = valid ? i : 0;
        const float px = light[j * 3], py = light[j * 3 + 1], pz = light[j * 3 + 2];
        const float ddx = hit[j * 3] - px, ddy = hit[j * 3 + 1] - py, ddz = hit[j * 3 + 2] - pz;
        const float srx = __builtin_amdgcn_rcpf(ddx), sry = __builtin_amdgcn_rcpf(ddy), srz = __builtin_amdgcn_rcpf(ddz);
        const bool sroot = valid && box_hit_verified(root.bmin, px, py, pz, ddx, ddy, ddz, srx, sry, srz);
        float t_unused = 0.f; int f_unused = -1; bool occ = false; uint32_t c0 = 0, c1 = 0, c2 = 0;
        packet_walk<true, false>(nodes, tris, chunks, leaf_chunk0, S.extent, stk, lane, walk_plain(), sroot, px, py, pz, ddx, ddy, ddz, ddx, ddy, ddz, srx, sry, srz, t_unused, f_unused, occ, c0, c1, c2);
        if (valid) vis[i] = occ ? 0 : 1;
    }
}

// ======================================================================================================
// Unit-parity probes (rt_box_intersect / rt_tree_probe / rt_primary_points): the PRODUCT's device functions on caller-given inputs,
// so that tests can pin them directly to outputs of the reference's own boundingBox.cpp / boxTree.cpp / camera.hpp.
// ======================================================================================================
__global__ __launch_bounds__(256) void k_box_probe(const int n, const float *__restrict__ box, const float *__restrict__ org, const float *__restrict__ dst,
                                                   uint8_t *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float b[6] = {box[i * 6], box[i * 6 + 1], box[i * 6 + 2], box[i * 6 + 3], box[i * 6 + 4], box[i * 6 + 5]};
    const float ox = org[i * 3], oy = org[i * 3 + 1], oz = org[i * 3 + 2];
    const float dx = dst[i * 3] - ox, dy = dst[i * 3 + 1] - oy, dz = dst[i * 3 + 2] - oz;        // Eigen: dir = dest - origin (boundingBox.cpp:51)
    out[i] = box_hit_verified(b, ox, oy, oz, dx, dy, dz, __builtin_amdgcn_rcpf(dx), __builtin_amdgcn_rcpf(dy), __builtin_amdgcn_rcpf(dz)) ? 1 : 0;
}

// BoxTree::intersect (boxTree.cpp:150-173) as the traversal kernels perform it, reference semantics (COUNT variant: no early-out, no
// culling): per ray the boxIntersect calls, the leaf face references and a signature of the set of intersected non-empty leaves
// (sum of device node index x 2654435761 mod 2^32).
__global__ __launch_bounds__(RT_WAVES * 64) void k_tree_probe(const DNode *__restrict__ nodes, const TriRec *__restrict__ tris,
                                                              const ChunkBound *__restrict__ chunks, const uint32_t *__restrict__ leaf_chunk0,
                                                              const DScene S, const int n, const float *__restrict__ org, const float *__restrict__ dst,
                                                              uint32_t *__restrict__ out_box, uint32_t *__restrict__ out_ref, uint32_t *__restrict__ out_sig) {
    __shared__ uint4 s_stage[RT_WAVES * RT_STAGE_TRIS * 5];
    __shared__ unsigned long long s_mask[RT_WAVES * RT_STACK];
    __shared__ uint32_t s_node[RT_WAVES * RT_STACK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const WaveStack stk{s_node + wave * RT_STACK, s_mask + wave * RT_STACK, s_stage + wave * RT_STAGE_TRIS * 5};
    const DNode root = nodes[0];
    const int waves_total = gridDim.x * RT_WAVES;
    for (int base = (blockIdx.x * RT_WAVES + wave) * 64; base < n; base += waves_total * 64) {
        const int i = base + lane;
        const bool valid = i < n;
        const int j = valid ? i : 0;
        const float ox = org[j * 3], oy = org[j * 3 + 1], oz = org[j * 3 + 2];
        const float dx = dst[j * 3] - ox, dy = dst[j * 3 + 1] - oy, dz = dst[j * 3 + 2] - oz;
        const float rx = __builtin_amdgcn_rcpf(dx), ry = __builtin_amdgcn_rcpf(dy), rz = __builtin_amdgcn_rcpf(dz);
        // BoxTree::intersect tests the root itself first (the callers' own pre-test of the root is theirs, not part of intersect)
        const bool in_root = valid && box_hit_verified(root.bmin, ox, oy, oz, dx, dy, dz, rx, ry, rz);
        float t_unused = 0.f; int f_unused = -1; bool occ = false; uint32_t c_box = 0, c_ref = 0, c_sig = 0;
        packet_walk<true, true>(nodes, tris, chunks, leaf_chunk0, S.extent, stk, lane, walk_plain(), in_root, ox, oy, oz, dx, dy, dz, dx, dy, dz, rx, ry, rz,
                                t_unused, f_unused, occ, c_box, c_ref, c_sig);
        if (valid) { out_box[i] = c_box + (in_root ? 0u : 1u); out_ref[i] = c_ref; out_sig[i] = c_sig; }
    }
}

__global__ __launch_bounds__(256) void k_primary_probe(const DCam *__restrict__ camp, const int W, const int H, float *__restrict__ out) {
    const DCam cam = *camp;
    const size_t total = static_cast<size_t>(W) * static_cast<size_t>(H);
    for (size_t p = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; p < total; p += static_cast<size_t>(gridDim.x) * blockDim.x) {
        float sx, sy, sz;
        screen_point(cam, static_cast<int>(p % static_cast<size_t>(W)), static_cast<int>(p / static_cast<size_t>(W)), sx, sy, sz);
        out[p * 3] = sx; out[p * 3 + 1] = sy; out[p * 3 + 2] = sz;
    }
}

void launch_box_probe(hipStream_t st, int n, const float *box, const float *org, const float *dst, uint8_t *out) {
    hipLaunchKernelGGL(k_box_probe, dim3((n + 255) / 256), dim3(256), 0, st, n, box, org, dst, out);
}
void launch_tree_probe(int grid, hipStream_t st, const DScene &S, int n, const float *org, const float *dst, uint32_t *out_box, uint32_t *out_ref, uint32_t *out_sig) {
    hipLaunchKernelGGL(k_tree_probe, dim3(grid), dim3(RT_WAVES * 64), 0, st, S.nodes, S.leaf_tris, S.chunks, S.leaf_chunk0, S, n, org, dst, out_box, out_ref, out_sig);
}
void launch_primary_probe(int grid, hipStream_t st, const DCam *cam, int W, int H, float *out) {
    hipLaunchKernelGGL(k_primary_probe, dim3(grid), dim3(256), 0, st, cam, W, H, out);
}

#ifdef RT_PROFILE
__global__ void k_set_prof(Control *ctl, uint32_t base) { g_prof = ctl->prof; g_prof_base = base; }
void launch_set_prof(hipStream_t st, Control *ctl, uint32_t base) { hipLaunchKernelGGL(k_set_prof, dim3(1), dim3(1), 0, st, ctl, base); }
#else
void launch_set_prof(hipStream_t, Control *, uint32_t) {}
#endif

// ------------------------------------------------------------------------------------------------------
// residency: blocks per CU for each persistent kernel (fast variants), queried once per scene
// ------------------------------------------------------------------------------------------------------
void query_occupancy(bool flat, int *trace_primary, int *trace_rays, int *shadow, int *shaft_out, int *shade) {
    int n = 0;
    auto q = [&](auto kernel, int threads, int fallback) {
        return (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, 0) == hipSuccess && n > 0) ? n : fallback;
    };
    if (flat) {
        *trace_primary = q(k_trace<true, false, true>, RT_WAVES * 64, 4);
        *trace_rays = q(k_trace<false, false, true>, RT_WAVES * 64, 4);
        *shadow = q(k_shadow<false, true, false>, RT_WAVES * 64, 4);
        *shaft_out = *shadow;
    } else {
        *trace_primary = q(k_stage<true, false, 0, false>, RT_WAVES * 64, 4);
        *trace_rays = q(k_stage<false, false, 0, false>, RT_WAVES * 64, 4);
        *shadow = q(k_shadow<false, false, false>, RT_WAVES * 64, 4);
        *shaft_out = q((k_shadow_shaft<false, false>), RT_WAVES * 64, 4);      // (its grid used to be the smaller of the two residencies: 4 of its 6 waves per SIMD)
    }
    *shade = flat ? q((k_shade<true, true>), 256, 2) : q((k_shade<true, false>), 256, 2);
}

// ------------------------------------------------------------------------------------------------------
// host-callable launchers (keep <<<>>> syntax inside this translation unit)
// ------------------------------------------------------------------------------------------------------
#define RT_LAUNCH_TRACE(P, C, F) hipLaunchKernelGGL((k_trace<P, C, F>), g, b, 0, st, S.nodes, S.leaf_tris, S.chunks, S.leaf_chunk0, S, camp, L, Fr, level, slot, rays_in, items, ctl, rec, out_hit, out_t)
void launch_trace(bool primary, bool count, bool flat, int grid, hipStream_t st, const DScene &S, const DCam *camp, const DLights &L, const DFrame &Fr,
                  int level, int slot, const RayItem *rays_in, ShadeItem *items, Control *ctl, float4 *rec, int32_t *out_hit, float *out_t) {
    const dim3 g(grid), b(RT_WAVES * 64);
    if (flat) {
        if (primary) { if (count) RT_LAUNCH_TRACE(true, true, true); else RT_LAUNCH_TRACE(true, false, true); }
        else { if (count) RT_LAUNCH_TRACE(false, true, true); else RT_LAUNCH_TRACE(false, false, true); }
    } else {               // fused kernel on a tree scene (the staged k_stage pipeline is the alternative)
        if (primary) { if (count) RT_LAUNCH_TRACE(true, true, false); else RT_LAUNCH_TRACE(true, false, false); }
        else { if (count) RT_LAUNCH_TRACE(false, true, false); else RT_LAUNCH_TRACE(false, false, false); }
    }
}

#define RT_LAUNCH_STAGE(P, C, ST, K) hipLaunchKernelGGL((k_stage<P, C, ST, K>), g, b, 0, st, S.nodes, S.leaf_tris, S.chunks, S.leaf_chunk0, S, camp, L, Fr, level, lslots, rays_in, items, ctl, rec, out_hit, out_t, best, lit, Q)
void launch_stage(bool primary, bool count, int stage, bool cont, int grid, hipStream_t st, const DScene &S, const DCam *camp, const DLights &L,
                  const DFrame &Fr, int level, int lslots, const RayItem *rays_in, ShadeItem *items, Control *ctl, float4 *rec, int32_t *out_hit,
                  float *out_t, unsigned long long *best, unsigned long long *lit, const TaskQueues &Q) {
    const dim3 g(grid), b(RT_WAVES * 64);
    if (cont) {            // continuations exist for the two traversal stages of the fast (non-counting) variants only
        if (primary) { if (stage == 0) RT_LAUNCH_STAGE(true, false, 0, true); else RT_LAUNCH_STAGE(true, false, 1, true); }
        else { if (stage == 0) RT_LAUNCH_STAGE(false, false, 0, true); else RT_LAUNCH_STAGE(false, false, 1, true); }
        return;
    }
    const int sel = (primary ? 6 : 0) + (count ? 3 : 0) + stage;
    switch (sel) {
        case 0: RT_LAUNCH_STAGE(false, false, 0, false); break;
        case 1: RT_LAUNCH_STAGE(false, false, 1, false); break;
        case 2: RT_LAUNCH_STAGE(false, false, 2, false); break;
        case 3: RT_LAUNCH_STAGE(false, true, 0, false); break;
        case 4: RT_LAUNCH_STAGE(false, true, 1, false); break;
        case 5: RT_LAUNCH_STAGE(false, true, 2, false); break;
        case 6: RT_LAUNCH_STAGE(true, false, 0, false); break;
        case 7: RT_LAUNCH_STAGE(true, false, 1, false); break;
        case 8: RT_LAUNCH_STAGE(true, false, 2, false); break;
        case 9: RT_LAUNCH_STAGE(true, true, 0, false); break;
        case 10: RT_LAUNCH_STAGE(true, true, 1, false); break;
        default: RT_LAUNCH_STAGE(true, true, 2, false); break;
    }
}

#define RT_LAUNCH_SHADOW(C, F, K) hipLaunchKernelGGL((k_shadow<C, F, K>), g, b, 0, st, S.nodes, S.leaf_tris, S.chunks, S.leaf_chunk0, S, L, level, slot, lslots, item_cap, items, ctl, vis, Q, sidx)
void launch_shadow(bool count, bool flat, int grid, hipStream_t st, const DScene &S, const DLights &L, int level, int slot, int lslots,
                   uint32_t item_cap, const ShadeItem *items, Control *ctl, unsigned long long *vis, ContTask *tasks_out, uint32_t cap, uint32_t budget, uint32_t target,
                   const uint32_t *sidx) {
    const dim3 g(grid), b(RT_WAVES * 64);
    const TaskQueues Q{nullptr, (flat || count) ? nullptr : tasks_out, 0u, 2u, cap, (flat || count) ? 0u : budget, target};
    if (count) { if (flat) RT_LAUNCH_SHADOW(true, true, false); else RT_LAUNCH_SHADOW(true, false, false); }
    else { if (flat) RT_LAUNCH_SHADOW(false, true, false); else RT_LAUNCH_SHADOW(false, false, false); }
}

void launch_shadow_shaft(int grid, hipStream_t st, const DScene &S, const DLights &L, int level, int slot, int lslots, uint32_t item_cap,
                         const ShadeItem *items, Control *ctl, unsigned long long *vis, ContTask *tasks_out, uint32_t cap, uint32_t budget, uint32_t target, const uint32_t *sidx,
                         const uint8_t *pair_done) {
    TaskQueues Q{nullptr, tasks_out, 0u, 2u, cap, budget, target};
    Q.pair_done = pair_done;
    // (the task emission costs the walking kernel 30 more spilled registers: it is compiled in only when a budget asks for it)
    if (budget != 0u && tasks_out != nullptr)
        hipLaunchKernelGGL((k_shadow_shaft<false, true>), dim3(grid), dim3(RT_WAVES * 64), 0, st, S.nodes, S.leaf_tris, S.chunks, S, L, level, slot, lslots, item_cap, items, ctl, vis, Q, sidx);
    else
        hipLaunchKernelGGL((k_shadow_shaft<false, false>), dim3(grid), dim3(RT_WAVES * 64), 0, st, S.nodes, S.leaf_tris, S.chunks, S, L, level, slot, lslots, item_cap, items, ctl, vis, Q, sidx);
}

// the leaf tasks of a shaft-walk launch: chunk ranges of big leaves, through the same leaf code (shaft_leaf)
void launch_shadow_shaft_cont(int grid, hipStream_t st, const DScene &S, const DLights &L, int level, int lslots, uint32_t item_cap, const ShadeItem *items,
                              Control *ctl, unsigned long long *vis, const ContTask *tasks_in, uint32_t cap, const uint32_t *sidx) {
    const TaskQueues Q{tasks_in, nullptr, 2u, 0u, cap, 0u};
    hipLaunchKernelGGL((k_shadow_shaft<true, false>), dim3(grid), dim3(RT_WAVES * 64), 0, st, S.nodes, S.leaf_tris, S.chunks, S, L, level, 0, lslots, item_cap, items, ctl, vis, Q, sidx);
}

// processes the leaf tasks of queue q_in (leaf tasks never create new tasks)
void launch_shadow_cont(int grid, hipStream_t st, const DScene &S, const DLights &L, int level, int lslots, uint32_t item_cap, const ShadeItem *items,
                        Control *ctl, unsigned long long *vis, const ContTask *tasks_in, ContTask *tasks_out, uint32_t q_in, uint32_t q_out,
                        uint32_t cap, uint32_t budget, const uint32_t *sidx) {
    const dim3 g(grid), b(RT_WAVES * 64);
    const int slot = 0;
    const TaskQueues Q{tasks_in, tasks_out, q_in, q_out, cap, tasks_out ? budget : 0u};
    RT_LAUNCH_SHADOW(false, false, true);
}

void launch_beam(int grid, hipStream_t st, const DScene &S, const DLights &L, int level, int lslots, uint32_t item_cap, const ShadeItem *items, Control *ctl,
                 unsigned long long *vis, uint32_t *sidx) {
    hipLaunchKernelGGL(k_beam, dim3(grid), dim3(RT_WAVES * 64), 0, st, S.nodes, S.leaf_tris, S.chunks, S, L, level, lslots, item_cap, items, ctl, vis, sidx);
}

void launch_pair_beam(int grid, hipStream_t st, const DScene &S, const DLights &L, int level, int lslots, uint32_t item_cap, const ShadeItem *items, Control *ctl,
                      unsigned long long *vis, uint32_t *sidx, uint8_t *done) {
    hipLaunchKernelGGL(k_pair_beam, dim3(grid), dim3(RT_WAVES * 64), 0, st, S.nodes, S.leaf_tris, S.chunks, S, L, level, lslots, item_cap, items, ctl, vis, sidx, done);
}

void launch_shade(int grid, hipStream_t st, const DScene &S, const DLights &L, const DFrame &F, int level, int slot, int lslots,
                  const ShadeItem *items, Control *ctl, const unsigned long long *vis, float4 *rec, float *fres, RayItem *rays_out, bool resolve_flat) {
    const bool simple = L.mode != RT_LIGHT_SPHERE && L.n_samples <= 64;
    const dim3 g(grid), b(256);
#define RT_LAUNCH_SHADE(SI, FL) hipLaunchKernelGGL((k_shade<SI, FL>), g, b, 0, st, S.nodes, S.leaf_tris, S, L, F, level, slot, lslots, items, ctl, vis, rec, fres, rays_out)
    if (simple) { if (resolve_flat) RT_LAUNCH_SHADE(true, true); else RT_LAUNCH_SHADE(true, false); }
    else { if (resolve_flat) RT_LAUNCH_SHADE(false, true); else RT_LAUNCH_SHADE(false, false); }
}

void launch_deep(int grid, hipStream_t st, const DScene &S, const DLights &L, const DFrame &F, int level0, const RayItem *rays_in, Control *ctl, float4 *rec0, float *fres0) {
    hipLaunchKernelGGL(k_deep, dim3(grid), dim3(256), 0, st, S.nodes, S.leaf_tris, S, L, F, level0, rays_in, ctl, rec0, fres0);
}

void launch_resolve(int grid, hipStream_t st, const DFrame &F, const float4 *rec, const float *fres, float *out_rgb, uint8_t *out_u8) {
    hipLaunchKernelGGL(k_resolve, dim3(grid), dim3(256), 0, st, F, rec, fres, out_rgb, out_u8);
}

void launch_segments(int grid, hipStream_t st, const DScene &S, int n, const float *hit, const float *light, uint8_t *vis) {
    hipLaunchKernelGGL(k_segments, dim3(grid), dim3(RT_WAVES * 64), 0, st, S.nodes, S.leaf_tris, S.chunks, S.leaf_chunk0, S, n, hit, light, vis);
}

}  // namespace rtamd
