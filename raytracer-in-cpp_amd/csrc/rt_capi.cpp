// rt_capi.cpp -- implementation of the C ABI declared in include/rt_mi355x.h: context, scene upload,
// frame orchestration (a fixed, host-sync-free launch sequence per frame) and the host-scene wrappers.
// No CPU fallback exists here: without a HIP device every entry point that needs one fails loudly.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "host_scene.hpp"
#include "rt_device.hpp"
#include "rt_mi355x.h"

namespace rtamd {
void launch_trace(bool primary, bool count, bool flat, int grid, hipStream_t st, const DScene &S, const DCam *camp, const DLights &L, const DFrame &F,
                  int level, int slot, const RayItem *rays_in, ShadeItem *items, Control *ctl, float4 *rec, int32_t *out_hit, float *out_t);
void launch_shadow(bool count, bool flat, int grid, hipStream_t st, const DScene &S, const DLights &L, int level, int slot, int lslots,
                   uint32_t item_cap, const ShadeItem *items, Control *ctl, unsigned long long *vis, ContTask *tasks_out, uint32_t cap, uint32_t budget, uint32_t target, const uint32_t *sidx);
void launch_shadow_shaft(int grid, hipStream_t st, const DScene &S, const DLights &L, int level, int slot, int lslots, uint32_t item_cap,
                         const ShadeItem *items, Control *ctl, unsigned long long *vis, ContTask *tasks_out, uint32_t cap, uint32_t budget, uint32_t target, const uint32_t *sidx,
                         const uint8_t *pair_done);
void launch_shadow_shaft_cont(int grid, hipStream_t st, const DScene &S, const DLights &L, int level, int lslots, uint32_t item_cap, const ShadeItem *items,
                              Control *ctl, unsigned long long *vis, const ContTask *tasks_in, uint32_t cap, const uint32_t *sidx);
void launch_beam(int grid, hipStream_t st, const DScene &S, const DLights &L, int level, int lslots, uint32_t item_cap, const ShadeItem *items, Control *ctl,
                 unsigned long long *vis, uint32_t *sidx);
void launch_pair_beam(int grid, hipStream_t st, const DScene &S, const DLights &L, int level, int lslots, uint32_t item_cap, const ShadeItem *items, Control *ctl,
                      unsigned long long *vis, uint32_t *sidx, uint8_t *done);
void launch_shadow_cont(int grid, hipStream_t st, const DScene &S, const DLights &L, int level, int lslots, uint32_t item_cap, const ShadeItem *items,
                        Control *ctl, unsigned long long *vis, const ContTask *tasks_in, ContTask *tasks_out, uint32_t q_in, uint32_t q_out,
                        uint32_t cap, uint32_t budget, const uint32_t *sidx);
void launch_shade(int grid, hipStream_t st, const DScene &S, const DLights &L, const DFrame &F, int level, int slot, int lslots,
                  const ShadeItem *items, Control *ctl, const unsigned long long *vis, float4 *rec, float *fres, RayItem *rays_out, bool resolve_flat);
void launch_resolve(int grid, hipStream_t st, const DFrame &F, const float4 *rec, const float *fres, float *out_rgb, uint8_t *out_u8);
void launch_deep(int grid, hipStream_t st, const DScene &S, const DLights &L, const DFrame &F, int level0, const RayItem *rays_in, Control *ctl, float4 *rec0, float *fres0);
void launch_stage(bool primary, bool count, int stage, bool cont, int grid, hipStream_t st, const DScene &S, const DCam *camp, const DLights &L,
                  const DFrame &Fr, int level, int lslots, const RayItem *rays_in, ShadeItem *items, Control *ctl, float4 *rec, int32_t *out_hit,
                  float *out_t, unsigned long long *best, unsigned long long *lit, const TaskQueues &Q);
void launch_segments(int grid, hipStream_t st, const DScene &S, int n, const float *hit, const float *light, uint8_t *vis);
void launch_box_probe(hipStream_t st, int n, const float *box, const float *org, const float *dst, uint8_t *out);
void launch_tree_probe(int grid, hipStream_t st, const DScene &S, int n, const float *org, const float *dst, uint32_t *out_box, uint32_t *out_ref, uint32_t *out_sig);
void launch_primary_probe(int grid, hipStream_t st, const DCam *cam, int W, int H, float *out);
bool gpu_build_octree(HostScene &hs, int cap, int depth, hipStream_t st, std::string *err);
void query_occupancy(bool flat, int *trace_primary, int *trace_rays, int *shadow, int *shaft, int *shade);
void launch_set_prof(hipStream_t st, Control *ctl, uint32_t base);
}  // namespace rtamd

using namespace rtamd;

struct rt_host_scene {
    HostScene hs;
};

struct rt_ctx {
    int device = 0;
    int cus = 256;
    int occ_trace_primary = 4, occ_trace_rays = 4, occ_shadow = 4, occ_shaft = 4, occ_shade = 2;   // resident blocks per CU
    hipStream_t stream = nullptr;
    std::string err;
    // scene
    bool has_scene = false;
    DScene S{};
    void *d_chunks = nullptr, *d_leaf_chunk0 = nullptr, *d_bad_leaves = nullptr;
    void *d_nodes = nullptr, *d_tris = nullptr, *d_tri_verts = nullptr, *d_face_normal = nullptr, *d_tri_vid = nullptr,
         *d_mat_id = nullptr, *d_vert_normal = nullptr, *d_mats = nullptr;
    bool reflective = false;     // some material spawns bounce rays (illum 3,4,5,6,9)
    bool flat = false;           // the root is a small leaf (cube.obj): specialised stack-free kernels
    int grid_mult = 1;
    int dyn_trace = 0;
    int staged_trace = 1;        // tree scenes: closest / centre / finish kernels with continuation tasks instead of the fused k_trace
    // frame buffers
    size_t cap_pix = 0;
    int cap_levels = 0;
    size_t cap_vis = 0;
    RayItem *d_rays[2] = {nullptr, nullptr};
    ShadeItem *d_items = nullptr;
    unsigned long long *d_vis = nullptr;
    uint32_t *d_sidx = nullptr;           // k_beam's survivors: item storage indices, 16 sub-lists like d_items
    uint8_t *d_done = nullptr;            // k_pair_beam with several lights: one byte per (item slot, light)
    size_t cap_done = 0;
    unsigned long long *d_best = nullptr, *d_lit = nullptr;   // staged trace of tree scenes: closest-hit keys, centre-visibility masks
    size_t cap_lit = 0, cap_best = 0;
    ContTask *d_tasks[2] = {nullptr, nullptr};   // continuation queues of k_shadow (tree scenes)
    uint32_t task_cap = 1u << 21;
    uint32_t trace_budget = 500u;               // leaves above this estimated cost (VALU instructions) become tasks (0 = off); round 3 sweep after the task
                                                // counters were sharded: dodge trace 0.250 / 0.239 / 0.243 ms at 1000 / 500 / 250
    uint32_t group_budget = 4u;                 // groups a trace unit pops before it hands the rest of its stack to the task launch (RT_GROUP_BUDGET, 0 = never)
    uint32_t shadow_budget = 3000u;
    bool beam_trees = false;
    uint32_t item_beam = 1;               // tree scenes, lights of more than 64 samples: the per-hit beam test (k_pair_beam) in front of k_shadow_shaft (RT_ITEM_BEAM=0: off, 2: also for one pass)
    int item_beam_blocks = 6;             // its workgroups per CU (6 waves per SIMD)
    bool deep = true;                     // flat scenes: levels 2 .. max_depth in ONE launch (k_deep); RT_NO_DEEP=1 keeps the four launches per level
    int shaft_min_samples = 33;           // tree scenes: sample counts from which a (hit, light) pair gets a wave of its own (k_shadow_shaft)
    uint32_t shaft_budget = 0u;           // the shaft walk culls per triangle: its leaves are cheap enough to stay inline (dodge 1080p: 1.31 -> 1.22 ms without tasks)
    uint32_t shaft_budget_deep = 3000u;   // ... but the bounce levels have few units and a heavy tail: their big leaves do go to a leaf-task launch (cfg4 29.2 -> 28.3 ms)
    int stage_mult = 2;                         // grid multiplier of the main k_stage launches (RT_STAGE_MULT): twice the resident grid lets
                                                // blocks of sky tiles retire early and evens out the object tiles (dodge trace 0.278 -> 0.254 ms)
    uint32_t task_target = 0u;                  // estimated cost of one leaf-task piece (0 = same as the budget); RT_TASK_TARGET, else set per scene at upload
    uint32_t trace_target = 1000u;              // ... of the trace stages: 1000 on small scenes, 4000 on big ones (cfg4 has > 130 k tasks per stage: trace 2.12 -> 1.76 ms;
                                                // dodge, 9 k tasks: 0.245 vs 0.256 ms the other way) -- by leaf references, see rt_upload_scene
    bool task_target_env = false;
    float4 *d_rec = nullptr;
    float *d_fres = nullptr;
    Control *d_ctl = nullptr;
    DCam *d_cam = nullptr;            // camera of the frame in flight (device memory: graph-replayable)
    DCam *h_cam_ring = nullptr;       // pinned staging ring for asynchronous camera uploads
    uint32_t cam_slot = 0;
    uint64_t frame_generation = 0;    // bumped whenever the frame buffers are reallocated (invalidates captured graphs)
    uint64_t scene_generation = 0;    // bumped by every rt_upload_scene: a captured graph holds the scene's device pointers by value
    hipEvent_t cam_events[512] = {};  // one per camera-ring slot: recorded after the slot's H2D copy, waited for before the slot is reused
    float *d_offsets = nullptr;  // RT_LIGHT_SPHERE sample offsets of the last eager call (a captured graph owns a copy of its own: rt_graph)
    size_t cap_offsets = 0;
    uint32_t frame_launches = 0;               // device operations (kernel launches + memsets) the last run_frame enqueued
    int frame_wide_levels = 0;                 // levels of the last frame that ran the per-level kernel groups (the deeper ones went to k_deep)
    hipStream_t last_frame_stream = nullptr;   // stream of the most recent eager frame (it may still read d_offsets)
    float *d_rgb = nullptr;      // staging for rt_render (host output)
    int32_t *d_hit = nullptr;
    float *d_t = nullptr;
    size_t cap_out = 0;
    std::vector<hipEvent_t> events;
    // deferred timing (collect_stats == 2): events are not reused until rt_timing_collect
    size_t ev_base = 0;                       // first free event index
    std::vector<std::pair<size_t, int>> pending;   // (first event, levels_run) per frame
    hipStream_t pending_stream = nullptr;
    DFrame pending_frame{};
};

static constexpr uint32_t kCamRing = 512;   // camera uploads that may be queued before one is consumed
static const char *k_no_ctx = "rt_mi355x: null context";

#define HIPCHK(ctx, call)                                                                                      \
    do {                                                                                                       \
        hipError_t e_ = (call);                                                                                \
        if (e_ != hipSuccess) {                                                                                \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                                    \
            return RT_ERR_HIP;                                                                                 \
        }                                                                                                      \
    } while (0)

extern "C" const char *rt_version(void) { return "rt_mi355x 0.1 (gfx950)"; }

extern "C" void *rt_stream(rt_ctx *ctx) { return ctx ? static_cast<void *>(ctx->stream) : nullptr; }

extern "C" const char *rt_last_error(const rt_ctx *ctx) { return ctx ? ctx->err.c_str() : k_no_ctx; }

extern "C" void rt_destroy(rt_ctx *c);
extern "C" rt_status rt_create(rt_ctx **out, int device) {
    if (!out) return RT_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        std::fprintf(stderr, "rt_mi355x: no HIP device visible -- this library has no CPU fallback\n");
        return RT_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) return RT_ERR_NO_DEVICE;
    rt_ctx *c = new rt_ctx();
    c->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete c; return RT_ERR_NO_DEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char *dt = std::getenv("RT_TRACE_DYNAMIC")) c->dyn_trace = std::atoi(dt) != 0;
    if (const char *sb = std::getenv("RT_SHADOW_BUDGET")) c->shadow_budget = static_cast<uint32_t>(std::atoi(sb));
    if (const char *sb = std::getenv("RT_SHAFT_BUDGET")) c->shaft_budget = c->shaft_budget_deep = static_cast<uint32_t>(std::atoi(sb));
    if (const char *sm = std::getenv("RT_SHAFT_MIN_SAMPLES")) c->shaft_min_samples = std::atoi(sm);
    if (const char *bt = std::getenv("RT_BEAM_TREES")) c->beam_trees = std::atoi(bt) != 0;
    if (const char *ib = std::getenv("RT_ITEM_BEAM")) c->item_beam = static_cast<uint32_t>(std::max(0, std::atoi(ib)));
    if (const char *ib = std::getenv("RT_ITEM_BEAM_BLOCKS")) c->item_beam_blocks = std::max(1, std::atoi(ib));
    if (std::getenv("RT_NO_DEEP")) c->deep = false;
    if (const char *sg = std::getenv("RT_STAGED_TRACE")) c->staged_trace = std::atoi(sg) != 0;
    if (const char *sm = std::getenv("RT_STAGE_MULT")) { const int v = std::atoi(sm); if (v >= 1 && v <= 8) c->stage_mult = v; }
    if (const char *tc = std::getenv("RT_TASK_CAP")) { const long v = std::atol(tc); if (v >= 64 && v <= (1l << 24)) c->task_cap = static_cast<uint32_t>(v); }
    if (const char *tt = std::getenv("RT_TASK_TARGET")) { c->task_target = static_cast<uint32_t>(std::atoi(tt)); c->task_target_env = true; }
    if (const char *tb = std::getenv("RT_TRACE_BUDGET")) c->trace_budget = static_cast<uint32_t>(std::atoi(tb));
    if (const char *gb = std::getenv("RT_GROUP_BUDGET")) c->group_budget = static_cast<uint32_t>(std::atoi(gb));
    if (const char *gm = std::getenv("RT_GRID_MULT")) {          // tuning knob: grid = CUs x residency x mult
        const int m = std::atoi(gm);
        if (m > 0 && m <= 64) c->grid_mult = m;
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return RT_ERR_HIP; }
    if (hipMalloc(reinterpret_cast<void **>(&c->d_ctl), sizeof(Control)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&c->d_cam), sizeof(DCam)) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **>(&c->h_cam_ring), sizeof(DCam) * kCamRing, hipHostMallocDefault) != hipSuccess) {
        (void)hipStreamDestroy(c->stream);
        delete c;
        return RT_ERR_HIP;
    }
    if (hipMemset(c->d_ctl, 0, sizeof(Control)) != hipSuccess) { rt_destroy(c); return RT_ERR_HIP; }
    *out = c;
    return RT_OK;
}

static void free_scene(rt_ctx *c) {
    void **p[] = {&c->d_bad_leaves, &c->d_chunks, &c->d_leaf_chunk0, &c->d_nodes, &c->d_tris, &c->d_tri_verts, &c->d_face_normal, &c->d_tri_vid, &c->d_mat_id, &c->d_vert_normal, &c->d_mats};
    for (void **q : p) { if (*q) (void)hipFree(*q); *q = nullptr; }
    c->has_scene = false;
}

static void free_frame(rt_ctx *c) {
    if (c->d_rays[0]) (void)hipFree(c->d_rays[0]);
    if (c->d_rays[1]) (void)hipFree(c->d_rays[1]);
    if (c->d_items) (void)hipFree(c->d_items);
    if (c->d_vis) (void)hipFree(c->d_vis);
    if (c->d_sidx) (void)hipFree(c->d_sidx);
    if (c->d_done) (void)hipFree(c->d_done);
    c->d_done = nullptr; c->cap_done = 0;
    if (c->d_best) (void)hipFree(c->d_best);
    if (c->d_lit) (void)hipFree(c->d_lit);
    c->d_best = c->d_lit = nullptr; c->cap_lit = 0; c->cap_best = 0;
    if (c->d_tasks[0]) (void)hipFree(c->d_tasks[0]);
    if (c->d_tasks[1]) (void)hipFree(c->d_tasks[1]);
    c->d_tasks[0] = c->d_tasks[1] = nullptr;
    if (c->d_rec) (void)hipFree(c->d_rec);
    if (c->d_fres) (void)hipFree(c->d_fres);
    c->d_rays[0] = c->d_rays[1] = nullptr; c->d_items = nullptr; c->d_vis = nullptr; c->d_sidx = nullptr; c->d_rec = nullptr; c->d_fres = nullptr;
    c->cap_pix = 0; c->cap_levels = 0; c->cap_vis = 0;
}

extern "C" void rt_destroy(rt_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    free_scene(c);
    free_frame(c);
    if (c->d_rgb) (void)hipFree(c->d_rgb);
    if (c->d_offsets) (void)hipFree(c->d_offsets);
    if (c->d_hit) (void)hipFree(c->d_hit);
    if (c->d_t) (void)hipFree(c->d_t);
    if (c->d_ctl) (void)hipFree(c->d_ctl);
    if (c->d_cam) (void)hipFree(c->d_cam);
    if (c->h_cam_ring) (void)hipHostFree(c->h_cam_ring);
    for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->cam_events) if (e) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

template <typename T>
static rt_status upload(rt_ctx *c, void **dst, const T *src, size_t n) {
    const size_t bytes = (n ? n : 1) * sizeof(T);
    HIPCHK(c, hipMalloc(dst, bytes));
    if (n) HIPCHK(c, hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return RT_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// Conservative chunk bounds for the lanes=triangles leaf mode.
//
// A (ray, chunk) pair may be skipped only if NO triangle of the chunk can pass Flyscene::rayTriangleIntersection as the
// reference evaluates it in float (flyscene.cpp:787-819).  That evaluation accepts a triangle when the barycentric
// coordinates of the COMPUTED point P = o + t*d (projected on the triangle's plane) pass u>=0, v>=0, u+v<1.
//   (1) P lies on the ray line up to rounding (~1e-7 * |P|), whatever the error of t.
//   (2) P is close to the triangle's plane whatever the angle between ray and plane.  With the computed num = n.A - o.n
//       (absolute error dn_ <= ~3e-7*(|A|+|o|)), den = d.n (absolute error dd_ <= ~3e-7*|d|) and t = num/den*(1+e), |e| <= 1e-7:
//           n.P - n.A = t*(den - dd_) - (num - dn_) = num*e - t*dd_ + dn_
//       so |dist(P, plane)| <= 1e-7*|num| + |t|*3e-7*|d| + 3e-7*(|A|+|o|) -- no division by den anywhere.  If |t||d| <= 4(|o|+extent)
//       this is <= ~2e-6*(|o|+extent).  If |t||d| > 4(|o|+extent), P is more than 2.3*extent away from the origin along some
//       axis, i.e. far outside every triangle, and so is its projection (P is within 3e-7*|t||d| of the plane): the true
//       barycentrics are >= ~1.3 in magnitude and their relative error (3) cannot flip a sign -- never accepted.
//   (3) With kappa = d00*d11/denom (conditioning of the reference's barycentric solve) the errors of u and v are
//       <= ~20*eps*kappa*(1+|u|+|v|), i.e. P's projection is inside the triangle grown by 1.2e-6*kappa*edge.
//   => an accepted hit implies the computed point P = o + t*d lies within  2e-6*(|o|+extent) + 1.2e-6*kappa*edge  of the
//      triangle, hence inside the chunk's inflated box, and t itself lies in the box's [t_in, t_out] of that line.
// So the chunk AABB is inflated by max(5e-6*kappa*edge) + 1e-3*max_edge + 1e-4*extent here (kappa <= 1e4 required) and the
// kernel adds 4e-4*(|o|_1+extent) per ray; a ray skips a chunk when its line misses that box or [t_in, t_out] lies outside
// the t range a hit can count in.  (An earlier version also demanded |d.n| > 0.002|d| for every triangle of the chunk, from
// a bound on dist(P, plane) that went through the RELATIVE error of d.n; the absolute form above makes that guard, and the
// per-chunk normal cone that short-cut it, unnecessary: -7 % instructions on dodgeColorTest.obj's k_shadow.)
// Chunks holding an ill-conditioned, degenerate, non-unit-normal or non-finite triangle are never cullable.
// ---------------------------------------------------------------------------------------------------------------
static uint32_t morton3(uint32_t x, uint32_t y, uint32_t z) {
    auto spread = [](uint32_t v) {
        v &= 0x3ffu; v = (v | (v << 16)) & 0x30000ffu; v = (v | (v << 8)) & 0x300f00fu;
        v = (v | (v << 4)) & 0x30c30c3u; v = (v | (v << 2)) & 0x9249249u; return v;
    };
    return spread(x) | (spread(y) << 1) | (spread(z) << 2);
}

// A triangle that rayTriangleIntersection (flyscene.cpp:787-819) can never accept, whatever the ray: a zero face normal gives
// dn = d.n = 0 for every finite direction (`dn == 0` -> rejected); a NaN normal makes t, u, v NaN (every comparison false); and when the
// float denominator d00*d11 - d01*d01 -- evaluated as the reference evaluates it -- is 0 or NaN, 1/denom is inf / NaN and u, v are each
// +-inf or NaN: u >= 0 && v >= 0 && u + v < 1 cannot hold.  Collinear triangles of real meshes are mostly of this kind; the others (a
// denominator that is one rounding error instead of zero) report hits wherever their plane is crossed and stay un-cullable.
static bool never_hit(const float *v, const float *nn) {
    if (!(nn[0] == nn[0]) || !(nn[1] == nn[1]) || !(nn[2] == nn[2])) return true;
    if (nn[0] == 0.0f && nn[1] == 0.0f && nn[2] == 0.0f) {
        for (int k = 0; k < 9; ++k) if (!std::isfinite(v[k])) return false;
        return true;
    }
    const V3 A{v[0], v[1], v[2]}, B{v[3], v[4], v[5]}, C{v[6], v[7], v[8]};
    const V3 e0 = C - A, e1 = B - A;
    const float d00 = dot(e0, e0), d01 = dot(e0, e1), d11 = dot(e1, e1);
    const float inv = 1 / (d00 * d11 - d01 * d01);
    return !std::isfinite(inv);
}

// A triangle the chunk bounds cannot vouch for: non-finite, a face normal that is not unit, or a barycentric solve conditioned worse than
// kappa = d00 d11 / denom = 1e4 (edges from vertex A, as the reference sets it up).  Its computed hit points are not tied to the triangle, so
// no region bounds them: the chunk that holds it is never culled as a whole and the triangle itself never by the per-triangle shaft test
// (TriRec::flags bit 1).  Otherwise: the in-plane growth (3) of the error analysis above and the longer of its two edges.
static bool tri_ill_conditioned(const float *v, const float *nn, double &edge, double &bary_infl) {
    edge = 0; bary_infl = 0;
    for (int k = 0; k < 9; ++k) if (!std::isfinite(v[k])) return true;
    const double nl = std::sqrt(double(nn[0]) * nn[0] + double(nn[1]) * nn[1] + double(nn[2]) * nn[2]);
    if (!std::isfinite(nl) || std::fabs(nl - 1.0) > 1e-3) return true;   // Face::normal is unit unless degenerate
    double e0[3], e1[3];
    for (int k = 0; k < 3; ++k) { e0[k] = double(v[6 + k]) - v[k]; e1[k] = double(v[3 + k]) - v[k]; }
    const double d00 = e0[0] * e0[0] + e0[1] * e0[1] + e0[2] * e0[2], d11 = e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2];
    const double d01 = e0[0] * e1[0] + e0[1] * e1[1] + e0[2] * e1[2];
    const double den = d00 * d11 - d01 * d01;
    if (!(d00 > 0) || !(d11 > 0) || !(den > 1e-4 * d00 * d11)) return true;   // kappa = d00*d11/den <= 1e4
    edge = std::sqrt(std::max(d00, d11));
    // (3): |error(u)|, |error(v)| <= ~20*eps*kappa = 1.2e-6*kappa  ->  in-plane growth 1.2e-6*kappa*edge (x4 safety)
    bary_infl = 5e-6 * (d00 * d11 / den) * edge;
    return false;
}

static void build_chunk_bounds(const rt_scene *sc, std::vector<uint32_t> &refs, std::vector<uint32_t> &leaf_chunk0,
                               std::vector<ChunkBound> &out, float extent, bool no_cull) {
    for (uint32_t ni = 0; ni < sc->n_nodes; ++ni) {
        const rt_node &nd = sc->nodes[ni];
        if (!(nd.count_flags & RT_NODE_LEAF)) continue;
        const uint32_t cnt = nd.count_flags & 0x7fffffffu;
        leaf_chunk0[ni] = static_cast<uint32_t>(out.size());
        if (cnt == 0) continue;
        uint32_t *r = refs.data() + nd.first;
        // Morton order of the centroids inside the leaf box
        float ext[3];
        for (int k = 0; k < 3; ++k) ext[k] = nd.bmax[k] - nd.bmin[k];
        std::vector<std::pair<uint32_t, uint32_t>> keyed(cnt);
        for (uint32_t i = 0; i < cnt; ++i) {
            const float *v = sc->tri_verts + static_cast<size_t>(r[i]) * 9;
            uint32_t q[3];
            for (int k = 0; k < 3; ++k) {
                const float cen = (v[k] + v[3 + k] + v[6 + k]) / 3.0f;
                float u = ext[k] > 0.f ? (cen - nd.bmin[k]) / ext[k] : 0.f;
                if (!(u > 0.f)) u = 0.f;
                if (u > 1.f) u = 1.f;
                q[k] = static_cast<uint32_t>(u * 1023.0f);
            }
            keyed[i] = {morton3(q[0], q[1], q[2]), r[i]};
        }
        std::sort(keyed.begin(), keyed.end());
        for (uint32_t i = 0; i < cnt; ++i) r[i] = keyed[i].second;
        for (uint32_t c0 = 0; c0 < cnt; c0 += 64) {
            const uint32_t n = cnt - c0 < 64 ? cnt - c0 : 64;
            ChunkBound cb{};
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, max_edge = 0, bary_infl = 0;
            bool ok = !no_cull;
            uint32_t live_tris = 0;
            double nsum[3] = {0, 0, 0}, nfirst[3] = {0, 0, 0};
            for (uint32_t i = 0; i < n; ++i) {
                const uint32_t f = r[c0 + i];
                const float *v = sc->tri_verts + static_cast<size_t>(f) * 9;
                const float *nn = sc->face_normal + static_cast<size_t>(f) * 3;
                if (never_hit(v, nn)) continue;         // cannot be hit by any ray AS THE REFERENCE COMPUTES IT: no bound needed for it
                double edge, binfl;
                if (tri_ill_conditioned(v, nn, edge, binfl)) { ok = false; continue; }      // (the well-conditioned ones still get their inflation: below)
                ++live_tris;
                for (int k = 0; k < 3; ++k) {
                    lo[k] = std::min(lo[k], double(std::min(v[k], std::min(v[3 + k], v[6 + k]))));
                    hi[k] = std::max(hi[k], double(std::max(v[k], std::max(v[3 + k], v[6 + k]))));
                }
                max_edge = std::max(max_edge, edge);
                bary_infl = std::max(bary_infl, binfl);
                // area-weighted mean normal (e1 x e0 has twice the area as its length), orientation of the chunk's first triangle
                double e0[3], e1[3];
                for (int k = 0; k < 3; ++k) { e0[k] = double(v[6 + k]) - v[k]; e1[k] = double(v[3 + k]) - v[k]; }
                const double cr[3] = {e1[1] * e0[2] - e1[2] * e0[1], e1[2] * e0[0] - e1[0] * e0[2], e1[0] * e0[1] - e1[1] * e0[0]};
                if (live_tris == 1) for (int k = 0; k < 3; ++k) nfirst[k] = cr[k];
                const double sg = (cr[0] * nfirst[0] + cr[1] * nfirst[1] + cr[2] * nfirst[2]) < 0 ? -1.0 : 1.0;
                for (int k = 0; k < 3; ++k) nsum[k] += sg * cr[k];
            }
            if (ok) {
                const double infl = bary_infl + 1e-3 * max_edge + 1e-4 * extent;
                for (int k = 0; k < 3; ++k) {
                    cb.lo[k] = std::nextafter(static_cast<float>(lo[k] - infl), -INFINITY);
                    cb.hi[k] = std::nextafter(static_cast<float>(hi[k] + infl), INFINITY);
                }
                cb.never = 0.0f;
                cb.infl = std::nextafter(static_cast<float>(infl), INFINITY);
                // the slab: any unit direction gives a valid bound (the point of an accepted hit lies within infl per axis of its triangle, a
                // convex combination of the vertices), the mean normal gives the thin one
                double nl = std::sqrt(nsum[0] * nsum[0] + nsum[1] * nsum[1] + nsum[2] * nsum[2]);
                double sn[3] = {0, 0, 1};
                if (std::isfinite(nl) && nl > 1e-30) for (int k = 0; k < 3; ++k) sn[k] = nsum[k] / nl;
                for (int k = 0; k < 3; ++k) cb.sn[k] = static_cast<float>(sn[k]);
                double slo = 1e300, shi = -1e300;
                for (uint32_t i = 0; i < n; ++i) {
                    const uint32_t f = r[c0 + i];
                    const float *v = sc->tri_verts + static_cast<size_t>(f) * 9;
                    if (never_hit(v, sc->face_normal + static_cast<size_t>(f) * 3)) continue;
                    for (int q = 0; q < 3; ++q) {
                        const double dq = double(cb.sn[0]) * v[3 * q] + double(cb.sn[1]) * v[3 * q + 1] + double(cb.sn[2]) * v[3 * q + 2];
                        slo = std::min(slo, dq); shi = std::max(shi, dq);
                    }
                }
                const double sinfl = 1.0625 * (std::fabs(double(cb.sn[0])) + std::fabs(double(cb.sn[1])) + std::fabs(double(cb.sn[2]))) * double(cb.infl);
                cb.slo = live_tris ? std::nextafter(static_cast<float>(slo - sinfl), -INFINITY) : -3e38f;
                cb.shi = live_tris ? std::nextafter(static_cast<float>(shi + sinfl), INFINITY) : 3e38f;
            }
            if (ok && live_tris == 0) { for (int k = 0; k < 3; ++k) cb.lo[k] = cb.hi[k] = 1e30f; cb.infl = 0.0f; }     // nothing hittable inside: a far-away point
            if (!ok) {
                // never culled as a whole; its well-conditioned triangles keep THEIR inflation for the per-triangle shaft test of the shadow units
                // (0: there is none -- no_cull, or every triangle of the chunk is ill-conditioned)
                const double infl = bary_infl + 1e-3 * max_edge + 1e-4 * extent;
                cb = ChunkBound{}; cb.never = 2.0f; cb.slo = -3e38f; cb.shi = 3e38f;
                cb.infl = (!no_cull && live_tris) ? std::nextafter(static_cast<float>(infl), INFINITY) : 0.0f;
            }
            out.push_back(cb);
        }
    }
    if (out.empty()) { ChunkBound cb{}; cb.never = 2.0f; cb.slo = -3e38f; cb.shi = 3e38f; out.push_back(cb); }
}

// host-only: builds the chunk bounds of a flattened scene and reports {chunks, cullable chunks, leaves, max chunks per leaf}
extern "C" rt_status rt_debug_chunk_stats(const rt_scene *sc, int32_t out[4]) {
    if (!sc || !out || !sc->nodes) return RT_ERR_INVALID;
    std::vector<uint32_t> refs(sc->face_refs, sc->face_refs + sc->n_face_refs);
    std::vector<uint32_t> leaf_chunk0(sc->n_nodes, 0u);
    std::vector<ChunkBound> cbs;
    float extent = 0.f;
    for (size_t i = 0; i < static_cast<size_t>(sc->n_faces) * 9; ++i) extent = std::fmax(extent, std::fabs(sc->tri_verts[i]));
    build_chunk_bounds(sc, refs, leaf_chunk0, cbs, extent, false);
    int cullable = 0, leaves = 0, maxc = 0;
    for (const ChunkBound &cb : cbs) cullable += cb.never < 1.5f ? 1 : 0;
    for (uint32_t i = 0; i < sc->n_nodes; ++i)
        if (sc->nodes[i].count_flags & RT_NODE_LEAF) {
            ++leaves;
            const int nc = static_cast<int>(((sc->nodes[i].count_flags & 0x7fffffffu) + 63u) / 64u);
            if (nc > maxc) maxc = nc;
        }
    out[0] = static_cast<int32_t>(cbs.size()); out[1] = cullable; out[2] = leaves; out[3] = maxc;
    return RT_OK;
}

// host-only: the chunk bounds themselves, 16 floats per chunk {lo[3], hi[3], never, infl, sn[3], slo, shi, 0, 0, 0}, the per-node index of a
// leaf's first chunk and the leaf face references in the order the chunks hold them (tests/test_host_scene.py checks the containment the
// culling rules rely on)
extern "C" rt_status rt_debug_chunk_bounds(const rt_scene *sc, float *bounds, int32_t cap_chunks, int32_t *n_chunks, uint32_t *leaf_chunk0_out, uint32_t *refs_out) {
    if (!sc || !sc->nodes || !n_chunks) return RT_ERR_INVALID;
    std::vector<uint32_t> refs(sc->face_refs, sc->face_refs + sc->n_face_refs);
    std::vector<uint32_t> leaf_chunk0(sc->n_nodes, 0u);
    std::vector<ChunkBound> cbs;
    float extent = 0.f;
    for (size_t i = 0; i < static_cast<size_t>(sc->n_faces) * 9; ++i) extent = std::fmax(extent, std::fabs(sc->tri_verts[i]));
    build_chunk_bounds(sc, refs, leaf_chunk0, cbs, extent, false);
    *n_chunks = static_cast<int32_t>(cbs.size());
    if (bounds) {
        if (cap_chunks < *n_chunks) return RT_ERR_INVALID;
        static_assert(sizeof(ChunkBound) == 16 * sizeof(float), "ChunkBound is 16 floats");
        std::memcpy(bounds, cbs.data(), cbs.size() * sizeof(ChunkBound));
    }
    if (leaf_chunk0_out) std::memcpy(leaf_chunk0_out, leaf_chunk0.data(), leaf_chunk0.size() * sizeof(uint32_t));
    if (refs_out) std::memcpy(refs_out, refs.data(), refs.size() * sizeof(uint32_t));
    return RT_OK;
}

extern "C" rt_status rt_upload_scene(rt_ctx *c, const rt_scene *sc) {
    if (!c) return RT_ERR_INVALID;
    if (!sc || !sc->nodes || sc->n_nodes == 0 || !sc->materials || sc->n_materials == 0) { c->err = "rt_upload_scene: empty scene"; return RT_ERR_INVALID; }
    if (sc->n_faces && (!sc->tri_verts || !sc->face_normal || !sc->tri_vid || !sc->mat_id || !sc->vert_normal)) { c->err = "rt_upload_scene: missing arrays"; return RT_ERR_INVALID; }
    HIPCHK(c, hipSetDevice(c->device));
    // validate indices and the depth bound of the traversal stack before anything reaches a kernel
    std::vector<int> depth(sc->n_nodes, 0);
    int max_depth = 0;
    for (uint32_t i = 0; i < sc->n_nodes; ++i) {
        const rt_node &n = sc->nodes[i];
        const uint32_t cnt = n.count_flags & 0x7fffffffu;
        if (n.count_flags & RT_NODE_LEAF) {
            if (static_cast<uint64_t>(n.first) + cnt > sc->n_face_refs) { c->err = "rt_upload_scene: leaf range outside face_refs"; return RT_ERR_INVALID; }
        } else {
            if (cnt > 8 || (cnt && (n.first <= i || static_cast<uint64_t>(n.first) + cnt > sc->n_nodes))) { c->err = "rt_upload_scene: bad child range"; return RT_ERR_INVALID; }
            for (uint32_t k = 0; k < cnt; ++k) depth[n.first + k] = depth[i] + 1;
        }
        if (depth[i] > max_depth) max_depth = depth[i];
    }
    if (max_depth > 16) { c->err = "rt_upload_scene: octree deeper than 16 levels (traversal stack bound)"; return RT_ERR_UNSUPPORTED; }
    for (uint32_t i = 0; i < sc->n_face_refs; ++i)
        if (sc->face_refs[i] >= sc->n_faces) { c->err = "rt_upload_scene: face ref out of range"; return RT_ERR_INVALID; }
    for (uint32_t f = 0; f < sc->n_faces; ++f) {
        if (sc->mat_id[f] < 0 || static_cast<uint32_t>(sc->mat_id[f]) >= sc->n_materials) { c->err = "rt_upload_scene: material id out of range"; return RT_ERR_INVALID; }
        for (int k = 0; k < 3; ++k)
            if (sc->tri_vid[f * 3 + k] >= sc->n_vert_normals) { c->err = "rt_upload_scene: vertex id outside vert_normal"; return RT_ERR_INVALID; }
    }

    // Leaf face lists are re-ordered along a Morton curve (the order inside a leaf cannot change any result: closest
    // hit is a minimum with a face-id tie-break, shadow rays are any-hit), so that every run of 64 references -- one
    // `chunk` of the lanes=triangles mode -- is spatially compact and its conservative bound is tight.
    std::vector<uint32_t> refs(sc->face_refs, sc->face_refs + sc->n_face_refs);
    std::vector<uint32_t> leaf_chunk0(sc->n_nodes, 0u);
    std::vector<ChunkBound> cbs;
    float extent = 0.f;
    for (size_t i = 0; i < static_cast<size_t>(sc->n_faces) * 9; ++i) extent = std::fmax(extent, std::fabs(sc->tri_verts[i]));
    const bool no_cull = std::getenv("RT_NO_CULL") != nullptr;
    build_chunk_bounds(sc, refs, leaf_chunk0, cbs, extent, no_cull);

    // leaf-ordered triangle records: the per-triangle constants of rayTriangleIntersection (flyscene.cpp:787-811),
    // evaluated once with the same float operations the reference performs on every call
    std::vector<TriRec> recs(sc->n_face_refs);
    for (uint32_t i = 0; i < sc->n_face_refs; ++i) {
        const uint32_t f = refs[i];
        const float *v = sc->tri_verts + static_cast<size_t>(f) * 9;
        const float *n = sc->face_normal + static_cast<size_t>(f) * 3;
        TriRec &r = recs[i];
        const V3 A{v[0], v[1], v[2]}, B{v[3], v[4], v[5]}, C{v[6], v[7], v[8]}, N{n[0], n[1], n[2]};
        const V3 e0 = C - A, e1 = B - A;
        r.ax = A.x; r.ay = A.y; r.az = A.z;
        r.e0x = e0.x; r.e0y = e0.y; r.e0z = e0.z;
        r.e1x = e1.x; r.e1y = e1.y; r.e1z = e1.z;
        r.nx = N.x; r.ny = N.y; r.nz = N.z;
        r.nA = dot(N, A);
        r.d00 = dot(e0, e0); r.d01 = dot(e0, e1); r.d11 = dot(e1, e1);
        r.inv_denom = 1 / (r.d00 * r.d11 - r.d01 * r.d01);
        r.face = f;
        double edge_, binfl_;
        r.flags = (sc->materials[sc->mat_id[f]].illum == 9 ? 1u : 0u) | ((!never_hit(v, n) && tri_ill_conditioned(v, n, edge_, binfl_)) ? 2u : 0u);
        r.pad = 0u;
    }
    c->reflective = false;
    for (uint32_t m = 0; m < sc->n_materials; ++m) {
        const int il = sc->materials[m].illum;
        if (il == 9 || il == 6 || (il > 2 && il < 6)) c->reflective = true;
    }

    HIPCHK(c, hipStreamSynchronize(c->stream));      // nothing in flight may still read the old scene
    free_scene(c);
    ++c->scene_generation;
    rt_status st;
    // device nodes = public nodes + content boxes, bottom-up (children always follow their parent in the array)
    std::vector<DNode> dnodes(sc->n_nodes);
    std::vector<float> bad_leaves;            // boxes of the leaves with a chunk that may never be culled (k_beam)
    for (uint32_t ii = sc->n_nodes; ii-- > 0;) {
        const rt_node &n = sc->nodes[ii];
        DNode &dn = dnodes[ii];
        std::memcpy(&dn, &n, sizeof(rt_node));
        const uint32_t cnt = n.count_flags & 0x7fffffffu;
        float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {1e30f, 1e30f, 1e30f};      // nothing inside: a far-away point
        bool any = false, open_box = false;
        auto grow = [&](const float *l, const float *h) {
            for (int k = 0; k < 3; ++k) {
                lo[k] = any ? std::fmin(lo[k], l[k]) : l[k];
                hi[k] = any ? std::fmax(hi[k], h[k]) : h[k];
            }
            any = true;
        };
        if (n.count_flags & RT_NODE_LEAF) {
            for (uint32_t k = 0; k < (cnt + 63u) / 64u; ++k) {
                const ChunkBound &cb = cbs[leaf_chunk0[ii] + k];
                if (cb.never >= 1.5f) open_box = true; else grow(cb.lo, cb.hi);
            }
            if (open_box) { for (int k = 0; k < 3; ++k) bad_leaves.push_back(n.bmin[k]); for (int k = 0; k < 3; ++k) bad_leaves.push_back(n.bmax[k]); }
        } else {
            for (uint32_t k = 0; k < cnt; ++k) {
                const DNode &ch = dnodes[n.first + k];
                if (ch.pad[1]) open_box = true;
                if (!(ch.clo[0] >= 1e30f)) grow(ch.clo, ch.chi);
            }
        }
        // The content box bounds the CULLABLE chunks below; pad[1] = 1 marks a subtree that also holds a chunk that may never be skipped
        // (a degenerate / ill-conditioned triangle inside): the walk then descends on the reference's own box test alone and the leaf's
        // cullable chunks are still skipped one by one.  (Opening the whole content box instead let 15 sliver triangles of
        // dodgeColorTest.obj -- 23 of its 437 chunks -- switch the content culling off for most of the top of the tree.)
        bool finite = true;
        for (int k = 0; k < 3; ++k) finite = finite && std::isfinite(lo[k]) && std::isfinite(hi[k]);
        if (!finite) open_box = true;
        for (int k = 0; k < 3; ++k) {
            dn.clo[k] = finite ? lo[k] : 1e30f;
            dn.chi[k] = finite ? hi[k] : 1e30f;
        }
        dn.pad[0] = (n.count_flags & RT_NODE_LEAF) ? leaf_chunk0[ii] : 0u;
        dn.pad[1] = open_box ? 1u : 0u;
    }
    if ((st = upload(c, &c->d_nodes, dnodes.data(), dnodes.size())) != RT_OK) return st;
    if ((st = upload(c, &c->d_tris, recs.data(), recs.size())) != RT_OK) return st;
    if ((st = upload(c, &c->d_chunks, cbs.data(), cbs.size())) != RT_OK) return st;
    if ((st = upload(c, &c->d_leaf_chunk0, leaf_chunk0.data(), leaf_chunk0.size())) != RT_OK) return st;
    const size_t n_bad = bad_leaves.size() / 6;
    if (bad_leaves.empty()) bad_leaves.assign(6, 0.0f);
    if ((st = upload(c, &c->d_bad_leaves, bad_leaves.data(), bad_leaves.size())) != RT_OK) return st;
    if ((st = upload(c, &c->d_tri_verts, sc->tri_verts, static_cast<size_t>(sc->n_faces) * 9)) != RT_OK) return st;
    if ((st = upload(c, &c->d_face_normal, sc->face_normal, static_cast<size_t>(sc->n_faces) * 3)) != RT_OK) return st;
    if ((st = upload(c, &c->d_tri_vid, sc->tri_vid, static_cast<size_t>(sc->n_faces) * 3)) != RT_OK) return st;
    if ((st = upload(c, &c->d_mat_id, sc->mat_id, sc->n_faces)) != RT_OK) return st;
    if ((st = upload(c, &c->d_vert_normal, sc->vert_normal, static_cast<size_t>(sc->n_vert_normals) * 3)) != RT_OK) return st;
    if ((st = upload(c, &c->d_mats, sc->materials, sc->n_materials)) != RT_OK) return st;
    c->S.nodes = static_cast<const DNode *>(c->d_nodes);
    c->S.leaf_tris = static_cast<const TriRec *>(c->d_tris);
    c->S.chunks = static_cast<const ChunkBound *>(c->d_chunks);
    c->S.leaf_chunk0 = static_cast<const uint32_t *>(c->d_leaf_chunk0);
    c->S.bad_leaves = static_cast<const float *>(c->d_bad_leaves);
    c->S.n_bad_leaves = n_bad <= 32 ? static_cast<uint32_t>(n_bad) : 0xffffffffu;
    c->S.extent = extent;
    c->S.tri_verts = static_cast<const float *>(c->d_tri_verts);
    c->S.face_normal = static_cast<const float *>(c->d_face_normal);
    c->S.tri_vid = static_cast<const uint32_t *>(c->d_tri_vid);
    c->S.mat_id = static_cast<const int32_t *>(c->d_mat_id);
    c->S.vert_normal = static_cast<const float *>(c->d_vert_normal);
    c->S.mats = static_cast<const rt_material *>(c->d_mats);
    std::memcpy(c->S.model, sc->model, sizeof(float) * 12);
    c->S.n_nodes = sc->n_nodes;
    c->S.n_faces = sc->n_faces;
    c->S.queue_local = -1;        // auto (rt_kernels.hip, k_shadow); RT_QUEUE_LOCAL=n forces chunks of n consecutive units, 0 the strided mode
    if (const char *ql = std::getenv("RT_QUEUE_LOCAL")) c->S.queue_local = std::atoi(ql);
    c->S.plane_cull = (no_cull || std::getenv("RT_NO_PLANE_CULL") != nullptr) ? 0 : 1;
    c->S.queue_div = 12;          // (units / (waves x 12) per chunk: dodge 1.125 -> 1.112 ms against 6, measured at the kernel's full residency)
    if (const char *qd = std::getenv("RT_QUEUE_DIV")) { const int v = std::atoi(qd); if (v >= 1 && v <= 4096) c->S.queue_div = v; }
    c->S.shaft = (no_cull || std::getenv("RT_NO_SHAFT") != nullptr) ? 0 : 1;
#ifdef RT_UNIT_HIST
    {   // (leaked at rt_destroy: diagnostic build)
        static uint32_t *dbg_buf = nullptr;
        if (!dbg_buf && hipMalloc(reinterpret_cast<void **>(&dbg_buf), (static_cast<size_t>(RT_UNIT_DBG_WORDS) + RT_UNIT_DBG_SHAFT) * sizeof(uint32_t)) == hipSuccess)
            (void)hipMemset(dbg_buf, 0, (static_cast<size_t>(RT_UNIT_DBG_WORDS) + RT_UNIT_DBG_SHAFT) * sizeof(uint32_t));
        c->S.dbg = dbg_buf;
    }
#endif
    c->S.beam = (no_cull || std::getenv("RT_NO_BEAM") != nullptr) ? 0 : 1;
    c->S.beam_budget = 1024;
    if (const char *bb = std::getenv("RT_BEAM_BUDGET")) { const int v = std::atoi(bb); if (v >= 1 && v <= (1 << 20)) c->S.beam_budget = v; }
    if (sc->n_nodes >= (1u << 28)) { c->err = "rt_upload_scene: more than 2^28 nodes"; return RT_ERR_UNSUPPORTED; }
    c->flat = (sc->nodes[0].count_flags & RT_NODE_LEAF) && (sc->nodes[0].count_flags & 0x7fffffffu) <= 64u;
    { const uint32_t t = sc->n_face_refs / 256u; c->trace_target = t < 1000u ? 1000u : (t > 4000u ? 4000u : t); }
    query_occupancy(c->flat, &c->occ_trace_primary, &c->occ_trace_rays, &c->occ_shadow, &c->occ_shaft, &c->occ_shade);
    // k_trace uses static tile striding: with more than ~4 blocks/CU a wave owns so few tiles (32,400 tiles at 1080p)
    // that heavy object tiles no longer average out (measured 0.26 ms at 4 blocks/CU vs 0.38 ms at 7-8)
    int trace_cap = 6;         // (round 3, RT_TRACE_OCC sweep on the cube frame: 3 / 4 / 5 / 6 / 8 blocks per CU -> trace group 76 / 80 / 76 / 68 / 78 us; round 1's cap of 4
                               // dated from the single-counter compaction lists)
    if (const char *tc = std::getenv("RT_TRACE_OCC")) { const int v = std::atoi(tc); if (v >= 1 && v <= 8) trace_cap = v; }
    if (c->occ_trace_primary > trace_cap) c->occ_trace_primary = trace_cap;
    if (c->occ_trace_rays > trace_cap) c->occ_trace_rays = trace_cap;
    c->occ_trace_primary *= c->grid_mult; c->occ_trace_rays *= c->grid_mult; c->occ_shadow *= c->grid_mult; c->occ_shaft *= c->grid_mult; c->occ_shade *= c->grid_mult;
    c->has_scene = true;
    return RT_OK;
}

extern "C" int32_t rt_local_rows(const rt_params *p) {
    if (!p || p->stripe <= 0 || p->nranks <= 0 || p->row1 < p->row0) return 0;
    int32_t n = 0;
    for (int32_t y = p->row0; y < p->row1; ++y)
        if (((y - p->row0) / p->stripe) % p->nranks == p->rank) ++n;
    return n;
}

// `own_offsets` (rt_graph_create): the sphere offsets go to a NEW device buffer handed to the caller instead of the context's buffer --
// a captured graph holds the pointer by value, so it must not be the buffer later calls rewrite or reallocate.
static rt_status check_lights(rt_ctx *c, const rt_lights *l, DLights *out, float **own_offsets = nullptr) {
    if (!l || l->n_lights < 1 || l->n_lights > RT_MAX_LIGHTS) { c->err = "lights: n_lights must be in 1..25 (the reference overflows bool[25] beyond)"; return RT_ERR_INVALID; }
    if (l->mode != RT_LIGHT_POINT && l->mode != RT_LIGHT_AREA && l->mode != RT_LIGHT_SPHERE) { c->err = "lights: mode must be point, area or sphere"; return RT_ERR_INVALID; }
    int ns = 1;
    if (l->mode == RT_LIGHT_SPHERE) {
        if (l->n_offsets < 1 || l->n_offsets > RT_MAX_SAMPLES || !l->offsets) { c->err = "lights: sphere mode needs 1..1024 offsets"; return RT_ERR_INVALID; }
        ns = l->n_offsets;
    }
    if (l->mode == RT_LIGHT_AREA) {
        if (l->usteps < 1 || l->vsteps < 1 || static_cast<long>(l->usteps) * l->vsteps > RT_MAX_SAMPLES) { c->err = "lights: usteps*vsteps must be in 1..1024"; return RT_ERR_INVALID; }
        ns = l->usteps * l->vsteps;
    }
    std::memset(out, 0, sizeof *out);
    std::memcpy(out->pos, l->pos, sizeof out->pos);
    std::memcpy(out->color, l->color, sizeof out->color);
    out->n_lights = l->n_lights; out->mode = l->mode;
    out->usteps = l->mode == RT_LIGHT_AREA ? l->usteps : 1;
    out->vsteps = l->mode == RT_LIGHT_AREA ? l->vsteps : 1;
    out->n_samples = ns; out->len_x = l->len_x; out->len_y = l->len_y;
    out->offsets = nullptr;
    if (l->mode == RT_LIGHT_SPHERE) {
        // the offsets travel to a device buffer (synchronous copy: sphere mode is not a latency path); their box bounds the samples
        const size_t bytes = static_cast<size_t>(ns) * 3 * sizeof(float);
        if (own_offsets) {
            HIPCHK(c, hipMalloc(reinterpret_cast<void **>(own_offsets), bytes));
            HIPCHK(c, hipMemcpy(*own_offsets, l->offsets, bytes, hipMemcpyHostToDevice));
            out->offsets = *own_offsets;
        } else {
            // a frame in flight -- on the context's stream or on the caller's stream of the last eager frame -- may still read the buffer
            HIPCHK(c, hipStreamSynchronize(c->stream));
            // (a caller's stream that has been destroyed since has no work left: its error is not ours)
            if (c->last_frame_stream && c->last_frame_stream != c->stream && hipStreamSynchronize(c->last_frame_stream) != hipSuccess) (void)hipGetLastError();
            c->last_frame_stream = nullptr;
            if (bytes > c->cap_offsets) {
                if (c->d_offsets) (void)hipFree(c->d_offsets);
                c->d_offsets = nullptr; c->cap_offsets = 0;
                HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->d_offsets), bytes));
                c->cap_offsets = bytes;
            }
            HIPCHK(c, hipMemcpy(c->d_offsets, l->offsets, bytes, hipMemcpyHostToDevice));
            out->offsets = c->d_offsets;
        }
        for (int k = 0; k < 3; ++k) { out->obox[k] = l->offsets[k]; out->obox[3 + k] = l->offsets[k]; }
        for (int i = 1; i < ns; ++i)
            for (int k = 0; k < 3; ++k) {
                out->obox[k] = std::fmin(out->obox[k], l->offsets[i * 3 + k]);
                out->obox[3 + k] = std::fmax(out->obox[3 + k], l->offsets[i * 3 + k]);
            }
    }
    return RT_OK;
}

// Sizes of the per-frame lists.  `tiles` bounds the dense tile/group numbering of every level: the primary 8x8 tiles, or
// ceil(n/64) + RT_LIST_SHARDS for a sharded list of n <= npix elements (each shard rounds up to whole groups of 64).  A
// list shard receives the elements of every RT_LIST_SHARDS-th tile, hence the per-shard capacity.
static size_t frame_tiles(const DFrame &F) {
    return std::max(static_cast<size_t>(F.tiles_x) * static_cast<size_t>(F.tiles_y), static_cast<size_t>(F.npix) / 64 + 1 + RT_LIST_SHARDS);
}
static uint32_t list_cap(size_t tiles) { return static_cast<uint32_t>(((tiles + RT_LIST_SHARDS - 1) / RT_LIST_SHARDS + 1) * 64); }

static rt_status ensure_frame(rt_ctx *c, size_t npix_frame, int levels, size_t samples_words, size_t tiles, size_t lslots) {
    const size_t lit_words = tiles * lslots, best_slots = tiles * 64;
    const size_t npix = std::max(npix_frame, static_cast<size_t>(list_cap(tiles)) * RT_LIST_SHARDS);   // list storage (all shards)
    const size_t vis_words = npix * lslots * samples_words;
    if (npix > c->cap_pix || levels > c->cap_levels || vis_words > c->cap_vis || lit_words > c->cap_lit || best_slots > c->cap_best) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        const size_t np = npix > c->cap_pix ? npix : c->cap_pix;
        const int lv = levels > c->cap_levels ? levels : c->cap_levels;
        const size_t vw = vis_words > c->cap_vis ? vis_words : c->cap_vis;
        const size_t lw = lit_words > c->cap_lit ? lit_words : c->cap_lit;
        const size_t bs = best_slots > c->cap_best ? best_slots : c->cap_best;
        free_frame(c);
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->d_rays[0]), np * sizeof(RayItem)));
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->d_rays[1]), np * sizeof(RayItem)));
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->d_items), np * sizeof(ShadeItem)));
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->d_vis), vw * sizeof(unsigned long long)));
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->d_sidx), np * sizeof(uint32_t)));
        // staged trace: one 64-bit closest-hit key per ray slot of every 8x8 tile (tiles are padded to 64 lanes), lit masks per (tile, light)
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->d_best), (bs ? bs : 64) * sizeof(unsigned long long)));
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->d_lit), (lw ? lw : 1) * sizeof(unsigned long long)));
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->d_tasks[0]), static_cast<size_t>(c->task_cap) * sizeof(ContTask)));
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->d_tasks[1]), static_cast<size_t>(c->task_cap) * sizeof(ContTask)));
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->d_rec), np * static_cast<size_t>(lv) * sizeof(float4)));
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->d_fres), np * static_cast<size_t>(lv) * sizeof(float)));
        c->cap_pix = np; c->cap_levels = lv; c->cap_vis = vw; c->cap_lit = lw; c->cap_best = bs;
        ++c->frame_generation;
    }
    // k_pair_beam's (item, light) bytes: only frames with several lights use them
    if (lslots > 1 && c->cap_pix * lslots > c->cap_done) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->d_done) (void)hipFree(c->d_done);
        c->d_done = nullptr; c->cap_done = 0;
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->d_done), c->cap_pix * lslots + 64));
        c->cap_done = c->cap_pix * lslots;
        ++c->frame_generation;
    }
    return RT_OK;
}

static hipEvent_t event_at(rt_ctx *c, size_t i) {
    while (c->events.size() <= i) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        c->events.push_back(e);
    }
    return c->events[i];
}

// asynchronous camera upload through the pinned ring: a slot is only rewritten after the copy that last read it has completed
static rt_status upload_camera(rt_ctx *c, const DCam &dc, hipStream_t st) {
    const uint32_t slot_i = c->cam_slot++ % kCamRing;
    if (c->cam_events[slot_i]) HIPCHK(c, hipEventSynchronize(c->cam_events[slot_i]));
    else HIPCHK(c, hipEventCreateWithFlags(&c->cam_events[slot_i], hipEventDisableTiming));
    DCam *slot = &c->h_cam_ring[slot_i];
    *slot = dc;
    HIPCHK(c, hipMemcpyAsync(c->d_cam, slot, sizeof(DCam), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipEventRecord(c->cam_events[slot_i], st));
    return RT_OK;
}

// One frame = memset(control) ; per level { trace ; shadow ; shade } ; resolve -- no host synchronisation inside.
static rt_status run_frame(rt_ctx *c, hipStream_t st, const DCam *cam, const DLights &L, DFrame F, bool primary, bool count,
                           float *d_rgb, uint8_t *d_u8, int32_t *d_hit, float *d_t, int timed, uint32_t n_input_rays) {
    const int D = F.max_depth;
    // bounce levels can only be populated when some material reflects/refracts
    const int levels_run = c->reflective ? D + 1 : 1;
    const int lslots = L.n_lights;
    const size_t P = (static_cast<size_t>(L.n_samples) + 63) / 64;
    const size_t tiles = frame_tiles(F);
    rt_status s = ensure_frame(c, F.npix, D + 1, P, tiles, static_cast<size_t>(lslots));
    if (s != RT_OK) return s;
    F.item_cap = F.ray_cap = list_cap(tiles);
    uint32_t nl = 1;             // device operations of this frame: this memset + every kernel launch below
    HIPCHK(c, hipMemsetAsync(c->d_ctl, 0, kFrameClearBytes, st));        // everything but the sticky overflow word
    launch_set_prof(st, c->d_ctl, 0u);   // no-op unless built with -DRT_PROFILE
    if (!primary) ++nl;
    if (!primary) HIPCHK(c, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(&c->d_ctl->n_rays[0][0]), static_cast<int>(n_input_rays), 1, st));
    size_t ev = c->ev_base;
    if (cam) {   // (skipped when replaying a captured graph)
        rt_status cs = upload_camera(c, *cam, st);
        if (cs != RT_OK) return cs;
    }
    // timed == 1: an event between every pair of launches (per-kernel breakdown; adds ~4 us per boundary)
    // timed == 2: lean set for timed loops -- frame start, around each k_shadow launch, frame end
    if (timed) HIPCHK(c, hipEventRecord(event_at(c, ev++), st));
    // flat scenes: the levels from 2 on are ONE launch (k_deep); the counting pass keeps the per-level kernels (its variants count per kernel)
    const bool deep = c->flat && c->deep && !count && levels_run > 2;
    const int wide_levels = deep ? 2 : levels_run;
    for (int level = 0; level < wide_levels; ++level) {
        float4 *rec_l = c->d_rec + static_cast<size_t>(level) * F.npix;
        float *fres_l = c->d_fres + static_cast<size_t>(level) * F.npix;
        const bool prim = primary && level == 0;
        const int tgrid = c->cus * (prim ? c->occ_trace_primary : c->occ_trace_rays);
        int32_t *hit_l = level == 0 ? d_hit : nullptr;
        float *t_l = level == 0 ? d_t : nullptr;
        if (c->flat || !c->staged_trace) {
            ++nl, launch_trace(prim, count, c->flat, tgrid, st, c->S, c->d_cam, L, F, level, 3 * level, c->d_rays[level & 1], c->d_items, c->d_ctl, rec_l, hit_l, t_l);
        } else {
            // tree scenes: closest hit -> light-centre visibility -> finish; each traversal stage writes its big leaves as
            // chunk-range tasks that a second launch spreads over all waves
            const uint32_t B = count ? 0u : c->trace_budget, cap = c->task_cap;
            for (int stage = 0; stage < 2; ++stage) {
                const uint32_t q0 = static_cast<uint32_t>(stage);
                ++nl, launch_stage(prim, count, stage, false, tgrid * c->stage_mult, st, c->S, c->d_cam, L, F, level, lslots, c->d_rays[level & 1], c->d_items, c->d_ctl, rec_l,
                             hit_l, t_l, c->d_best, c->d_lit, TaskQueues{nullptr, B ? c->d_tasks[stage] : nullptr, 0u, q0, cap, B, c->task_target_env ? c->task_target : c->trace_target, count ? 0u : c->group_budget});
                if (B != 0u)
                    ++nl, launch_stage(prim, false, stage, true, tgrid, st, c->S, c->d_cam, L, F, level, lslots, c->d_rays[level & 1], c->d_items, c->d_ctl, rec_l,
                                 hit_l, t_l, c->d_best, c->d_lit, TaskQueues{c->d_tasks[stage], nullptr, q0, 0u, cap, 0u});
            }
            ++nl, launch_stage(prim, count, 2, false, tgrid, st, c->S, c->d_cam, L, F, level, lslots, c->d_rays[level & 1], c->d_items, c->d_ctl, rec_l, hit_l, t_l,
                         c->d_best, c->d_lit, TaskQueues{nullptr, nullptr, 0u, 0u, cap, 0u});
        }
        if (timed) HIPCHK(c, hipEventRecord(event_at(c, ev++), st));
        launch_set_prof(st, c->d_ctl, RT_WORK_SHADOW);
        // first the beam test of whole tiles of 64 hits (k_beam: the hits whose sample rays nothing can block get their visibility words
        // there and never become shadow units), then the survivors
        // (tree scenes: off unless RT_BEAM_TREES=1.  Measured on dodgeColorTest.obj 1080p/64: 84 % of the 3,332 tiles come out unblocked -- none
        // of their hits can reach a leaf with one of the model's degenerate triangles -- and the shadow units drop from 213k to 33k, but a beam
        // walks ~370 steps alone in its wave (k_beam 0.30 ms) and the units that remain are the expensive ones (penumbra, cluttered parts:
        // 0.60 ms of the former 0.77): 0.90 ms against 0.77.  cfg4: 9 % unblocked; the launch's own brake stops testing after 4k of 18k tiles.)
        // tree scenes with one (hit, light) pair per wave: the shaft walk (rt_kernels.hip, k_shadow_shaft), behind the per-hit beam test (k_beam_items)
        const bool shaft = !c->flat && !count && c->S.shaft != 0 && L.n_samples >= c->shaft_min_samples;
        // (one pass per pair -- 64 samples or fewer: a beam costs about 1.7 units and replaces one; dodge 1080p/64: 1.05 -> 1.50 ms.  Four passes, cfg4:
        //  nine pairs in ten are decided by their beam, k_shadow_shaft 22.6 -> 11.7 ms behind 6.5 ms of beams)
        const bool item_beam = shaft && c->S.beam != 0 && !c->beam_trees && (c->item_beam >= 2u || (c->item_beam == 1u && P > 1));
        const bool beam = !count && c->S.beam != 0 && (c->flat || c->beam_trees);
        const uint32_t *sidx = (beam || item_beam) ? c->d_sidx : nullptr;
        if (beam) ++nl, launch_beam(c->cus * 4, st, c->S, L, level, lslots, F.item_cap, c->d_items, c->d_ctl, c->d_vis, c->d_sidx);
        const uint8_t *pair_done = (item_beam && lslots > 1) ? c->d_done : nullptr;
        if (item_beam) ++nl, launch_pair_beam(c->cus * c->item_beam_blocks, st, c->S, L, level, lslots, F.item_cap, c->d_items, c->d_ctl, c->d_vis, c->d_sidx, lslots > 1 ? c->d_done : nullptr);
        const uint32_t shaft_b = level == 0 ? c->shaft_budget : c->shaft_budget_deep;
        if (shaft)
            ++nl, launch_shadow_shaft(c->cus * c->occ_shaft, st, c->S, L, level, 3 * level + 1, lslots, F.item_cap, c->d_items, c->d_ctl, c->d_vis,
                                c->d_tasks[0], c->task_cap, shaft_b, c->task_target, sidx, pair_done);
        else
            ++nl, launch_shadow(count, c->flat, c->cus * c->occ_shadow, st, c->S, L, level, 3 * level + 1, lslots, F.item_cap, c->d_items, c->d_ctl, c->d_vis,
                          c->d_tasks[0], c->task_cap, c->shadow_budget, c->task_target, sidx);
        if (shaft && shaft_b != 0u)
            ++nl, launch_shadow_shaft_cont(c->cus * c->occ_shaft, st, c->S, L, level, lslots, F.item_cap, c->d_items, c->d_ctl, c->d_vis, c->d_tasks[0], c->task_cap, sidx);
        else if (!shaft && !c->flat && !count && c->shadow_budget != 0u)      // the big leaves of the shadow units, spread over all waves
            ++nl, launch_shadow_cont(c->cus * c->occ_shadow, st, c->S, L, level, lslots, F.item_cap, c->d_items, c->d_ctl, c->d_vis, c->d_tasks[0], nullptr, 2u, 0u,
                               c->task_cap, 0u, sidx);
        if (timed) HIPCHK(c, hipEventRecord(event_at(c, ev++), st));   // after the whole shadow group (incl. continuations)
        launch_set_prof(st, c->d_ctl, 0u);
        ++nl, launch_shade(c->cus * c->occ_shade, st, c->S, L, F, level, 3 * level + 2, lslots, c->d_items, c->d_ctl, c->d_vis, rec_l, fres_l, c->d_rays[(level + 1) & 1],
                           c->flat && c->deep && !count && level + 1 < levels_run);
        if (timed) HIPCHK(c, hipEventRecord(event_at(c, ev++), st));        // after k_shade (lean timing too: the shade interval is a single kernel)
    }
    if (deep) {
        ++nl, launch_deep(c->cus * c->occ_shade, st, c->S, L, F, 2, c->d_rays[0], c->d_ctl, c->d_rec + 2 * static_cast<size_t>(F.npix), c->d_fres + 2 * static_cast<size_t>(F.npix));
        // (the event layout stays three per level: the deep launch is booked as the trace interval of level 2, the other intervals are empty)
        if (timed) for (int k = 0; k < 3 * (levels_run - 2); ++k) HIPCHK(c, hipEventRecord(event_at(c, ev++), st));
    }
    DFrame Fr = F;
    Fr.max_depth = levels_run - 1;
    ++nl, launch_resolve(c->cus * 8, st, Fr, c->d_rec, c->d_fres, d_rgb, d_u8);
    if (timed) HIPCHK(c, hipEventRecord(event_at(c, ev++), st));
    c->frame_launches = nl;
    c->frame_wide_levels = wide_levels;
    HIPCHK(c, hipGetLastError());
    return RT_OK;
}

static rt_status sum_frame_times(rt_ctx *c, size_t ev, int levels_run, rt_stats *out, bool lean) {
    float ms = 0.f;
    const size_t first = ev;
    for (int level = 0; level < levels_run; ++level) {
        if (lean) {
            // events: [.. trace ..] E [beam, shadow] E [shade] E [.. next trace ..]
            HIPCHK(c, hipEventElapsedTime(&ms, c->events[ev], c->events[ev + 1])); out->ms_trace += ms; ++ev;
            HIPCHK(c, hipEventElapsedTime(&ms, c->events[ev], c->events[ev + 1])); out->ms_shadow += ms; ++ev;
            HIPCHK(c, hipEventElapsedTime(&ms, c->events[ev], c->events[ev + 1])); out->ms_shade += ms; ++ev;
        } else {
            HIPCHK(c, hipEventElapsedTime(&ms, c->events[ev], c->events[ev + 1])); out->ms_trace += ms; ++ev;
            HIPCHK(c, hipEventElapsedTime(&ms, c->events[ev], c->events[ev + 1])); out->ms_shadow += ms; ++ev;
            HIPCHK(c, hipEventElapsedTime(&ms, c->events[ev], c->events[ev + 1])); out->ms_shade += ms; ++ev;
        }
    }
    if (!lean) { HIPCHK(c, hipEventElapsedTime(&ms, c->events[ev], c->events[ev + 1])); out->ms_resolve += ms; }
    HIPCHK(c, hipEventElapsedTime(&ms, c->events[first], c->events[ev + 1])); out->ms_total += ms;
    const uint32_t wide = static_cast<uint32_t>(c->frame_wide_levels > 0 && c->frame_wide_levels < levels_run ? c->frame_wide_levels : levels_run);
    out->launches_trace += wide;
    out->launches_shadow += wide;
    out->launches_shade += wide;
    out->launches_total = c->frame_launches;
    return RT_OK;
}

// after a synchronise: did a kernel of the last frame fail to reserve list space?  (never expected; see list_cap)
static rt_status check_overflow(rt_ctx *c) {
    uint32_t ov = 0;
    HIPCHK(c, hipMemcpy(&ov, &c->d_ctl->overflow, sizeof ov, hipMemcpyDeviceToHost));
    if (ov) {
        HIPCHK(c, hipMemset(&c->d_ctl->overflow, 0, sizeof ov));
        c->err = "internal: a compaction list overflowed its capacity; a frame since the last synchronising call is incomplete";
        return RT_ERR_HIP;
    }
    return RT_OK;
}

static rt_status fill_stats(rt_ctx *c, hipStream_t st, const DFrame &F, int levels_run, bool timed, rt_stats *out, bool counted) {
    HIPCHK(c, hipStreamSynchronize(st));
    Control h;
    HIPCHK(c, hipMemcpy(&h, c->d_ctl, sizeof h, hipMemcpyDeviceToHost));
    { const rt_status os_ = check_overflow(c); if (os_ != RT_OK) return os_; }
    fold_stats(h);
    out->launches_total = c->frame_launches;
    if (std::getenv("RT_DEBUG")) std::fprintf(stderr, "RT_DEBUG level0: items %u tasks closest %u %u centre %u %u shadow %u %u\n", [&] { uint32_t t = 0; for (int sh = 0; sh < RT_LIST_SHARDS; ++sh) t += h.n_items[0][sh * 16]; return t; }(), [&] { uint32_t t = 0; for (int sh = 0; sh < RT_LIST_SHARDS; ++sh) t += h.n_task_tr[0][0][sh * 16]; return t; }(), 0u, [&] { uint32_t t = 0; for (int sh = 0; sh < RT_LIST_SHARDS; ++sh) t += h.n_task_tr[0][1][sh * 16]; return t; }(), 0u, [&] { uint32_t t = 0; for (int sh = 0; sh < RT_LIST_SHARDS; ++sh) t += h.n_task_sh[0][sh * 16]; return t; }(), 0u);
    out->rays_primary = h.rays_primary; out->rays_bounce = h.rays_bounce; out->rays_centre = h.rays_centre; out->rays_sample = h.rays_sample; out->rays_sample_walked = h.sample_walked;
    out->pixels = F.npix; out->pixels_culled = h.pixels_culled; out->shaded_hits = h.shaded_hits;
#ifdef RT_UNIT_HIST
    if (!counted && c->S.dbg != nullptr) {
        if (const char *dump = std::getenv("RT_UNIT_DUMP")) {
            std::vector<uint32_t> hbuf(static_cast<size_t>(RT_UNIT_DBG_WORDS) + RT_UNIT_DBG_SHAFT);
            HIPCHK(c, hipMemcpy(hbuf.data(), c->S.dbg, hbuf.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
            if (FILE *f = std::fopen(dump, "wb")) { std::fwrite(hbuf.data(), sizeof(uint32_t), hbuf.size(), f); std::fclose(f); }
        }
        HIPCHK(c, hipMemset(c->S.dbg, 0, (static_cast<size_t>(RT_UNIT_DBG_WORDS) + RT_UNIT_DBG_SHAFT) * sizeof(uint32_t)));
    }
#endif
#ifdef RT_PROFILE
    if (!counted) {
        std::fprintf(stderr, "RT_PROFILE ray-mode steps %llu useful %llu | tri-mode steps %llu useful %llu | box steps %llu useful %llu | leaves ray %llu tri %llu live-at-tri %llu | chunk-culled (ray,chunk) pairs %llu\n",
                     h.prof[0], h.prof[1], h.prof[2], h.prof[3], h.prof[4], h.prof[5], h.prof[6], h.prof[7], h.prof[8], h.prof[14]);
        std::fprintf(stderr, "RT_PROFILE plane cull: ray-mode triangles skipped %llu, tri-mode chunks skipped %llu; content-box culled (ray,node) pairs %llu\n", h.prof[64], h.prof[65], h.prof[66]);
        std::fprintf(stderr, "RT_PROFILE shadow units %llu: cycles max %llu mean %.0f; log2 histogram:", h.prof[11], h.prof[9], h.prof[11] ? double(h.prof[10]) / double(h.prof[11]) : 0.0);
        for (int b = 8; b <= 30; ++b) std::fprintf(stderr, " [2^%d]=%llu", b, h.prof[16 + b]);
        std::fprintf(stderr, "\n");
        {
            static const char *nm[8] = {"walk-other", "pop+node-load(shaft: group load+shaft test)", "inner-children(shaft: survivors per-ray)", "leaf-tri-mode(shaft: leaves)", "leaf-scalar", "leaf-staged", "unit-setup(+queue)", "unit-finish"};
            for (int k = 0; k < 2; ++k) {
                unsigned long long tot = 0; for (int i = 0; i < 8; ++i) tot += h.prof[480 + 8 * k + i];
                std::fprintf(stderr, "RT_PROFILE k_shadow%s wave-cycles by phase (total %llu):", k ? "<CONT>" : "", tot);
                for (int i = 0; i < 8; ++i) std::fprintf(stderr, " %s=%.1f%%", nm[i], tot ? 100.0 * double(h.prof[480 + 8 * k + i]) / double(tot) : 0.0);
                std::fprintf(stderr, "\n");
            }
        }
        {
            static const char *nm2[8] = {"other", "group-load+cone-test / pop", "per-ray-children", "leaf-tri-mode", "leaf-scalar", "leaf-staged", "tile-setup", "-"};
            for (int k = 0; k < 4; ++k) {         // stage 0 walk, stage 0 tasks, stage 1 walk, stage 1 tasks (level 0)
                unsigned long long tot = 0; for (int i = 0; i < 8; ++i) tot += h.prof[592 + 8 * k + i];
                const int b = 624 + 4 * k;
                const double span = (h.prof[b + 2] && h.prof[b + 3]) ? double(h.prof[b + 3] - ~h.prof[b + 2]) / 100.0 : 0.0;
                std::fprintf(stderr, "RT_PROFILE k_stage<%d%s> waves %llu, lifetimes sum %.1f us, launch span %.1f us (busy %.0f%%), wave-cycles by phase (total %llu):", k / 2, (k & 1) ? ",CONT" : "",
                             h.prof[b + 1], double(h.prof[b]) / 100.0, span, (span > 0 && h.prof[b + 1]) ? 100.0 * double(h.prof[b]) / 100.0 / (span * double(h.prof[b + 1])) : 0.0, tot);
                for (int i = 0; i < 7; ++i) std::fprintf(stderr, " %s=%.1f%%", nm2[i], tot ? 100.0 * double(h.prof[592 + 8 * k + i]) / double(tot) : 0.0);
                std::fprintf(stderr, "\n");
            }
        }
        {
            std::fprintf(stderr, "RT_PROFILE k_shadow_shaft waves %llu: sum of wave lifetimes %.1f us, last end - origin: see bins (50 us each):", h.prof[103], double(h.prof[102]) / 100.0);
            for (int b = 0; b < 40; ++b) if (h.prof[110 + b]) std::fprintf(stderr, " [%d]=%llu", b, h.prof[110 + b]);
            std::fprintf(stderr, "\n");
        }
        std::fprintf(stderr, "RT_PROFILE   unit durations (10 ns ticks), max %llu; log2 bins:", h.prof[559]);
        for (int b = 0; b < 31; ++b) if (h.prof[560 + b]) std::fprintf(stderr, " [2^%d]=%llu", b, h.prof[560 + b]);
        std::fprintf(stderr, "\n");
        std::fprintf(stderr, "RT_PROFILE   wave START bins:");
        for (int b = 0; b < 40; ++b) if (h.prof[520 + b]) std::fprintf(stderr, " [%d]=%llu", b, h.prof[520 + b]);
        std::fprintf(stderr, "\n");
        for (int x = 0; x < 8; ++x) {
            std::fprintf(stderr, "RT_PROFILE   XCC %d (rays %llu):", x, h.prof[500 + x]);
            for (int b = 0; b < 40; ++b) if (h.prof[160 + x * 40 + b]) std::fprintf(stderr, " [%d]=%llu", b, h.prof[160 + x * 40 + b]);
            std::fprintf(stderr, "\n");
        }
        std::fprintf(stderr, "RT_PROFILE shaft walk: groups %llu children %llu shaft-survivors %llu per-ray-survivors %llu | leaf visits %llu chunks %llu shaft-kept %llu with-todo %llu\n",
                     h.prof[RT_WORK_SHADOW + 88], h.prof[RT_WORK_SHADOW + 89], h.prof[RT_WORK_SHADOW + 90], h.prof[RT_WORK_SHADOW + 91], h.prof[RT_WORK_SHADOW + 92], h.prof[RT_WORK_SHADOW + 93], h.prof[RT_WORK_SHADOW + 94], h.prof[RT_WORK_SHADOW + 95]);
        std::fprintf(stderr, "RT_PROFILE slowest trace tile %llu: ray-mode leaf triangles %llu, tri-mode leaf triangles %llu, tri-mode (ray,chunk) tests %llu, child boxes %llu\n",
                     h.prof[56], h.prof[57], h.prof[58], h.prof[59], h.prof[60]);
        std::fprintf(stderr, "RT_PROFILE trace tiles: cycles max %llu sum %llu; log2 histogram:", h.prof[38], h.prof[39]);
        for (int b = 8; b <= 23; ++b) std::fprintf(stderr, " [2^%d]=%llu", b, h.prof[40 + b]);
        std::fprintf(stderr, "\n");
    }
#endif
    if (counted) {
        out->box_tests = h.box_tests + h.box_tests_shadow; out->leaf_tri_refs = h.leaf_tri_refs + h.leaf_tri_refs_shadow;
        out->box_tests_shadow = h.box_tests_shadow; out->leaf_tri_refs_shadow = h.leaf_tri_refs_shadow;
    }
    if (timed) {
        out->ms_trace = out->ms_shadow = out->ms_shade = out->ms_resolve = out->ms_total = 0.f;
        out->launches_trace = out->launches_shadow = out->launches_shade = 0;
        rt_status s = sum_frame_times(c, c->ev_base, levels_run, out, false);
        if (s != RT_OK) return s;
    }
    return RT_OK;
}

static rt_status make_frame(rt_ctx *c, const rt_params *p, DFrame *F) {
    if (!p || p->width <= 0 || p->height <= 0) { c->err = "params: width/height must be positive"; return RT_ERR_INVALID; }
    if (p->stripe <= 0 || p->nranks <= 0 || p->rank < 0 || p->rank >= p->nranks || p->row0 < 0 || p->row1 > p->height || p->row0 > p->row1) {
        c->err = "params: bad row shard (row0,row1,stripe,rank,nranks)";
        return RT_ERR_INVALID;
    }
    if (p->max_depth > RT_MAX_DEPTH) { c->err = "params: max_depth above RT_MAX_DEPTH"; return RT_ERR_UNSUPPORTED; }
    F->width = p->width; F->height = p->height;
    F->local_rows = rt_local_rows(p);
    F->row0 = p->row0; F->stripe = p->stripe; F->rank = p->rank; F->nranks = p->nranks;
    F->tiles_x = (p->width + 7) / 8; F->tiles_y = (F->local_rows + 7) / 8;
    F->npix = static_cast<uint32_t>(F->local_rows) * static_cast<uint32_t>(p->width);
    F->max_depth = p->max_depth < 0 ? RT_MAX_DEPTH : p->max_depth;
    F->dyn_trace = c->dyn_trace;
    return RT_OK;
}

static void make_cam(const rt_camera *cam, DCam *d) {
    std::memcpy(d->center, cam->center, sizeof d->center);
    std::memcpy(d->inv_view, cam->inv_view, sizeof d->inv_view);
    std::memcpy(d->vp, cam->viewport, sizeof d->vp);
    // getPerspectiveScale / scale (camera.hpp:164-166, 263-266): host double arithmetic, as the reference
    const float persp = static_cast<float>(static_cast<double>(1.0f) / std::tan(static_cast<double>(cam->fovy / 2.0f) * (M_PI / static_cast<double>(180.0f))));
    const float scale = static_cast<float>(1.0 / static_cast<double>(persp));
    d->k0 = cam->aspect * scale;
    d->k1 = scale;
}

// waits for every frame the context has enqueued (its own stream and the stream of the most recent rt_render_device call) and reports a
// work-list overflow of any of them
extern "C" rt_status rt_synchronize(rt_ctx *c) {
    if (!c) return RT_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->last_frame_stream && c->last_frame_stream != c->stream) HIPCHK(c, hipStreamSynchronize(c->last_frame_stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return c->d_ctl ? check_overflow(c) : RT_OK;
}

extern "C" rt_status rt_render_device(rt_ctx *c, const rt_camera *cam, const rt_lights *lights, const rt_params *p,
                                      float *d_out_rgb, uint8_t *d_out_u8, int32_t *d_out_hit, void *stream, rt_stats *stats) {
    if (!c) return RT_ERR_INVALID;
    if (!c->has_scene) { c->err = "render before rt_upload_scene"; return RT_ERR_NO_SCENE; }
    if (!cam) { c->err = "camera is null"; return RT_ERR_INVALID; }
    HIPCHK(c, hipSetDevice(c->device));
    DLights L;
    rt_status s = check_lights(c, lights, &L);
    if (s != RT_OK) return s;
    DFrame F;
    if ((s = make_frame(c, p, &F)) != RT_OK) return s;
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (F.npix == 0) return RT_OK;
    DCam dc;
    make_cam(cam, &dc);
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : c->stream;
    c->last_frame_stream = st;
    const int levels_run = c->reflective ? F.max_depth + 1 : 1;
    if (stats && p->collect_stats == 1) {
        // counting pass: same frame with the no-early-out traversal variants (never part of a timed region)
        if ((s = run_frame(c, st, &dc, L, F, true, true, d_out_rgb, d_out_u8, d_out_hit, nullptr, 0, 0)) != RT_OK) return s;
        if ((s = fill_stats(c, st, F, levels_run, false, stats, true)) != RT_OK) return s;
    }
    if (p->collect_stats == 2) {
        const size_t first = c->ev_base;
        if ((s = run_frame(c, st, &dc, L, F, true, false, d_out_rgb, d_out_u8, d_out_hit, nullptr, 2, 0)) != RT_OK) return s;
        c->pending.emplace_back(first, levels_run);
        c->ev_base = first + static_cast<size_t>(3 * levels_run + 2);
        c->pending_stream = st;
        c->pending_frame = F;
        return RT_OK;
    }
    if ((s = run_frame(c, st, &dc, L, F, true, false, d_out_rgb, d_out_u8, d_out_hit, nullptr, stats != nullptr ? 1 : 0, 0)) != RT_OK) return s;
    if (stats) {
        const uint64_t bt = stats->box_tests, lr = stats->leaf_tri_refs, bts = stats->box_tests_shadow, lrs = stats->leaf_tri_refs_shadow;
        if ((s = fill_stats(c, st, F, levels_run, true, stats, false)) != RT_OK) return s;
        stats->box_tests = bt; stats->leaf_tri_refs = lr; stats->box_tests_shadow = bts; stats->leaf_tri_refs_shadow = lrs;
    }
    return RT_OK;
}

struct rt_graph {
    rt_ctx *ctx = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    DFrame F{};
    uint64_t generation = 0, scene_generation = 0;
    hipStream_t last_stream = nullptr;
    float *d_offsets = nullptr;       // RT_LIGHT_SPHERE: the graph's own copy of the sample offsets (the kernel arguments hold this pointer)
};

extern "C" rt_status rt_graph_create(rt_ctx *c, const rt_lights *lights, const rt_params *p, float *d_out_rgb, uint8_t *d_out_u8,
                                     rt_graph **out) {
    if (!c || !out) return RT_ERR_INVALID;
    *out = nullptr;
    if (!c->has_scene) { c->err = "rt_graph_create before rt_upload_scene"; return RT_ERR_NO_SCENE; }
    if (!d_out_rgb && !d_out_u8) { c->err = "rt_graph_create: no output buffer"; return RT_ERR_INVALID; }
    HIPCHK(c, hipSetDevice(c->device));
    DLights L;
    float *own_offsets = nullptr;
    rt_status s = check_lights(c, lights, &L, &own_offsets);
    if (s != RT_OK) { if (own_offsets) (void)hipFree(own_offsets); return s; }
    DFrame F;
    if ((s = make_frame(c, p, &F)) != RT_OK) { if (own_offsets) (void)hipFree(own_offsets); return s; }
    if (F.npix == 0) { if (own_offsets) (void)hipFree(own_offsets); c->err = "rt_graph_create: empty shard"; return RT_ERR_INVALID; }
    // every allocation happens BEFORE the capture
    const size_t P = (static_cast<size_t>(L.n_samples) + 63) / 64;
    if ((s = ensure_frame(c, F.npix, F.max_depth + 1, P, frame_tiles(F), static_cast<size_t>(L.n_lights))) != RT_OK) { if (own_offsets) (void)hipFree(own_offsets); return s; }
    if (hipStreamSynchronize(c->stream) != hipSuccess) { if (own_offsets) (void)hipFree(own_offsets); c->err = "rt_graph_create: hipStreamSynchronize failed"; return RT_ERR_HIP; }
    rt_graph *g = new rt_graph();
    g->d_offsets = own_offsets;
    g->ctx = c; g->F = F; g->generation = c->frame_generation; g->scene_generation = c->scene_generation;
    if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { if (g->d_offsets) (void)hipFree(g->d_offsets); delete g; c->err = "hipStreamBeginCapture failed"; return RT_ERR_HIP; }
    s = run_frame(c, c->stream, nullptr, L, F, true, false, d_out_rgb, d_out_u8, nullptr, nullptr, 0, 0);
    const hipError_t e = hipStreamEndCapture(c->stream, &g->graph);
    if (s != RT_OK || e != hipSuccess || !g->graph) {
        if (g->graph) (void)hipGraphDestroy(g->graph);
        if (g->d_offsets) (void)hipFree(g->d_offsets);
        delete g;
        if (s == RT_OK) { c->err = std::string("hipStreamEndCapture: ") + hipGetErrorString(e); s = RT_ERR_HIP; }
        return s;
    }
    if (hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGraphDestroy(g->graph);
        if (g->d_offsets) (void)hipFree(g->d_offsets);
        delete g;
        c->err = "hipGraphInstantiate failed";
        return RT_ERR_HIP;
    }
    *out = g;
    return RT_OK;
}

extern "C" rt_status rt_graph_launch(rt_graph *g, const rt_camera *cam, void *stream) {
    if (!g || !cam) return RT_ERR_INVALID;
    rt_ctx *c = g->ctx;
    if (g->generation != c->frame_generation) { c->err = "rt_graph_launch: the frame buffers were reallocated after capture; re-create the graph"; return RT_ERR_INVALID; }
    if (g->scene_generation != c->scene_generation) { c->err = "rt_graph_launch: a scene was uploaded after capture (the graph holds the old scene's device pointers); re-create the graph"; return RT_ERR_INVALID; }
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : c->stream;
    DCam dc;
    make_cam(cam, &dc);
    rt_status cs = upload_camera(c, dc, st);
    if (cs != RT_OK) return cs;
    HIPCHK(c, hipGraphLaunch(g->exec, st));
    g->last_stream = st;
    return RT_OK;
}

extern "C" rt_status rt_graph_stats(rt_graph *g, rt_stats *out) {
    if (!g || !out) return RT_ERR_INVALID;
    rt_ctx *c = g->ctx;
    std::memset(out, 0, sizeof *out);
    if (g->scene_generation != c->scene_generation) { c->err = "rt_graph_stats: a scene was uploaded after capture; re-create the graph"; return RT_ERR_INVALID; }
    HIPCHK(c, hipSetDevice(c->device));
    const int levels_run = c->reflective ? g->F.max_depth + 1 : 1;
    return fill_stats(c, g->last_stream ? g->last_stream : c->stream, g->F, levels_run, false, out, false);
}

extern "C" void rt_graph_destroy(rt_graph *g) {
    if (!g) return;
    (void)hipSetDevice(g->ctx->device);
    if (g->last_stream) (void)hipStreamSynchronize(g->last_stream);
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    if (g->d_offsets) (void)hipFree(g->d_offsets);
    delete g;
}

extern "C" rt_status rt_timing_collect(rt_ctx *c, rt_stats *out) {
    if (!c || !out) return RT_ERR_INVALID;
    std::memset(out, 0, sizeof *out);
    if (c->pending.empty()) return RT_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->pending_stream));
    Control h;
    HIPCHK(c, hipMemcpy(&h, c->d_ctl, sizeof h, hipMemcpyDeviceToHost));
    { const rt_status os_ = check_overflow(c); if (os_ != RT_OK) return os_; }
    fold_stats(h);
    out->rays_primary = h.rays_primary; out->rays_bounce = h.rays_bounce; out->rays_centre = h.rays_centre; out->rays_sample = h.rays_sample; out->rays_sample_walked = h.sample_walked;
    out->pixels = c->pending_frame.npix; out->pixels_culled = h.pixels_culled; out->shaded_hits = h.shaded_hits;
    rt_status s = RT_OK;
    for (const auto &fr : c->pending)
        if ((s = sum_frame_times(c, fr.first, fr.second, out, true)) != RT_OK) break;
    c->pending.clear();
    c->ev_base = 0;
    return s;
}

extern "C" rt_status rt_render(rt_ctx *c, const rt_camera *cam, const rt_lights *lights, const rt_params *p,
                               float *out_rgb, int32_t *out_hit, rt_stats *stats) {
    if (!c) return RT_ERR_INVALID;
    if (!out_rgb || !p) { c->err = "rt_render: null output or params"; return RT_ERR_INVALID; }
    HIPCHK(c, hipSetDevice(c->device));
    const size_t npix = static_cast<size_t>(rt_local_rows(p)) * static_cast<size_t>(p->width > 0 ? p->width : 0);
    if (npix > c->cap_out) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->d_rgb) (void)hipFree(c->d_rgb);
        if (c->d_hit) (void)hipFree(c->d_hit);
        if (c->d_t) (void)hipFree(c->d_t);
        c->d_rgb = nullptr; c->d_hit = nullptr; c->d_t = nullptr; c->cap_out = 0;
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->d_rgb), npix * 3 * sizeof(float)));
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->d_hit), npix * sizeof(int32_t)));
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->d_t), npix * sizeof(float)));
        c->cap_out = npix;
    }
    rt_status s = rt_render_device(c, cam, lights, p, c->d_rgb, nullptr, out_hit ? c->d_hit : nullptr, nullptr, stats);
    if (s != RT_OK) return s;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if ((s = check_overflow(c)) != RT_OK) return s;
    if (npix) {
        HIPCHK(c, hipMemcpy(out_rgb, c->d_rgb, npix * 3 * sizeof(float), hipMemcpyDeviceToHost));
        if (out_hit) HIPCHK(c, hipMemcpy(out_hit, c->d_hit, npix * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    return RT_OK;
}

extern "C" rt_status rt_trace_rays(rt_ctx *c, const rt_lights *lights, int32_t max_depth, int32_t n, const float *origin,
                                   const float *dir, float *out_rgb, int32_t *out_face, float *out_t) {
    if (!c) return RT_ERR_INVALID;
    if (!c->has_scene) { c->err = "rt_trace_rays before rt_upload_scene"; return RT_ERR_NO_SCENE; }
    if (n < 0 || (n && (!origin || !dir || !out_rgb))) { c->err = "rt_trace_rays: bad arguments"; return RT_ERR_INVALID; }
    if (max_depth > RT_MAX_DEPTH) { c->err = "rt_trace_rays: max_depth above RT_MAX_DEPTH"; return RT_ERR_UNSUPPORTED; }
    if (n == 0) return RT_OK;
    HIPCHK(c, hipSetDevice(c->device));
    DLights L;
    rt_status s = check_lights(c, lights, &L);
    if (s != RT_OK) return s;
    DFrame F{};
    F.width = n; F.height = 1; F.local_rows = 1; F.row0 = 0; F.stripe = 1; F.rank = 0; F.nranks = 1;
    F.tiles_x = (n + 7) / 8; F.tiles_y = 1; F.npix = static_cast<uint32_t>(n);
    F.max_depth = max_depth < 0 ? RT_MAX_DEPTH : max_depth;
    F.dyn_trace = c->dyn_trace;
    const size_t P = (static_cast<size_t>(L.n_samples) + 63) / 64;
    if ((s = ensure_frame(c, F.npix, F.max_depth + 1, P, frame_tiles(F), static_cast<size_t>(L.n_lights))) != RT_OK) return s;
    std::vector<RayItem> rays(static_cast<size_t>(n));
    for (int32_t i = 0; i < n; ++i) {
        RayItem &r = rays[static_cast<size_t>(i)];
        r.ox = origin[i * 3]; r.oy = origin[i * 3 + 1]; r.oz = origin[i * 3 + 2];
        r.dx = dir[i * 3]; r.dy = dir[i * 3 + 1]; r.dz = dir[i * 3 + 2];
        r.lx = r.ly = r.lz = 0.f; r.lmode = 0u; r.pix = static_cast<uint32_t>(i); r.pad = 0u;
    }
    const size_t need = static_cast<size_t>(n);
    if (need > c->cap_out) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->d_rgb) (void)hipFree(c->d_rgb);
        if (c->d_hit) (void)hipFree(c->d_hit);
        if (c->d_t) (void)hipFree(c->d_t);
        c->d_rgb = nullptr; c->d_hit = nullptr; c->d_t = nullptr; c->cap_out = 0;
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->d_rgb), need * 3 * sizeof(float)));
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->d_hit), need * sizeof(int32_t)));
        HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->d_t), need * sizeof(float)));
        c->cap_out = need;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_rays[0], rays.data(), need * sizeof(RayItem), hipMemcpyHostToDevice, c->stream));
    if ((s = run_frame(c, c->stream, nullptr, L, F, false, false, c->d_rgb, nullptr, c->d_hit, c->d_t, 0, static_cast<uint32_t>(n))) != RT_OK) return s;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if ((s = check_overflow(c)) != RT_OK) return s;
    HIPCHK(c, hipMemcpy(out_rgb, c->d_rgb, need * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (out_face) HIPCHK(c, hipMemcpy(out_face, c->d_hit, need * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (out_t) HIPCHK(c, hipMemcpy(out_t, c->d_t, need * sizeof(float), hipMemcpyDeviceToHost));
    return RT_OK;
}

extern "C" rt_status rt_light_strikes(rt_ctx *c, int32_t n, const float *hit, const float *light, uint8_t *vis) {
    if (!c) return RT_ERR_INVALID;
    if (!c->has_scene) { c->err = "rt_light_strikes before rt_upload_scene"; return RT_ERR_NO_SCENE; }
    if (n < 0 || (n && (!hit || !light || !vis))) { c->err = "rt_light_strikes: bad arguments"; return RT_ERR_INVALID; }
    if (n == 0) return RT_OK;
    HIPCHK(c, hipSetDevice(c->device));
    float *d_hit = nullptr, *d_light = nullptr;
    uint8_t *d_vis = nullptr;
    const size_t bytes = static_cast<size_t>(n) * 3 * sizeof(float);
    HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&d_hit), bytes));
    HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&d_light), bytes));
    HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&d_vis), static_cast<size_t>(n)));
    rt_status s = RT_OK;
    do {
        if (hipMemcpy(d_hit, hit, bytes, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_light, light, bytes, hipMemcpyHostToDevice) != hipSuccess) { s = RT_ERR_HIP; c->err = "rt_light_strikes: upload failed"; break; }
        const int blocks = (n + 255) / 256;
        launch_segments(blocks < c->cus * 8 ? blocks : c->cus * 8, c->stream, c->S, n, d_hit, d_light, d_vis);
        if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(vis, d_vis, static_cast<size_t>(n), hipMemcpyDeviceToHost) != hipSuccess) { s = RT_ERR_HIP; c->err = "rt_light_strikes: kernel or download failed"; break; }
    } while (0);
    (void)hipFree(d_hit); (void)hipFree(d_light); (void)hipFree(d_vis);
    return s;
}

// ---- debug ray (createDebugRay / recursiveDebugRay without the GL shapes): a host composition of the entry points above ------------
extern "C" rt_status rt_debug_ray(rt_ctx *c, const rt_camera *cam, const rt_lights *lights, float px, float py, int32_t max_levels, rt_debug_hit *out,
                                  int32_t *n_out) {
    if (!c) return RT_ERR_INVALID;
    if (!cam || !lights || !out || !n_out || max_levels < 1) { c->err = "rt_debug_ray: bad arguments"; return RT_ERR_INVALID; }
    if (!c->has_scene) { c->err = "rt_debug_ray before rt_upload_scene"; return RT_ERR_NO_SCENE; }
    *n_out = 0;
    float scr[3];
    screen_to_world(cam, px, py, scr);                                              // flyscene.cpp:439
    V3 pos{scr[0], scr[1], scr[2]};
    V3 dir = unit_fixed(pos - V3{cam->center[0], cam->center[1], cam->center[2]});   // flyscene.cpp:441
    for (int32_t n = 0; n < max_levels; ++n) {
        rt_debug_hit &r = out[n];
        std::memset(&r, 0, sizeof r);
        r.level = n; r.face = -1;
        r.pos[0] = pos.x; r.pos[1] = pos.y; r.pos[2] = pos.z; r.dir[0] = dir.x; r.dir[1] = dir.y; r.dir[2] = dir.z;
        // root box: boxIntersect(origin, origin + direction), as traceRay tests it (flyscene.cpp:655)
        const float dest[3] = {pos.x + dir.x, pos.y + dir.y, pos.z + dir.z};
        DNode root;
        HIPCHK(c, hipMemcpy(&root, c->d_nodes, sizeof root, hipMemcpyDeviceToHost));
        uint8_t in_box = 0;
        const float boxes[6] = {root.bmin[0], root.bmin[1], root.bmin[2], root.bmax[0], root.bmax[1], root.bmax[2]};
        rt_status s = rt_box_intersect(c, 1, boxes, r.pos, dest, &in_box);
        if (s != RT_OK) return s;
        int32_t face = -1; float t = -1.0f;
        s = rt_trace_rays(c, lights, 0, 1, r.pos, r.dir, r.color, &face, &t);
        if (s != RT_OK) return s;
        *n_out = n + 1;
        r.status = !in_box ? 0 : (face < 0 ? 1 : 2);
        if (face < 0) break;
        r.face = face; r.t = t;
        const V3 p0 = pos + V3{t * dir.x, t * dir.y, t * dir.z};                   // flyscene.cpp:270
        r.hit_point[0] = p0.x; r.hit_point[1] = p0.y; r.hit_point[2] = p0.z;
        float nrm[3];
        HIPCHK(c, hipMemcpy(nrm, static_cast<const float *>(c->d_face_normal) + static_cast<size_t>(face) * 3, sizeof nrm, hipMemcpyDeviceToHost));
        std::memcpy(r.normal, nrm, sizeof nrm);
        const V3 nv{nrm[0], nrm[1], nrm[2]};
        float hit[RT_MAX_LIGHTS * 3];
        const int nl = lights->n_lights < 1 ? 0 : (lights->n_lights > RT_MAX_LIGHTS ? RT_MAX_LIGHTS : lights->n_lights);
        for (int i = 0; i < nl; ++i) { hit[i * 3] = p0.x; hit[i * 3 + 1] = p0.y; hit[i * 3 + 2] = p0.z; }
        if (nl && (s = rt_light_strikes(c, nl, hit, &lights->pos[0][0], r.light_visible)) != RT_OK) return s;      // flyscene.cpp:273
        const float two = 2 * dot(dir, nv);
        const V3 refl = dir - V3{two * nv.x, two * nv.y, two * nv.z};              // flyscene.cpp:349
        r.reflected[0] = refl.x; r.reflected[1] = refl.y; r.reflected[2] = refl.z;
        pos = p0; dir = refl;
    }
    return RT_OK;
}

// ---- unit-parity probes ---------------------------------------------------------------------------------------------
namespace {
struct DevBuf {          // scoped device allocation for the probe entry points
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    bool alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1) == hipSuccess; }
};
}  // namespace

extern "C" rt_status rt_box_intersect(rt_ctx *c, int32_t n, const float *boxes, const float *origin, const float *dest, uint8_t *hit) {
    if (!c) return RT_ERR_INVALID;
    if (n < 0 || (n && (!boxes || !origin || !dest || !hit))) { c->err = "rt_box_intersect: bad arguments"; return RT_ERR_INVALID; }
    if (n == 0) return RT_OK;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf b, o, d, h;
    const size_t nn = static_cast<size_t>(n);
    if (!b.alloc(nn * 24) || !o.alloc(nn * 12) || !d.alloc(nn * 12) || !h.alloc(nn)) { c->err = "rt_box_intersect: hipMalloc failed"; return RT_ERR_HIP; }
    HIPCHK(c, hipMemcpy(b.p, boxes, nn * 24, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(o.p, origin, nn * 12, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(d.p, dest, nn * 12, hipMemcpyHostToDevice));
    launch_box_probe(c->stream, n, static_cast<const float *>(b.p), static_cast<const float *>(o.p), static_cast<const float *>(d.p), static_cast<uint8_t *>(h.p));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(hit, h.p, nn, hipMemcpyDeviceToHost));
    return RT_OK;
}

extern "C" rt_status rt_tree_probe(rt_ctx *c, int32_t n, const float *origin, const float *dest, uint32_t *box_tests, uint32_t *leaf_refs, uint32_t *leaf_sig) {
    if (!c) return RT_ERR_INVALID;
    if (!c->has_scene) { c->err = "rt_tree_probe before rt_upload_scene"; return RT_ERR_NO_SCENE; }
    if (n < 0 || (n && (!origin || !dest || !box_tests || !leaf_refs || !leaf_sig))) { c->err = "rt_tree_probe: bad arguments"; return RT_ERR_INVALID; }
    if (n == 0) return RT_OK;
    if (c->flat) { c->err = "rt_tree_probe: the scene is a single leaf (no tree)"; return RT_ERR_UNSUPPORTED; }
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf o, d, ob, orf, os;
    const size_t nn = static_cast<size_t>(n);
    if (!o.alloc(nn * 12) || !d.alloc(nn * 12) || !ob.alloc(nn * 4) || !orf.alloc(nn * 4) || !os.alloc(nn * 4)) { c->err = "rt_tree_probe: hipMalloc failed"; return RT_ERR_HIP; }
    HIPCHK(c, hipMemcpy(o.p, origin, nn * 12, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(d.p, dest, nn * 12, hipMemcpyHostToDevice));
    const int blocks = (n + 255) / 256;
    launch_tree_probe(blocks < c->cus * 4 ? blocks : c->cus * 4, c->stream, c->S, n, static_cast<const float *>(o.p), static_cast<const float *>(d.p),
                      static_cast<uint32_t *>(ob.p), static_cast<uint32_t *>(orf.p), static_cast<uint32_t *>(os.p));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(box_tests, ob.p, nn * 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(leaf_refs, orf.p, nn * 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(leaf_sig, os.p, nn * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

extern "C" rt_status rt_primary_points(rt_ctx *c, const rt_camera *cam, int32_t w, int32_t h, float *out) {
    if (!c) return RT_ERR_INVALID;
    if (!cam || !out || w <= 0 || h <= 0) { c->err = "rt_primary_points: bad arguments"; return RT_ERR_INVALID; }
    HIPCHK(c, hipSetDevice(c->device));
    DCam dc;
    make_cam(cam, &dc);
    DevBuf dcam, dout;
    const size_t nn = static_cast<size_t>(w) * static_cast<size_t>(h) * 3;
    if (!dcam.alloc(sizeof(DCam)) || !dout.alloc(nn * 4)) { c->err = "rt_primary_points: hipMalloc failed"; return RT_ERR_HIP; }
    HIPCHK(c, hipMemcpy(dcam.p, &dc, sizeof dc, hipMemcpyHostToDevice));
    launch_primary_probe(c->cus * 4, c->stream, static_cast<const DCam *>(dcam.p), w, h, static_cast<float *>(dout.p));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, dout.p, nn * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

// executed-work counters of the last frame (diagnostic builds only: -DRT_PROFILE -DRT_PROFILE_STEPS, `make work`)
extern "C" rt_status rt_debug_work_counters(rt_ctx *c, uint64_t *out, int32_t n) {
    if (!c || !out || n < 0) return RT_ERR_INVALID;
#if defined(RT_PROFILE) && defined(RT_PROFILE_STEPS)
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const size_t cnt = static_cast<size_t>(n) < sizeof(Control::prof) / sizeof(unsigned long long) ? static_cast<size_t>(n) : sizeof(Control::prof) / sizeof(unsigned long long);
    HIPCHK(c, hipMemcpy(out, c->d_ctl->prof, cnt * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (size_t i = cnt; i < static_cast<size_t>(n); ++i) out[i] = 0;
    return RT_OK;
#else
    c->err = "rt_debug_work_counters: this build carries no step counters (use librt_mi355x_work.so)";
    return RT_ERR_UNSUPPORTED;
#endif
}

// ---- host scene wrappers ------------------------------------------------------------------------------------
extern "C" rt_status rt_host_scene_load(const char *obj_path, int32_t leaf_capacity, int32_t max_depth, rt_host_scene **out) {
    if (!obj_path || !out || leaf_capacity < 1 || max_depth < 0 || max_depth > 15) return RT_ERR_INVALID;
    *out = nullptr;
    rt_host_scene *h = new rt_host_scene();
    std::string err;
    if (!h->hs.load_obj(obj_path, &err)) {
        std::fprintf(stderr, "rt_mi355x: %s\n", err.c_str());
        delete h;
        return RT_ERR_IO;
    }
    h->hs.build_octree(leaf_capacity, max_depth);
    if (h->hs.overflow) {
        std::fprintf(stderr, "rt_mi355x: the reference's octree construction does not terminate in bounded memory for this mesh at "
                             "capacity %d (more than %zu nodes / %zu face references); choose a larger capacity\n",
                     leaf_capacity, HostScene::kMaxNodes, HostScene::kMaxRefs);
        delete h;
        return RT_ERR_UNSUPPORTED;
    }
    h->hs.flatten();
    *out = h;
    return RT_OK;
}

extern "C" void rt_host_scene_free(rt_host_scene *hs) { delete hs; }

extern "C" rt_status rt_host_scene_view(const rt_host_scene *hs, rt_scene *out) {
    if (!hs || !out) return RT_ERR_INVALID;
    hs->hs.view(out);
    return RT_OK;
}

extern "C" rt_status rt_host_scene_set_model(rt_host_scene *hs, const float model[12], int32_t rebuild_tree) {
    if (!hs || !model) return RT_ERR_INVALID;
    hs->hs.set_model(model, rebuild_tree != 0);
    return hs->hs.overflow ? RT_ERR_UNSUPPORTED : RT_OK;
}

extern "C" rt_status rt_host_scene_build_gpu(rt_host_scene *hs, rt_ctx *c, int32_t leaf_capacity, int32_t max_depth) {
    if (!hs || !c || leaf_capacity < 1 || max_depth < 0 || max_depth > 15) return RT_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    std::string err;
    if (!gpu_build_octree(hs->hs, leaf_capacity, max_depth, c->stream, &err)) { c->err = "rt_host_scene_build_gpu: " + err; return RT_ERR_HIP; }
    if (hs->hs.overflow) return RT_ERR_UNSUPPORTED;
    hs->hs.flatten();
    return RT_OK;
}

extern "C" rt_status rt_host_scene_info(const rt_host_scene *hs, int32_t out[8], float root_box[6]) {
    if (!hs || !out) return RT_ERR_INVALID;
    hs->hs.info(out, root_box);
    return RT_OK;
}

extern "C" void rt_default_camera(rt_camera *cam, int32_t w, int32_t h) { if (cam && w > 0 && h > 0) default_camera(cam, w, h); }
extern "C" void rt_yaw_camera(rt_camera *cam, int32_t w, int32_t h, float yaw) { if (cam && w > 0 && h > 0) yaw_camera(cam, w, h, yaw); }
extern "C" void rt_screen_to_world(const rt_camera *cam, float i, float j, float out[3]) { if (cam && out) screen_to_world(cam, i, j, out); }
extern "C" void rt_default_lights(rt_lights *l, int32_t area) { if (l) default_lights(l, area); }
extern "C" void rt_sphere_offsets(uint32_t seed, float radius, int32_t n, float *out) { if (out && n > 0) sphere_offsets(seed, radius, n, out); }

extern "C" rt_status rt_write_ppm(const char *path, const float *rgb, int32_t w, int32_t h) {
    if (!path || !rgb || w <= 0 || h <= 0) return RT_ERR_INVALID;
    return write_ppm(path, rgb, w, h) ? RT_OK : RT_ERR_IO;
}
extern "C" rt_status rt_write_pfm(const char *path, const float *rgb, int32_t w, int32_t h) {
    if (!path || !rgb || w <= 0 || h <= 0) return RT_ERR_INVALID;
    std::FILE *f = std::fopen(path, "wb");
    if (!f) return RT_ERR_IO;
    bool ok = std::fprintf(f, "PF\n%d %d\n-1.0\n", w, h) > 0;
    for (int32_t y = h - 1; ok && y >= 0; --y)
        ok = std::fwrite(rgb + static_cast<size_t>(y) * w * 3, sizeof(float), static_cast<size_t>(w) * 3, f) == static_cast<size_t>(w) * 3;
    ok = (std::fclose(f) == 0) && ok;
    return ok ? RT_OK : RT_ERR_IO;
}
extern "C" rt_status rt_write_ppm_u8(const char *path, const uint8_t *rgb, int32_t w, int32_t h) {
    if (!path || !rgb || w <= 0 || h <= 0) return RT_ERR_INVALID;
    return write_ppm_u8(path, rgb, w, h) ? RT_OK : RT_ERR_IO;
}
