// rt_device.hpp -- device-side data layout shared by the HIP kernels and the C-ABI host code.
//
// HBM layout (all arrays resident for the lifetime of the uploaded scene; 288 GB per MI355X makes the
// per-frame working buffers -- one slot per pixel per recursion level -- a non-issue):
//   nodes      rt_node[n_nodes]        32 B   breadth-first, live children contiguous, read wave-uniformly (s_load)
//   leaf_tris  TriRec[n_face_refs]     80 B   leaf-ordered, per-triangle constants of rayTriangleIntersection
//                                             hoisted (bit-identical float ops evaluated once on the host)
//   tri_verts / face_normal / tri_vid / mat_id / vert_normal / mats   shading inputs, gathered per shaded hit
//   rays[2]    RayItem[npix]           48 B   ping-pong bounce rays (level >= 1), compacted per level
//   items      ShadeItem[npix]         64 B   lit closest hits of the current level (compacted)
//   vis        u64[npix * L * words]          sample-visibility masks written by the shadow kernel
//   rec        float4[(D+1) * npix]           per pixel, per level: Phong RGB + blend kind (folded by resolve)
//   fres       float [(D+1) * npix]           Fresnel factor of illum-5 hits
#pragma once

#include <cstddef>
#include <cstdint>

#include "rt_mi355x.h"

namespace rtamd {

// Per-(leaf, triangle) record, 80 B = s_load_dwordx16 + s_load_dwordx4.  Everything here is a pure function of
// the triangle that Flyscene::rayTriangleIntersection (flyscene.cpp:787-819) recomputes on every call.
struct alignas(16) TriRec {
    float ax, ay, az;        // vertices[0]
    float e0x, e0y, e0z;     // v0 = vertices[2] - vertices[0]
    float e1x, e1y, e1z;     // v1 = vertices[1] - vertices[0]
    float nx, ny, nz;        // triangle.normal
    float nA;                // triangleNormal.dot(vertices[0])
    float d00, d01, d11;     // v0.v0, v0.v1, v1.v1
    float inv_denom;         // 1 / (d00*d11 - d01*d01)
    uint32_t face;           // face id (tie-break: lowest id wins, std::set order + strict '<')
    uint32_t flags;          // bit 0: material illum == 9 (skipped by lightStrikes, flyscene.cpp:934-936)
    uint32_t pad;
};
static_assert(sizeof(TriRec) == 80, "TriRec must be 80 bytes");

// Conservative bound of one 64-triangle chunk of a leaf.  A ray may skip the chunk only when it is provably impossible for ANY
// triangle of the chunk to pass rayTriangleIntersection AS THE REFERENCE COMPUTES IT IN FLOAT with a t the caller still counts:
// the point of an accepted hit lies inside this inflated box (rt_capi.cpp: build_chunk_bounds has the error analysis).
struct alignas(16) ChunkBound {
    float lo[3], hi[3];      // AABB of the chunk's triangles, inflated
    float never;             // 0: the chunk may be culled; 2: never (ill-conditioned / degenerate / non-finite triangle inside)
    float infl;              // the inflation: the point of an accepted hit lies within `infl` (per axis) of the TRIANGLE it was accepted for
    // the same bound along ONE more direction: sn . P lies in [slo, shi] for the point P of every accepted hit (sn: the chunk's mean face
    // normal; the interval is the chunk's extent along it, inflated like the box).  A patch of a smooth surface is a thin plate in a fat
    // axis-aligned box: rays that graze the surface cross the box and miss the plate.  Never-cullable chunks: sn = 0, (-3e38, 3e38).
    float sn[3], slo, shi;
    float pad_[3];
};
static_assert(sizeof(ChunkBound) == 64, "ChunkBound must be 64 bytes");

struct alignas(16) RayItem {     // a bounce ray (level >= 1) or an rt_trace_rays input ray
    float ox, oy, oz, dx;
    float dy, dz, lx, ly;
    float lz;
    uint32_t lmode;              // 0: sees the scene lights; 1: sees the single light (lx,ly,lz) (flyscene.cpp:735-738)
    uint32_t pix;
    uint32_t pad;
};
static_assert(sizeof(RayItem) == 48, "RayItem must be 48 bytes");

struct alignas(16) ShadeItem {   // a lit closest hit waiting for its sample shadow rays and Phong
    float ox, oy, oz, dx;
    float dy, dz, lx, ly;
    float lz;
    uint32_t lmode;
    uint32_t pix;
    int32_t face;
    float t;
    uint32_t pad0, pad1, pad2;
};
static_assert(sizeof(ShadeItem) == 64, "ShadeItem must be 64 bytes");

// A piece of a unit's work handed to other waves: the chunks [c_begin, c_end) (64 triangles each) of leaf `node` for the
// rays in `mask` (lane mask of the unit's wave).  Big leaves are never processed inline by the walking wave: one wave
// grinding through a 979-triangle leaf for 64 rays was the critical path of whole kernels.
struct alignas(16) ContTask {
    uint32_t unit;
    uint32_t node;
    unsigned long long mask;
    uint32_t c_begin, c_end;
    uint32_t pad0, pad1;
};
static_assert(sizeof(ContTask) == 32, "ContTask must be 32 bytes");

// queues of one traversal launch: where its continuation tasks come from / go to (indices into Control::n_tasks[level])
struct TaskQueues {
    const ContTask *tasks_in;     // continuation launches only
    ContTask *tasks_out;          // nullptr: never hand work away
    uint32_t q_in, q_out;
    uint32_t cap, budget;         // queue capacity; leaves whose estimated cost (VALU instructions) exceeds `budget` are split into tasks (0 = off)
    uint32_t target = 0;          // estimated cost of one task piece (0: same as budget)
    uint32_t group_budget = 0;    // cone walk: groups a unit pops before it hands the rest of its stack to the task launch (0 = never)
    const uint8_t *pair_done = nullptr;   // k_shadow_shaft behind k_pair_beam, several lights: byte (item, light) = 1 -- the beam has written that pair's words, skip its units
};

// blend kinds stored in rec[].w (bit pattern of a uint32)
enum : uint32_t {
    KIND_CONST = 0,      // terminal: rgb is the value (BACKGROUND, SHADOW or plain Phong)
    KIND_PASS = 1,       // illum 9:      0.10*phong + 0.90*child   (flyscene.cpp:718)
    KIND_REFRACT = 2,    // illum 6:      0.2*phong  + 0.8*child    (flyscene.cpp:754)
    KIND_MIRROR = 3,     // illum 3,4:    0.15*phong + 0.85*child   (flyscene.cpp:738)
    KIND_FRESNEL = 4     // illum 5:      fresnel * (0.15*phong + 0.85*child)  (flyscene.cpp:739-743)
};

// Device copy of a node: the public rt_node (the reference's box, used for the bit-exact boxIntersect) followed by the node's
// CONTENT box -- the union of the inflated chunk boxes of every leaf below it (rt_capi.cpp, build_chunk_bounds), or
// (-3e38, 3e38) when some chunk below is not cullable.  A ray whose line has no countable point inside the content box
// cannot have an accepted hit anywhere in the subtree, whatever the reference's own box test says.  64 B = one
// s_load_dwordx16 per child.
struct alignas(16) DNode {
    float bmin[3], bmax[3];
    uint32_t first, count_flags;
    float clo[3], chi[3];
    uint32_t pad[2];               // pad[0]: leaves -- index of the leaf's first ChunkBound (chunks[]); pad[1]: unused
};
static_assert(sizeof(DNode) == 64, "DNode must be 64 bytes");

struct DScene {
    const DNode *nodes;
    const TriRec *leaf_tris;
    const ChunkBound *chunks;          // per leaf: ceil(count/64) bounds starting at leaf_chunk0[node]
    const uint32_t *leaf_chunk0;
    float extent;                      // max |coordinate| of the scene (for the per-ray slab padding)
    const float *tri_verts;
    const float *face_normal;
    const uint32_t *tri_vid;
    const int32_t *mat_id;
    const float *vert_normal;
    const rt_material *mats;
    float model[12];
    uint32_t n_nodes, n_faces;
    int32_t queue_local;               // k_shadow queue: -1 auto, 0 strided chunks (balance first), n chunks of n consecutive units
    int32_t plane_cull;                // k_shadow: per-unit plane culling (rt_kernels.hip, SegPacket); RT_NO_PLANE_CULL=1 turns it off
    int32_t queue_div;                 // k_shadow_shaft: units are handed out in chunks of units / (waves x queue_div)
    const float *bad_leaves;           // boxes (min, max) of the leaves that hold a chunk which may never be culled (k_beam tests them per hit)
    uint32_t n_bad_leaves;             // 0xffffffff: too many for the per-hit test -- such a chunk then blocks every beam that meets it
    int32_t beam;                      // k_beam before the shadow kernels: whole tiles of 64 lit hits whose sample rays nothing can block; RT_NO_BEAM=1 turns it off
    int32_t beam_budget;               // k_beam: group steps + chunk batches + chunks a beam may spend before it leaves its hits to the shadow units (RT_BEAM_BUDGET)
    int32_t shaft;                     // k_shadow on tree scenes: shaft-culled group walk (rt_kernels.hip, shaft_walk); RT_NO_SHAFT=1 turns it off
#ifdef RT_UNIT_HIST
    uint32_t *dbg;                     // diagnostic build only: per-unit / per-wave records of the trace stages (rt_capi.cpp: RT_UNIT_DUMP)
#endif
};
#define RT_UNIT_DBG_WORDS (4u * 65536u * 8u + 4u * 16384u * 4u)          // trace-stage unit and wave records ...
#define RT_UNIT_DBG_SHAFT (65536u * 16u)                                   // ... followed by the unit records of the level-0 k_shadow_shaft launch

struct DCam {
    float center[3];
    float inv_view[12];
    float vp[4];
    float k0, k1;            // aspect*scale, scale  (camera.hpp:164-166), evaluated on the host
};

struct DLights {
    float pos[RT_MAX_LIGHTS][3];
    float color[3];
    int32_t n_lights, mode, usteps, vsteps, n_samples;
    float len_x, len_y;
    const float *offsets;    // RT_LIGHT_SPHERE: n_samples * 3 sample offsets (device); sample s of a light at p = offsets[s] + p
    float obox[6];           // ... and their bounding box (min, max): p + obox bounds the samples (float addition is monotone)
};

struct DFrame {              // which pixels this launch covers
    int32_t width, height;   // full frame
    int32_t local_rows;      // rows rendered by this shard
    int32_t row0, stripe, rank, nranks;
    int32_t tiles_x, tiles_y;
    uint32_t npix;           // local_rows * width
    int32_t max_depth;
    int32_t dyn_trace;       // k_trace pulls tiles from the sharded queue instead of static striding
    uint32_t item_cap, ray_cap;   // per-shard capacity of the shade-item and bounce-ray lists
};

#define RT_WORK_SHADOW 640
#define RT_QUEUE_SHARDS 8
#define RT_STAT_SHARDS 64
// Compaction lists (shade items, bounce rays) are split into RT_LIST_SHARDS sub-lists, each with its own counter on its
// own 64-byte line: one returning atomicAdd per 64-pixel tile on a SINGLE counter serialises in the memory-side atomic
// unit at ~88 per us (9,000 tiles of cube.obj at 1080p = 0.1 ms, measured as the floor of k_trace and of k_shade).
// Element i of shard s lives at index s * cap + i; the producing tile/group number picks the shard (tile % RT_LIST_SHARDS),
// so the per-shard capacity is known up front.
#ifndef RT_LIST_SHARDS
#define RT_LIST_SHARDS 16
#endif
enum : int { ST_RAYS_PRIMARY = 0, ST_RAYS_BOUNCE, ST_RAYS_CENTRE, ST_RAYS_SAMPLE, ST_PIXELS_CULLED, ST_SHADED_HITS,
             ST_BOX_TESTS, ST_LEAF_TRI_REFS, ST_BOX_TESTS_SHADOW, ST_LEAF_TRI_REFS_SHADOW, ST_SAMPLE_WALKED };

// control block in device memory (zeroed once per frame by a memset node on the render stream)
struct Control {
    // work-queue heads: one set of RT_QUEUE_SHARDS counters per launch, each counter alone on a 64-byte line
    uint32_t queue[3 * (RT_MAX_DEPTH + 1) + 4][RT_QUEUE_SHARDS * 16];
    uint32_t n_items[RT_MAX_DEPTH + 1][RT_LIST_SHARDS * 16];   // lit hits per level and shard (counter s at [s * 16])
    uint32_t n_rays[RT_MAX_DEPTH + 2][RT_LIST_SHARDS * 16];    // bounce rays per level and shard (n_rays[0][0] = rt_trace_rays input count)
    // leaf tasks of the two traversal stages of the staged trace (closest hit q0 | light centre q1): RT_LIST_SHARDS sub-queues like
    // n_task_sh below, each counter on its own 64-byte line.  (ONE word per stage made the walking launches atomic-bound: dodge at 1080p
    // emits 5,878 + 3,417 tasks, one returning atomicAdd each, and one word serves ~88 of them per microsecond -- 67 and 39 us of the 80
    // and 91 us those launches took, whatever the walk itself cost.)
    uint32_t n_task_tr[RT_MAX_DEPTH + 1][2][RT_LIST_SHARDS * 16];
    // leaf tasks of the shadow kernels: RT_LIST_SHARDS sub-queues (producer block % RT_LIST_SHARDS), each counter on its own line --
    // one returning atomic per emitting leaf visit on a SINGLE word (~60k per dodge launch) ran into the ~88 per us limit
    uint32_t n_task_sh[RT_MAX_DEPTH + 1][RT_LIST_SHARDS * 16];
    uint32_t n_sitems[RT_MAX_DEPTH + 1][RT_LIST_SHARDS * 16];
    uint32_t beam_yield[RT_MAX_DEPTH + 1][RT_LIST_SHARDS * 16]; // k_beam's own brake: shard s (a line of its own) holds beams tested [16 s] / unblocked [16 s + 1] so far  // lit hits per level and shard that still need their sample shadow rays (k_beam's survivors)
    // totals, filled on the HOST by fold_stats() from the sharded counters below
    unsigned long long rays_primary, rays_bounce, rays_centre, rays_sample, pixels_culled, shaded_hits;
    unsigned long long box_tests, leaf_tri_refs;              // k_trace (closest hit + light-centre rays)
    unsigned long long box_tests_shadow, leaf_tri_refs_shadow; // k_shadow (area-light sample rays)
    unsigned long long sample_walked;                          // sample shadow segments that were actually formed (not decided by k_beam / the per-unit culling tests)
    // what the kernels add to: one 128-byte line per shard, shard = blockIdx.x % RT_STAT_SHARDS.  (4096 waves adding
    // to ONE line at kernel end serialise in the memory-side atomic unit: measured 176 us for the 1080p primary k_trace
    // whose arithmetic needs < 20 us.)
    unsigned long long stat[RT_STAT_SHARDS][16];
    // -DRT_PROFILE builds only: executed work (wave steps) and useful lane work per leaf mode / box tests
    unsigned long long prof[768];            // [0, 96): step counters of the trace kernels; [RT_WORK_SHADOW, +96): of the shadow kernels; between: histograms
    // LAST member, NOT covered by the per-frame memset (kFrameClearBytes): set by a kernel whose list reservation did not fit (never
    // expected: the capacities are derived from the tile counts).  Sticky, so that asynchronous frames (rt_render_device without stats,
    // graph replays) cannot lose it; every synchronising entry point turns it into an error and clears it.
    uint32_t overflow;
};
static const size_t kFrameClearBytes = offsetof(Control, overflow);

inline void fold_stats(Control &h) {
    unsigned long long t[16] = {0};
    for (int sh = 0; sh < RT_STAT_SHARDS; ++sh)
        for (int k = 0; k < 16; ++k) t[k] += h.stat[sh][k];
    h.rays_primary = t[ST_RAYS_PRIMARY]; h.rays_bounce = t[ST_RAYS_BOUNCE]; h.rays_centre = t[ST_RAYS_CENTRE]; h.rays_sample = t[ST_RAYS_SAMPLE];
    h.pixels_culled = t[ST_PIXELS_CULLED]; h.shaded_hits = t[ST_SHADED_HITS];
    h.box_tests = t[ST_BOX_TESTS]; h.leaf_tri_refs = t[ST_LEAF_TRI_REFS];
    h.box_tests_shadow = t[ST_BOX_TESTS_SHADOW]; h.leaf_tri_refs_shadow = t[ST_LEAF_TRI_REFS_SHADOW];
    h.sample_walked = t[ST_SAMPLE_WALKED];
}

}  // namespace rtamd
