/*
 * rt_mi355x.h -- C ABI of the MI355X-native primary/shadow ray-trace path.
 *
 * Drop-in boundary for the reference's CPU ThreadPool render (Sh-Anand/Raytracer-in-CPP).  The reference
 * has no FFI layer; the boundary is two C++ members (citations relative to /root/reference):
 *     void            Flyscene::raytraceScene(int width = 0, int height = 0)           src/flyscene.hpp:84
 *                                                                                      src/flyscene.cpp:519-648
 *     Eigen::Vector3f Flyscene::traceRay(Vector3f& origin, Vector3f& direction, int level,
 *                                        vector<Vector3f>& lights, bool countRay)      src/flyscene.hpp:94
 *                                                                                      src/flyscene.cpp:651-771
 * called from src/main.cpp:70 (key 'T') and from the pool lambda src/flyscene.cpp:615-623.
 * Each entry point below names the reference interface it replaces.  Plain pointers and sizes only; no C++,
 * Eigen, Tucano or torch types cross this boundary.  All functions return RT_OK (0) or a negative rt_status;
 * nothing throws and nothing calls exit().  A context belongs to one HIP device; rt_render* is blocking and not
 * re-entrant per context; distinct contexts may be used from distinct threads.
 *
 * The library FAILS LOUDLY (RT_ERR_NO_DEVICE / RT_ERR_HIP) when there is no MI355X-class HIP device: there is
 * no CPU fallback in the product.
 */
#ifndef RT_MI355X_H
#define RT_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int rt_status;
enum {
    RT_OK = 0,
    RT_ERR_INVALID = -1,      /* bad argument (NULL, W/H <= 0, L > 25, N not a square grid, ...) */
    RT_ERR_NO_DEVICE = -2,    /* no HIP device / wrong device index */
    RT_ERR_HIP = -3,          /* a HIP runtime call failed (see rt_last_error) */
    RT_ERR_IO = -4,           /* file could not be opened / written */
    RT_ERR_UNSUPPORTED = -5,  /* scene exceeds a compiled limit (tree depth, samples) */
    RT_ERR_NO_SCENE = -6      /* render before rt_upload_scene */
};

#define RT_MAX_LIGHTS 25      /* the reference's fixed bool visibleLights[25]  (flyscene.cpp:699,835) */
#define RT_MAX_SAMPLES 1024   /* per light; the reference hard-codes 5x5 = 25  (flyscene.cpp:971) */
#define RT_MAX_DEPTH 15       /* recursion levels kept per pixel (the reference is unbounded) */

/* ---- flattened scene: the device-friendly restatement of BoxTree / Tucano::Mesh / Material::Mtl ------------- */
/* replaces: class BoxTree (src/boxTree.hpp:15-62), BoundingBox (src/boundingBox.hpp:21-43),
 *           Tucano::Face (tucano/mesh.hpp:238-247), Material::Mtl (tucano/materials/mtl.hpp:16-116)             */
typedef struct rt_node {
    float    bmin[3], bmax[3];
    uint32_t first;           /* leaf: first index into face_refs; inner: index of first child node            */
    uint32_t count_flags;     /* low 31 bits: #faces (leaf) or #children (inner); bit 31 set = leaf            */
} rt_node;                    /* 32 bytes */
#define RT_NODE_LEAF 0x80000000u

typedef struct rt_material {
    float kd[3], ks[3];
    float shininess, optical_density;
    int32_t illum;
} rt_material;                /* 36 bytes */

typedef struct rt_scene {
    uint32_t n_nodes;      const rt_node  *nodes;        /* node 0 = root; children of a node are contiguous    */
    uint32_t n_face_refs;  const uint32_t *face_refs;    /* leaf face lists, concatenated                       */
    uint32_t n_faces;
    const float    *tri_verts;      /* n_faces*9  world-space A,B,C = ((model*shape)*v).head<3>()               */
    const float    *face_normal;    /* n_faces*3  Face::normal (object space, used untransformed)               */
    const uint32_t *tri_vid;        /* n_faces*3  vertex ids (index vert_normal)                                */
    const int32_t  *mat_id;         /* n_faces                                                                  */
    uint32_t n_vert_normals; const float *vert_normal;   /* *3, mesh.getNormal(vertex_id)                       */
    uint32_t n_materials;    const rt_material *materials;
    float model[12];                /* Mesh::getModelMatrix() 3x4 row-major (identity unless modifyTriangle)    */
} rt_scene;

/* replaces: Tucano::Flycamera state read by raytraceScene (flyscene.cpp:551,575; camera.hpp:115-118,155-173)    */
typedef struct rt_camera {
    float center[3];          /* flycamera.getCenter()                                                          */
    float inv_view[12];       /* getViewMatrix().inverse(), 3x4 row-major                                       */
    float fovy;               /* degrees; 60 in the reference (flyscene.cpp:46)                                 */
    float aspect;             /* width/(float)height                                                            */
    float viewport[4];        /* (0,0,W,H)                                                                      */
} rt_camera;

enum { RT_LIGHT_POINT = 0, RT_LIGHT_AREA = 1, RT_LIGHT_SPHERE = 2 };
/* replaces: Flyscene::lights, lightrep colour, the stdin switches areaLight/pointLight (flyscene.cpp:31-34,68,72)
 * and the literals of createSpherePoint/createAreaLight (flyscene.cpp:956-972, arealight.hpp:15-25)             */
typedef struct rt_lights {
    int32_t n_lights;                       /* 1..RT_MAX_LIGHTS                                                 */
    float   pos[RT_MAX_LIGHTS][3];
    float   color[3];                       /* (1,1,0)                                                          */
    int32_t mode;                           /* RT_LIGHT_POINT (1 sample) | RT_LIGHT_AREA (usteps*vsteps)        */
    int32_t usteps, vsteps;                 /* 5,5 in the reference; 8,8 / 16,16 for the 64 / 256 sample configs */
    float   len_x, len_y;                   /* 0.3, 0.15                                                        */
    /* RT_LIGHT_SPHERE: the third branch of createSpherePoint (flyscene.cpp:974-995, neither stdin switch set): sample s of a light at p
       is offsets[s] + p, offsets[s] = Vector3f(x, y, z) / 5 of that branch.  The reference draws them from an unseeded
       std::random_device for every shaded hit; here the caller fixes them per frame (rt_sphere_offsets gives the seeded restatement).  */
    int32_t      n_offsets;                 /* 1..RT_MAX_SAMPLES (25 in the reference)                          */
    const float *offsets;                   /* host, n_offsets * 3; copied by the call                           */
} rt_lights;

/* replaces: the literals inside traceRay / lightStrikes and raytraceScene's image size + thread partitioning    */
typedef struct rt_params {
    int32_t width, height;    /* full frame                                                                     */
    int32_t max_depth;        /* levels 0..max_depth are traced (a hit at level == max_depth is plain Phong);
                                 <0 = RT_MAX_DEPTH.  The reference never tests `level` (flyscene.cpp:651-771)   */
    /* row shard (multi-GPU): this call renders the rows y in [row0,row1) with ((y-row0)/stripe) % nranks == rank,
       in increasing y.  Single GPU: row0=0,row1=height,stripe=1,rank=0,nranks=1.                               */
    int32_t row0, row1, stripe, rank, nranks;
    int32_t collect_stats;    /* 0: ray counters + per-kernel times of this call (when stats != NULL; syncs)
                                 1: also run the counting (no early-out) traversal variants first and fill the
                                    algorithmic counters; costs a second frame, never set inside a timed region
                                 2: deferred timing -- record per-kernel HIP events on the launch stream, do NOT
                                    synchronise; sums are fetched later with rt_timing_collect (bench loops)   */
} rt_params;

typedef struct rt_stats {
    /* ray = one traversal query (root AABB test + tree walk)                                                    */
    uint64_t rays_primary, rays_bounce, rays_centre, rays_sample;
    uint64_t pixels, pixels_culled;       /* pixels whose primary ray misses the root box (flyscene.cpp:576-581) */
    uint64_t shaded_hits;                 /* phongShade calls                                                    */
    /* algorithmic (reference-semantics) counters, only with collect_stats: boxIntersect calls and leaf face
       references exactly as BoxTree::intersect/traceRay/lightStrikes would perform them (no early-out)         */
    uint64_t box_tests, leaf_tri_refs;                 /* whole frame (closest-hit + centre + sample rays)      */
    uint64_t box_tests_shadow, leaf_tri_refs_shadow;   /* the share of the sample-shadow kernel (k_shadow)      */
    /* per-kernel device time of the last render, milliseconds (HIP events on the render stream)                 */
    float ms_trace, ms_shadow, ms_shade, ms_resolve, ms_total;
    uint32_t launches_trace, launches_shadow, launches_shade;   /* levels that launched the group (historic name)                 */
    uint32_t launches_total;              /* device operations one frame enqueues: kernel launches + the control-block memset */
    uint64_t rays_sample_walked;          /* sample shadow segments actually FORMED and walked: rays_sample minus those whole tiles (k_beam) or whole
                                             (hit, light) units were proven unblocked for before a ray existed (0 in the counting pass)  */
} rt_stats;

typedef struct rt_ctx rt_ctx;

/* ---- context --------------------------------------------------------------------------------------------- */
rt_status   rt_create(rt_ctx **out, int device);                     /* replaces: ThreadPool pool(n) flyscene.cpp:609 */
void        rt_destroy(rt_ctx *ctx);                                 /* replaces: pool.~ThreadPool() flyscene.cpp:634 */
const char *rt_last_error(const rt_ctx *ctx);                        /* NULL-safe; static string when ctx == NULL     */
const char *rt_version(void);
void       *rt_stream(rt_ctx *ctx);                                  /* the context's own hipStream_t (what a NULL `stream` argument means)   */

/* replaces: the scene state traceRay reads through `this` (octree, mesh, materials).  Arrays are copied.          */
rt_status rt_upload_scene(rt_ctx *ctx, const rt_scene *scene);

/* replaces: Flyscene::raytraceScene's pixel loop + pool execution (flyscene.cpp:573-629).
 * out_rgb: host, [n_local_rows * W * 3] float, row-major (y,x); out_hit (optional): level-0 closest face id or -1 */
rt_status rt_render(rt_ctx *ctx, const rt_camera *cam, const rt_lights *lights, const rt_params *p,
                    float *out_rgb, int32_t *out_hit, rt_stats *stats);

/* Same, results stay in device memory (for RCCL gathers / chained launches).  d_out_rgb: device float
 * [n_local_rows*W*3]; d_out_u8 (optional): device uint8 [n_local_rows*W*3] quantised as ppmIO.hpp:145;
 * stream: hipStream_t (NULL = the context's own stream).  Asynchronous when stats == NULL.                        */
rt_status rt_render_device(rt_ctx *ctx, const rt_camera *cam, const rt_lights *lights, const rt_params *p,
                           float *d_out_rgb, uint8_t *d_out_u8, int32_t *d_out_hit, void *stream, rt_stats *stats);
/* waits for the frames rt_render_device has enqueued (the context's stream and the stream of the latest call); reports a work-list overflow */
rt_status rt_synchronize(rt_ctx *ctx);

/* Sums the per-kernel device times (ms_* = SUM over frames, launches_* = total launches) of every frame rendered
 * with collect_stats == 2 since the last call, plus the ray counters of the last frame.  Synchronises the stream.  */
rt_status rt_timing_collect(rt_ctx *ctx, rt_stats *out);

/* ---- captured frames (hipGraph) -------------------------------------------------------------------------------
 * The launch sequence of a frame has no host round trip, so it is captured once into a hipGraph and replayed per frame
 * (animation paths: BASELINE cfg5).  Lights, frame size, shard and output buffers are frozen at capture; the camera is
 * read from device memory and may change on every launch.  No reference counterpart (the reference re-runs
 * raytraceScene per key press, main.cpp:69-70).                                                                    */
typedef struct rt_graph rt_graph;
rt_status rt_graph_create(rt_ctx *ctx, const rt_lights *lights, const rt_params *p, float *d_out_rgb, uint8_t *d_out_u8,
                          rt_graph **out);
/* asynchronous: uploads `cam`, then replays the captured frame on `stream` (NULL = the context's stream)             */
rt_status rt_graph_launch(rt_graph *g, const rt_camera *cam, void *stream);
/* synchronises and returns the ray counters of the last replayed frame                                             */
rt_status rt_graph_stats(rt_graph *g, rt_stats *out);
void      rt_graph_destroy(rt_graph *g);

/* ---- multi-GPU: one process per GPU, row stripes (rt_params.stripe / rank / nranks), ONE RCCL gather over xGMI -----------------------
 * No reference counterpart (single process, std::thread pool: src/flyscene.cpp:558-629).  RCCL is bound at run time (dlopen); a
 * single-GPU user never loads it.  Typical use on every rank:
 *     rank 0: rt_comm_unique_id(id); broadcast id to the other processes out of band (MPI, a file, torch.distributed ...)
 *     rt_comm_create(&comm, device, id, nranks, rank);
 *     per frame: rt_render_gather(ctx, comm, &cam, &lights, &params, d_local_u8, block_bytes, d_gathered_u8 /-root only-/, 0, stream);
 *     root: copy d_gathered_u8 to the host, rt_stitch_rows(...), rt_write_ppm_u8("result.ppm", ...)                                    */
#define RT_COMM_ID_BYTES 128          /* sizeof(ncclUniqueId) */
typedef struct rt_comm rt_comm;
rt_status   rt_comm_unique_id(uint8_t id[RT_COMM_ID_BYTES]);
rt_status   rt_comm_create(rt_comm **out, int device, const uint8_t id[RT_COMM_ID_BYTES], int32_t nranks, int32_t rank);
void        rt_comm_destroy(rt_comm *comm);
const char *rt_comm_last_error(const rt_comm *comm);
/* the single exchange of a frame: `bytes` bytes from every rank to `root` (rank r's block at d_gathered + r * bytes); asynchronous on
 * `stream` (hipStream_t), no host synchronisation, capturable                                                                       */
rt_status   rt_comm_gather_rows(rt_comm *comm, const void *d_local, size_t bytes, void *d_gathered, int32_t root, void *stream);
/* rt_render_device(d_out_u8 = d_local_u8) + rt_comm_gather_rows on the same stream; local_bytes = the common block size
 * (>= rt_local_rows(p) * width * 3 on every rank)                                                                                   */
rt_status   rt_render_gather(rt_ctx *ctx, rt_comm *comm, const rt_camera *cam, const rt_lights *lights, const rt_params *p, uint8_t *d_local_u8,
                             size_t local_bytes, uint8_t *d_gathered_u8, int32_t root, void *stream);
/* root, host side: de-interleaves the gathered blocks (row0 = 0, row1 = height) into frame[height][width][3]                          */
rt_status   rt_stitch_rows(const uint8_t *gathered, size_t block_bytes, int32_t width, int32_t height, int32_t stripe, int32_t nranks, uint8_t *frame);

/* number of rows rt_render produces for p */
int32_t rt_local_rows(const rt_params *p);

/* replaces: Flyscene::traceRay called directly (debug ray, flyscene.cpp:286; unit parity).  n rays, origin/dir
 * [n*3]; every ray sees the scene lights.  out_rgb [n*3]; out_face/out_t optional (level-0 closest hit).          */
rt_status rt_trace_rays(rt_ctx *ctx, const rt_lights *lights, int32_t max_depth, int32_t n,
                        const float *origin, const float *dir, float *out_rgb, int32_t *out_face, float *out_t);

/* replaces: the computational part of Flyscene::createDebugRay / recursiveDebugRay (flyscene.cpp:241-430, 433-470; the cylinders, spheres
 * and console prints are GL / GUI and out of scope).  Level 0 starts at the screen point of the pixel with dir = (screen - centre).normalized();
 * every level records the closest hit of (pos, dir), the hit point p0 = pos + t * dir, the face normal, lightStrikes(p0, lights), the colour
 * traceRay returns for that ray, and continues along reflectedDir = dir - 2 * dir.dot(n) * n from p0 (flyscene.cpp:349), until a miss or
 * max_levels records.  (The reference re-uses the PRIMARY ray's root test and candidate set at every level and keeps negative t,
 * flyscene.cpp:247-259 -- a visualiser quirk that is not reproduced: each level here traces its own ray.)                              */
typedef struct rt_debug_hit {
    int32_t level;
    int32_t status;                 /* 0: the ray misses the root box (the reference draws it red), 1: box but no triangle (blue), 2: hit (green) */
    int32_t face;                   /* closest face id, -1 without a hit                                                          */
    float   t;
    float   pos[3], dir[3];         /* the ray of this level                                                                      */
    float   hit_point[3], normal[3], reflected[3], color[3];
    uint8_t light_visible[RT_MAX_LIGHTS];
    uint8_t pad[3];
} rt_debug_hit;
rt_status rt_debug_ray(rt_ctx *ctx, const rt_camera *cam, const rt_lights *lights, float pixel_x, float pixel_y, int32_t max_levels,
                       rt_debug_hit *out, int32_t *n_out);

/* replaces: Flyscene::lightStrikes (flyscene.cpp:912-954): n segments light[i] -> hit[i]; vis[i] = 1 iff visible   */
rt_status rt_light_strikes(rt_ctx *ctx, int32_t n, const float *hit, const float *light, uint8_t *vis);

/* ---- unit-parity entry points: the device functions of the path on caller-given inputs ------------------------------------------
 * replaces: BoundingBox::boxIntersect (src/boundingBox.cpp:48-83) -- n boxes [n*6: min, max], segments origin/dest [n*3]; hit[i] = the
 *           decision of the kernels' slab test (approximate-then-verify form, bit-identical to the reference by construction)       */
rt_status rt_box_intersect(rt_ctx *ctx, int32_t n, const float *boxes, const float *origin, const float *dest, uint8_t *hit);
/* replaces: BoxTree::intersect (src/boxTree.cpp:150-173) on the uploaded tree, reference semantics (no culling, no early-out): per ray
 *           the number of boxIntersect calls, the sum of faces.size() over the intersected non-empty leaves, and a signature of that
 *           leaf set: sum over its leaves of (index into rt_scene.nodes) * 2654435761 mod 2^32                                    */
rt_status rt_tree_probe(rt_ctx *ctx, int32_t n, const float *origin, const float *dest, uint32_t *box_tests, uint32_t *leaf_refs, uint32_t *leaf_sig);
/* replaces: Camera::screenToWorld (camera.hpp:155-173) for every pixel, evaluated by the device's primary-ray generator:
 *           out[(j*W + i)*3 ..] = screenToWorld(Vector2f(i, j))                                                                   */
rt_status rt_primary_points(rt_ctx *ctx, const rt_camera *cam, int32_t width, int32_t height, float *out);

/* ---- host-side scene preparation (GL-free restatement of the Tucano loader + BoxTree builder) --------------- */
typedef struct rt_host_scene rt_host_scene;
/* replaces: MeshImporter::loadObjFile + mesh.normalizeModelMatrix() + BoxTree(mesh, capacity)
 *           (flyscene.cpp:50-56,86-93; objimporter.hpp:83-284; boxTree.cpp:11-31)                                 */
rt_status rt_host_scene_load(const char *obj_path, int32_t leaf_capacity, int32_t max_depth, rt_host_scene **out);
void      rt_host_scene_free(rt_host_scene *hs);
/* borrowed view of the flattened arrays (valid until rt_host_scene_free / rt_host_scene_set_model)               */
rt_status rt_host_scene_view(const rt_host_scene *hs, rt_scene *out);
/* replaces: Flyscene::modifyTriangle (flyscene.cpp:998-1015): sets the model matrix; rebuild != 0 also rebuilds
 * the octree (the reference leaves it stale)                                                                     */
rt_status rt_host_scene_set_model(rt_host_scene *hs, const float model[12], int32_t rebuild_tree);
/* replaces: BoxTree(mesh, capacity) / split / clasifyFace (src/boxTree.cpp:11-31,88-147,203-336) evaluated ON THE DEVICE of `ctx`
 * (level-synchronous classify + stable compaction, rt_build.hip): rebuilds the octree of the host scene's current world vertices and
 * re-flattens it.  The result equals the host build (rt_host_scene_load / _set_model(rebuild)) array for array.                   */
rt_status rt_host_scene_build_gpu(rt_host_scene *hs, rt_ctx *ctx, int32_t leaf_capacity, int32_t max_depth);
/* tree summary: nodes, non-empty leaves, face refs, largest leaf, depth, "lost" faces                             */
rt_status rt_host_scene_info(const rt_host_scene *hs, int32_t out[8], float root_box[6]);

/* diagnostic, host only: {chunks, cullable chunks, leaves, max chunks per leaf} of the lanes=triangles chunk bounds   */
rt_status rt_debug_chunk_stats(const rt_scene *scene, int32_t out[4]);
/* diagnostic, host only: the bounds themselves -- 16 floats per chunk {lo[3], hi[3], never, infl, sn[3], slo, shi, 0, 0, 0} (box, slab along
 * the chunk's mean normal) -- with the first chunk of every leaf (n_nodes words) and the leaf face references in chunk order (n_face_refs
 * words).  bounds may be NULL to query *n_chunks.                                                                                         */
rt_status rt_debug_chunk_bounds(const rt_scene *scene, float *bounds, int32_t cap_chunks, int32_t *n_chunks, uint32_t *leaf_chunk0, uint32_t *refs);

/* diagnostic: wave-level step counters of the last frame, as executed by the shipped kernels (box-test steps, shaft steps, (ray, chunk)
 * triangle steps ...; layout in DESIGN.md 6).  Only the counting build librt_mi355x_work.so (same sources, -DRT_PROFILE_STEPS) fills them;
 * the product library returns RT_ERR_UNSUPPORTED.  bench.py uses it, outside the timed region, for the executed-work roofline.          */
rt_status rt_debug_work_counters(rt_ctx *ctx, uint64_t *out, int32_t n);

/* replaces: Flycamera defaults + setPerspectiveMatrix/setViewport (flyscene.cpp:46-47, flycamera.hpp:76-86)       */
void rt_default_camera(rt_camera *cam, int32_t width, int32_t height);
/* fly-camera yaw (rotation_Y_axis) for animation paths (flycamera.hpp:166-191)                                    */
void rt_yaw_camera(rt_camera *cam, int32_t width, int32_t height, float yaw);
/* replaces: Camera::screenToWorld (camera.hpp:155-173); host evaluation, used to check the device's              */
void rt_screen_to_world(const rt_camera *cam, float i, float j, float out[3]);
/* replaces: the sphere-point loop of createSpherePoint (flyscene.cpp:976-993) with std::random_device replaced by the seed: point i
 * uses std::mt19937 gen(seed + i); std::uniform_real_distribution<> dis(0, 1) (libstdc++: generate_canonical<double, 53>), then
 * theta = 2.0f * M_PI * r, phi = acos(2.0 * r - 1.0), (x, y, z) = radius * (sin(phi) cos(theta), sin(phi) sin(theta), cos(phi)) in float,
 * out[i] = Vector3f(x, y, z) / 5.  radius = lightrep.getBoundingSphereRadius() (1.0: the unit sphere shape).  Host only.              */
void rt_sphere_offsets(uint32_t seed, float radius, int32_t n, float *out);
/* replaces: lights.push_back((-1,1,1)), lightrep colour, areaLight/pointLight stdin (flyscene.cpp:31-34,68,72)    */
void rt_default_lights(rt_lights *l, int32_t area);

/* replaces: Tucano::ImageImporter::writePPMImage (ppmIO.hpp:130-151): byte-exact ASCII P3                         */
rt_status rt_write_ppm(const char *path, const float *rgb, int32_t width, int32_t height);
rt_status rt_write_ppm_u8(const char *path, const uint8_t *rgb8, int32_t width, int32_t height);
/* binary side channel beside result.ppm (no reference counterpart): PFM "PF\nW H\n-1.0\n" + little-endian
 * float RGB rows, bottom row first as the format demands -- the un-quantised frame, 12 bytes per pixel                     */
rt_status rt_write_pfm(const char *path, const float *rgb, int32_t width, int32_t height);

#ifdef __cplusplus
}
#endif
#endif /* RT_MI355X_H */
