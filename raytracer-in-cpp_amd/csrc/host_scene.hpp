// host_scene.hpp -- GL-free host-side scene preparation for the MI355X ray-trace path.
//
// Restates, without Tucano/Eigen/OpenGL, the data semantics the reference's tracer depends on
// (citations relative to /root/reference):
//   OBJ/MTL import ........ dependencies/tucano/tucano/utils/objimporter.hpp:83-284, utils/mtlIO.hpp:45-125
//   vertex-normal quirk ... objimporter.hpp:50-74 (normals appended, accumulated by vertex id)
//   face normals .......... dependencies/tucano/tucano/mesh.hpp:441-468
//   normalisation ......... mesh.hpp:578-644, model.hpp:102-105,169-173, src/flyscene.cpp:56
//   octree build .......... src/boxTree.cpp:11-31,88-147,203-456, src/boundingBox.cpp:14-43
// and adds the flattener that turns the pointer-rich BoxTree into the rt_scene arrays of include/rt_mi355x.h.
#pragma once

#include <array>
#include <cstdint>
#include <string>
#include <vector>

#include "rt_mi355x.h"

namespace rtamd {

struct V3 {
    float x = 0.f, y = 0.f, z = 0.f;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float &at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};

// Eigen 3.3.7 evaluation orders for 3-vectors (Eigen/src/Core/Redux.h:91-105): a0*b0 + (a1*b1 + a2*b2)
inline float dot(const V3 &a, const V3 &b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
inline V3 operator-(const V3 &a, const V3 &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator+(const V3 &a, const V3 &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 cross(const V3 &a, const V3 &b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
V3 unit_fixed(const V3 &v);    // Vector3f::normalized(): z = x*x + (y*y + z*z)
V3 unit_dynamic(const V3 &v);  // normalized() of a head(3) block expression: z = (x*x + y*y) + z*z
inline float min_std(float a, float b) { return (b < a) ? b : a; }  // std::min
inline float max_std(float a, float b) { return (a < b) ? b : a; }  // std::max

struct Material {
    V3 ka{0.3f, 0.3f, 0.3f}, kd{0.5f, 0.5f, 0.5f}, ks{1.f, 1.f, 1.f};
    float ns = 10.f, ni = 0.f, d = 1.f;
    int illum = 0;
    std::string name;
};

struct Triangle {
    std::array<uint32_t, 3> vid{};
    int material = -1;
    V3 normal;
};

struct Affine {  // 3x4 row-major
    float m[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
};

struct AABB {
    V3 lo, hi;
    bool hit_by(const V3 &origin, const V3 &dest) const;  // BoundingBox::boxIntersect, boundingBox.cpp:48-83
};

struct OctNode {
    AABB box;
    bool leaf = false, empty = false;
    std::vector<int> kids;   // 0 or 8 pool indices
    std::vector<int> faces;
    int level = 0;
};

class HostScene {
public:
    // import + normalise (flyscene.cpp:50-56)
    struct MeshGroup { std::vector<uint32_t> ids; int mat = -1; };       // one index group (a usemtl block; PLY: the whole file)
    bool load_obj(const std::string &path, std::string *err);            // .obj, or .ply by extension
    bool load_ply(const std::string &path, std::string *err);
    bool finish_mesh(const std::vector<MeshGroup> &groups, const std::string &path, std::string *err);
    // BoxTree(mesh, capacity) with MAX_DEPTH (flyscene.cpp:86-93, boxTree.cpp:3,11-31)
    void build_octree(int capacity, int max_depth);
    // BoxTree -> rt_scene arrays
    void flatten();
    void set_model(const float model[12], bool rebuild);
    void view(rt_scene *out) const;
    void info(int32_t out[8], float root_box[6]) const;

    // mesh
    std::vector<std::array<float, 4>> verts;  // object space
    std::vector<V3> normals;                  // file vn + nverts appended (quirk)
    std::vector<Triangle> tris;
    std::vector<Material> mats;
    V3 centroid;
    float radius = 1.f, norm_scale = 1.f;
    Affine shape, model;
    std::vector<V3> world;                    // world-space vertices
    // octree
    std::vector<OctNode> pool;
    int capacity = 1000, max_depth = 15;
    // flattened
    std::vector<rt_node> f_nodes;
    std::vector<uint32_t> f_refs;
    std::vector<float> f_tri_verts, f_face_normal, f_vert_normal;
    std::vector<uint32_t> f_tri_vid;
    std::vector<int32_t> f_mat_id;
    std::vector<rt_material> f_mats;
    int flat_depth = 0;
    // build guard: the reference's octree can grow exponentially (see subdivide)
    bool overflow = false;
    size_t total_refs = 0;
    static constexpr size_t kMaxNodes = 4u << 20, kMaxRefs = 256u << 20;

private:
    void recompute_world();
    bool face_touches(const AABB &box, int face) const;  // BoxTree::clasifyFace
    void subdivide(int node, int depth);                 // BoxTree::split
    bool load_mtl(const std::string &path);
};

// camera / lights helpers shared by the C ABI and the Flyscene facade
void default_camera(rt_camera *cam, int w, int h);
void yaw_camera(rt_camera *cam, int w, int h, float yaw);
void screen_to_world(const rt_camera *cam, float i, float j, float out[3]);
void default_lights(rt_lights *l, int area);
void sphere_offsets(uint32_t seed, float radius, int n, float *out);
int write_ppm(const char *path, const float *rgb, int w, int h);
int write_ppm_u8(const char *path, const uint8_t *rgb, int w, int h);

}  // namespace rtamd
