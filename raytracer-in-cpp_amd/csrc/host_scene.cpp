// host_scene.cpp -- see host_scene.hpp.  Compiled with -ffp-contract=off on baseline x86-64 so that every
// float operation rounds as in the reference's g++ -O2 build; the values computed here (world vertices,
// face/vertex normals, octree boxes and face lists) are consumed verbatim by the HIP kernels.
#include "host_scene.hpp"

#include <algorithm>
#include <cctype>
#include <cfloat>
#include <cmath>
#include <random>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace rtamd {

V3 unit_fixed(const V3 &v) {
    // Eigen/src/Core/Dot.h:124-134 on a fixed-size Vector3f
    const float z = dot(v, v);
    if (z > 0.f) {
        const float s = std::sqrt(z);
        return {v.x / s, v.y / s, v.z / s};
    }
    return v;
}

V3 unit_dynamic(const V3 &v) {
    // the same normalized(), but on the dynamic-size difference of two .head(3) blocks (mesh.hpp:461-462):
    // Eigen's un-unrolled reduction sums left to right (Eigen/src/Core/Redux.h:200-245)
    const float z = (v.x * v.x + v.y * v.y) + v.z * v.z;
    if (z > 0.f) {
        const float s = std::sqrt(z);
        return {v.x / s, v.y / s, v.z / s};
    }
    return v;
}

bool AABB::hit_by(const V3 &origin, const V3 &dest) const {
    // slab test on the unbounded line origin->dest; division by zero and NaNs resolved only by the
    // std::min/std::max argument order, exactly as boundingBox.cpp:48-83
    const V3 dir = dest - origin;
    const float tx0 = (lo.x - origin.x) / dir.x, tx1 = (hi.x - origin.x) / dir.x;
    const float ty0 = (lo.y - origin.y) / dir.y, ty1 = (hi.y - origin.y) / dir.y;
    const float tz0 = (lo.z - origin.z) / dir.z, tz1 = (hi.z - origin.z) / dir.z;
    const float enter = max_std(max_std(min_std(tx0, tx1), min_std(ty0, ty1)), min_std(tz0, tz1));
    const float leave = min_std(min_std(max_std(tx0, tx1), max_std(ty0, ty1)), max_std(tz0, tz1));
    return !((enter > leave) || (leave < 0));
}

// ---------------------------------------------------------------------------------------------------------
// MTL (mtlIO.hpp:45-125): tokens split on a single ' ', values through atof/atoi, defaults of mtl.hpp:21-39
// ---------------------------------------------------------------------------------------------------------
bool HostScene::load_mtl(const std::string &path) {
    std::ifstream in(path.c_str(), std::ios::in);
    if (!in) {
        std::fprintf(stderr, "rt_mi355x: cannot open %s\n", path.c_str());
        return false;
    }
    for (std::string line; std::getline(in, line);) {
        if (line.empty()) continue;
        std::stringstream ss(line);
        std::vector<std::string> tok;
        for (std::string piece; std::getline(ss, piece, ' ');) tok.push_back(piece);
        if (tok.empty() || tok[0] == "#") continue;
        auto num = [&](size_t i) { return i < tok.size() ? std::atof(tok[i].c_str()) : 0.0; };
        if (tok[0] == "newmtl") {
            mats.emplace_back();
            mats.back().name = tok.size() > 1 ? tok[1] : std::string();
            continue;
        }
        if (mats.empty()) continue;
        Material &m = mats.back();
        if (tok[0] == "Ns") m.ns = static_cast<float>(num(1));
        else if (tok[0] == "Ka") m.ka = {static_cast<float>(num(1)), static_cast<float>(num(2)), static_cast<float>(num(3))};
        else if (tok[0] == "Kd") m.kd = {static_cast<float>(num(1)), static_cast<float>(num(2)), static_cast<float>(num(3))};
        else if (tok[0] == "Ks") m.ks = {static_cast<float>(num(1)), static_cast<float>(num(2)), static_cast<float>(num(3))};
        else if (tok[0] == "Ni") m.ni = static_cast<float>(num(1));
        else if (tok[0] == "d") m.d = static_cast<float>(num(1));
        else if (tok[0] == "illum") m.illum = tok.size() > 1 ? std::atoi(tok[1].c_str()) : 0;
    }
    if (mats.empty()) mats.emplace_back();  // mtlIO.hpp:112-116
    return true;
}

// ---------------------------------------------------------------------------------------------------------
// OBJ (objimporter.hpp:83-284)
// ---------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------
// PLY (plyimporter.hpp:186-262 through rply).  Tucano's PLY path only fills GL buffers -- it never calls storeVertexData / createFaces,
// so the reference cannot ray-trace a PLY mesh at all (mesh.getNumberOfFaces() == 0).  Defined here as: the SAME mesh state
// loadObjFile would build from the same data -- positions (x, y, z, 1), the file's per-vertex normals as the `vn` list (then the
// importer's normal-accumulation quirk), the first three indices of every face list (plyimporter.hpp:104-118), no materials
// (-> the default Mtl).  ascii and binary_little_endian; parity unpinned (no reference render can exist).
// ---------------------------------------------------------------------------------------------------------
namespace {
struct PlyProp { std::string name; int size = 0; char kind = 'f'; bool list = false; int count_size = 0; char count_kind = 'u'; };
struct PlyElem { std::string name; size_t count = 0; std::vector<PlyProp> props; };

bool ply_type(const std::string &t, int *size, char *kind) {
    static const struct { const char *n; int s; char k; } T[] = {
        {"char", 1, 'i'}, {"int8", 1, 'i'}, {"uchar", 1, 'u'}, {"uint8", 1, 'u'}, {"short", 2, 'i'}, {"int16", 2, 'i'}, {"ushort", 2, 'u'}, {"uint16", 2, 'u'},
        {"int", 4, 'i'}, {"int32", 4, 'i'}, {"uint", 4, 'u'}, {"uint32", 4, 'u'}, {"float", 4, 'f'}, {"float32", 4, 'f'}, {"double", 8, 'f'}, {"float64", 8, 'f'}};
    for (const auto &e : T) if (t == e.n) { *size = e.s; *kind = e.k; return true; }
    return false;
}
// one scalar as the double rply's ply_get_argument_value returns
bool ply_scalar(std::istream &in, bool ascii, int size, char kind, double *out) {
    if (ascii) { return static_cast<bool>(in >> *out); }
    unsigned char b[8];
    if (!in.read(reinterpret_cast<char *>(b), size)) return false;
    if (kind == 'f') {
        if (size == 4) { float f; std::memcpy(&f, b, 4); *out = f; } else { double d; std::memcpy(&d, b, 8); *out = d; }
    } else {
        unsigned long long u = 0;
        for (int i = size - 1; i >= 0; --i) u = (u << 8) | b[i];
        if (kind == 'i') {
            const unsigned long long sign = 1ull << (size * 8 - 1);
            *out = (u & sign) ? -static_cast<double>((~u + 1ull) & ((sign << 1) - 1ull)) : static_cast<double>(u);
        } else *out = static_cast<double>(u);
    }
    return true;
}
}  // namespace

bool HostScene::load_ply(const std::string &path, std::string *err) {
    std::ifstream in(path.c_str(), std::ios::in | std::ios::binary);
    if (!in) { if (err) *err = "cannot open " + path; return false; }
    std::string line;
    if (!std::getline(in, line) || line.substr(0, 3) != "ply") { if (err) *err = "not a PLY file: " + path; return false; }
    bool ascii = false, have_format = false;
    std::vector<PlyElem> elems;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ls(line);
        std::string key;
        ls >> key;
        if (key == "end_header") break;
        if (key == "format") {
            std::string f; ls >> f;
            if (f == "ascii") ascii = true;
            else if (f != "binary_little_endian") { if (err) *err = "unsupported PLY format " + f; return false; }
            have_format = true;
        } else if (key == "element") {
            PlyElem e;
            long long cnt = -1;
            if (!(ls >> e.name >> cnt) || cnt < 0) { if (err) *err = "bad PLY element line: " + line; return false; }
            e.count = static_cast<size_t>(cnt);
            elems.push_back(e);
        } else if (key == "property" && !elems.empty()) {
            PlyProp p; std::string t; ls >> t;
            if (t == "list") {
                std::string ct, it;
                if (!(ls >> ct >> it >> p.name)) { if (err) *err = "bad PLY property line: " + line; return false; }
                p.list = true;
                if (!ply_type(ct, &p.count_size, &p.count_kind) || !ply_type(it, &p.size, &p.kind)) { if (err) *err = "bad PLY list type"; return false; }
            } else {
                if (!(ls >> p.name)) { if (err) *err = "bad PLY property line: " + line; return false; }
                if (!ply_type(t, &p.size, &p.kind)) { if (err) *err = "bad PLY type " + t; return false; }
            }
            elems.back().props.push_back(p);
        }
    }
    if (!have_format) { if (err) *err = "PLY header without a format line"; return false; }
    // The file is not trusted: every element count is bounded by what the rest of the file can hold (a record takes at least one byte per
    // property -- binary: the scalar sizes, a list its count field; ascii: at least one digit -- and an element without properties
    // takes none, so it may not claim records at all).
    const std::streamoff body0 = in.tellg();
    in.seekg(0, std::ios::end);
    const std::streamoff file_end = in.tellg();
    in.seekg(body0, std::ios::beg);
    if (body0 < 0 || file_end < body0) { if (err) *err = "cannot size " + path; return false; }
    unsigned long long budget = static_cast<unsigned long long>(file_end - body0);
    for (const PlyElem &e : elems) {
        unsigned long long rec = 0;
        for (const PlyProp &p : e.props) rec += ascii ? 1ull : static_cast<unsigned long long>(p.list ? p.count_size : p.size);
        if (e.count != 0 && (rec == 0 || e.count > budget / rec)) { if (err) *err = "PLY element '" + e.name + "' claims more records than the file holds"; return false; }
        budget -= rec * e.count;
    }
    verts.clear(); normals.clear(); tris.clear(); mats.clear();
    std::vector<MeshGroup> groups(1);
    for (const PlyElem &e : elems) {
        for (size_t i = 0; i < e.count; ++i) {
            std::array<float, 4> v{0.f, 0.f, 0.f, 1.f};
            V3 n; bool have_nz = false;
            for (const PlyProp &p : e.props) {
                if (p.list) {
                    double cnt = 0;
                    if (!ply_scalar(in, ascii, p.count_size, p.count_kind, &cnt)) { if (err) *err = "truncated PLY"; return false; }
                    // (a list longer than the rest of the file is refused before the loop runs; NaN compares false)
                    if (!(cnt >= 0.0 && cnt <= static_cast<double>(file_end - body0))) { if (err) *err = "bad PLY list length"; return false; }
                    for (long k = 0; k < static_cast<long>(cnt); ++k) {
                        double val = 0;
                        if (!ply_scalar(in, ascii, p.size, p.kind, &val)) { if (err) *err = "truncated PLY"; return false; }
                        if (e.name == "face" && p.name == "vertex_indices" && k < 3) {     // face_cb: value_index 0..2
                            // the cast of a negative, non-finite or >= 2^32 double to uint32_t is undefined: refuse it here
                            if (!(val >= 0.0 && val < 4294967296.0)) { if (err) *err = "PLY face index out of range"; return false; }
                            groups[0].ids.push_back(static_cast<uint32_t>(val));
                        }
                    }
                } else {
                    double val = 0;
                    if (!ply_scalar(in, ascii, p.size, p.kind, &val)) { if (err) *err = "truncated PLY"; return false; }
                    if (e.name == "vertex") {
                        if (p.name == "x") v[0] = static_cast<float>(val);
                        else if (p.name == "y") v[1] = static_cast<float>(val);
                        else if (p.name == "z") v[2] = static_cast<float>(val);
                        else if (p.name == "nx") n.x = static_cast<float>(val);
                        else if (p.name == "ny") n.y = static_cast<float>(val);
                        else if (p.name == "nz") { n.z = static_cast<float>(val); have_nz = true; }
                    }
                }
            }
            if (e.name == "vertex") {
                verts.push_back(v);
                if (have_nz) normals.push_back(n);            // normal_cb pushes when nz arrives
            }
        }
    }
    return finish_mesh(groups, path, err);
}

bool HostScene::load_obj(const std::string &path, std::string *err) {
    const size_t dot = path.find_last_of('.');
    if (dot != std::string::npos) {
        std::string ext = path.substr(dot);
        for (char &ch : ext) ch = static_cast<char>(std::tolower(static_cast<unsigned char>(ch)));
        if (ext == ".ply") return load_ply(path, err);
    }
    std::ifstream in(path.c_str(), std::ios::in);
    if (!in) {
        if (err) *err = "cannot open " + path;
        return false;
    }
    const size_t cut = path.find_last_of("/\\");
    const std::string dir = cut == std::string::npos ? std::string() : path.substr(0, cut + 1);

    verts.clear(); normals.clear(); tris.clear(); mats.clear();
    typedef MeshGroup Group;
    std::vector<Group> groups(1);
    int current = -1;

    for (std::string line; std::getline(in, line);) {
        const std::string key6 = line.substr(0, 6), key2 = line.substr(0, 2);
        if (key6 == "mtllib") {
            if (line.size() < 7) continue;
            std::string fn = dir + line.substr(7);
            fn.erase(std::remove(fn.begin(), fn.end(), '\n'), fn.end());
            fn.erase(std::remove(fn.begin(), fn.end(), '\r'), fn.end());
            load_mtl(fn);
        } else if (key6 == "usemtl") {
            if (!groups.back().ids.empty()) groups.emplace_back();
            const std::string want = line.size() >= 7 ? line.substr(7) : std::string();
            for (size_t i = 0; i < mats.size(); ++i)
                if (mats[i].name.compare(want) == 0) current = static_cast<int>(i);  // exact match, '\r' included
            groups.back().mat = current;
        } else if (key2 == "v ") {
            std::istringstream s(line.substr(2));
            std::array<float, 4> v{0.f, 0.f, 0.f, 1.f};
            s >> v[0]; s >> v[1]; s >> v[2];
            verts.push_back(v);
        } else if (key2 == "vn") {
            std::istringstream s(line.size() >= 3 ? line.substr(3) : std::string());
            V3 n;
            s >> n.x; s >> n.y; s >> n.z;
            normals.push_back(n);
        } else if (key2 == "f ") {
            // every whitespace-separated element contributes ONE vertex id (the text before the first '/');
            // faces are the consecutive triples of the group's id stream -- no polygon triangulation
            std::stringstream ls(line.substr(2));
            for (std::string elem; ls >> elem;) {
                const std::string head = elem.substr(0, elem.find('/'));
                groups.back().ids.push_back(static_cast<uint32_t>(std::atoi(head.c_str()) - 1));
            }
        }
    }
    return finish_mesh(groups, path, err);
}

// everything after the file has been parsed: the same for OBJ and PLY
bool HostScene::finish_mesh(const std::vector<MeshGroup> &groups, const std::string &path, std::string *err) {
    typedef MeshGroup Group;
    if (verts.empty()) {
        if (err) *err = "no vertices in " + path;
        return false;
    }
    for (const Group &g : groups) {
        if (g.ids.size() % 3 != 0) {
            if (err) *err = "face index stream is not a multiple of 3 (non-triangle faces are undefined behaviour in the reference)";
            return false;
        }
        for (uint32_t id : g.ids)
            if (id >= verts.size()) {
                if (err) *err = "face references a missing vertex";
                return false;
            }
    }

    // vertex normals with the importer's quirk (objimporter.hpp:50-74): nverts zero normals are APPENDED to the
    // file's vn list and unit face normals are accumulated at normals[vertex_id]
    const size_t n_file = normals.size();
    normals.resize(n_file + verts.size());
    auto P = [&](uint32_t i) { return V3{verts[i][0], verts[i][1], verts[i][2]}; };
    for (const Group &g : groups)
        for (size_t i = 0; i + 2 < g.ids.size(); i += 3) {
            const V3 a = P(g.ids[i]);
            const V3 e1 = unit_fixed(P(g.ids[i + 2]) - a);
            const V3 e0 = unit_fixed(P(g.ids[i + 1]) - a);
            const V3 n = unit_fixed(cross(e0, e1));
            for (int k = 0; k < 3; ++k) {
                V3 &acc = normals[g.ids[i + k]];
                acc = acc + n;
            }
        }
    for (V3 &n : normals) n = unit_fixed(n);

    // centroid, bounding-sphere radius, normalisation scale (mesh.hpp:627-642)
    V3 c;
    for (const auto &v : verts) c = c + V3{v[0], v[1], v[2]};
    const float count = static_cast<float>(static_cast<unsigned int>(verts.size()));
    c = {c.x / count, c.y / count, c.z / count};
    float r = 0.f;
    for (const auto &v : verts) {
        const V3 d = V3{v[0], v[1], v[2]} - c;
        // (vert[i].head(3) - centroid).norm() is a dynamic-size block expression in Eigen: left-to-right sum
        r = max_std(r, std::sqrt((d.x * d.x + d.y * d.y) + d.z * d.z));
    }
    centroid = c;
    radius = r;
    norm_scale = static_cast<float>(1.0 / static_cast<double>(r));

    // faces with their object-space normals (mesh.hpp:441-468); empty index groups are skipped (objimporter.hpp:262-269)
    for (const Group &g : groups)
        for (size_t i = 0; i + 2 < g.ids.size(); i += 3) {
            Triangle t;
            t.vid = {g.ids[i], g.ids[i + 1], g.ids[i + 2]};
            t.material = g.mat;
            const V3 a = P(t.vid[0]);
            const V3 e1 = unit_dynamic(P(t.vid[2]) - a);
            const V3 e0 = unit_dynamic(P(t.vid[1]) - a);
            t.normal = unit_fixed(cross(e0, e1));
            tris.push_back(t);
        }

    // EXTENSION: the reference indexes materials[-1] (undefined behaviour, flyscene.cpp:712) when an OBJ has no
    // usable material; we fall back to Tucano's default Mtl instead.
    if (mats.empty()) mats.emplace_back();
    for (Triangle &t : tris)
        if (t.material < 0) t.material = 0;

    // shape = Identity.scale(s).translate(-centroid) (model.hpp:169-173); model = identity
    shape = Affine();
    model = Affine();
    shape.m[0] = 1.f * norm_scale; shape.m[5] = 1.f * norm_scale; shape.m[10] = 1.f * norm_scale;
    shape.m[3] = 0.f + norm_scale * (-centroid.x);
    shape.m[7] = 0.f + norm_scale * (-centroid.y);
    shape.m[11] = 0.f + norm_scale * (-centroid.z);
    recompute_world();
    return true;
}

void HostScene::recompute_world() {
    // ((model * shape) * v4).head<3>()  (model.hpp:102-105, flyscene.cpp:788-790)
    float ms[12];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c)
            ms[r * 4 + c] = (model.m[r * 4] * shape.m[c] + model.m[r * 4 + 1] * shape.m[4 + c]) + model.m[r * 4 + 2] * shape.m[8 + c];
        ms[r * 4 + 3] = ((model.m[r * 4] * shape.m[3] + model.m[r * 4 + 1] * shape.m[7]) + model.m[r * 4 + 2] * shape.m[11]) + model.m[r * 4 + 3];
    }
    world.resize(verts.size());
    for (size_t i = 0; i < verts.size(); ++i) {
        const auto &v = verts[i];
        float o[3];
        for (int r = 0; r < 3; ++r)
            o[r] = ((ms[r * 4] * v[0] + ms[r * 4 + 1] * v[1]) + ms[r * 4 + 2] * v[2]) + ms[r * 4 + 3] * v[3];
        world[i] = {o[0], o[1], o[2]};
    }
}

void HostScene::set_model(const float m[12], bool rebuild) {
    std::memcpy(model.m, m, sizeof(float) * 12);
    recompute_world();
    if (rebuild) build_octree(capacity, max_depth);
    flatten();
}

// ---------------------------------------------------------------------------------------------------------
// Octree.  Bug-compatible with BoxTree: the SAT test runs on NORMALISED centre-relative vectors
// (boxTree.cpp:236-240) and therefore drops triangles; faces are lost when a child holds exactly `capacity`
// faces (boxTree.cpp:140-145).  A "correct" tree would fail parity with the reference.
// ---------------------------------------------------------------------------------------------------------
namespace {

inline bool separated(float p_first, float p_second, float rad) {
    // axisTestX01/Y02/Z0/X02/Y1: std::max(p_second, p_first), std::min(p_second, p_first)
    const float hi = max_std(p_second, p_first), lo = min_std(p_second, p_first);
    return lo > rad || hi < -rad;
}
inline bool separated_z12(float p1, float p2, float rad) {
    // axisTestZ12 passes (p1, p2) in the other order (boxTree.cpp:403-404)
    const float hi = max_std(p1, p2), lo = min_std(p1, p2);
    return lo > rad || hi < -rad;
}

bool plane_overlaps(const V3 &n, const V3 &vert, const V3 &half) {
    // BoxTree::planeBoxOverlap, boxTree.cpp:345-366
    V3 lo, hi;
    for (int i = 0; i < 3; ++i) {
        const float v = vert[i];
        if (n[i] > 0.0f) { lo.at(i) = -half[i] - v; hi.at(i) = half[i] - v; }
        else { lo.at(i) = half[i] - v; hi.at(i) = -half[i] - v; }
    }
    if (dot(n, lo) > 0.0f) return false;
    return dot(n, hi) >= 0.0f;
}

}  // namespace

bool HostScene::face_touches(const AABB &b, int face) const {
    // BoxTree::clasifyFace, boxTree.cpp:203-336
    const Triangle &t = tris[face];
    const V3 p[3] = {world[t.vid[0]], world[t.vid[1]], world[t.vid[2]]};
    for (const V3 &v : p)
        if (b.lo.x <= v.x && b.hi.x >= v.x && b.lo.y <= v.y && b.hi.y >= v.y && b.lo.z <= v.z && b.hi.z >= v.z) return true;

    const V3 mid{b.lo.x + (b.hi.x - b.lo.x) / 2.f, b.lo.y + (b.hi.y - b.lo.y) / 2.f, b.lo.z + (b.hi.z - b.lo.z) / 2.f};
    const V3 h = unit_fixed(b.hi - mid);
    const V3 A = unit_fixed(p[0] - mid), B = unit_fixed(p[1] - mid), C = unit_fixed(p[2] - mid);
    const V3 e[3] = {B - A, C - B, A - C};

    {   // edge 0: X01(A,C) Y02(A,C) Z12(B,C)
        const float fx = std::fabs(e[0].x), fy = std::fabs(e[0].y), fz = std::fabs(e[0].z);
        if (separated(e[0].z * A.y - e[0].y * A.z, e[0].z * C.y - e[0].y * C.z, fz * h.y + fy * h.z)) return false;
        if (separated(-e[0].z * A.x + e[0].x * A.z, -e[0].z * C.x + e[0].x * C.z, fz * h.x + fx * h.z)) return false;
        if (separated_z12(e[0].y * B.x - e[0].x * B.y, e[0].y * C.x - e[0].x * C.y, fy * h.x + fx * h.y)) return false;
    }
    {   // edge 1: X01(A,C) Y02(A,C) Z0(A,B)
        const float fx = std::fabs(e[1].x), fy = std::fabs(e[1].y), fz = std::fabs(e[1].z);
        if (separated(e[1].z * A.y - e[1].y * A.z, e[1].z * C.y - e[1].y * C.z, fz * h.y + fy * h.z)) return false;
        if (separated(-e[1].z * A.x + e[1].x * A.z, -e[1].z * C.x + e[1].x * C.z, fz * h.x + fx * h.z)) return false;
        if (separated(e[1].y * A.x - e[1].x * A.y, e[1].y * B.x - e[1].x * B.y, fy * h.x + fx * h.y)) return false;
    }
    {   // edge 2: X02(A,B) Y1(A,B) Z12(B,C)
        const float fx = std::fabs(e[2].x), fy = std::fabs(e[2].y), fz = std::fabs(e[2].z);
        if (separated(e[2].z * A.y - e[2].y * A.z, e[2].z * B.y - e[2].y * B.z, fz * h.y + fy * h.z)) return false;
        if (separated(-e[2].z * A.x + e[2].x * A.z, -e[2].z * B.x + e[2].x * B.z, fz * h.x + fx * h.z)) return false;
        if (separated_z12(e[2].y * B.x - e[2].x * B.y, e[2].y * C.x - e[2].x * C.y, fy * h.x + fx * h.y)) return false;
    }
    for (int k = 0; k < 3; ++k) {  // findMinMax + slab per axis (boxTree.cpp:302-322)
        const float lo = min_std(min_std(A[k], B[k]), C[k]);
        const float hi = max_std(max_std(A[k], B[k]), C[k]);
        if (lo > h[k] || hi < -h[k]) return false;
    }
    const V3 n = unit_fixed(cross(A - B, A - C));
    return plane_overlaps(n, A, h);
}

void HostScene::subdivide(int node, int depth) {
    // BoxTree::split, boxTree.cpp:88-147
    // Guard (not in the reference, which simply runs out of memory): because clasifyFace tests NORMALISED vectors it can
    // accept a triangle in all 8 octants at every level, so small capacities make the tree grow like 8^15 on some
    // meshes (dodgeColorTest.obj at capacity 64).  We stop and report instead of exhausting host memory.
    if (overflow || pool.size() > kMaxNodes || total_refs > kMaxRefs) { overflow = true; return; }
    pool[node].leaf = false;
    const V3 lo = pool[node].box.lo, hi = pool[node].box.hi;
    const float dx = (hi.x - lo.x) / 2, dy = (hi.y - lo.y) / 2, dz = (hi.z - lo.z) / 2;
    const V3 ex{dx, 0, 0}, ey{0, dy, 0}, ez{0, 0, dz};
    auto twice = [](const V3 &v) { return V3{2.f * v.x, 2.f * v.y, 2.f * v.z}; };
    const AABB oct[8] = {
        {lo, ((lo + ex) + ey) + ez},
        {lo + ez, ((lo + ex) + ey) + twice(ez)},
        {lo + ey, ((lo + ex) + twice(ey)) + ez},
        {(lo + ey) + ez, ((lo + ex) + twice(ey)) + twice(ez)},
        {lo + ex, ((lo + twice(ex)) + ey) + ez},
        {(lo + ex) + ez, hi - ey},
        {(lo + ex) + ey, hi - ez},
        {((lo + ex) + ey) + ez, hi},
    };
    const int level = pool[node].level + 1;
    std::vector<int> kids(8);
    for (int k = 0; k < 8; ++k) {
        OctNode c;
        c.box = oct[k];
        c.level = level;
        kids[k] = static_cast<int>(pool.size());
        pool.push_back(std::move(c));
    }
    const std::vector<int> parent_faces = std::move(pool[node].faces);
    pool[node].faces.clear();
    pool[node].kids = kids;
    for (int k = 0; k < 8; ++k) {
        OctNode &c = pool[kids[k]];
        for (int f : parent_faces)
            if (face_touches(c.box, f)) c.faces.push_back(f);
        total_refs += c.faces.size();
    }
    for (int k = 0; k < 8; ++k) {
        const int ci = kids[k];
        const size_t nf = pool[ci].faces.size();
        if (nf == 0 && pool[ci].kids.empty()) pool[ci].empty = true;
        if (nf < static_cast<size_t>(capacity) || depth <= 0) pool[ci].leaf = true;
        if (nf > static_cast<size_t>(capacity) && depth > 0) subdivide(ci, depth - 1);
    }
}

void HostScene::build_octree(int cap, int depth) {
    capacity = cap;
    max_depth = depth;
    pool.clear();
    overflow = false;
    total_refs = 0;
    // BoundingBox(Mesh&): the running maximum starts at FLT_MIN, the smallest POSITIVE float (boundingBox.cpp:20-22)
    V3 lo{FLT_MAX, FLT_MAX, FLT_MAX}, hi{FLT_MIN, FLT_MIN, FLT_MIN};
    for (const Triangle &t : tris)
        for (uint32_t id : t.vid) {
            const V3 &v = world[id];
            lo = {min_std(lo.x, v.x), min_std(lo.y, v.y), min_std(lo.z, v.z)};
            hi = {max_std(hi.x, v.x), max_std(hi.y, v.y), max_std(hi.z, v.z)};
        }
    OctNode root;
    root.box = {lo, hi};
    root.faces.resize(tris.size());
    for (size_t i = 0; i < tris.size(); ++i) root.faces[i] = static_cast<int>(i);
    pool.push_back(std::move(root));
    if (pool[0].faces.size() > static_cast<size_t>(capacity)) subdivide(0, max_depth);
    else if (pool[0].faces.empty()) pool[0].empty = true;
    else pool[0].leaf = true;
}

// ---------------------------------------------------------------------------------------------------------
// Flattener (new): breadth-first node array, the live children of a node contiguous.  Children flagged isEmpty
// are dropped -- BoxTree::intersect skips them before any box test (boxTree.cpp:164).  A node that is neither
// leaf nor empty but has no children (the "exactly capacity faces" case) is kept as an inner node with zero
// children: the reference tests its box and then finds nothing, and so do we.
// ---------------------------------------------------------------------------------------------------------
void HostScene::flatten() {
    f_nodes.clear(); f_refs.clear();
    flat_depth = 0;
    if (!pool.empty()) {
        std::vector<int> order{0};
        f_nodes.push_back(rt_node{});
        for (size_t head = 0; head < order.size(); ++head) {
            const OctNode &n = pool[order[head]];
            rt_node out{};
            out.bmin[0] = n.box.lo.x; out.bmin[1] = n.box.lo.y; out.bmin[2] = n.box.lo.z;
            out.bmax[0] = n.box.hi.x; out.bmax[1] = n.box.hi.y; out.bmax[2] = n.box.hi.z;
            flat_depth = std::max(flat_depth, n.level);
            if (n.leaf && !n.empty) {
                out.first = static_cast<uint32_t>(f_refs.size());
                out.count_flags = RT_NODE_LEAF | static_cast<uint32_t>(n.faces.size());
                for (int f : n.faces) f_refs.push_back(static_cast<uint32_t>(f));
            } else if (n.empty) {
                // only reachable for an empty root: nothing to traverse
                out.first = 0;
                out.count_flags = 0;
            } else {
                out.first = static_cast<uint32_t>(order.size());
                uint32_t live = 0;
                for (int k : n.kids)
                    if (!pool[k].empty) {
                        order.push_back(k);
                        f_nodes.push_back(rt_node{});
                        ++live;
                    }
                out.count_flags = live;
            }
            f_nodes[head] = out;
        }
    }
    const size_t nf = tris.size();
    f_tri_verts.resize(nf * 9); f_face_normal.resize(nf * 3); f_tri_vid.resize(nf * 3); f_mat_id.resize(nf);
    for (size_t i = 0; i < nf; ++i) {
        for (int k = 0; k < 3; ++k) {
            const V3 &w = world[tris[i].vid[k]];
            f_tri_verts[i * 9 + k * 3] = w.x; f_tri_verts[i * 9 + k * 3 + 1] = w.y; f_tri_verts[i * 9 + k * 3 + 2] = w.z;
            f_tri_vid[i * 3 + k] = tris[i].vid[k];
        }
        f_face_normal[i * 3] = tris[i].normal.x; f_face_normal[i * 3 + 1] = tris[i].normal.y; f_face_normal[i * 3 + 2] = tris[i].normal.z;
        f_mat_id[i] = tris[i].material;
    }
    f_vert_normal.resize(normals.size() * 3);
    for (size_t i = 0; i < normals.size(); ++i) {
        f_vert_normal[i * 3] = normals[i].x; f_vert_normal[i * 3 + 1] = normals[i].y; f_vert_normal[i * 3 + 2] = normals[i].z;
    }
    f_mats.resize(mats.size());
    for (size_t i = 0; i < mats.size(); ++i) {
        rt_material &m = f_mats[i];
        m.kd[0] = mats[i].kd.x; m.kd[1] = mats[i].kd.y; m.kd[2] = mats[i].kd.z;
        m.ks[0] = mats[i].ks.x; m.ks[1] = mats[i].ks.y; m.ks[2] = mats[i].ks.z;
        m.shininess = mats[i].ns; m.optical_density = mats[i].ni; m.illum = mats[i].illum;
    }
}

void HostScene::view(rt_scene *o) const {
    std::memset(o, 0, sizeof *o);
    o->n_nodes = static_cast<uint32_t>(f_nodes.size()); o->nodes = f_nodes.data();
    o->n_face_refs = static_cast<uint32_t>(f_refs.size()); o->face_refs = f_refs.data();
    o->n_faces = static_cast<uint32_t>(tris.size());
    o->tri_verts = f_tri_verts.data(); o->face_normal = f_face_normal.data();
    o->tri_vid = f_tri_vid.data(); o->mat_id = f_mat_id.data();
    o->n_vert_normals = static_cast<uint32_t>(normals.size()); o->vert_normal = f_vert_normal.data();
    o->n_materials = static_cast<uint32_t>(f_mats.size()); o->materials = f_mats.data();
    std::memcpy(o->model, model.m, sizeof(float) * 12);
}

void HostScene::info(int32_t out[8], float root_box[6]) const {
    int leaves = 0, refs = 0, biggest = 0, depth = 0, lost_nodes = 0;
    std::vector<char> seen(tris.size(), 0);
    for (const OctNode &n : pool) {
        depth = std::max(depth, n.level);
        if (n.leaf && !n.empty) {
            ++leaves;
            refs += static_cast<int>(n.faces.size());
            biggest = std::max(biggest, static_cast<int>(n.faces.size()));
            for (int f : n.faces) seen[f] = 1;
        } else if (!n.leaf && !n.empty && n.kids.empty() && !n.faces.empty()) {
            ++lost_nodes;
        }
    }
    int unreachable = 0;
    for (char s : seen) unreachable += s ? 0 : 1;
    out[0] = static_cast<int32_t>(pool.size()); out[1] = leaves; out[2] = refs; out[3] = biggest; out[4] = depth;
    out[5] = unreachable; out[6] = static_cast<int32_t>(f_nodes.size()); out[7] = lost_nodes;
    if (root_box && !pool.empty()) {
        root_box[0] = pool[0].box.lo.x; root_box[1] = pool[0].box.lo.y; root_box[2] = pool[0].box.lo.z;
        root_box[3] = pool[0].box.hi.x; root_box[4] = pool[0].box.hi.y; root_box[5] = pool[0].box.hi.z;
    }
}

// ---------------------------------------------------------------------------------------------------------
// camera / lights / PPM
// ---------------------------------------------------------------------------------------------------------
void default_camera(rt_camera *cam, int w, int h) {
    // setPerspectiveMatrix(60, w/(float)h, .1, 100); setViewport((w,h)); view = T(0,0,-2) (flyscene.cpp:46-47,
    // flycamera.hpp:76-86,166-191) => inverse view = T(0,0,2), centre (0,0,2)
    std::memset(cam, 0, sizeof *cam);
    cam->fovy = 60.0f;
    cam->aspect = static_cast<float>(w) / static_cast<float>(h);
    cam->viewport[2] = static_cast<float>(w);
    cam->viewport[3] = static_cast<float>(h);
    cam->inv_view[0] = cam->inv_view[5] = cam->inv_view[10] = 1.0f;
    cam->inv_view[11] = 2.0f;
    cam->center[2] = 2.0f;
}

// The fly camera's view for camera paths: Flycamera::updateViewMatrix with rotation_Y_axis = yaw, rotation_X_axis = 0
// (flycamera.hpp:166-191), Camera::getCenter (camera.hpp:115-118) and getViewMatrix().inverse() (camera.hpp:170), evaluated with the
// operation order of the vendored Eigen 3.3.7: AngleAxis::toRotationMatrix (Geometry/AngleAxis.h), the 3x3 inverse by cofactors
// (LU/InverseImpl.h:126-170) and the affine inverse (Geometry/Transform.h: translation = -linear_inv * translation).
namespace {
void angle_axis_matrix(float angle, const float ax[3], float R[9]) {
    const float s = std::sin(angle), c = std::cos(angle);
    const float sa[3] = {s * ax[0], s * ax[1], s * ax[2]};
    const float c1 = 1.0f - c;
    const float ca[3] = {c1 * ax[0], c1 * ax[1], c1 * ax[2]};
    float tmp;
    tmp = ca[0] * ax[1]; R[1] = tmp - sa[2]; R[3] = tmp + sa[2];
    tmp = ca[0] * ax[2]; R[2] = tmp + sa[1]; R[6] = tmp - sa[1];
    tmp = ca[1] * ax[2]; R[5] = tmp - sa[0]; R[7] = tmp + sa[0];
    R[0] = ca[0] * ax[0] + c; R[4] = ca[1] * ax[1] + c; R[8] = ca[2] * ax[2] + c;
}
void mat3_vec(const float M[9], const float v[3], float o[3]) {
    for (int r = 0; r < 3; ++r) o[r] = M[r * 3] * v[0] + (M[r * 3 + 1] * v[1] + M[r * 3 + 2] * v[2]);
}
float cofactor3(const float m[9], int i, int j) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return m[i1 * 3 + j1] * m[i2 * 3 + j2] - m[i1 * 3 + j2] * m[i2 * 3 + j1];
}
void mat3_inverse(const float m[9], float inv[9]) {
    const float c0 = cofactor3(m, 0, 0), c1 = cofactor3(m, 1, 0), c2 = cofactor3(m, 2, 0);
    const float det = c0 * m[0] + (c1 * m[3] + c2 * m[6]);
    const float invdet = 1.0f / det;
    inv[0] = c0 * invdet; inv[1] = c1 * invdet; inv[2] = c2 * invdet;
    inv[3] = cofactor3(m, 0, 1) * invdet; inv[4] = cofactor3(m, 1, 1) * invdet; inv[5] = cofactor3(m, 2, 1) * invdet;
    inv[6] = cofactor3(m, 0, 2) * invdet; inv[7] = cofactor3(m, 1, 2) * invdet; inv[8] = cofactor3(m, 2, 2) * invdet;
}
void unit3(float v[3]) { const V3 u = unit_fixed({v[0], v[1], v[2]}); v[0] = u.x; v[1] = u.y; v[2] = u.z; }
}  // namespace

void yaw_camera(rt_camera *cam, int w, int h, float yaw) {
    default_camera(cam, w, h);
    if (yaw == 0.0f) return;
    const float uy[3] = {0.f, 1.f, 0.f}, ux[3] = {1.f, 0.f, 0.f}, uz[3] = {0.f, 0.f, 1.f};
    float Ry[9], R0[9], rx[3], ry[3], rz[3], tmpv[3];
    angle_axis_matrix(yaw, uy, Ry);
    mat3_vec(Ry, ux, rx); unit3(rx);
    mat3_vec(Ry, uz, tmpv);
    angle_axis_matrix(0.0f, rx, R0);
    mat3_vec(R0, tmpv, rz); unit3(rz);
    mat3_vec(R0, uy, ry); unit3(ry);
    const float R[9] = {rx[0], rx[1], rx[2], ry[0], ry[1], ry[2], rz[0], rz[1], rz[2]};      // rotation_matrix rows; view.linear = I * I * R
    const float push[3] = {0.f, 0.f, -2.f};
    float t[3], Linv[9];
    mat3_vec(R, push, t);                                                                     // translate(default_translation)
    mat3_inverse(R, Linv);
    for (int r = 0; r < 3; ++r) {
        for (int k = 0; k < 3; ++k) cam->inv_view[r * 4 + k] = Linv[r * 3 + k];
        cam->inv_view[r * 4 + 3] = (-Linv[r * 3]) * t[0] + ((-Linv[r * 3 + 1]) * t[1] + (-Linv[r * 3 + 2]) * t[2]);
        cam->center[r] = Linv[r * 3] * (-t[0]) + (Linv[r * 3 + 1] * (-t[1]) + Linv[r * 3 + 2] * (-t[2]));
    }
}

void screen_to_world(const rt_camera *cam, float i, float j, float out[3]) {
    // Camera::screenToWorld (camera.hpp:155-173): raster -> [-1,1] in DOUBLE, cast to float, scale by the
    // perspective factor, then inverse view.  getPerspectiveScale (camera.hpp:263-266) mixes float and double.
    float n0 = static_cast<float>(2.0 * static_cast<double>(i - cam->viewport[0]) / static_cast<double>(cam->viewport[2]) - 1.0);
    float n1 = static_cast<float>(1.0 - 2.0 * static_cast<double>(j - cam->viewport[1]) / static_cast<double>(cam->viewport[3]));
    const float n2 = -1.0f;
    const float persp = static_cast<float>(static_cast<double>(1.0f) / std::tan(static_cast<double>(cam->fovy / 2.0f) * (M_PI / static_cast<double>(180.0f))));
    const float scale = static_cast<float>(1.0 / static_cast<double>(persp));
    n0 = n0 * (cam->aspect * scale);
    n1 = n1 * scale;
    const float *m = cam->inv_view;
    for (int r = 0; r < 3; ++r) out[r] = ((m[r * 4] * n0 + m[r * 4 + 1] * n1) + m[r * 4 + 2] * n2) + m[r * 4 + 3] * 1.0f;
}

void sphere_offsets(uint32_t seed, float radius, int n, float *out) {
    // the sphere-point loop of Flyscene::createSpherePoint (flyscene.cpp:976-993) with std::random_device replaced by seed + i; the same
    // standard-library generator and distribution as the reference, so the offsets are what its code yields for that generator state
    for (int i = 0; i < n; ++i) {
        std::mt19937 gen(seed + static_cast<uint32_t>(i));
        std::uniform_real_distribution<> dis(0, 1);
        float randomno = dis(gen);
        float theta = 2.0f * M_PI * randomno;
        float phi = std::acos(2.0 * randomno - 1.0);
        float x = radius * std::sin(phi) * std::cos(theta);
        float y = radius * std::sin(phi) * std::sin(theta);
        float z = radius * std::cos(phi);
        out[i * 3] = x / 5; out[i * 3 + 1] = y / 5; out[i * 3 + 2] = z / 5;       // Vector3f / int: the int becomes a float, true division
    }
}

void default_lights(rt_lights *l, int area) {
    std::memset(l, 0, sizeof *l);
    l->n_lights = 1;
    l->pos[0][0] = -1.0f; l->pos[0][1] = 1.0f; l->pos[0][2] = 1.0f;   // flyscene.cpp:72
    l->color[0] = 1.0f; l->color[1] = 1.0f; l->color[2] = 0.0f;       // flyscene.cpp:68
    l->mode = area ? RT_LIGHT_AREA : RT_LIGHT_POINT;
    l->usteps = 5; l->vsteps = 5;                                     // flyscene.cpp:971
    l->len_x = static_cast<float>(0.3); l->len_y = static_cast<float>(0.15);
}

// Byte-exact ASCII P3 (ppmIO.hpp:130-151) through a digit table and one large buffer: the reference's
// ofstream << int path costs ~0.3 s per 1080p frame, which would dwarf the GPU frame time.
namespace {
struct DigitTable {
    char txt[256][4];
    unsigned char len[256];
    DigitTable() {
        for (int v = 0; v < 256; ++v) len[v] = static_cast<unsigned char>(std::snprintf(txt[v], 4, "%d", v));
    }
};
inline int quantise(float c) {
    const int v = static_cast<int>(255 * c);  // truncation toward zero, ppmIO.hpp:145
    return v < 255 ? v : 255;
}
}  // namespace

int write_ppm(const char *path, const float *rgb, int w, int h) {
    static const DigitTable table;
    FILE *f = std::fopen(path, "wb");
    if (!f) return 0;
    std::fprintf(f, "P3\n%d %d\n255\n", w, h);
    std::vector<char> row(static_cast<size_t>(w) * 3 * 12 + 2);
    for (int j = 0; j < h; ++j) {
        char *p = row.data();
        const float *src = rgb + static_cast<size_t>(j) * w * 3;
        for (int i = 0; i < w * 3; ++i) {
            const int v = quantise(src[i]);
            if (v >= 0) {
                std::memcpy(p, table.txt[v], table.len[v]);
                p += table.len[v];
            } else {
                p += std::sprintf(p, "%d", v);  // negatives are not clamped by the reference
            }
            *p++ = ' ';
        }
        *p++ = '\n';
        std::fwrite(row.data(), 1, static_cast<size_t>(p - row.data()), f);
    }
    std::fclose(f);
    return 1;
}

int write_ppm_u8(const char *path, const uint8_t *rgb, int w, int h) {
    static const DigitTable table;
    FILE *f = std::fopen(path, "wb");
    if (!f) return 0;
    std::fprintf(f, "P3\n%d %d\n255\n", w, h);
    std::vector<char> row(static_cast<size_t>(w) * 3 * 4 + 2);
    for (int j = 0; j < h; ++j) {
        char *p = row.data();
        const uint8_t *src = rgb + static_cast<size_t>(j) * w * 3;
        for (int i = 0; i < w * 3; ++i) {
            std::memcpy(p, table.txt[src[i]], table.len[src[i]]);
            p += table.len[src[i]];
            *p++ = ' ';
        }
        *p++ = '\n';
        std::fwrite(row.data(), 1, static_cast<size_t>(p - row.data()), f);
    }
    std::fclose(f);
    return 1;
}

}  // namespace rtamd
