// ref_probe2.cpp -- ORACLE-SIDE probe no. 2 (test infrastructure, NOT the product).
//
// Compiled IN PLACE, together with the reference's own src/boundingBox.cpp and src/boxTree.cpp, against the reference's headers
// (include paths only: src/, dependencies/{tucano, eigen/include, glew/include, glfw/include}) by `make -C oracle ref` into
// oracle/_ref/ (git-ignored).  No GL symbol is referenced by what is called here, so the binary links with NO stand-in of any kind:
//   * BoundingBox::boxIntersect                      src/boundingBox.cpp:48-83      (the reference's compiled code)
//   * BoxTree::intersect                             src/boxTree.cpp:150-173        (on a tree re-assembled through the public fields,
//                                                                                    src/boxTree.hpp:17-22)
//   * BoxTree::planeBoxOverlap / axisTest* / findMinMax   src/boxTree.cpp:338-456   (members of a default-constructed BoxTree)
//   * Tucano::Camera::screenToWorld / getCenter      tucano/camera.hpp:115-118,155-173
//   * Tucano::ImageImporter::writePPMImage           tucano/utils/ppmIO.hpp:130-151
// What can NOT be built here: Flyscene (flyscene.cpp), Tucano::Mesh construction and Flycamera construction call GLEW entry points;
// GLEW/GLFW libraries are absent from this image and writing stand-ins is not allowed.  So BoxTree(Mesh&, int), split and clasifyFace
// (they take a Mesh&) cannot be CALLED; `sat` below re-walks clasifyFace's control flow (boxTree.cpp:203-336) with every arithmetic
// step evaluated by the reference's own members and by Eigen -- it pins the arithmetic, not the flow.
//
// Nothing from the reference is copied into this repository: this driver reads binary inputs written by oracle/make_ref_fixtures.py,
// EVALUATES the reference's functions on them and writes the results; the script packs them into tests/golden/ref_pins.npz.
//
//   ref_probe2 box  <in: n x 12 f32 (bmin3 bmax3 origin3 dest3)>            <out: n bytes>
//   ref_probe2 tree <tree file> <rays: n x 6 f32 (origin3 dest3)>           <out: per ray  int32 count, then count int32 ids>
//   ref_probe2 sat  <in: n x 15 f32 (bmin3 bmax3 A3 B3 C3)>                 <out: n bytes (clasifyFace decision)>
//   ref_probe2 prim <in: n x 16 f32>                                        <out: n x 12 bytes (primitive decisions) + n x 2 f32 (findMinMax)>
//   ref_probe2 cam  <W> <H> <yaw (float bits, hex)> <out: center3 f32, then W*H*3 f32 (pixel (i,j) at [(j*W+i)*3])>
//   ref_probe2 ppm  <in: W*H*3 f32 row-major [y][x]> <W> <H> <out.ppm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <set>
#include <string>
#include <vector>
using namespace std;
#include "boxTree.hpp"            // pulls boundingBox.hpp, Eigen, tucano/mesh.hpp (declarations only are used)
#include <tucano/camera.hpp>
#include <tucano/utils/ppmIO.hpp>

static vector<float> read_f32(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    vector<float> v(n / 4);
    if (fread(v.data(), 4, v.size(), f) != v.size()) { fprintf(stderr, "short read %s\n", path); exit(2); }
    fclose(f);
    return v;
}
static Eigen::Vector3f v3(const float *p) { return Eigen::Vector3f(p[0], p[1], p[2]); }

// ---- tree file: int32 n_nodes; per node: 6 f32 box, int32 isLeaf, isEmpty, nchildren, 8 x int32 child ids, int32 nfaces, faces ----
struct NodeRec { float box[6]; int leaf, empty, nch, ch[8], nf; vector<int> faces; };
static BoxTree assemble(const vector<NodeRec> &N, int i, bool leaf_ids, int *leaf_counter) {
    BoxTree t;                                         // BoxTree(void) {} -- public fields, src/boxTree.hpp:17-24
    t.box = BoundingBox(v3(N[i].box), v3(N[i].box + 3));
    t.capacity = 1000;
    t.isLeaf = N[i].leaf != 0;
    t.isEmpty = N[i].empty != 0;
    if (leaf_ids) {
        // every leaf that holds faces carries ONE pseudo face = its node index: intersect() then returns the set of intersected leaves
        if (N[i].leaf && !N[i].empty && N[i].nf > 0) t.faces.push_back(i);
        (void)leaf_counter;
    } else {
        t.faces = N[i].faces;
    }
    for (int k = 0; k < N[i].nch; ++k) t.children.push_back(assemble(N, N[i].ch[k], leaf_ids, leaf_counter));
    return t;
}

// Camera with Flycamera's view matrix (flycamera.hpp:76-86 reset(), :166-191 updateViewMatrix()): Flycamera itself cannot be
// constructed without GL (its CoordinateAxes member builds meshes), so the same Eigen statements are issued on the base class.
struct ProbeCamera : public Tucano::Camera {
    void fly_view(float rotation_Y_axis, float rotation_X_axis) {
        Eigen::Matrix3f rotation_matrix = Eigen::Matrix3f::Identity(), default_rotation = Eigen::Matrix3f::Identity();
        Eigen::Vector3f default_translation(0.0, 0.0, -2.0), translation_vector = Eigen::Vector3f::Zero();
        resetViewMatrix();
        Eigen::Vector3f rotX = Eigen::AngleAxisf(rotation_Y_axis, Eigen::Vector3f::UnitY()) * Eigen::Vector3f::UnitX();
        rotX.normalize();
        Eigen::Vector3f rotZ = Eigen::AngleAxisf(rotation_Y_axis, Eigen::Vector3f::UnitY()) * Eigen::Vector3f::UnitZ();
        rotZ = Eigen::AngleAxisf(rotation_X_axis, rotX) * rotZ;
        rotZ.normalize();
        Eigen::Vector3f rotY = Eigen::AngleAxisf(rotation_X_axis, rotX) * Eigen::Vector3f::UnitY();
        rotY.normalize();
        rotation_matrix.row(0) = rotX;
        rotation_matrix.row(1) = rotY;
        rotation_matrix.row(2) = rotZ;
        view_matrix.rotate(default_rotation);
        view_matrix.rotate(rotation_matrix);
        view_matrix.translate(default_translation);
        view_matrix.translate(translation_vector);
    }
};

int main(int argc, char **argv) {
    if (argc < 2) return 1;
    const string cmd = argv[1];
    if (cmd == "box" && argc == 4) {
        vector<float> in = read_f32(argv[2]);
        const size_t n = in.size() / 12;
        vector<unsigned char> out(n);
        for (size_t i = 0; i < n; ++i) {
            const float *p = &in[i * 12];
            BoundingBox b(v3(p), v3(p + 3));
            out[i] = b.boxIntersect(v3(p + 6), v3(p + 9)) ? 1 : 0;
        }
        FILE *f = fopen(argv[3], "wb"); fwrite(out.data(), 1, n, f); fclose(f);
        return 0;
    }
    if (cmd == "tree" && argc == 6) {
        const bool leaf_ids = string(argv[5]) == "leaves";
        FILE *f = fopen(argv[2], "rb");
        if (!f) return 2;
        int nn = 0;
        if (fread(&nn, 4, 1, f) != 1) return 2;
        vector<NodeRec> N(nn);
        for (int i = 0; i < nn; ++i) {
            NodeRec &r = N[i];
            if (fread(r.box, 4, 6, f) != 6) return 2;
            int hdr[3]; if (fread(hdr, 4, 3, f) != 3) return 2;
            r.leaf = hdr[0]; r.empty = hdr[1]; r.nch = hdr[2];
            if (fread(r.ch, 4, 8, f) != 8) return 2;
            if (fread(&r.nf, 4, 1, f) != 1) return 2;
            r.faces.resize(r.nf);
            if (r.nf && fread(r.faces.data(), 4, r.nf, f) != (size_t)r.nf) return 2;
        }
        fclose(f);
        int lc = 0;
        BoxTree root = assemble(N, 0, leaf_ids, &lc);
        vector<float> rays = read_f32(argv[3]);
        const size_t n = rays.size() / 6;
        FILE *o = fopen(argv[4], "wb");
        for (size_t i = 0; i < n; ++i) {
            std::set<int> s = root.intersect(v3(&rays[i * 6]), v3(&rays[i * 6 + 3]));       // src/boxTree.cpp:150-173
            int cnt = (int)s.size();
            fwrite(&cnt, 4, 1, o);
            for (int id : s) fwrite(&id, 4, 1, o);
        }
        fclose(o);
        return 0;
    }
    if (cmd == "sat" && argc == 4) {
        // clasifyFace's flow (src/boxTree.cpp:203-336) with the reference's members doing the arithmetic
        vector<float> in = read_f32(argv[2]);
        const size_t n = in.size() / 15;
        vector<unsigned char> out(n);
        BoxTree T;
        for (size_t i = 0; i < n; ++i) {
            const float *p = &in[i * 15];
            BoundingBox box(v3(p), v3(p + 3));
            Eigen::Vector3f vertices[3] = {v3(p + 6), v3(p + 9), v3(p + 12)};
            int countVertexesInBox = 0;
            for (Eigen::Vector3f vertex : vertices) {
                if (box.getMin().x() <= vertex.x() && box.getMax().x() >= vertex.x() && box.getMin().y() <= vertex.y() && box.getMax().y() >= vertex.y() &&
                    box.getMin().z() <= vertex.z() && box.getMax().z() >= vertex.z())
                    countVertexesInBox++;
            }
            bool res;
            if (countVertexesInBox > 0) res = true;
            else {
                res = false;
                do {
                    Eigen::Vector3f boxcenter = Eigen::Vector3f(box.getMin().x() + (box.getMax().x() - box.getMin().x()) / 2.f,
                                                                box.getMin().y() + (box.getMax().y() - box.getMin().y()) / 2.f,
                                                                box.getMin().z() + (box.getMax().z() - box.getMin().z()) / 2.f);
                    Eigen::Vector3f boxhalfsize = (box.getMax() - boxcenter).normalized();
                    Eigen::Vector3f a_origin = (vertices[0] - boxcenter).normalized();
                    Eigen::Vector3f b_origin = (vertices[1] - boxcenter).normalized();
                    Eigen::Vector3f c_origin = (vertices[2] - boxcenter).normalized();
                    Eigen::Vector3f e_0 = b_origin - a_origin, e_1 = c_origin - b_origin, e_2 = a_origin - c_origin;
                    float fex = fabsf(e_0.x()), fey = fabsf(e_0.y()), fez = fabsf(e_0.z());
                    if (!T.axisTestX01(e_0.z(), e_0.y(), fez, fey, a_origin, c_origin, boxhalfsize)) break;
                    if (!T.axisTestY02(e_0.z(), e_0.x(), fez, fex, a_origin, c_origin, boxhalfsize)) break;
                    if (!T.axisTestZ12(e_0.y(), e_0.x(), fey, fex, b_origin, c_origin, boxhalfsize)) break;
                    fex = fabsf(e_1.x()); fey = fabsf(e_1.y()); fez = fabsf(e_1.z());
                    if (!T.axisTestX01(e_1.z(), e_1.y(), fez, fey, a_origin, c_origin, boxhalfsize)) break;
                    if (!T.axisTestY02(e_1.z(), e_1.x(), fez, fex, a_origin, c_origin, boxhalfsize)) break;
                    if (!T.axisTestZ0(e_1.y(), e_1.x(), fey, fex, a_origin, b_origin, boxhalfsize)) break;
                    fex = fabsf(e_2.x()); fey = fabsf(e_2.y()); fez = fabsf(e_2.z());
                    if (!T.axisTestX02(e_2.z(), e_2.y(), fez, fey, a_origin, b_origin, boxhalfsize)) break;
                    if (!T.axisTestY1(e_2.z(), e_2.x(), fez, fex, a_origin, b_origin, boxhalfsize)) break;
                    if (!T.axisTestZ12(e_2.y(), e_2.x(), fey, fex, b_origin, c_origin, boxhalfsize)) break;
                    pair<float, float> mm = T.findMinMax(a_origin.x(), b_origin.x(), c_origin.x());
                    if (mm.first > boxhalfsize.x() || mm.second < -boxhalfsize.x()) break;
                    mm = T.findMinMax(a_origin.y(), b_origin.y(), c_origin.y());
                    if (mm.first > boxhalfsize.y() || mm.second < -boxhalfsize.y()) break;
                    mm = T.findMinMax(a_origin.z(), b_origin.z(), c_origin.z());
                    if (mm.first > boxhalfsize.z() || mm.second < -boxhalfsize.z()) break;
                    Eigen::Vector3f edge1 = a_origin - b_origin, edge2 = a_origin - c_origin;
                    Eigen::Vector3f normal = (edge1.cross(edge2)).normalized();
                    if (!T.planeBoxOverlap(normal, a_origin, boxhalfsize)) break;
                    res = true;
                } while (0);
            }
            out[i] = res ? 1 : 0;
        }
        FILE *f = fopen(argv[3], "wb"); fwrite(out.data(), 1, n, f); fclose(f);
        return 0;
    }
    if (cmd == "prim" && argc == 4) {
        // 16 floats per case: a b fa fb | v0(3) | v1(3) | boxhalfsize(3) | 3 spare  -- the members straight, no flow
        vector<float> in = read_f32(argv[2]);
        const size_t n = in.size() / 16;
        vector<unsigned char> dec(n * 12, 0);
        vector<float> mm(n * 2);
        BoxTree T;
        for (size_t i = 0; i < n; ++i) {
            const float *p = &in[i * 16];
            Eigen::Vector3f v0 = v3(p + 4), v1 = v3(p + 7), bh = v3(p + 10);
            unsigned char *d = &dec[i * 12];
            d[0] = T.axisTestX01(p[0], p[1], p[2], p[3], v0, v1, bh);
            d[1] = T.axisTestY02(p[0], p[1], p[2], p[3], v0, v1, bh);
            d[2] = T.axisTestZ12(p[0], p[1], p[2], p[3], v0, v1, bh);
            d[3] = T.axisTestZ0(p[0], p[1], p[2], p[3], v0, v1, bh);
            d[4] = T.axisTestX02(p[0], p[1], p[2], p[3], v0, v1, bh);
            d[5] = T.axisTestY1(p[0], p[1], p[2], p[3], v0, v1, bh);
            d[6] = T.planeBoxOverlap(v0, v1, bh);                 // (normal, vert, maxbox)
            d[7] = T.planeBoxOverlap(v1, v0, bh);
            pair<float, float> r = T.findMinMax(p[13], p[14], p[15]);
            mm[i * 2] = r.first; mm[i * 2 + 1] = r.second;
        }
        FILE *f = fopen(argv[3], "wb"); fwrite(dec.data(), 1, dec.size(), f); fwrite(mm.data(), 4, mm.size(), f); fclose(f);
        return 0;
    }
    if (cmd == "cam" && argc == 6) {
        const int W = atoi(argv[2]), H = atoi(argv[3]);
        uint32_t bits = (uint32_t)strtoul(argv[4], nullptr, 16);
        float yaw; memcpy(&yaw, &bits, 4);
        ProbeCamera cam;
        cam.fly_view(yaw, 0.0f);
        cam.setPerspectiveMatrix(60.0, W / (float)H, 0.1f, 100.0f);            // flyscene.cpp:46
        cam.setViewport(Eigen::Vector2f((float)W, (float)H));                   // flyscene.cpp:47
        vector<float> out(3 + (size_t)W * H * 3);
        Eigen::Vector3f c = cam.getCenter();
        out[0] = c[0]; out[1] = c[1]; out[2] = c[2];
        for (int j = 0; j < H; ++j)
            for (int i = 0; i < W; ++i) {
                Eigen::Vector3f w = cam.screenToWorld(Eigen::Vector2f(i, j));    // flyscene.cpp:575 passes the pixel as floats
                float *o = &out[3 + ((size_t)j * W + i) * 3];
                o[0] = w[0]; o[1] = w[1]; o[2] = w[2];
            }
        FILE *f = fopen(argv[5], "wb"); fwrite(out.data(), 4, out.size(), f); fclose(f);
        return 0;
    }
    if (cmd == "ppm" && argc == 6) {
        vector<float> in = read_f32(argv[2]);
        const int W = atoi(argv[3]), H = atoi(argv[4]);
        // writePPMImage reads width = data[0].size(), height = data.size() and indexes data[i][j] with i < width, j < height
        // (ppmIO.hpp:132-145): it is only self-consistent for square images -- the reference's raytraceScene relies on that
        // (flyscene.cpp:525-527,620).  The probe therefore takes square inputs: data[i][j] = pixel (column i, row j).
        if (W != H) return 3;
        vector<vector<Eigen::Vector3f>> data(H, vector<Eigen::Vector3f>(W));
        for (int j = 0; j < H; ++j)
            for (int i = 0; i < W; ++i) data[i][j] = v3(&in[((size_t)j * W + i) * 3]);
        Tucano::ImageImporter::writePPMImage(argv[5], data);
        return 0;
    }
    fprintf(stderr, "usage: see the header of oracle/ref_probe2.cpp\n");
    return 1;
}
