/*
 * rt_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).  See rt_oracle.h.
 *
 * Plain C99 restatement of the reference's ray-trace path.  Build with
 *   gcc -O2 -std=gnu99 -ffp-contract=off -fno-fast-math   (baseline x86-64, no FMA)
 * so that every float operation rounds exactly like the reference's g++ -O2 build.
 *
 * Arithmetic conventions restated from the vendored Eigen 3.3.7
 * (dependencies/eigen/include/Eigen/src/Core/Redux.h, Dot.h):
 *   - fixed-size Vector3f dot / squaredNorm:  a0*b0 + (a1*b1 + a2*b2)     [Redux.h:91-105]
 *   - dynamic-size (VectorBlock<Vector4f,Dynamic>) squaredNorm: (c0 + c1) + c2 [Redux.h:200-245]
 *   - normalized()/normalize(): z = squaredNorm(); z > 0 ? v / sqrt(z) : v    [Dot.h:124-156]
 * Both are checked bit-for-bit against the real Eigen by oracle/ref_probe.cpp.
 */
#define _GNU_SOURCE
#include "rt_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------ */
/* small vector helpers (Eigen evaluation orders, see header comment)  */
/* ------------------------------------------------------------------ */
static inline float dot3(const float a[3], const float b[3]) {
    return a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]);
}
static inline void sub3(const float a[3], const float b[3], float o[3]) {
    o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2];
}
static inline void cross3(const float a[3], const float b[3], float o[3]) {
    /* Eigen/src/Geometry/OrthoMethods.h:34-40 */
    float x = a[1] * b[2] - a[2] * b[1];
    float y = a[2] * b[0] - a[0] * b[2];
    float z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline void normalize3_fixed(float v[3]) {
    float z = dot3(v, v);
    if (z > 0.0f) { float s = sqrtf(z); v[0] = v[0] / s; v[1] = v[1] / s; v[2] = v[2] / s; }
}
static inline void normalize3_dyn(float v[3]) {
    /* squaredNorm of a dynamic-size (max 4) expression: sequential sum */
    float z = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
    if (z > 0.0f) { float s = sqrtf(z); v[0] = v[0] / s; v[1] = v[1] / s; v[2] = v[2] / s; }
}
/* std::min / std::max exactly as libstdc++ defines them (NaN behaviour matters, boundingBox.cpp:63-72) */
static inline float stdminf(float a, float b) { return (b < a) ? b : a; }
static inline float stdmaxf(float a, float b) { return (a < b) ? b : a; }

/* ------------------------------------------------------------------ */
/* MTL loader   dependencies/tucano/tucano/utils/mtlIO.hpp:45-125      */
/* ------------------------------------------------------------------ */
static void mtl_default(omtl *m) {
    /* mtl.hpp:21-39 */
    m->ka[0] = m->ka[1] = m->ka[2] = (float)0.3;
    m->kd[0] = m->kd[1] = m->kd[2] = (float)0.5;
    m->ks[0] = m->ks[1] = m->ks[2] = (float)1.0;
    m->shininess = 10; m->optical_density = 0.0f; m->dissolve = 1.0f; m->illum = 0;
    m->name[0] = 0;
}

/* split on single ' ' with std::getline semantics (mtlIO.hpp:63-67) */
static int split_spaces(const char *line, char tok[][256], int maxtok) {
    int n = 0; const char *p = line;
    if (!*p) return 0;
    for (;;) {
        const char *q = strchr(p, ' ');
        size_t len = q ? (size_t)(q - p) : strlen(p);
        if (n < maxtok) { if (len > 255) len = 255; memcpy(tok[n], p, len); tok[n][len] = 0; n++; }
        if (!q) break;
        p = q + 1;
        if (!*p) break; /* delimiter was the last char: next getline extracts nothing and fails */
    }
    return n;
}

static int read_line(FILE *f, char **buf, size_t *cap) {
    /* std::getline(in, line): strips '\n', keeps '\r' */
    ssize_t n = getline(buf, cap, f);
    if (n < 0) return 0;
    if (n > 0 && (*buf)[n - 1] == '\n') (*buf)[n - 1] = 0;
    return 1;
}

static int load_mtl(const char *fn, omtl **mtls, int *nm) {
    FILE *f = fopen(fn, "r");
    if (!f) { fprintf(stderr, "oracle: cannot open %s\n", fn); return 0; }
    char *line = NULL; size_t cap = 0;
    static char tok[16][256];
    char (*t)[256] = malloc(16 * 256);
    (void)tok;
    while (read_line(f, &line, &cap)) {
        if (!line[0]) continue;
        int nt = split_spaces(line, t, 16);
        if (nt == 0) continue;
        if (strcmp(t[0], "#") == 0) continue;
        if (strcmp(t[0], "newmtl") == 0) {
            *mtls = realloc(*mtls, sizeof(omtl) * (size_t)(*nm + 1));
            mtl_default(&(*mtls)[*nm]);
            if (nt > 1) { strncpy((*mtls)[*nm].name, t[1], 127); (*mtls)[*nm].name[127] = 0; }
            (*nm)++;
            continue;
        }
        if (*nm == 0) continue; /* materials.back() on empty vector is UB in the reference; ignore */
        omtl *m = &(*mtls)[*nm - 1];
        if (strcmp(t[0], "Ns") == 0 && nt > 1) m->shininess = (float)atof(t[1]);
        else if (strcmp(t[0], "Ka") == 0 && nt > 3) { for (int k = 0; k < 3; k++) m->ka[k] = (float)atof(t[1 + k]); }
        else if (strcmp(t[0], "Kd") == 0 && nt > 3) { for (int k = 0; k < 3; k++) m->kd[k] = (float)atof(t[1 + k]); }
        else if (strcmp(t[0], "Ks") == 0 && nt > 3) { for (int k = 0; k < 3; k++) m->ks[k] = (float)atof(t[1 + k]); }
        else if (strcmp(t[0], "Ni") == 0 && nt > 1) m->optical_density = (float)atof(t[1]);
        else if (strcmp(t[0], "d") == 0 && nt > 1) m->dissolve = (float)atof(t[1]);
        else if (strcmp(t[0], "illum") == 0 && nt > 1) m->illum = atoi(t[1]);
    }
    free(t); free(line); fclose(f);
    if (*nm == 0) { /* mtlIO.hpp:112-116 */
        *mtls = realloc(*mtls, sizeof(omtl)); mtl_default(&(*mtls)[0]); *nm = 1;
    }
    return 1;
}

/* ------------------------------------------------------------------ */
/* OBJ loader   dependencies/tucano/tucano/utils/objimporter.hpp:83-284 */
/* ------------------------------------------------------------------ */
typedef struct { unsigned *ids; int n, cap; int mat; } ogroup;

static void grp_push(ogroup *g, unsigned v) {
    if (g->n == g->cap) { g->cap = g->cap ? g->cap * 2 : 64; g->ids = realloc(g->ids, sizeof(unsigned) * (size_t)g->cap); }
    g->ids[g->n++] = v;
}

/* operator>>(istream&, float&): skip whitespace, strtof; failure leaves 0 (C++11) */
static const char *parse_float(const char *p, float *out, int *ok) {
    if (!*ok) { *out = 0.0f; return p; } /* failbit already set: further extractions are no-ops... value untouched */
    char *e; float v = strtof(p, &e);
    if (e == p) { *ok = 0; *out = 0.0f; return p; }
    *out = v; return e;
}

static void affine_identity(float m[12]) {
    memset(m, 0, sizeof(float) * 12); m[0] = m[5] = m[10] = 1.0f;
}

static void compute_world(oscene *s);

/* ---- PLY (plyimporter.hpp:186-262 via rply): the reference's PLY path fills GL buffers only and never builds the faces the ray tracer
   reads, so no reference semantics exist; defined as "the mesh loadObjFile would build from the same data": x,y,z (w = 1), the
   file's nx,ny,nz as the vn list, the first three indices of every face list (face_cb, plyimporter.hpp:104-118), no materials. ---- */
typedef struct { char name[64]; int size; char kind; int list; int csize; char ckind; } plyprop;
typedef struct { char name[64]; long count; plyprop props[32]; int nprops; } plyelem;
static int ply_type(const char *t, int *size, char *kind) {
    static const struct { const char *n; int s; char k; } T[] = {
        {"char", 1, 'i'}, {"int8", 1, 'i'}, {"uchar", 1, 'u'}, {"uint8", 1, 'u'}, {"short", 2, 'i'}, {"int16", 2, 'i'}, {"ushort", 2, 'u'}, {"uint16", 2, 'u'},
        {"int", 4, 'i'}, {"int32", 4, 'i'}, {"uint", 4, 'u'}, {"uint32", 4, 'u'}, {"float", 4, 'f'}, {"float32", 4, 'f'}, {"double", 8, 'f'}, {"float64", 8, 'f'}};
    for (size_t i = 0; i < sizeof T / sizeof T[0]; i++) if (strcmp(t, T[i].n) == 0) { *size = T[i].s; *kind = T[i].k; return 1; }
    return 0;
}
static int ply_scalar(FILE *f, int ascii, int size, char kind, double *out) {
    if (ascii) return fscanf(f, "%lf", out) == 1;
    unsigned char b[8];
    if (fread(b, 1, (size_t)size, f) != (size_t)size) return 0;
    if (kind == 'f') { if (size == 4) { float v; memcpy(&v, b, 4); *out = v; } else { double v; memcpy(&v, b, 8); *out = v; } return 1; }
    uint64_t u = 0;
    for (int i = size - 1; i >= 0; i--) u = (u << 8) | b[i];
    if (kind == 'i') {
        int64_t v = size == 1 ? (int8_t)u : size == 2 ? (int16_t)u : (int32_t)u;
        *out = (double)v;
    } else *out = (double)u;
    return 1;
}
/* fills vert (x,y,z,1), norm (nx,ny,nz) and one index group; returns 0 on a malformed file */
static int load_ply_arrays(const char *path, float **vert, int *nv, float **norm, int *nn, ogroup *grp) {
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "oracle: cannot open %s\n", path); return 0; }
    char line[1024];
    if (!fgets(line, sizeof line, f) || strncmp(line, "ply", 3) != 0) { fclose(f); return 0; }
    int ascii = 0, have_format = 0, ne = 0;
    plyelem *el = calloc(16, sizeof(plyelem));
    while (fgets(line, sizeof line, f)) {
        char key[64] = "", a[64] = "", b[64] = "", c[64] = "", d[64] = "";
        int n = sscanf(line, "%63s %63s %63s %63s %63s", key, a, b, c, d);
        if (n < 1) continue;
        if (strcmp(key, "end_header") == 0) break;
        if (strcmp(key, "format") == 0) { ascii = strcmp(a, "ascii") == 0; if (!ascii && strcmp(a, "binary_little_endian") != 0) { fclose(f); free(el); return 0; } have_format = 1; }
        else if (strcmp(key, "element") == 0 && ne < 16) { snprintf(el[ne].name, 64, "%s", a); el[ne].count = atol(b); el[ne].nprops = 0; ne++; }
        else if (strcmp(key, "property") == 0 && ne > 0 && el[ne - 1].nprops < 32) {
            plyprop *p = &el[ne - 1].props[el[ne - 1].nprops++];
            memset(p, 0, sizeof *p);
            if (strcmp(a, "list") == 0) { p->list = 1; if (!ply_type(b, &p->csize, &p->ckind) || !ply_type(c, &p->size, &p->kind)) { fclose(f); free(el); return 0; } snprintf(p->name, 64, "%s", d); }
            else { if (!ply_type(a, &p->size, &p->kind)) { fclose(f); free(el); return 0; } snprintf(p->name, 64, "%s", b); }
        }
    }
    if (!have_format) { fclose(f); free(el); return 0; }
    int cv = 0, cn = 0, ok = 1;
    *nv = 0; *nn = 0; *vert = NULL; *norm = NULL;
    for (int e = 0; e < ne && ok; e++) {
        int is_v = strcmp(el[e].name, "vertex") == 0, is_f = strcmp(el[e].name, "face") == 0;
        for (long i = 0; i < el[e].count && ok; i++) {
            float v[3] = {0, 0, 0}, nr[3] = {0, 0, 0}; int have_nz = 0;
            for (int k = 0; k < el[e].nprops && ok; k++) {
                const plyprop *p = &el[e].props[k];
                double val = 0;
                if (p->list) {
                    double cnt = 0;
                    if (!ply_scalar(f, ascii, p->csize, p->ckind, &cnt)) { ok = 0; break; }
                    for (long j = 0; j < (long)cnt; j++) {
                        if (!ply_scalar(f, ascii, p->size, p->kind, &val)) { ok = 0; break; }
                        if (is_f && strcmp(p->name, "vertex_indices") == 0 && j < 3) grp_push(grp, (unsigned)val);
                    }
                } else {
                    if (!ply_scalar(f, ascii, p->size, p->kind, &val)) { ok = 0; break; }
                    if (is_v) {
                        if (strcmp(p->name, "x") == 0) v[0] = (float)val; else if (strcmp(p->name, "y") == 0) v[1] = (float)val;
                        else if (strcmp(p->name, "z") == 0) v[2] = (float)val; else if (strcmp(p->name, "nx") == 0) nr[0] = (float)val;
                        else if (strcmp(p->name, "ny") == 0) nr[1] = (float)val; else if (strcmp(p->name, "nz") == 0) { nr[2] = (float)val; have_nz = 1; }
                    }
                }
            }
            if (is_v && ok) {
                if (*nv == cv) { cv = cv ? cv * 2 : 1024; *vert = realloc(*vert, sizeof(float) * 4 * (size_t)cv); }
                (*vert)[*nv * 4] = v[0]; (*vert)[*nv * 4 + 1] = v[1]; (*vert)[*nv * 4 + 2] = v[2]; (*vert)[*nv * 4 + 3] = 1.0f; (*nv)++;
                if (have_nz) {
                    if (*nn == cn) { cn = cn ? cn * 2 : 1024; *norm = realloc(*norm, sizeof(float) * 3 * (size_t)cn); }
                    (*norm)[*nn * 3] = nr[0]; (*norm)[*nn * 3 + 1] = nr[1]; (*norm)[*nn * 3 + 2] = nr[2]; (*nn)++;
                }
            }
        }
    }
    fclose(f); free(el);
    return ok;
}

oscene *orc_load_obj(const char *obj_path) {
    const char *ext = strrchr(obj_path, '.');
    const int is_ply = ext && (strcmp(ext, ".ply") == 0 || strcmp(ext, ".PLY") == 0);
    FILE *f = is_ply ? NULL : fopen(obj_path, "r");
    if (!f && !is_ply) { fprintf(stderr, "oracle: cannot open %s\n", obj_path); return NULL; }
    oscene *s = calloc(1, sizeof(oscene));
    char dir[1024]; { /* getPathName, objimporter.hpp:44-48 */
        const char *sl = strrchr(obj_path, '/'); const char *bs = strrchr(obj_path, '\\');
        const char *last = sl > bs ? sl : bs;
        size_t n = last ? (size_t)(last - obj_path + 1) : 0; if (n > 1000) n = 1000;
        memcpy(dir, obj_path, n); dir[n] = 0;
    }
    float *vert = NULL; int nv = 0, cv = 0;
    float *norm = NULL; int nn = 0, cn = 0;
    ogroup *grp = calloc(1, sizeof(ogroup)); int ng = 1; grp[0].mat = -1;
    int current_mat = -1;
    char *line = NULL; size_t cap = 0;
    if (is_ply && !load_ply_arrays(obj_path, &vert, &nv, &norm, &nn, &grp[0])) { fprintf(stderr, "oracle: bad PLY %s\n", obj_path); free(grp); free(s); return NULL; }
    while (!is_ply && read_line(f, &line, &cap)) {
        size_t len = strlen(line);
        if (len >= 6 && strncmp(line, "mtllib", 6) == 0) {
            if (len < 7) continue;
            char fn[2048]; snprintf(fn, sizeof fn, "%s%s", dir, line + 7);
            /* remove '\n' and '\r' anywhere (objimporter.hpp:128-129) */
            char *w = fn; for (char *r = fn; *r; r++) if (*r != '\n' && *r != '\r') *w++ = *r; *w = 0;
            load_mtl(fn, &s->mtls, &s->nmtls);
        } else if (len >= 6 && strncmp(line, "usemtl", 6) == 0) {
            if (grp[ng - 1].n != 0) {
                grp = realloc(grp, sizeof(ogroup) * (size_t)(ng + 1));
                memset(&grp[ng], 0, sizeof(ogroup)); grp[ng].mat = -1; ng++;
            }
            const char *nm = len >= 7 ? line + 7 : "";
            for (int i = 0; i < s->nmtls; i++) if (strcmp(s->mtls[i].name, nm) == 0) current_mat = i;
            grp[ng - 1].mat = current_mat;
        } else if (len >= 2 && line[0] == 'v' && line[1] == ' ') {
            float v[3]; int ok = 1; const char *p = line + 2;
            p = parse_float(p, &v[0], &ok); p = parse_float(p, &v[1], &ok); p = parse_float(p, &v[2], &ok);
            if (nv == cv) { cv = cv ? cv * 2 : 1024; vert = realloc(vert, sizeof(float) * 4 * (size_t)cv); }
            vert[nv * 4 + 0] = v[0]; vert[nv * 4 + 1] = v[1]; vert[nv * 4 + 2] = v[2]; vert[nv * 4 + 3] = 1.0f; nv++;
        } else if (len >= 2 && line[0] == 'v' && line[1] == 'n') {
            float v[3] = {0, 0, 0}; int ok = 1; const char *p = len >= 3 ? line + 3 : "";
            p = parse_float(p, &v[0], &ok); p = parse_float(p, &v[1], &ok); p = parse_float(p, &v[2], &ok);
            if (nn == cn) { cn = cn ? cn * 2 : 1024; norm = realloc(norm, sizeof(float) * 3 * (size_t)cn); }
            norm[nn * 3 + 0] = v[0]; norm[nn * 3 + 1] = v[1]; norm[nn * 3 + 2] = v[2]; nn++;
        } else if (len >= 2 && line[0] == 'f' && line[1] == ' ') {
            /* whitespace-separated elements; only the vertex id before the first '/' is used (objimporter.hpp:196-207) */
            const char *p = line + 2;
            while (*p) {
                while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\v' || *p == '\f') p++;
                if (!*p) break;
                int vid = atoi(p); /* stoi of the text before '/' */
                grp_push(&grp[ng - 1], (unsigned)(vid - 1));
                while (*p && !(*p == ' ' || *p == '\t' || *p == '\r' || *p == '\v' || *p == '\f')) p++;
            }
        }
    }
    free(line); if (f) fclose(f);

    /* computeNormals (objimporter.hpp:50-74): APPENDS nverts zero normals to the file's vn list and accumulates
       unit face normals at normals[vertex_id] -- i.e. into the file's vn slots when the file has any. */
    s->nnormals = nn + nv;
    s->normals = calloc((size_t)s->nnormals * 3 + 3, sizeof(float));
    memcpy(s->normals, norm, sizeof(float) * 3 * (size_t)nn);
    for (int g = 0; g < ng; g++) {
        for (int i = 0; i + 2 < grp[g].n; i += 3) {
            unsigned a = grp[g].ids[i], b = grp[g].ids[i + 1], c = grp[g].ids[i + 2];
            float v1[3], v0[3], n[3];
            sub3(&vert[c * 4], &vert[a * 4], v1);
            sub3(&vert[b * 4], &vert[a * 4], v0);
            normalize3_fixed(v0); normalize3_fixed(v1);
            cross3(v0, v1, n); normalize3_fixed(n);
            unsigned id3[3] = {a, b, c};
            for (int k = 0; k < 3; k++) {
                float *d = &s->normals[id3[k] * 3];
                d[0] = d[0] + n[0]; d[1] = d[1] + n[1]; d[2] = d[2] + n[2];
            }
        }
    }
    for (int i = 0; i < s->nnormals; i++) normalize3_fixed(&s->normals[i * 3]);
    free(norm);

    s->nverts = nv; s->verts = vert;

    /* loadVertices (mesh.hpp:578-644): centroid, radius, normalization_scale = 1/radius */
    {
        float c[3] = {0, 0, 0};
        for (int i = 0; i < nv; i++) { c[0] = c[0] + vert[i * 4]; c[1] = c[1] + vert[i * 4 + 1]; c[2] = c[2] + vert[i * 4 + 2]; }
        float fn = (float)(unsigned)nv;
        c[0] = c[0] / fn; c[1] = c[1] / fn; c[2] = c[2] / fn;
        float radius = 0.0f;
        for (int i = 0; i < nv; i++) {
            float d[3]; sub3(&vert[i * 4], c, d);
            /* ( vert[i].head(3) - centroid ).norm(): a dynamic-size block expression -> left-to-right sum
               (pinned by oracle/ref_probe.cpp, key head3_minus_fixed_norm) */
            float nrm = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
            radius = stdmaxf(radius, nrm);
        }
        s->centroid[0] = c[0]; s->centroid[1] = c[1]; s->centroid[2] = c[2];
        s->radius = radius;
        s->norm_scale = (float)(1.0 / (double)radius);
    }

    /* faces: createFaces (mesh.hpp:441-468), one index group per usemtl block, non-empty groups only (objimporter.hpp:262-269) */
    int nf = 0; for (int g = 0; g < ng; g++) nf += grp[g].n / 3;
    s->nfaces = nf;
    s->face_vid = malloc(sizeof(unsigned) * 3 * (size_t)(nf ? nf : 1));
    s->face_mat = malloc(sizeof(int) * (size_t)(nf ? nf : 1));
    s->face_normal = malloc(sizeof(float) * 3 * (size_t)(nf ? nf : 1));
    int fi = 0;
    for (int g = 0; g < ng; g++) {
        for (int i = 0; i + 2 < grp[g].n; i += 3) {
            unsigned a = grp[g].ids[i], b = grp[g].ids[i + 1], c = grp[g].ids[i + 2];
            s->face_vid[fi * 3] = a; s->face_vid[fi * 3 + 1] = b; s->face_vid[fi * 3 + 2] = c;
            s->face_mat[fi] = grp[g].mat;
            float v1[3], v0[3], n[3];
            sub3(&vert[c * 4], &vert[a * 4], v1); normalize3_dyn(v1);  /* .head(3) difference: dynamic-size normalized() */
            sub3(&vert[b * 4], &vert[a * 4], v0); normalize3_dyn(v0);
            cross3(v0, v1, n); normalize3_fixed(n);
            s->face_normal[fi * 3] = n[0]; s->face_normal[fi * 3 + 1] = n[1]; s->face_normal[fi * 3 + 2] = n[2];
            fi++;
        }
        free(grp[g].ids);
    }
    free(grp);

    /* EXTENSION (reference: UB, materials[-1] at flyscene.cpp:712): OBJ without usable material -> default Mtl */
    if (s->nmtls == 0) { s->mtls = malloc(sizeof(omtl)); mtl_default(&s->mtls[0]); s->nmtls = 1; }
    for (int i = 0; i < nf; i++) if (s->face_mat[i] < 0) s->face_mat[i] = 0;

    /* normalizeModelMatrix (flyscene.cpp:56, model.hpp:169-173): shape = Identity.scale(s).translate(-centroid) */
    affine_identity(s->shape); affine_identity(s->model);
    {
        float sc = s->norm_scale;
        s->shape[0] = 1.0f * sc; s->shape[5] = 1.0f * sc; s->shape[10] = 1.0f * sc;
        /* translationExt() += linearExt() * (-centroid): off-diagonal products are exact zeros */
        s->shape[3]  = 0.0f + sc * (-s->centroid[0]);
        s->shape[7]  = 0.0f + sc * (-s->centroid[1]);
        s->shape[11] = 0.0f + sc * (-s->centroid[2]);
    }
    s->wverts = malloc(sizeof(float) * 3 * (size_t)(nv ? nv : 1));
    compute_world(s);
    return s;
}

/* (model_matrix * shape_matrix) * v4, head<3>   (model.hpp:102-105; flyscene.cpp:788-790)
   Affine*Affine then 3x4 * 4-vector; rows evaluated left to right.  With the default identity model
   matrix and uniform-scale shape matrix every cross term is an exact zero, so association cannot matter. */
static void compute_world(oscene *s) {
    float ms[12];
    const float *M = s->model, *S = s->shape;
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++)
            ms[r * 4 + c] = (M[r * 4 + 0] * S[0 * 4 + c] + M[r * 4 + 1] * S[1 * 4 + c]) + M[r * 4 + 2] * S[2 * 4 + c];
        ms[r * 4 + 3] = ((M[r * 4 + 0] * S[3] + M[r * 4 + 1] * S[7]) + M[r * 4 + 2] * S[11]) + M[r * 4 + 3];
    }
    for (int i = 0; i < s->nverts; i++) {
        const float *v = &s->verts[i * 4];
        for (int r = 0; r < 3; r++)
            s->wverts[i * 3 + r] = ((ms[r * 4 + 0] * v[0] + ms[r * 4 + 1] * v[1]) + ms[r * 4 + 2] * v[2]) + ms[r * 4 + 3] * v[3];
    }
}

void orc_set_model_matrix(oscene *s, const float m[12]) {
    memcpy(s->model, m, sizeof(float) * 12);
    compute_world(s);
}

void orc_free_scene(oscene *s) {
    if (!s) return;
    for (int i = 0; i < s->nnodes; i++) free(s->nodes[i].faces);
    free(s->nodes); free(s->verts); free(s->normals); free(s->face_vid); free(s->face_mat);
    free(s->face_normal); free(s->mtls); free(s->wverts); free(s);
}

/* ------------------------------------------------------------------ */
/* BoundingBox    src/boundingBox.cpp                                  */
/* ------------------------------------------------------------------ */
int orc_box_intersect(const float vmin[3], const float vmax[3], const float origin[3], const float dest[3]) {
    /* boundingBox.cpp:48-83 */
    float dir[3]; sub3(dest, origin, dir);
    float txmin = (vmin[0] - origin[0]) / dir[0];
    float txmax = (vmax[0] - origin[0]) / dir[0];
    float tymin = (vmin[1] - origin[1]) / dir[1];
    float tymax = (vmax[1] - origin[1]) / dir[1];
    float tzmin = (vmin[2] - origin[2]) / dir[2];
    float tzmax = (vmax[2] - origin[2]) / dir[2];
    float tinx = stdminf(txmin, txmax), toutx = stdmaxf(txmin, txmax);
    float tiny = stdminf(tymin, tymax), touty = stdmaxf(tymin, tymax);
    float tinz = stdminf(tzmin, tzmax), toutz = stdmaxf(tzmin, tzmax);
    float tin = stdmaxf(stdmaxf(tinx, tiny), tinz);
    float tout = stdminf(stdminf(toutx, touty), toutz);
    if ((tin > tout) || (tout < 0)) return 0;
    return 1;
}

/* ------------------------------------------------------------------ */
/* BoxTree build   src/boxTree.cpp:11-31, 88-147, 203-456              */
/* ------------------------------------------------------------------ */
static int node_new(oscene *s, const float bmin[3], const float bmax[3], int depth) {
    if (s->nnodes == s->cap_nodes) {
        s->cap_nodes = s->cap_nodes ? s->cap_nodes * 2 : 256;
        s->nodes = realloc(s->nodes, sizeof(onode) * (size_t)s->cap_nodes);
    }
    onode *n = &s->nodes[s->nnodes];
    memset(n, 0, sizeof *n);
    memcpy(n->bmin, bmin, sizeof(float) * 3); memcpy(n->bmax, bmax, sizeof(float) * 3);
    for (int i = 0; i < 8; i++) n->child[i] = -1;
    n->depth = depth;
    return s->nnodes++;
}

static int axis_test(float p0, float p1, float rad) {
    /* common tail of the six axisTest* members (boxTree.cpp:368-456) */
    float mx = stdmaxf(p1, p0), mn = stdminf(p1, p0);
    if (mn > rad || mx < -rad) return 0;
    return 1;
}
static int axis_test_z12(float p1, float p2, float rad) {
    /* axisTestZ12 takes std::max(p1, p2) (argument order differs, boxTree.cpp:403-404) */
    float mx = stdmaxf(p1, p2), mn = stdminf(p1, p2);
    if (mn > rad || mx < -rad) return 0;
    return 1;
}

static int plane_box_overlap(const float normal[3], const float vert[3], const float maxbox[3]) {
    /* boxTree.cpp:345-366 */
    float vmin[3], vmax[3];
    for (int i = 0; i < 3; i++) {
        float v = vert[i];
        if (normal[i] > 0.0f) { vmin[i] = -maxbox[i] - v; vmax[i] = maxbox[i] - v; }
        else { vmin[i] = maxbox[i] - v; vmax[i] = -maxbox[i] - v; }
    }
    if (dot3(normal, vmin) > 0.0f) return 0;
    if (dot3(normal, vmax) >= 0.0f) return 1;
    return 0;
}

/* the six axisTest* members (boxTree.cpp:368-456): p = projection of the two given vertices on the axis, rad = box extent on it.
   X01 / X02: p = a*v.y - b*v.z, rad = fa*bh.y + fb*bh.z;  Y02 / Y1: p = -a*v.x + b*v.z, rad = fa*bh.x + fb*bh.z;
   Z12 / Z0:  p = a*v.x - b*v.y, rad = fa*bh.x + fb*bh.y  (Z12 takes std::max(p1, p2), the others std::max(p1, p0)) */
static int ax_x(float a, float b, float fa, float fb, const float v0[3], const float v1[3], const float bh[3]) {
    return axis_test(a * v0[1] - b * v0[2], a * v1[1] - b * v1[2], fa * bh[1] + fb * bh[2]);
}
static int ax_y(float a, float b, float fa, float fb, const float v0[3], const float v1[3], const float bh[3]) {
    return axis_test(-a * v0[0] + b * v0[2], -a * v1[0] + b * v1[2], fa * bh[0] + fb * bh[2]);
}
static int ax_z12(float a, float b, float fa, float fb, const float v1[3], const float v2[3], const float bh[3]) {
    return axis_test_z12(a * v1[0] - b * v1[1], a * v2[0] - b * v2[1], fa * bh[0] + fb * bh[1]);
}
static int ax_z0(float a, float b, float fa, float fb, const float v0[3], const float v1[3], const float bh[3]) {
    return axis_test(a * v0[0] - b * v0[1], a * v1[0] - b * v1[1], fa * bh[0] + fb * bh[1]);
}

static int classify_tri(const float bmin[3], const float bmax[3], const float *V[3]) {
    /* BoxTree::clasifyFace, boxTree.cpp:203-336 */
    int count = 0;
    for (int k = 0; k < 3; k++) {
        const float *v = V[k];
        if (bmin[0] <= v[0] && bmax[0] >= v[0] && bmin[1] <= v[1] && bmax[1] >= v[1] && bmin[2] <= v[2] && bmax[2] >= v[2]) count++;
    }
    if (count > 0) return 1;

    float bc[3];
    for (int k = 0; k < 3; k++) bc[k] = bmin[k] + (bmax[k] - bmin[k]) / 2.f;
    float bh[3]; sub3(bmax, bc, bh); normalize3_fixed(bh);          /* boxhalfsize NORMALISED (:236) */
    float a[3], b[3], c[3];
    sub3(V[0], bc, a); normalize3_fixed(a);                          /* vertices NORMALISED (:238-240) */
    sub3(V[1], bc, b); normalize3_fixed(b);
    sub3(V[2], bc, c); normalize3_fixed(c);
    float e0[3], e1[3], e2[3];
    sub3(b, a, e0); sub3(c, b, e1); sub3(a, c, e2);
    float fex, fey, fez;

    fex = fabsf(e0[0]); fey = fabsf(e0[1]); fez = fabsf(e0[2]);
    if (!ax_x(e0[2], e0[1], fez, fey, a, c, bh)) return 0;        /* axisTestX01(e0.z, e0.y, fez, fey, a, c) */
    if (!ax_y(e0[2], e0[0], fez, fex, a, c, bh)) return 0;        /* axisTestY02(e0.z, e0.x, fez, fex, a, c) */
    if (!ax_z12(e0[1], e0[0], fey, fex, b, c, bh)) return 0;      /* axisTestZ12(e0.y, e0.x, fey, fex, b, c) */

    fex = fabsf(e1[0]); fey = fabsf(e1[1]); fez = fabsf(e1[2]);
    if (!ax_x(e1[2], e1[1], fez, fey, a, c, bh)) return 0;        /* X01(a,c) */
    if (!ax_y(e1[2], e1[0], fez, fex, a, c, bh)) return 0;        /* Y02(a,c) */
    if (!ax_z0(e1[1], e1[0], fey, fex, a, b, bh)) return 0;       /* axisTestZ0(e1.y, e1.x, fey, fex, a, b) */

    fex = fabsf(e2[0]); fey = fabsf(e2[1]); fez = fabsf(e2[2]);
    if (!ax_x(e2[2], e2[1], fez, fey, a, b, bh)) return 0;        /* axisTestX02(e2.z, e2.y, fez, fey, a, b) */
    if (!ax_y(e2[2], e2[0], fez, fex, a, b, bh)) return 0;        /* axisTestY1(e2.z, e2.x, fez, fex, a, b) */
    if (!ax_z12(e2[1], e2[0], fey, fex, b, c, bh)) return 0;      /* axisTestZ12(e2.y, e2.x, fey, fex, b, c) */

    for (int k = 0; k < 3; k++) { /* findMinMax per axis, :302-322 */
        float mn = stdminf(stdminf(a[k], b[k]), c[k]);
        float mx = stdmaxf(stdmaxf(a[k], b[k]), c[k]);
        if (mn > bh[k] || mx < -bh[k]) return 0;
    }
    float ed1[3], ed2[3], nrm[3];
    sub3(a, b, ed1); sub3(a, c, ed2);
    cross3(ed1, ed2, nrm); normalize3_fixed(nrm);
    if (!plane_box_overlap(nrm, a, bh)) return 0;
    return 1;
}

static int classify_face(const oscene *s, const float bmin[3], const float bmax[3], int face) {
    const float *V[3];
    for (int k = 0; k < 3; k++) V[k] = &s->wverts[s->face_vid[face * 3 + k] * 3];
    return classify_tri(bmin, bmax, V);
}

/* unit-parity entry points for the reference pins (tests/test_ref_pins.py) */
int orc_classify_tri(const float bmin[3], const float bmax[3], const float tri[9]) {
    const float *V[3] = {tri, tri + 3, tri + 6};
    return classify_tri(bmin, bmax, V);
}
void orc_sat_prims(const float in[16], unsigned char dec[8], float mm[2]) {
    /* in: a b fa fb | v0 | v1 | boxhalfsize | 3 spare -- the same call pattern as oracle/ref_probe2.cpp `prim` */
    const float *v0 = in + 4, *v1 = in + 7, *bh = in + 10;
    dec[0] = (unsigned char)ax_x(in[0], in[1], in[2], in[3], v0, v1, bh);      /* axisTestX01 */
    dec[1] = (unsigned char)ax_y(in[0], in[1], in[2], in[3], v0, v1, bh);      /* axisTestY02 */
    dec[2] = (unsigned char)ax_z12(in[0], in[1], in[2], in[3], v0, v1, bh);    /* axisTestZ12 */
    dec[3] = (unsigned char)ax_z0(in[0], in[1], in[2], in[3], v0, v1, bh);     /* axisTestZ0 */
    dec[4] = (unsigned char)ax_x(in[0], in[1], in[2], in[3], v0, v1, bh);      /* axisTestX02 */
    dec[5] = (unsigned char)ax_y(in[0], in[1], in[2], in[3], v0, v1, bh);      /* axisTestY1 */
    dec[6] = (unsigned char)plane_box_overlap(v0, v1, bh);
    dec[7] = (unsigned char)plane_box_overlap(v1, v0, bh);
    mm[0] = stdminf(stdminf(in[13], in[14]), in[15]);                          /* findMinMax, boxTree.cpp:338-343 */
    mm[1] = stdmaxf(stdmaxf(in[13], in[14]), in[15]);
}

static void node_split(oscene *s, int ni, int depth) {
    /* BoxTree::split, boxTree.cpp:88-147 */
    /* guard (the reference has none and exhausts memory): the normalised-vector SAT can accept a face in all 8 octants,
       so small capacities grow the tree like 8^15 on some meshes; stop at 4M nodes and flag the tree as unusable */
    if (s->nnodes > (4 << 20)) { s->tree_capacity = -1; return; }
    s->nodes[ni].is_leaf = 0;
    float mn[3], mx[3];
    memcpy(mn, s->nodes[ni].bmin, sizeof mn); memcpy(mx, s->nodes[ni].bmax, sizeof mx);
    float dx = (mx[0] - mn[0]) / 2, dy = (mx[1] - mn[1]) / 2, dz = (mx[2] - mn[2]) / 2;
    float vx[3] = {dx, 0, 0}, vy[3] = {0, dy, 0}, vz[3] = {0, 0, dz};
    float cmin[8][3], cmax[8][3];
#define E3(out, expr) for (int k = 0; k < 3; k++) { out[k] = (expr); }
    E3(cmin[0], mn[k]);                               E3(cmax[0], ((mn[k] + vx[k]) + vy[k]) + vz[k]);
    E3(cmin[1], mn[k] + vz[k]);                       E3(cmax[1], ((mn[k] + vx[k]) + vy[k]) + 2.0f * vz[k]);
    E3(cmin[2], mn[k] + vy[k]);                       E3(cmax[2], ((mn[k] + vx[k]) + 2.0f * vy[k]) + vz[k]);
    E3(cmin[3], (mn[k] + vy[k]) + vz[k]);             E3(cmax[3], ((mn[k] + vx[k]) + 2.0f * vy[k]) + 2.0f * vz[k]);
    E3(cmin[4], mn[k] + vx[k]);                       E3(cmax[4], ((mn[k] + 2.0f * vx[k]) + vy[k]) + vz[k]);
    E3(cmin[5], (mn[k] + vx[k]) + vz[k]);             E3(cmax[5], mx[k] - vy[k]);
    E3(cmin[6], (mn[k] + vx[k]) + vy[k]);             E3(cmax[6], mx[k] - vz[k]);
    E3(cmin[7], ((mn[k] + vx[k]) + vy[k]) + vz[k]);   E3(cmax[7], mx[k]);
#undef E3
    int ch[8];
    for (int b = 0; b < 8; b++) {
        ch[b] = node_new(s, cmin[b], cmax[b], s->nodes[ni].depth + 1); /* may realloc s->nodes */
    }
    onode *par = &s->nodes[ni];
    for (int b = 0; b < 8; b++) par->child[b] = ch[b];
    par->nchildren = 8;
    for (int b = 0; b < 8; b++) {
        onode *c = &s->nodes[ch[b]];
        c->faces = malloc(sizeof(int) * (size_t)(par->nfaces ? par->nfaces : 1));
        c->nfaces = 0;
        for (int i = 0; i < par->nfaces; i++)
            if (classify_face(s, c->bmin, c->bmax, par->faces[i])) c->faces[c->nfaces++] = par->faces[i];
        c->faces = realloc(c->faces, sizeof(int) * (size_t)(c->nfaces ? c->nfaces : 1));
    }
    free(par->faces); par->faces = NULL; par->nfaces = 0;
    int cap = s->tree_capacity;
    for (int b = 0; b < 8; b++) {
        int ci = ch[b];
        if (s->nodes[ci].nfaces == 0 && s->nodes[ci].nchildren == 0) s->nodes[ci].is_empty = 1;
        if (s->nodes[ci].nfaces < cap || depth <= 0) s->nodes[ci].is_leaf = 1;
        if (s->nodes[ci].nfaces > cap && depth > 0) node_split(s, ci, depth - 1);
    }
}

void orc_build_tree(oscene *s, int capacity, int maxdepth) {
    for (int i = 0; i < s->nnodes; i++) free(s->nodes[i].faces);
    s->nnodes = 0;
    s->tree_capacity = capacity; s->tree_maxdepth = maxdepth;
    /* BoundingBox(Mesh&), boundingBox.cpp:14-43: note max starts at FLT_MIN (smallest positive normal) */
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {FLT_MIN, FLT_MIN, FLT_MIN};
    for (int f = 0; f < s->nfaces; f++)
        for (int k = 0; k < 3; k++) {
            const float *v = &s->wverts[s->face_vid[f * 3 + k] * 3];
            for (int a = 0; a < 3; a++) { mn[a] = stdminf(mn[a], v[a]); mx[a] = stdmaxf(mx[a], v[a]); }
        }
    int root = node_new(s, mn, mx, 0);
    onode *r = &s->nodes[root];
    r->nfaces = s->nfaces;
    r->faces = malloc(sizeof(int) * (size_t)(s->nfaces ? s->nfaces : 1));
    for (int i = 0; i < s->nfaces; i++) r->faces[i] = i;
    if (r->nfaces > capacity) node_split(s, root, maxdepth);
    else if (r->nfaces == 0) s->nodes[root].is_empty = 1;
    else s->nodes[root].is_leaf = 1;
}

/* ------------------------------------------------------------------ */
/* powf of phongShade (flyscene.cpp:852)                                */
/* ------------------------------------------------------------------ */
/* The reference calls libm's powf.  In glibc 2.35 (this image, and the one the reference outputs of SURVEY Appendix A were made
   with) that is the ARM "optimized routines" powf (sysdeps/ieee754/flt-32/e_powf.c, POWF_LOG2_TABLE_BITS 4, EXP2F_TABLE_BITS 5,
   no TOINT intrinsics) and on x86-64 CPUs with FMA+AVX2 the ifunc picks the build with fused multiply-adds.  Its published algorithm
   is restated here with EXPLICIT fma() at exactly the places that build contracts (read off this image's libm.so.6), so that the
   oracle no longer depends on which variant the host's ifunc picks and the device can mirror it bit for bit.
   tests/test_powf.py pins orc_powf to the host's powf exhaustively over the ranges the shipped materials use. */
static const struct { double invc, logc; } POWF_LOG2[16] = {
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2}, {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},
    {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2}, {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4}, {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5},
    {0x1p+0, 0x0p+0}, {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4}, {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3}, {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2}, {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},
    {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}};
static const double POWF_A[5] = {0x1.27616c9496e0bp-2, -0x1.71969a075c67ap-2, 0x1.ec70a6ca7baddp-2, -0x1.7154748bef6c8p-1, 0x1.71547652ab82bp+0};
static const uint64_t POWF_EXP2[32] = {
    0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51, 0x3fef72b83c7d517b, 0x3fef54873168b9aa, 0x3fef387a6e756238,
    0x3fef1e9df51fdee1, 0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d, 0x3feebfdad5362a27, 0x3feeb42b569d4f82,
    0x3feeab07dd485429, 0x3feea47eb03a5585, 0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74, 0x3feea11473eb0187, 0x3feea589994cce13, 0x3feeace5422aa0db,
    0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d, 0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069,
    0x3fef5818dcfba487, 0x3fef7c97337b9b5f, 0x3fefa4afa2a490da, 0x3fefd0765b6e4540};
static const double POWF_C[3] = {0x1.c6af84b912394p-5, 0x1.ebfce50fac4f3p-3, 0x1.62e42ff0c52d6p-1};

float orc_powf(float x, float y) {
    uint32_t ix, iy;
    memcpy(&ix, &x, 4); memcpy(&iy, &y, 4);
    /* The fast path needs a positive normal or subnormal x and a finite non-zero y: everything phongShade can pass except y == 0,
       x == 0, x == 1-with-special-y and non-finite values, which go to libm (identical in every variant: no arithmetic involved). */
    if (!(ix - 0x00800000u < 0x7f800000u - 0x00800000u) || (2u * iy - 1u >= 2u * 0x7f800000u - 1u)) {
        if (ix - 1u < 0x007fffffu && !(2u * iy - 1u >= 2u * 0x7f800000u - 1u)) {
            /* subnormal x: normalise as e_powf.c does */
            float xs = x * 0x1p23f;
            memcpy(&ix, &xs, 4);
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        } else {
            return powf(x, y);
        }
    }
    /* log2_inline */
    uint32_t tmp = ix - 0x3f330000u;
    int i = (int)((tmp >> 19) & 15u);
    uint32_t top = tmp & 0xff800000u;
    uint32_t iz = ix - top;
    int k = (int32_t)top >> 23;
    float zf; memcpy(&zf, &iz, 4);
    double z = (double)zf;
    double r = fma(z, POWF_LOG2[i].invc, -1.0);
    double y0 = POWF_LOG2[i].logc + (double)k;
    double r2 = r * r;
    double yy = fma(POWF_A[0], r, POWF_A[1]);
    double p = fma(POWF_A[2], r, POWF_A[3]);
    double r4 = r2 * r2;
    double q = fma(POWF_A[4], r, y0);
    q = fma(p, r2, q);
    double logx = fma(yy, r4, q);
    double ylogx = (double)y * logx;
    uint64_t u; memcpy(&u, &ylogx, 8);
    if (((u >> 47) & 0xffffu) >= 0x80bfu) {            /* |y * log2(x)| >= 126 */
        if (ylogx > 0x1.fffffffd1d571p+6) return powf(x, y);      /* overflow (and the rounding-mode check just below it): not reachable */
        if (ylogx > 0x1.fffffffa3aae2p+6) return powf(x, y);      /* from phongShade (x <= 1 + ulp, y > 0); libm decides */
        if (ylogx <= -150.0) return 0.0f;                          /* __math_uflowf(0): 0x1p-95f * 0x1p-95f */
        if (ylogx < -149.0) return 0x1p-149f;                      /* __math_may_uflowf(0): 0x1.4p-75f * 0x1.4p-75f rounds to the smallest subnormal */
    }
    /* exp2_inline */
    double kd = ylogx + 0x1.8p+47;
    uint64_t ki; memcpy(&ki, &kd, 8);
    kd -= 0x1.8p+47;
    double rr = ylogx - kd;
    uint64_t t = POWF_EXP2[ki & 31u];
    t += ki << 47;
    double sc; memcpy(&sc, &t, 8);
    double zz = fma(POWF_C[0], rr, POWF_C[1]);
    double rr2 = rr * rr;
    double yv = fma(POWF_C[2], rr, 1.0);
    yv = fma(zz, rr2, yv);
    yv = yv * sc;
    return (float)yv;
}

/* compares orc_powf with the host libm's powf for every float bit pattern in [lo_bits, hi_bits] (step `stride`); returns the number of
   mismatching results and stores the first one */
long orc_powf_compare(float expo, uint32_t lo_bits, uint32_t hi_bits, uint32_t stride, uint32_t *first_bad) {
    long bad = 0;
    for (uint64_t b = lo_bits; b <= hi_bits; b += stride) {
        uint32_t bb = (uint32_t)b; float x; memcpy(&x, &bb, 4);
        float a = orc_powf(x, expo), c = powf(x, expo);
        uint32_t ua, uc; memcpy(&ua, &a, 4); memcpy(&uc, &c, 4);
        if (ua != uc && !(a != a && c != c)) { if (!bad && first_bad) *first_bad = bb; bad++; }
    }
    return bad;
}

/* ------------------------------------------------------------------ */
/* per-thread scratch for BoxTree::intersect's std::set<int>           */
/* ------------------------------------------------------------------ */
typedef struct {
    int *stamp; int cur; int *list; int nlist; int *queue; int qcap;
} oscratch;

static void scratch_init(oscratch *sc, const oscene *s) {
    sc->stamp = calloc((size_t)(s->nfaces ? s->nfaces : 1), sizeof(int));
    sc->list = malloc(sizeof(int) * (size_t)(s->nfaces ? s->nfaces : 1));
    sc->qcap = s->nnodes > 0 ? s->nnodes + 8 : 8;
    sc->queue = malloc(sizeof(int) * (size_t)sc->qcap);
    sc->cur = 0; sc->nlist = 0;
}
static void scratch_free(oscratch *sc) { free(sc->stamp); free(sc->list); free(sc->queue); }

/* BoxTree::intersect (boxTree.cpp:150-173): BFS; a popped node is re-tested (:158); children tested on push (:164).
   Leaves the unique candidate faces in sc->list (unsorted; consumers are order-independent, see closest hit). */
static void tree_collect(const oscene *s, const float o[3], const float dest[3], oscratch *sc, ostats *st) {
    sc->nlist = 0; sc->cur++;
    if (sc->cur == 0x7fffffff) { memset(sc->stamp, 0, sizeof(int) * (size_t)s->nfaces); sc->cur = 1; }
    int qh = 0, qt = 0;
    sc->queue[qt++] = 0;
    while (qh < qt) {
        const onode *n = &s->nodes[sc->queue[qh++]];
        if (st) st->box_tests++;
        if (!orc_box_intersect(n->bmin, n->bmax, o, dest)) continue;
        if (n->is_leaf && !n->is_empty) {
            if (st) st->leaf_tri_refs += (uint64_t)n->nfaces;
            for (int i = 0; i < n->nfaces; i++) {
                int f = n->faces[i];
                if (sc->stamp[f] != sc->cur) { sc->stamp[f] = sc->cur; sc->list[sc->nlist++] = f; }
            }
        } else if (!n->is_empty) {
            for (int c = 0; c < n->nchildren; c++) {
                const onode *ch = &s->nodes[n->child[c]];
                if (ch->is_empty) continue;
                if (st) st->box_tests++;
                if (orc_box_intersect(ch->bmin, ch->bmax, o, dest)) sc->queue[qt++] = n->child[c];
            }
        }
    }
}

static int cmp_int(const void *a, const void *b) { int x = *(const int *)a, y = *(const int *)b; return (x > y) - (x < y); }

int orc_tree_intersect(const oscene *s, const float o[3], const float dest[3], int *out_faces, int cap, ostats *st) {
    oscratch sc; scratch_init(&sc, s);
    tree_collect(s, o, dest, &sc, st);
    qsort(sc.list, (size_t)sc.nlist, sizeof(int), cmp_int); /* std::set iteration order */
    int n = sc.nlist < cap ? sc.nlist : cap;
    memcpy(out_faces, sc.list, sizeof(int) * (size_t)n);
    int total = sc.nlist;
    scratch_free(&sc);
    return total;
}

/* the intersected non-empty leaves themselves (node indices, ascending): the same walk as tree_collect.  Pinned against the
   reference's BoxTree::intersect run on a tree whose leaves carry their own index as the only "face" (oracle/ref_probe2.cpp). */
int orc_tree_leaves(const oscene *s, const float o[3], const float dest[3], int *out_nodes, int cap) {
    int *queue = malloc(sizeof(int) * (size_t)(s->nnodes + 1));
    int qh = 0, qt = 0, n_out = 0;
    queue[qt++] = 0;
    while (qh < qt) {
        const int ni = queue[qh++];
        const onode *n = &s->nodes[ni];
        if (!orc_box_intersect(n->bmin, n->bmax, o, dest)) continue;
        if (n->is_leaf && !n->is_empty) {
            if (n->nfaces > 0) { if (n_out < cap) out_nodes[n_out] = ni; n_out++; }
        } else if (!n->is_empty) {
            for (int c = 0; c < n->nchildren; c++) {
                const onode *ch = &s->nodes[n->child[c]];
                if (ch->is_empty) continue;
                if (orc_box_intersect(ch->bmin, ch->bmax, o, dest)) queue[qt++] = n->child[c];
            }
        }
    }
    free(queue);
    qsort(out_nodes, (size_t)(n_out < cap ? n_out : cap), sizeof(int), cmp_int);
    return n_out;
}

/* ------------------------------------------------------------------ */
/* rayTriangleIntersection   src/flyscene.cpp:787-819                  */
/* ------------------------------------------------------------------ */
static float ray_triangle(const oscene *s, const float o[3], const float d[3], int face) {
    const float *A = &s->wverts[s->face_vid[face * 3 + 0] * 3];
    const float *B = &s->wverts[s->face_vid[face * 3 + 1] * 3];
    const float *C = &s->wverts[s->face_vid[face * 3 + 2] * 3];
    const float *n = &s->face_normal[face * 3];
    float dn = dot3(d, n);
    if (dn == 0) return -72;
    float t = (dot3(n, A) - dot3(o, n)) / dn;
    float P[3] = {o[0] + t * d[0], o[1] + t * d[1], o[2] + t * d[2]};
    float v0[3], v1[3], v2[3];
    sub3(C, A, v0); sub3(B, A, v1); sub3(P, A, v2);
    float d00 = dot3(v0, v0), d01 = dot3(v0, v1), d11 = dot3(v1, v1), d02 = dot3(v0, v2), d12 = dot3(v1, v2);
    float invDenom = 1 / (d00 * d11 - d01 * d01);
    float u = (d11 * d02 - d01 * d12) * invDenom;
    float v = (d00 * d12 - d01 * d02) * invDenom;
    if ((u >= 0) && (v >= 0) && (u + v < 1)) return t;
    return -72;
}
float orc_ray_triangle(const oscene *s, const float o[3], const float d[3], int face) { return ray_triangle(s, o, d, face); }

/* closest hit, flyscene.cpp:655-691.  The reference walks the std::set in ascending face id with a strict '<',
   so ties go to the lowest id; an unordered walk with (t < best || (t == best && id < best_id)) is identical. */
static int closest_hit(const oscene *s, const float o[3], const float d[3], float *t_out, oscratch *sc, ostats *st) {
    float dest[3] = {o[0] + d[0], o[1] + d[1], o[2] + d[2]};
    if (st) st->box_tests++;
    if (!orc_box_intersect(s->nodes[0].bmin, s->nodes[0].bmax, o, dest)) return -2; /* root miss */
    tree_collect(s, o, dest, sc, st);
    int best = -1; float t = FLT_MAX;
    for (int i = 0; i < sc->nlist; i++) {
        int f = sc->list[i];
        float x = ray_triangle(s, o, d, f);
        if (st) st->tri_tests++;
        if (x != -72 && x > 0.00001f) {
            if (x < t || (x == t && best >= 0 && f < best)) { t = x; best = f; }
        }
    }
    *t_out = t;
    return best;
}
int orc_closest_hit(const oscene *s, const float o[3], const float d[3], float *t_out, ostats *st) {
    oscratch sc; scratch_init(&sc, s);
    int r = closest_hit(s, o, d, t_out, &sc, st);
    scratch_free(&sc);
    return r < 0 ? -1 : r;
}

/* ------------------------------------------------------------------ */
/* lightStrikes   src/flyscene.cpp:912-954                             */
/* ------------------------------------------------------------------ */
static int light_strikes(const oscene *s, const float hit[3], const float *pts, int n, unsigned char *vis,
                         oscratch *sc, ostats *st, int is_sample) {
    int any = 0;
    for (int l = 0; l < n; l++) {
        float t = FLT_MAX;
        const float *origin = &pts[l * 3];
        float dir[3]; sub3(hit, origin, dir);
        if (st) { if (is_sample) st->rays_sample++; else st->rays_centre++; st->box_tests++; }
        if (orc_box_intersect(s->nodes[0].bmin, s->nodes[0].bmax, origin, hit)) {
            tree_collect(s, origin, hit, sc, st);
            for (int i = 0; i < sc->nlist; i++) {
                int f = sc->list[i];
                if (s->mtls[s->face_mat[f]].illum == 9) continue;
                float x = ray_triangle(s, origin, dir, f);
                if (st) st->tri_tests++;
                if (x != -72 && x < t && x > 0.00001) t = x;
            }
        }
        if (t >= 0.98) { any = 1; vis[l] = 1; } else vis[l] = 0;
    }
    return any;
}
int orc_light_strikes(const oscene *s, const float hit[3], const float *pts, int n, unsigned char *vis, ostats *st, int is_sample) {
    oscratch sc; scratch_init(&sc, s);
    int r = light_strikes(s, hit, pts, n, vis, &sc, st, is_sample);
    scratch_free(&sc);
    return r;
}

/* ------------------------------------------------------------------ */
/* createSpherePoint / createAreaLight / arealight::getPointLights     */
/* src/flyscene.cpp:956-972, arealight.hpp:15-25                       */
/* ------------------------------------------------------------------ */
int orc_light_samples(const olights *l, const float p[3], float *out) {
    if (l->mode == OLIGHT_POINT) { out[0] = p[0]; out[1] = p[1]; out[2] = p[2]; return 1; }
    if (l->mode == OLIGHT_SPHERE) {       /* pointOnSphere = Vector3f(x, y, z) / 5 + lightPoint  (flyscene.cpp:990) */
        for (int i = 0; i < l->n_offsets; i++)
            for (int k = 0; k < 3; k++) out[i * 3 + k] = l->offsets[i * 3 + k] + p[k];
        return l->n_offsets;
    }
    /* uvec = corner + lengthX*(1,0,0); vvec = corner + lengthY*(0,1,0) */
    float uvec[3] = {p[0] + l->len_x * 1.0f, p[1] + l->len_x * 0.0f, p[2] + l->len_x * 0.0f};
    float vvec[3] = {p[0] + l->len_y * 0.0f, p[1] + l->len_y * 1.0f, p[2] + l->len_y * 0.0f};
    int n = 0;
    for (int i = 0; i < l->usteps; i++)
        for (int j = 0; j < l->vsteps; j++) {
            /* ((i + 0.5) * (uvec/usteps)).x(): the double scalar is converted to float, then multiplied */
            out[n * 3 + 0] = (float)(i + 0.5) * (uvec[0] / (float)l->usteps);
            out[n * 3 + 1] = (float)(j + 0.5) * (vvec[1] / (float)l->vsteps);
            out[n * 3 + 2] = uvec[2];
            n++;
        }
    return n;
}

/* ------------------------------------------------------------------ */
/* getInterpolatedNormal   src/flyscene.cpp:864-888                    */
/* ------------------------------------------------------------------ */
void orc_interp_normal(const oscene *s, const float p[3], int face, float out[3]) {
    unsigned ia = s->face_vid[face * 3], ib = s->face_vid[face * 3 + 1], ic = s->face_vid[face * 3 + 2];
    const float *A = &s->wverts[ia * 3], *B = &s->wverts[ib * 3], *C = &s->wverts[ic * 3];
    float v0[3], v1[3], v2[3];
    sub3(B, A, v0); sub3(C, A, v1); sub3(p, A, v2);
    const float *nA = &s->normals[ia * 3], *nB = &s->normals[ib * 3], *nC = &s->normals[ic * 3]; /* mesh.getNormal(vertex_id) */
    float d00 = dot3(v0, v0), d01 = dot3(v0, v1), d11 = dot3(v1, v1), d20 = dot3(v2, v0), d21 = dot3(v2, v1);
    float denom = d00 * d11 - d01 * d01;
    float v = (d11 * d20 - d01 * d21) / denom;
    float w = (d00 * d21 - d01 * d20) / denom;
    float u = 1.0f - v - w;
    for (int k = 0; k < 3; k++) out[k] = (u * nA[k] + v * nB[k]) + w * nC[k];
}

/* ------------------------------------------------------------------ */
/* phongShade   src/flyscene.cpp:822-859                               */
/* ------------------------------------------------------------------ */
#define OMAXS 1024
static void phong(const oscene *s, const olights *L, const float origin[3], const float hit[3], int face,
                  const float *lightpts, int nl, float out[3], oscratch *sc, ostats *st) {
    const omtl *m = &s->mtls[s->face_mat[face]];
    float fin[3] = {0, 0, 0};
    float nrm[3]; orc_interp_normal(s, hit, face, nrm);
    { /* mesh.getModelMatrix() * n : Affine * Vector3f ADDS the translation (flyscene.cpp:829) */
        const float *M = s->model; float t[3];
        for (int r = 0; r < 3; r++) t[r] = ((M[r * 4] * nrm[0] + M[r * 4 + 1] * nrm[1]) + M[r * 4 + 2] * nrm[2]) + M[r * 4 + 3] * 1.0f;
        nrm[0] = t[0]; nrm[1] = t[1]; nrm[2] = t[2];
    }
    normalize3_fixed(nrm);
    if (st) st->shaded_hits++;
    static __thread float pts[OMAXS * 3];
    static __thread unsigned char vis[OMAXS];
    for (int l = 0; l < nl; l++) {
        float sum = 0; float col[3] = {0, 0, 0};
        int n = orc_light_samples(L, &lightpts[l * 3], pts);
        light_strikes(s, hit, pts, n, vis, sc, st, 1);
        for (int i = 0; i < n; i++) {
            if (!vis[i]) continue;
            sum++;
            float ld[3]; sub3(&pts[i * 3], hit, ld); normalize3_fixed(ld);
            float ldn = dot3(ld, nrm);
            float costheta = stdmaxf(0.0f, ldn);
            float two = 2 * dot3(ld, nrm);
            float rl[3] = {ld[0] - two * nrm[0], ld[1] - two * nrm[1], ld[2] - two * nrm[2]};
            normalize3_fixed(rl);
            float eh[3]; sub3(hit, origin, eh);
            float eye[3] = {-1.0f * eh[0], -1.0f * eh[1], -1.0f * eh[2]};
            normalize3_fixed(eye);
            float mr[3] = {-1.0f * rl[0], -1.0f * rl[1], -1.0f * rl[2]};
            float cosphi = stdmaxf(0.0f, dot3(eye, mr));
            float pw = orc_powf(cosphi, m->shininess);     /* = libm powf (see orc_powf) */
            for (int k = 0; k < 3; k++) {
                float diffuse = (L->color[k] * m->kd[k]) * costheta;
                float specular = (L->color[k] * m->ks[k]) * pw;
                col[k] = col[k] + (diffuse + specular);
            }
        }
        float a = sum / (float)n, b = 1.3f / (float)n;
        for (int k = 0; k < 3; k++) fin[k] = fin[k] + (col[k] * a) * b;
    }
    out[0] = fin[0]; out[1] = fin[1]; out[2] = fin[2];
}
void orc_phong(const oscene *s, const olights *l, const float origin[3], const float hit[3], int face,
               const float *lightpts, int nl, float out[3], ostats *st) {
    oscratch sc; scratch_init(&sc, s);
    phong(s, l, origin, hit, face, lightpts, nl, out, &sc, st);
    scratch_free(&sc);
}

/* fresnel   src/flyscene.cpp:890-910 */
float orc_fresnel(const float I[3], const float N[3], float ior) {
    float cosi = dot3(I, N);
    float etai = 1, etat = ior;
    if (cosi > 0) { float tmp = etai; etai = etat; etat = tmp; }
    float sint = etai / etat * sqrtf(stdmaxf(0.f, 1 - cosi * cosi));
    if (sint >= 1) return 1;
    float cost = sqrtf(stdmaxf(0.f, 1 - sint * sint));
    cosi = fabsf(cosi);
    float Rs = ((etat * cosi) - (etai * cost)) / ((etat * cosi) + (etai * cost));
    float Rp = ((etai * cosi) - (etat * cost)) / ((etai * cosi) + (etat * cost));
    return (Rs * Rs + Rp * Rp) / 2;
}

/* ------------------------------------------------------------------ */
/* traceRay   src/flyscene.cpp:651-771                                 */
/* EXTENSION max_depth: a hit at level == max_depth is     */
/* shaded as plain Phong whatever its illum; max_depth < 0: unbounded. */
/* ------------------------------------------------------------------ */
static void refracted_dir(const float d[3], const float n[3], float Ni, float out[3]) {
    /* flyscene.cpp:747-749: c1 float; pow()/sqrt() in double; result rounded to float */
    float c1 = fabsf(dot3(d, n));
    float inv = 1 / Ni;
    double p1 = (double)inv * (double)inv;       /* pow((1/Ni), 2): exact product in double */
    double p2 = (double)c1 * (double)c1;         /* pow(c1, 2) */
    float c2 = (float)sqrt(1 - p1 * (1 - p2));
    float k = inv * c1 - c2;
    for (int i = 0; i < 3; i++) out[i] = inv * d[i] + k * n[i];
}

static void trace_ray(const oscene *s, const olights *L, const float o[3], const float d[3], int level, int max_depth,
                      const float *lightpts, int nl, float out[3], oscratch *sc, ostats *st) {
    float t;
    if (st) { if (level == 0) st->rays_primary++; else st->rays_bounce++; }
    int face = closest_hit(s, o, d, &t, sc, st);
    if (face < 0) { out[0] = out[1] = out[2] = 1.f; return; }        /* BACKGROUND */
    float hit[3] = {o[0] + t * d[0], o[1] + t * d[1], o[2] + t * d[2]};
    const float *fn = &s->face_normal[face * 3];
    unsigned char vis[25];
    if (!light_strikes(s, hit, lightpts, nl, vis, sc, st, 0)) { out[0] = out[1] = out[2] = 0.f; return; } /* SHADOW */
    const omtl *m = &s->mtls[s->face_mat[face]];
    int imodel = m->illum;
    int cut = (max_depth >= 0 && level >= max_depth);
    float ph[3], ch[3];
    if (cut || imodel == 7 || !(imodel == 9 || imodel == 6 || (imodel > 2 && imodel < 7))) {
        /* plain Phong.  illum 7: Color stays (-1,-1,-1) after "1*Color + 0*child" (child finite) -> Phong (:726,:751,:758) */
        phong(s, L, o, hit, face, lightpts, nl, out, sc, st);
        return;
    }
    if (imodel == 9) {
        phong(s, L, o, hit, face, lightpts, nl, ph, sc, st);
        trace_ray(s, L, hit, d, level + 1, max_depth, lightpts, nl, ch, sc, st);
        for (int k = 0; k < 3; k++) out[k] = 0.10f * ph[k] + 0.90f * ch[k];
        return;
    }
    if (imodel == 6) {
        /* first block's traceRay result (:729) is overwritten by :754 and has no side effects: not traced here */
        float rd[3]; refracted_dir(d, fn, m->optical_density, rd);
        phong(s, L, o, hit, face, lightpts, nl, ph, sc, st);
        trace_ray(s, L, hit, rd, level + 1, max_depth, lightpts, nl, ch, sc, st);
        for (int k = 0; k < 3; k++) out[k] = 0.2f * ph[k] + 0.8f * ch[k];
        return;
    }
    /* imodel 3,4,5: mirror, child sees {hitPoint} as its only light (:734-738) */
    {
        float two = 2 * dot3(d, fn);
        float rd[3] = {d[0] - two * fn[0], d[1] - two * fn[1], d[2] - two * fn[2]};
        phong(s, L, o, hit, face, lightpts, nl, ph, sc, st);
        trace_ray(s, L, hit, rd, level + 1, max_depth, hit, 1, ch, sc, st);
        for (int k = 0; k < 3; k++) out[k] = 0.15f * ph[k] + 0.85f * ch[k];
        if (imodel == 5) {
            float fr = orc_fresnel(rd, fn, m->optical_density);
            for (int k = 0; k < 3; k++) out[k] = fr * out[k];
        }
    }
}
void orc_trace_ray(const oscene *s, const olights *l, const float o[3], const float d[3], int level, int max_depth,
                   const float *lightpts, int nl, float out[3], ostats *st) {
    oscratch sc; scratch_init(&sc, s);
    trace_ray(s, l, o, d, level, max_depth, lightpts, nl, out, &sc, st);
    scratch_free(&sc);
}

/* ------------------------------------------------------------------ */
/* camera   dependencies/tucano/tucano/camera.hpp, utils/flycamera.hpp */
/* ------------------------------------------------------------------ */
void orc_default_camera(ocamera *c, int w, int h) {
    /* flyscene.cpp:46-47: setPerspectiveMatrix(60, w/(float)h, .1, 100); setViewport(w,h)
       flycamera.hpp:76-86,166-191: view = R(I) * T(0,0,-2) -> inverse = T(0,0,2), centre (0,0,2) */
    memset(c, 0, sizeof *c);
    c->fovy = 60.0f; c->aspect = (float)w / (float)h;
    c->viewport[0] = 0.0f; c->viewport[1] = 0.0f; c->viewport[2] = (float)w; c->viewport[3] = (float)h;
    affine_identity(c->inv_view); c->inv_view[11] = 2.0f;
    c->center[0] = 0.0f; c->center[1] = 0.0f; c->center[2] = 2.0f;
}

/* Eigen 3.3.7 pieces that Flycamera::updateViewMatrix / Camera::getCenter / screenToWorld evaluate (vendored source):
   AngleAxis::toRotationMatrix (Geometry/AngleAxis.h), 3x3 inverse by cofactors (LU/InverseImpl.h:126-170), Affine inverse
   (Geometry/Transform.h: linear().inverse(), translation = -linear_inv * translation).  Pinned bit for bit by
   tests/test_ref_pins.py against the reference's Camera compiled in place (oracle/ref_probe2.cpp). */
static void eig_angle_axis(float angle, const float ax[3], float R[9]) {
    float s = sinf(angle), c = cosf(angle);
    float sa[3] = {s * ax[0], s * ax[1], s * ax[2]};
    float c1 = 1.0f - c;
    float ca[3] = {c1 * ax[0], c1 * ax[1], c1 * ax[2]};
    float tmp;
    tmp = ca[0] * ax[1]; R[0 * 3 + 1] = tmp - sa[2]; R[1 * 3 + 0] = tmp + sa[2];
    tmp = ca[0] * ax[2]; R[0 * 3 + 2] = tmp + sa[1]; R[2 * 3 + 0] = tmp - sa[1];
    tmp = ca[1] * ax[2]; R[1 * 3 + 2] = tmp - sa[0]; R[2 * 3 + 1] = tmp + sa[0];
    R[0] = ca[0] * ax[0] + c; R[4] = ca[1] * ax[1] + c; R[8] = ca[2] * ax[2] + c;
}
static void eig_mat3_vec(const float M[9], const float v[3], float o[3]) {
    for (int r = 0; r < 3; r++) o[r] = M[r * 3] * v[0] + (M[r * 3 + 1] * v[1] + M[r * 3 + 2] * v[2]);
}
static float eig_cof(const float m[9], int i, int j) {
    int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return m[i1 * 3 + j1] * m[i2 * 3 + j2] - m[i1 * 3 + j2] * m[i2 * 3 + j1];
}
static void eig_mat3_inverse(const float m[9], float inv[9]) {
    float c0 = eig_cof(m, 0, 0), c1 = eig_cof(m, 1, 0), c2 = eig_cof(m, 2, 0);
    float det = c0 * m[0] + (c1 * m[3] + c2 * m[6]);
    float invdet = 1.0f / det;
    inv[0] = c0 * invdet; inv[1] = c1 * invdet; inv[2] = c2 * invdet;
    inv[3] = eig_cof(m, 0, 1) * invdet; inv[4] = eig_cof(m, 1, 1) * invdet; inv[5] = eig_cof(m, 2, 1) * invdet;
    inv[6] = eig_cof(m, 0, 2) * invdet; inv[7] = eig_cof(m, 1, 2) * invdet; inv[8] = eig_cof(m, 2, 2) * invdet;
}

void orc_yaw_camera(ocamera *c, int w, int h, float yaw) {
    /* Flycamera::updateViewMatrix with rotation_Y_axis = yaw, rotation_X_axis = 0 (flycamera.hpp:166-191), then
       Camera::getCenter (camera.hpp:115-118) and getViewMatrix().inverse() (camera.hpp:170). */
    orc_default_camera(c, w, h);
    if (yaw == 0.0f) return;
    const float uy[3] = {0.0f, 1.0f, 0.0f}, ux[3] = {1.0f, 0.0f, 0.0f}, uz[3] = {0.0f, 0.0f, 1.0f};
    float Ry[9], R0[9], rx[3], ry[3], rz[3], tmpv[3];
    eig_angle_axis(yaw, uy, Ry);
    eig_mat3_vec(Ry, ux, rx); normalize3_fixed(rx);
    eig_mat3_vec(Ry, uz, tmpv);
    eig_angle_axis(0.0f, rx, R0);
    eig_mat3_vec(R0, tmpv, rz); normalize3_fixed(rz);
    eig_mat3_vec(R0, uy, ry); normalize3_fixed(ry);
    float R[9] = {rx[0], rx[1], rx[2], ry[0], ry[1], ry[2], rz[0], rz[1], rz[2]};   /* rotation_matrix rows; view.linear = I * I * R */
    float dt[3] = {0.0f, 0.0f, -2.0f}, t[3];
    eig_mat3_vec(R, dt, t);                                                          /* translate(default_translation) */
    float Linv[9];
    eig_mat3_inverse(R, Linv);
    for (int r = 0; r < 3; r++) {
        for (int k = 0; k < 3; k++) c->inv_view[r * 4 + k] = Linv[r * 3 + k];
        /* translation of the inverse: (-Linv) * t ; centre: Linv * (-t) -- the same values */
        float v = (-Linv[r * 3]) * t[0] + ((-Linv[r * 3 + 1]) * t[1] + (-Linv[r * 3 + 2]) * t[2]);
        c->inv_view[r * 4 + 3] = v;
        c->center[r] = Linv[r * 3] * (-t[0]) + (Linv[r * 3 + 1] * (-t[1]) + Linv[r * 3 + 2] * (-t[2]));
    }
}

void orc_screen_to_world(const ocamera *c, float i, float j, float out[3]) {
    /* camera.hpp:155-173, 263-266 */
    float n0 = (float)(2.0 * (double)(i - c->viewport[0]) / (double)c->viewport[2] - 1.0);
    float n1 = (float)(1.0 - 2.0 * (double)(j - c->viewport[1]) / (double)c->viewport[3]);
    float n2 = (float)-1.0;
    float persp = (float)((double)1.0f / tan((double)(c->fovy / 2.0f) * (M_PI / (double)180.0f)));
    float scale = (float)(1.0 / (double)persp);
    n0 = n0 * (c->aspect * scale);
    n1 = n1 * scale;
    const float *m = c->inv_view;
    for (int r = 0; r < 3; r++)
        out[r] = ((m[r * 4] * n0 + m[r * 4 + 1] * n1) + m[r * 4 + 2] * n2) + m[r * 4 + 3] * 1.0f;
}

/* std::mt19937 (the first two outputs after seeding) and libstdc++'s generate_canonical<double, 53>: what
   `std::mt19937 gen(seed); std::uniform_real_distribution<> dis(0, 1); dis(gen)` returns */
static double mt19937_canonical(uint32_t seed) {
    uint32_t x[400];
    x[0] = seed;
    for (int i = 1; i < 400; i++) x[i] = 1812433253u * (x[i - 1] ^ (x[i - 1] >> 30)) + (uint32_t)i;
    uint32_t u[2];
    for (int k = 0; k < 2; k++) {
        uint32_t y = (x[k] & 0x80000000u) | (x[k + 1] & 0x7fffffffu);
        uint32_t v = x[k + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        v ^= v >> 11; v ^= (v << 7) & 0x9d2c5680u; v ^= (v << 15) & 0xefc60000u; v ^= v >> 18;
        u[k] = v;
    }
    double sum = 0.0, tmp = 1.0;
    for (int k = 0; k < 2; k++) { sum += (double)u[k] * tmp; tmp *= 4294967296.0; }
    double ret = sum / tmp;
    if (ret >= 1.0) ret = nextafter(1.0, 0.0);
    return ret;
}

void orc_sphere_offsets(uint32_t seed, float radius, int n, float *out) {
    /* flyscene.cpp:976-993 with std::random_device replaced by seed + i */
    for (int i = 0; i < n; i++) {
        float randomno = (float)mt19937_canonical(seed + (uint32_t)i);
        float theta = (float)((double)2.0f * M_PI * (double)randomno);
        float phi = (float)acos(2.0 * (double)randomno - 1.0);
        float x = (radius * sinf(phi)) * cosf(theta);
        float y = (radius * sinf(phi)) * sinf(theta);
        float z = radius * cosf(phi);
        out[i * 3] = x / 5.0f; out[i * 3 + 1] = y / 5.0f; out[i * 3 + 2] = z / 5.0f;
    }
}

void orc_default_lights(olights *l, int area) {
    memset(l, 0, sizeof *l);
    l->nlights = 1; l->pos[0][0] = -1.0f; l->pos[0][1] = 1.0f; l->pos[0][2] = 1.0f;  /* flyscene.cpp:72 */
    l->color[0] = 1.0f; l->color[1] = 1.0f; l->color[2] = 0.0f;                      /* flyscene.cpp:68 */
    l->mode = area ? OLIGHT_AREA : OLIGHT_POINT;
    l->usteps = 5; l->vsteps = 5; l->len_x = (float)0.3; l->len_y = (float)0.15;    /* flyscene.cpp:971 */
}

/* ------------------------------------------------------------------ */
/* raytraceScene   src/flyscene.cpp:519-648                            */
/* ------------------------------------------------------------------ */
typedef struct {
    const oscene *s; const ocamera *c; const olights *l; const oparams *p;
    int row0, row1, stride; float *out_rgb; int32_t *out_hit; ostats st;
    volatile int *next_row; long npix;
} ojob;

static void render_pixel(const ojob *J, int i, int j, float rgb[3], int32_t *hitid, oscratch *sc, ostats *st) {
    const oscene *s = J->s;
    float scr[3]; orc_screen_to_world(J->c, (float)i, (float)j, scr);
    st->precull_tests++;
    if (hitid) *hitid = -1;
    if (!orc_box_intersect(s->nodes[0].bmin, s->nodes[0].bmax, J->c->center, scr)) { /* flyscene.cpp:576-581 */
        rgb[0] = rgb[1] = rgb[2] = 1.f; return;
    }
    float d[3]; sub3(scr, J->c->center, d);                                           /* flyscene.cpp:619 */
    if (hitid) { float t; ostats tmp; memset(&tmp, 0, sizeof tmp); int f = closest_hit(s, J->c->center, d, &t, sc, &tmp); *hitid = f < 0 ? -1 : f; }
    trace_ray(s, J->l, J->c->center, d, 0, J->p->max_depth, &J->l->pos[0][0], J->l->nlights, rgb, sc, st);
}

static void *render_worker(void *arg) {
    ojob *J = arg;
    oscratch sc; scratch_init(&sc, J->s);
    int W = J->p->width;
    for (;;) {
        int j = __sync_fetch_and_add(J->next_row, J->stride);
        if (j >= J->row1) break;
        for (int i = 0; i < W; i += J->stride) {
            float rgb[3]; int32_t hid;
            render_pixel(J, i, j, rgb, J->out_hit ? &hid : NULL, &sc, &J->st);
            J->npix++;
            if (J->out_rgb) {
                long idx = (long)(j - J->row0) * W + i;
                J->out_rgb[idx * 3] = rgb[0]; J->out_rgb[idx * 3 + 1] = rgb[1]; J->out_rgb[idx * 3 + 2] = rgb[2];
                if (J->out_hit) J->out_hit[idx] = hid;
            }
        }
    }
    scratch_free(&sc);
    return NULL;
}

static void stats_add(ostats *a, const ostats *b) {
    a->rays_primary += b->rays_primary; a->rays_bounce += b->rays_bounce; a->rays_centre += b->rays_centre;
    a->rays_sample += b->rays_sample; a->box_tests += b->box_tests; a->leaf_tri_refs += b->leaf_tri_refs;
    a->tri_tests += b->tri_tests; a->shaded_hits += b->shaded_hits; a->precull_tests += b->precull_tests;
}

static long run_jobs(const oscene *s, const ocamera *c, const olights *l, const oparams *p, int row0, int row1, int stride,
                     float *out_rgb, int32_t *out_hit, ostats *st) {
    int nt = p->nthreads > 0 ? p->nthreads : 1; if (nt > 256) nt = 256;
    volatile int next = row0;
    ojob *jobs = calloc((size_t)nt, sizeof(ojob));
    pthread_t *th = malloc(sizeof(pthread_t) * (size_t)nt);
    for (int t = 0; t < nt; t++) {
        jobs[t].s = s; jobs[t].c = c; jobs[t].l = l; jobs[t].p = p; jobs[t].row0 = row0; jobs[t].row1 = row1;
        jobs[t].stride = stride; jobs[t].out_rgb = out_rgb; jobs[t].out_hit = out_hit; jobs[t].next_row = &next;
        if (nt > 1) pthread_create(&th[t], NULL, render_worker, &jobs[t]);
    }
    if (nt == 1) render_worker(&jobs[0]);
    long npix = 0;
    for (int t = 0; t < nt; t++) {
        if (nt > 1) pthread_join(th[t], NULL);
        if (st) stats_add(st, &jobs[t].st);
        npix += jobs[t].npix;
    }
    free(jobs); free(th);
    return npix;
}

void orc_render(const oscene *s, const ocamera *c, const olights *l, const oparams *p,
                int row0, int row1, float *out_rgb, int32_t *out_hit, ostats *st) {
    if (st) memset(st, 0, sizeof *st);
    run_jobs(s, c, l, p, row0, row1, 1, out_rgb, out_hit, st);
}

long orc_render_subsample(const oscene *s, const ocamera *c, const olights *l, const oparams *p,
                          int stride, ostats *st, double *seconds) {
    if (st) memset(st, 0, sizeof *st);
    struct timespec a, b; clock_gettime(CLOCK_MONOTONIC, &a);
    long n = run_jobs(s, c, l, p, 0, p->height, stride < 1 ? 1 : stride, NULL, NULL, st);
    clock_gettime(CLOCK_MONOTONIC, &b);
    if (seconds) *seconds = (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
    return n;
}

/* ppmIO.hpp:130-151 */
void orc_quantise(const float *rgb, long n, int32_t *out) {
    for (long i = 0; i < n; i++) { int v = (int)(255 * rgb[i]); out[i] = v < 255 ? v : 255; }
}

int orc_write_ppm(const char *path, const float *rgb, int w, int h) {
    FILE *f = fopen(path, "w");
    if (!f) return 0;
    fprintf(f, "P3\n%d %d\n255\n", w, h);
    for (int j = 0; j < h; j++) {
        for (int i = 0; i < w; i++) {
            const float *p = &rgb[((long)j * w + i) * 3];
            int r = (int)(255 * p[0]), g = (int)(255 * p[1]), b = (int)(255 * p[2]);
            fprintf(f, "%d %d %d ", r < 255 ? r : 255, g < 255 ? g : 255, b < 255 ? b : 255);
        }
        fputc('\n', f);
    }
    fclose(f);
    return 1;
}

/* ------------------------------------------------------------------ */
/* accessors for the Python test harness (tests/oracle_lib.py)         */
/* ------------------------------------------------------------------ */
void orc_scene_counts(const oscene *s, int out[8]) {
    out[0] = s->nverts; out[1] = s->nnormals; out[2] = s->nfaces; out[3] = s->nmtls; out[4] = s->nnodes;
    out[5] = s->tree_capacity; out[6] = s->tree_maxdepth; out[7] = 0;
}
const float *orc_scene_wverts(const oscene *s) { return s->wverts; }
const float *orc_scene_normals(const oscene *s) { return s->normals; }
const float *orc_scene_face_normals(const oscene *s) { return s->face_normal; }
const unsigned *orc_scene_face_vid(const oscene *s) { return s->face_vid; }
const int *orc_scene_face_mat(const oscene *s) { return s->face_mat; }
void orc_scene_mtl(const oscene *s, int i, float out[8], int *illum) {
    const omtl *m = &s->mtls[i];
    out[0] = m->kd[0]; out[1] = m->kd[1]; out[2] = m->kd[2]; out[3] = m->ks[0]; out[4] = m->ks[1]; out[5] = m->ks[2];
    out[6] = m->shininess; out[7] = m->optical_density; *illum = m->illum;
}
/* node i: box[6], flags {is_leaf,is_empty,nchildren,nfaces,depth}, children[8]; faces copied into out_faces (cap) */
int orc_scene_node(const oscene *s, int i, float box[6], int flags[5], int children[8], int *out_faces, int cap) {
    const onode *n = &s->nodes[i];
    for (int k = 0; k < 3; k++) { box[k] = n->bmin[k]; box[3 + k] = n->bmax[k]; }
    flags[0] = n->is_leaf; flags[1] = n->is_empty; flags[2] = n->nchildren; flags[3] = n->nfaces; flags[4] = n->depth;
    for (int k = 0; k < 8; k++) children[k] = n->child[k];
    int m = n->nfaces < cap ? n->nfaces : cap;
    if (out_faces && m > 0) memcpy(out_faces, n->faces, sizeof(int) * (size_t)m);
    return n->nfaces;
}

/* ------------------------------------------------------------------ */
/* The oracle's vector conventions, exposed so tests can pin them bit-for-bit against the reference's real       */
/* Eigen 3.3.7 / Tucano headers (tests/golden/eigen_probe.json, produced by oracle/ref_probe.cpp).              */
/* out layout: see tests/test_eigen_probe.py                                                                     */
/* ------------------------------------------------------------------ */
void orc_vec_ops(const float a[3], const float b[3], const float c[3], float u, float v, float w, float sum, int steps,
                 int idx, float scale, float out[64]) {
    int k = 0;
    out[k++] = dot3(a, b);
    out[k++] = dot3(a, a);
    { float t[3]; cross3(a, b, t); memcpy(&out[k], t, 12); k += 3; }
    { float t[3] = {a[0], a[1], a[2]}; normalize3_fixed(t); memcpy(&out[k], t, 12); k += 3; }
    { float t[3]; sub3(a, b, t); normalize3_dyn(t); memcpy(&out[k], t, 12); k += 3; }             /* head3_diff_normalized */
    { float t[3]; sub3(a, c, t); out[k++] = sqrtf((t[0] * t[0] + t[1] * t[1]) + t[2] * t[2]); }     /* head3_minus_fixed_norm */
    for (int i = 0; i < 3; i++) out[k++] = 0.15f * a[i] + 0.85f * b[i];
    for (int i = 0; i < 3; i++) out[k++] = 0.10f * a[i] + 0.90f * b[i];
    for (int i = 0; i < 3; i++) out[k++] = 0.2f * a[i] + 0.8f * b[i];
    { float two = 2 * dot3(a, b); for (int i = 0; i < 3; i++) out[k++] = a[i] - two * b[i]; }     /* reflect */
    for (int i = 0; i < 3; i++) out[k++] = (u * a[i] + v * b[i]) + w * c[i];                       /* bary */
    { float n = 25.0f; float A = sum / n, B = 1.3f / n; for (int i = 0; i < 3; i++) out[k++] = c[i] + (a[i] * A) * B; }
    for (int i = 0; i < 3; i++) out[k++] = (float)(idx + 0.5) * (a[i] / (float)steps);             /* area_scale */
    { float vx[3] = {a[0], 0, 0}, vy[3] = {0, a[1], 0}, vz[3] = {0, 0, a[2]};
      for (int i = 0; i < 3; i++) out[k++] = ((b[i] + vx[i]) + vy[i]) + 2.0f * vz[i]; }            /* octant_max */
    { /* world vertex: shape = I.scale(s).translate(-c); (I * shape) * (a,1) */
      for (int i = 0; i < 3; i++) { float t = 0.0f + scale * (-c[i]); out[k++] = scale * a[i] + t; } }
    out[k++] = 0.f;
}
