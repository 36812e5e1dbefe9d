/*
 * rt_oracle_cli.c -- CPU ORACLE command line (test infrastructure, NOT the product).
 * Renders a frame with the C restatement and writes the reference's ASCII P3 result.ppm format.
 * usage: rt_oracle_cli <scene.obj> <W> <H> <area 0|1> <usteps> <max_depth|-1> <threads> <out.ppm>
 */
#include "rt_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

int main(int argc, char **argv) {
    if (argc < 9) { fprintf(stderr, "usage: %s scene.obj W H area usteps max_depth threads out.ppm\n", argv[0]); return 2; }
    int W = atoi(argv[2]), H = atoi(argv[3]), area = atoi(argv[4]), us = atoi(argv[5]), md = atoi(argv[6]), nt = atoi(argv[7]);
    oscene *s = orc_load_obj(argv[1]);
    if (!s) return 1;
    struct timespec a, b; clock_gettime(CLOCK_MONOTONIC, &a);
    orc_build_tree(s, 1000, 15);
    clock_gettime(CLOCK_MONOTONIC, &b);
    int leaves = 0, refs = 0, maxleaf = 0, maxd = 0;
    for (int i = 0; i < s->nnodes; i++) {
        if (s->nodes[i].is_leaf && !s->nodes[i].is_empty) { leaves++; refs += s->nodes[i].nfaces; if (s->nodes[i].nfaces > maxleaf) maxleaf = s->nodes[i].nfaces; }
        if (s->nodes[i].depth > maxd) maxd = s->nodes[i].depth;
    }
    fprintf(stderr, "scene: %d verts %d faces %d mtls; tree: %d nodes %d leaves %d refs maxleaf %d depth %d (%.3f s)\n",
            s->nverts, s->nfaces, s->nmtls, s->nnodes, leaves, refs, maxleaf, maxd,
            (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec));
    fprintf(stderr, "root box: min %.9g %.9g %.9g max %.9g %.9g %.9g\n", s->nodes[0].bmin[0], s->nodes[0].bmin[1], s->nodes[0].bmin[2],
            s->nodes[0].bmax[0], s->nodes[0].bmax[1], s->nodes[0].bmax[2]);
    ocamera c; orc_default_camera(&c, W, H);
    olights l; orc_default_lights(&l, area); l.usteps = l.vsteps = us;
    oparams p = {W, H, md, nt};
    float *rgb = malloc(sizeof(float) * 3 * (size_t)W * (size_t)H);
    ostats st;
    clock_gettime(CLOCK_MONOTONIC, &a);
    orc_render(s, &c, &l, &p, 0, H, rgb, NULL, &st);
    clock_gettime(CLOCK_MONOTONIC, &b);
    double sec = (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
    unsigned long long rays = st.rays_primary + st.rays_bounce + st.rays_centre + st.rays_sample + (st.precull_tests - st.rays_primary);
    fprintf(stderr, "render %.3f s; rays: primary %llu bounce %llu centre %llu sample %llu (total incl. culled pixels %llu, %.3f Mrays/s); box %llu refs %llu tri %llu shaded %llu\n",
            sec, (unsigned long long)st.rays_primary, (unsigned long long)st.rays_bounce, (unsigned long long)st.rays_centre,
            (unsigned long long)st.rays_sample, rays, rays / sec * 1e-6, (unsigned long long)st.box_tests,
            (unsigned long long)st.leaf_tri_refs, (unsigned long long)st.tri_tests, (unsigned long long)st.shaded_hits);
    if (strcmp(argv[8], "-") != 0) orc_write_ppm(argv[8], rgb, W, H);
    free(rgb); orc_free_scene(s);
    return 0;
}
