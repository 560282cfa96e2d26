/* Sanitizer driver for the CPU oracle (oracle/mcpt_oracle.c): reads a scene dumped by tests/test_oracle_asan.py (triangles, materials,
 * objects, camera, optional environment map), renders a small frame, intersects and casts a few rays, and frees everything, under
 * -fsanitize=address,undefined. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/mcpt_oracle.h"

int main(int argc, char **argv) {
    if (argc != 2) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 3;
    int32_t hdr[5];
    if (fread(hdr, sizeof hdr, 1, f) != 1) return 3;
    orc_triangle *tris = malloc(sizeof *tris * (size_t)(hdr[0] ? hdr[0] : 1));
    orc_material *mats = malloc(sizeof *mats * (size_t)hdr[1]);
    orc_object *objs = malloc(sizeof *objs * (size_t)hdr[2]);
    orc_camera cam;
    float rr;
    if ((hdr[0] && fread(tris, sizeof *tris, (size_t)hdr[0], f) != (size_t)hdr[0]) || fread(mats, sizeof *mats, (size_t)hdr[1], f) != (size_t)hdr[1] ||
        fread(objs, sizeof *objs, (size_t)hdr[2], f) != (size_t)hdr[2] || fread(&cam, sizeof cam, 1, f) != 1 || fread(&rr, sizeof rr, 1, f) != 1)
        return 3;
    float *env = NULL;
    if (hdr[3] > 0) {
        env = malloc(sizeof(float) * 3 * (size_t)hdr[3] * (size_t)hdr[4]);
        if (fread(env, sizeof(float) * 3, (size_t)hdr[3] * (size_t)hdr[4], f) != (size_t)hdr[3] * (size_t)hdr[4]) return 3;
    }
    fclose(f);
    orc_scene_desc d;
    memset(&d, 0, sizeof d);
    d.n_triangles = hdr[0]; d.n_materials = hdr[1]; d.n_objects = hdr[2];
    d.triangles = tris; d.materials = mats; d.objects = objs;
    d.env_w = hdr[3]; d.env_h = hdr[4]; d.env_pixels = env;
    d.background[0] = 0.2f; d.background[1] = 0.3f; d.background[2] = 0.4f;
    orc_scene *s = NULL;
    if (orc_scene_create(&d, &s) != 0) return 4;
    orc_params p;
    memset(&p, 0, sizeof p);
    p.spp = 3; p.rr_rate = rr; p.n_dir_sample = 4; p.enable_shadow = 1; p.seed = 1; p.n_threads = 2; p.tile_size = 8; p.nranks = 1;
    float *fb = calloc((size_t)cam.width * cam.height * 3, sizeof(float));
    orc_stats st;
    if (orc_render(s, &cam, &p, fb, &st) != 0) return 5;
    double sum = 0;
    for (int i = 0; i < cam.width * cam.height * 3; ++i) sum += fb[i] == fb[i] ? fb[i] : 0;
    enum { N = 256 };
    float o[3 * N], dir[3 * N], out[N];
    double t[N];
    int32_t prim[N], ch[N];
    uint32_t pix[N], smp[N];
    for (int i = 0; i < N; ++i) {
        orc_camera_ray(&cam, 7, (uint32_t)(i % (cam.width * cam.height)), (uint32_t)i, &o[3 * i], &dir[3 * i]);
        pix[i] = (uint32_t)i; smp[i] = (uint32_t)(i * 7); ch[i] = i % 3;
    }
    dir[0] = dir[1] = dir[2] = 0.f; /* the all-zero direction of a total internal reflection */
    orc_intersect(s, N, o, dir, t, prim);
    orc_cast_rays(s, &p, N, o, dir, pix, smp, ch, out);
    int32_t *vis = malloc(sizeof(int32_t) * (size_t)cam.width * cam.height * 3);
    if (orc_primary_hits(s, &cam, 1, 3, vis) != 0) return 6; /* primary visibility (the chess geometry pin) */
    long seen = 0;
    for (int i = 0; i < cam.width * cam.height * 3; ++i) seen += vis[i] >= 0;
    free(vis);
    uint8_t *rgba = malloc((size_t)cam.width * cam.height * 4);
    orc_tonemap(fb, (int64_t)cam.width * cam.height, rgba);
    printf("samples %llu rays %llu mean %.6f hit0 %d primary hits %ld\n", (unsigned long long)st.samples, (unsigned long long)st.scene_rays, sum / (cam.width * cam.height * 3), prim[1], seen);
    orc_scene_destroy(s);
    free(tris); free(mats); free(objs); free(env); free(fb); free(rgba);
    return 0;
}
