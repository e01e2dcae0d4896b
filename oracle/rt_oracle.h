/* TEST INFRASTRUCTURE ONLY -- never linked into, imported by or executed from the product path.
 *
 * Plain-C CPU restatement of the rendering hot path, used as the checker in tests/, in
 * __graft_entry__.smoke() and as bench.py's cpu_baseline leg.
 *
 * Two layers:
 *  - orc_trace_primary(): the reference's software backend pixel kernel
 *    (src/software_raytracer.cpp:68-91,205-240 and the leaf functions it calls).  PINNED: it is
 *    checked byte-for-byte against images and per-pixel floats produced by the reference's own
 *    code (oracle/_ref, fixtures under tests/golden/).
 *  - orc_trace_paths(): the extended renderer (shape list with closest hit, materials, counter
 *    based RNG, samples per pixel, bounce loop).  The reference has no such code, so this layer
 *    is the build's own specification: PARITY UNPINNED beyond the case where it degenerates to
 *    the reference (default scene, 1 sample at the pixel centre), which is pinned as above.
 *
 * All arithmetic is IEEE-754 binary32 in source order, no contraction (build flags in Makefile).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_MAT_NORMAL = 0, ORC_MAT_LAMBERT = 1, ORC_MAT_METAL = 2, ORC_MAT_DIELECTRIC = 3 };
enum { ORC_FLAG_JITTER = 1u };
enum { ORC_SAMPLE_BLOCK = 64 };     /* a pixel's samples are summed in blocks of this many */

/* 48 bytes; same field order as aehip_sphere so tests can hand one numpy array to both sides */
typedef struct orc_sphere {
    float center[3];
    float radius;
    float albedo[3];
    float param;        /* METAL: fuzz; DIELECTRIC: index of refraction */
    uint32_t material;
    uint32_t reserved[3];
} orc_sphere;

typedef struct orc_scene {
    float background0[4];   /* r,g,b,(unused)  colour at image row 0      (src/raytracer.cpp:10) */
    float background1[4];   /* r,g,b,(unused)  colour towards row H       (src/raytracer.cpp:11) */
    float camera_pos[4];    /* x,y,z,(0)                                  (src/raytracer.cpp:12) */
    uint32_t sphere_count;
    uint32_t reserved;
    const orc_sphere *spheres;
} orc_scene;

typedef struct orc_params {
    uint32_t width, height;         /* full image size */
    uint32_t row_begin, row_count;  /* band of rows to render; outputs hold row_count rows */
    uint32_t spp;                   /* samples per pixel (>=1) */
    uint32_t max_depth;             /* ray segments per path (>=1) */
    uint32_t seed;
    uint32_t flags;                 /* ORC_FLAG_JITTER */
} orc_params;

/* src/software_raytracer.cpp:68-91 */
void orc_setup_viewport(uint32_t width, uint32_t height, float out_viewport[2], float out_pixel[2]);

/* src/color.cpp:96-105 */
uint32_t orc_pack_argb32(float a, float r, float g, float b);

/* Reference pixel kernel over rows [row_begin,row_begin+row_count).  Uses scene->spheres[0]
 * only, with the reference's semantics (near root, no t range test).  argb_out: row_count*width
 * words; rgba_out (nullable): 4 floats per pixel = the r,g,b handed to the pack and alpha 1. */
void orc_trace_primary(const orc_scene *scene, uint32_t width, uint32_t height,
                       uint32_t row_begin, uint32_t row_count,
                       uint32_t *argb_out, float *rgba_out);

/* Extended renderer; single thread.  ray_count (nullable) receives the number of ray segments
 * tested against the shape list. */
void orc_trace_paths(const orc_scene *scene, const orc_params *params,
                     uint32_t *argb_out, float *rgba_out, uint64_t *ray_count);

/* Row-parallel drivers (race free: each thread owns whole rows).  Return wall milliseconds. */
double orc_trace_paths_mt(const orc_scene *scene, const orc_params *params, int threads,
                          uint32_t *argb_out, float *rgba_out, uint64_t *ray_count);
double orc_trace_primary_mt(const orc_scene *scene, uint32_t width, uint32_t height, int threads,
                            uint32_t *argb_out);

/* RNG and sampling primitives, exported so tests can pin them individually. */
uint32_t orc_pcg_hash(uint32_t v);
uint32_t orc_sample_term(uint32_t seed, uint32_t sample_index);
uint32_t orc_pixel_term(uint32_t seed, uint32_t pixel_index);
uint32_t orc_path_key(uint32_t seed, uint32_t pixel_index, uint32_t sample_index);
uint32_t orc_rng_next(uint32_t *state);
uint32_t orc_sample_row_length_log2(uint32_t spp);
void orc_sample_offset(uint32_t spp, uint32_t sample, uint32_t pixel_term, float *jx, float *jy);
void orc_cossin_2pi(float u, float *out_cos, float *out_sin);
void orc_unit_vector(uint32_t bits, float out[3]);

int orc_online_cpus(void);

#ifdef __cplusplus
}
#endif
#endif
