/* TEST INFRASTRUCTURE ONLY -- see rt_oracle.h.  Plain-C restatement of the rendering hot path.
 *
 * Every function that follows the reference cites the file:line it restates (paths relative to
 * the reference checkout).  Functions marked [spec] have no counterpart in the reference; they
 * are this build's own definition of the extended renderer and are what the HIP kernels are
 * compared against.
 */
#include "rt_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

typedef struct { float x, y, z; } v3;

/* ---------------------------------------------------------------- reference leaf functions */

/* src/vec.h:136-138 dot3: x*x' + y*y' + z*z', left to right */
static inline float dot3(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

/* src/vec.h:140-166 normalize of a vec4 whose w lane is 0: the 4-lane sum
 * ((x*x+y*y)+z*z)+w*w adds +0 last, which changes nothing; true division per lane. */
static inline v3 normalize3(v3 v) {
    const float n = sqrtf(((v.x * v.x + v.y * v.y) + v.z * v.z) + 0.0f * 0.0f);
    if(n != 0.0f) {
        v.x /= n;
        v.y /= n;
        v.z /= n;
    }
    return v;
}

/* src/aemath.h:24-26 */
static inline float lerp_f(float t, float v0, float v1) { return (1.0f - t) * v0 + (t * v1); }

/* src/aemath.h:34-42 with input (-1,1), output (0,1) as called at src/software_raytracer.cpp:227-232 */
static inline float remap_unit(float value) {
    const float in0 = -1.0f, in1 = 1.0f, out0 = 0.0f, out1 = 1.0f;
    return out0 + (value - in0) * (out1 - out0) / (in1 - in0);
}

/* src/color.cpp:96-105: truncating pack, only blue is masked */
uint32_t orc_pack_argb32(float a, float r, float g, float b) {
    return ((uint32_t)(a * 255.0f) << 24) | ((uint32_t)(r * 255.0f) << 16) |
           ((uint32_t)(g * 255.0f) << 8) | ((uint32_t)(b * 255.0f) & 0xffu);
}

/* src/software_raytracer.cpp:68-91 */
void orc_setup_viewport(uint32_t width, uint32_t height, float out_viewport[2], float out_pixel[2]) {
    float w, h;
    if(width > height) {
        w = (float)width / (float)height;
        h = 1.0f;
    } else {
        w = 1.0f;
        h = (float)height / (float)width;
    }
    out_viewport[0] = w;
    out_viewport[1] = h;
    out_pixel[0] = w / (float)width;
    out_pixel[1] = h / (float)height;
}

typedef struct {
    float vw, vh, psx, psy;
    v3 cam;
    v3 bg0, bg1;
    uint32_t width, height;
} view_t;

static view_t make_view(const orc_scene *scene, uint32_t width, uint32_t height) {
    view_t v;
    float vp[2], ps[2];
    orc_setup_viewport(width, height, vp, ps);
    v.vw = vp[0]; v.vh = vp[1]; v.psx = ps[0]; v.psy = ps[1];
    v.cam = (v3){scene->camera_pos[0], scene->camera_pos[1], scene->camera_pos[2]};
    v.bg0 = (v3){scene->background0[0], scene->background0[1], scene->background0[2]};
    v.bg1 = (v3){scene->background1[0], scene->background1[1], scene->background1[2]};
    v.width = width; v.height = height;
    return v;
}

/* src/software_raytracer.cpp:218-221 + src/ray.h:18-20: point on the z=0 viewport plane for
 * sample position (px,py) in pixel units, minus the camera position, normalised.
 * uv.z = 0*0 - 0*1 = 0 in the reference, so d.z = 0 - cam.z. */
static inline v3 primary_direction(const view_t *v, float px, float py) {
    const float uvx = px * v->psx - v->vw * 0.5f;
    const float uvy = py * v->psy - v->vh * 0.5f;
    v3 d = {uvx - v->cam.x, uvy - v->cam.y, 0.0f - v->cam.z};
    return normalize3(d);
}

/* src/software_raytracer.cpp:210-215: row gradient, t = row/height (no half-pixel offset) */
static inline v3 row_background(const view_t *v, uint32_t y) {
    const float t = (float)y / (float)v->height;
    return (v3){lerp_f(t, v->bg0.x, v->bg1.x), lerp_f(t, v->bg0.y, v->bg1.y), lerp_f(t, v->bg0.z, v->bg1.z)};
}

/* ---------------------------------------------------------------- reference pixel kernel */

void orc_trace_primary(const orc_scene *scene, uint32_t width, uint32_t height,
                       uint32_t row_begin, uint32_t row_count,
                       uint32_t *argb_out, float *rgba_out) {
    const view_t v = make_view(scene, width, height);
    const orc_sphere *s = &scene->spheres[0];
    const v3 center = {s->center[0], s->center[1], s->center[2]};
    const float radius = fabsf(s->radius);          /* src/shapes.cpp:10-12 */

    for(uint32_t row = 0; row < row_count; row++) {
        const uint32_t y = row_begin + row;
        const v3 bg = row_background(&v, y);
        const uint32_t bg_packed = orc_pack_argb32(1.0f, bg.x, bg.y, bg.z);

        for(uint32_t x = 0; x < width; x++) {
            const v3 d = primary_direction(&v, (float)x + 0.5f, (float)y + 0.5f);

            /* src/shapes.cpp:14-30 */
            const v3 oc = {center.x - v.cam.x, center.y - v.cam.y, center.z - v.cam.z};
            const float a = dot3(d, d);
            const float b = -2.0f * dot3(d, oc);
            const float c = dot3(oc, oc) - radius * radius;
            const float disc = b * b - 4 * a * c;

            uint32_t packed = bg_packed;
            v3 rgb = bg;
            if(disc >= 0) {
                const float t = (-b - sqrtf(disc)) / (2.0f * a);
                const v3 p = {v.cam.x + d.x * t, v.cam.y + d.y * t, v.cam.z + d.z * t};   /* src/ray.h:25 */
                const v3 n = normalize3((v3){p.x - center.x, p.y - center.y, p.z - center.z});
                rgb = (v3){remap_unit(n.x), remap_unit(n.y), remap_unit(n.z)};
                packed = orc_pack_argb32(1.0f, rgb.x, rgb.y, rgb.z);
            }

            const size_t idx = (size_t)row * width + x;
            argb_out[idx] = packed;
            if(rgba_out) {
                rgba_out[idx * 4 + 0] = rgb.x;
                rgba_out[idx * 4 + 1] = rgb.y;
                rgba_out[idx * 4 + 2] = rgb.z;
                rgba_out[idx * 4 + 3] = 1.0f;
            }
        }
    }
}

/* ---------------------------------------------------------------- [spec] counter-based RNG
 *
 * Replaces the reference's stateful xorshift32 (src/random.cpp:18-29, dead code there) by a
 * stream that is a pure function of (seed, absolute pixel index, sample index), so the image
 * does not depend on how pixels are assigned to lanes, workgroups or GPUs.
 * Mixing function: PCG RXS-M-XS 32 (O'Neill), used as a hash and as the per-path generator.
 *
 * Key of path (pixel, sample) = pixel_term + sample_term (mod 2^32):
 *   sample_term = hash(sample + hash(seed) ^ 0x9e3779b9)   one pseudo-random word per sample
 *                                                          index, shared by all pixels;
 *   pixel_term  = hash(pixel  + hash(seed))                a pseudo-random rotation per pixel.
 * The key seeds the path's generator (material scatter).  Where the samples of a pixel sit
 * is a separate matter: orc_sample_offset below. */

uint32_t orc_pcg_hash(uint32_t v) {
    const uint32_t s = v * 747796405u + 2891336453u;
    const uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (w >> 22) ^ w;
}

uint32_t orc_sample_term(uint32_t seed, uint32_t sample_index) {
    return orc_pcg_hash(sample_index + (orc_pcg_hash(seed) ^ 0x9e3779b9u));
}

uint32_t orc_pixel_term(uint32_t seed, uint32_t pixel_index) {
    return orc_pcg_hash(pixel_index + orc_pcg_hash(seed));
}

uint32_t orc_path_key(uint32_t seed, uint32_t pixel_index, uint32_t sample_index) {
    return orc_pixel_term(seed, pixel_index) + orc_sample_term(seed, sample_index);
}

/* [spec] Sample positions inside a pixel: a grid of cells, one sample per cell, every cell's
 * sample at the same place inside its cell -- the pixel's own random offset (rx, ry), the two
 * halves of its pixel term.  Each sample is uniform over its cell, the cells tile the pixel, so
 * the mean over samples is an unbiased estimate of the pixel's box-filtered value, with the
 * variance of a stratified pattern.  Layout for spp samples: rows of nx = 2^floor(log2(spp)/2)
 * cells (32 x 32 for 1024, 8 x 8 for 64); sample s sits in row r = s / nx, cell i = s % nx; a last
 * row that is not full holds its n_r = spp - r nx cells at width 1/n_r; a row's height is its
 * share n_r / spp of the samples.
 *     jx = (i + rx) * (1 / n_r)            jy = (r nx + ry n_r) * (1 / spp)
 * in binary32, the two reciprocals rounded once on the host (exact for powers of two).  On the
 * GPU the cell index is the same in every lane, so a sample costs no hash, no integer-to-float
 * conversion and no scalar-cache read: one vector add moves the ray to the next cell. */
uint32_t orc_sample_row_length_log2(uint32_t spp) {
    uint32_t bits = 0;
    while((2u << bits) <= spp && bits < 31) bits++;        /* bits = floor(log2(spp)) */
    return bits / 2;
}

void orc_sample_offset(uint32_t spp, uint32_t sample, uint32_t pixel_term, float *jx, float *jy) {
    const uint32_t nxl = orc_sample_row_length_log2(spp);
    const uint32_t nx = 1u << nxl;
    const uint32_t r = sample >> nxl, i = sample & (nx - 1u);
    const uint32_t left = spp - (r << nxl);
    const uint32_t n_r = left < nx ? left : nx;
    const float rx = (float)(pixel_term >> 16) * (1.0f / 65536.0f);
    const float ry = (float)(pixel_term & 0xffffu) * (1.0f / 65536.0f);
    const float inv_row = 1.0f / (float)n_r;
    const float inv_spp = 1.0f / (float)spp;
    *jx = ((float)i + rx) * inv_row;
    *jy = ((float)(r << nxl) + ry * (float)n_r) * inv_spp;
}

/* The path's generator is PCG RXS-M-XS 32 proper: the state starts at the path key, every draw
 * steps the LCG and returns the permuted state (so the first draw equals orc_pcg_hash(key)). */
uint32_t orc_rng_next(uint32_t *state) {
    const uint32_t s = *state * 747796405u + 2891336453u;
    *state = s;
    const uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (w >> 22) ^ w;
}

/* ---------------------------------------------------------------- [spec] sampling helpers */

/* cos and sin of 2*pi*u for u in [0,1), from +,* only so that a GPU evaluating the same
 * expression tree gets the same bits: octant reduction, then Taylor polynomials on [0,pi/4]
 * (|error| < 4e-7). */
void orc_cossin_2pi(float u, float *out_cos, float *out_sin) {
    const float k = u * 8.0f;
    int oct = (int)k;
    if(oct > 7) oct = 7;
    const float f = k - (float)oct;
    const float g = (oct & 1) ? (1.0f - f) : f;
    const float x = g * 0.78539816339744830962f;
    const float x2 = x * x;

    float p = -1.0f / 5040.0f;
    p = p * x2 + (1.0f / 120.0f);
    p = p * x2 + (-1.0f / 6.0f);
    p = p * x2 + 1.0f;
    const float sx = x * p;

    float q = 1.0f / 40320.0f;
    q = q * x2 + (-1.0f / 720.0f);
    q = q * x2 + (1.0f / 24.0f);
    q = q * x2 + (-0.5f);
    q = q * x2 + 1.0f;
    const float cx = q;

    const int swap = ((oct + 1) >> 1) & 1;          /* octants 1,2,5,6 */
    float c = swap ? sx : cx;
    float s = swap ? cx : sx;
    if(oct >= 2 && oct <= 5) c = -c;
    if(oct >= 4) s = -s;
    *out_cos = c;
    *out_sin = s;
}

/* Uniform direction on the unit sphere from one 32-bit draw: z from the high half,
 * azimuth from the low half. */
void orc_unit_vector(uint32_t bits, float out[3]) {
    const float u1 = (float)(bits >> 16) * (1.0f / 65536.0f);
    const float u2 = (float)(bits & 0xffffu) * (1.0f / 65536.0f);
    const float z = 1.0f - 2.0f * u1;
    const float r = sqrtf(fmaxf(0.0f, 1.0f - z * z));
    float c, s;
    orc_cossin_2pi(u2, &c, &s);
    out[0] = r * c;
    out[1] = r * s;
    out[2] = z;
}

/* ---------------------------------------------------------------- [spec] extended renderer */

#define ORC_T_MIN 0.001f

static inline v3 v3_add(v3 a, v3 b) { return (v3){a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline v3 v3_sub(v3 a, v3 b) { return (v3){a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline v3 v3_scale(v3 a, float s) { return (v3){a.x * s, a.y * s, a.z * s}; }

/* One camera sample followed to termination.  Returns radiance in L and adds the number of ray
 * segments to *rays. */
static void trace_one_path(const orc_scene *scene, const view_t *v, const orc_params *p,
                           uint32_t x, uint32_t y, uint32_t sample, float L[3], uint64_t *rays) {
    const uint32_t key = orc_path_key(p->seed, y * v->width + x, sample);
    uint32_t rng = key;

    float jx = 0.5f, jy = 0.5f;
    if(p->flags & ORC_FLAG_JITTER) orc_sample_offset(p->spp, sample, orc_pixel_term(p->seed, y * v->width + x), &jx, &jy);

    v3 o = v->cam;
    v3 d = primary_direction(v, (float)x + jx, (float)y + jy);
    v3 thr = {1.0f, 1.0f, 1.0f};

    L[0] = L[1] = L[2] = 0.0f;

    for(uint32_t bounce = 0; bounce < p->max_depth; bounce++) {
        (*rays)++;

        /* closest hit over the shape list; per-sphere quadratic as src/shapes.cpp:14-23,
         * extended with a (t_min, closest) range test and the far root */
        const float a = dot3(d, d);
        const float a4 = 4 * a;
        const float a2 = 2.0f * a;
        float closest = FLT_MAX;
        int hit = -1;
        for(uint32_t i = 0; i < scene->sphere_count; i++) {
            const orc_sphere *s = &scene->spheres[i];
            const float radius = fabsf(s->radius);
            const v3 oc = {s->center[0] - o.x, s->center[1] - o.y, s->center[2] - o.z};
            const float b = -2.0f * dot3(d, oc);
            const float c = dot3(oc, oc) - radius * radius;
            const float disc = b * b - a4 * c;
            if(disc >= 0) {
                const float sq = sqrtf(disc);
                float t = (-b - sq) / a2;
                if(!(t > ORC_T_MIN && t < closest)) {
                    t = (-b + sq) / a2;
                    if(!(t > ORC_T_MIN && t < closest)) continue;
                }
                closest = t;
                hit = (int)i;
            }
        }

        if(hit < 0) {
            v3 sky;
            if(bounce == 0) {
                sky = row_background(v, y);         /* primary miss: the reference's row gradient */
            } else {
                const float t = 0.5f * (d.y + 1.0f);
                sky = (v3){lerp_f(t, v->bg0.x, v->bg1.x), lerp_f(t, v->bg0.y, v->bg1.y), lerp_f(t, v->bg0.z, v->bg1.z)};
            }
            L[0] = thr.x * sky.x; L[1] = thr.y * sky.y; L[2] = thr.z * sky.z;
            return;
        }

        const orc_sphere *s = &scene->spheres[hit];
        const v3 center = {s->center[0], s->center[1], s->center[2]};
        const v3 pt = {o.x + d.x * closest, o.y + d.y * closest, o.z + d.z * closest};   /* src/ray.h:25 */
        const v3 n_out = normalize3(v3_sub(pt, center));                                /* src/shapes.cpp:25 */

        if(s->material == ORC_MAT_NORMAL) {
            /* src/software_raytracer.cpp:227-234: surface normal as colour, path ends */
            L[0] = thr.x * remap_unit(n_out.x);
            L[1] = thr.y * remap_unit(n_out.y);
            L[2] = thr.z * remap_unit(n_out.z);
            return;
        }

        const float dn = dot3(d, n_out);
        const int front = dn < 0.0f;
        const v3 n = front ? n_out : (v3){-n_out.x, -n_out.y, -n_out.z};
        const v3 albedo = {s->albedo[0], s->albedo[1], s->albedo[2]};
        v3 nd;

        if(s->material == ORC_MAT_LAMBERT) {
            float u[3];
            orc_unit_vector(orc_rng_next(&rng), u);
            nd = (v3){n.x + u[0], n.y + u[1], n.z + u[2]};
            if(fabsf(nd.x) < 1e-8f && fabsf(nd.y) < 1e-8f && fabsf(nd.z) < 1e-8f) nd = n;
        } else if(s->material == ORC_MAT_METAL) {
            const float k = 2.0f * dot3(d, n);
            nd = (v3){d.x - k * n.x, d.y - k * n.y, d.z - k * n.z};
            if(s->param > 0.0f) {
                float u[3];
                orc_unit_vector(orc_rng_next(&rng), u);
                nd = (v3){nd.x + s->param * u[0], nd.y + s->param * u[1], nd.z + s->param * u[2]};
            }
            if(!(dot3(nd, n) > 0.0f)) return;       /* absorbed: L stays 0 */
        } else {                                     /* ORC_MAT_DIELECTRIC */
            const float ri = front ? (1.0f / s->param) : s->param;
            const v3 md = {-d.x, -d.y, -d.z};
            const float cos_t = fminf(dot3(md, n), 1.0f);
            const float sin_t = sqrtf(fmaxf(0.0f, 1.0f - cos_t * cos_t));
            float r0 = (1.0f - ri) / (1.0f + ri);
            r0 = r0 * r0;
            const float w = 1.0f - cos_t;
            const float w2 = w * w;
            const float reflectance = r0 + (1.0f - r0) * ((w2 * w2) * w);
            const float u = (float)(orc_rng_next(&rng) >> 8) * (1.0f / 16777216.0f);
            if(ri * sin_t > 1.0f || reflectance > u) {
                const float k = 2.0f * dot3(d, n);
                nd = (v3){d.x - k * n.x, d.y - k * n.y, d.z - k * n.z};
            } else {
                const v3 perp = v3_scale(v3_add(d, v3_scale(n, cos_t)), ri);
                const float par = -sqrtf(fabsf(1.0f - dot3(perp, perp)));
                nd = v3_add(perp, v3_scale(n, par));
            }
        }

        thr.x *= albedo.x; thr.y *= albedo.y; thr.z *= albedo.z;
        o = pt;
        d = normalize3(nd);
    }
    /* depth exhausted: no light */
}

static inline float clamp01(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }

static void trace_rows(const orc_scene *scene, const orc_params *p, const view_t *v,
                       uint32_t row0, uint32_t row1, uint32_t *argb_out, float *rgba_out,
                       uint64_t *rays) {
    for(uint32_t row = row0; row < row1; row++) {
        const uint32_t y = p->row_begin + row;
        for(uint32_t x = 0; x < p->width; x++) {
            /* samples accumulate in blocks of ORC_SAMPLE_BLOCK: index order inside a block, block
               sums in block order (blocks are independent, so a GPU can spread them over waves) */
            float sum[3] = {0.0f, 0.0f, 0.0f};
            for(uint32_t s0 = 0; s0 < p->spp; s0 += ORC_SAMPLE_BLOCK) {
                const uint32_t s1 = p->spp - s0 < ORC_SAMPLE_BLOCK ? p->spp : s0 + ORC_SAMPLE_BLOCK;
                float block[3] = {0.0f, 0.0f, 0.0f};
                for(uint32_t s = s0; s < s1; s++) {
                    float L[3];
                    trace_one_path(scene, v, p, x, y, s, L, rays);
                    block[0] += L[0]; block[1] += L[1]; block[2] += L[2];
                }
                sum[0] += block[0]; sum[1] += block[1]; sum[2] += block[2];
            }
            const float n = (float)p->spp;
            const float r = clamp01(sum[0] / n), g = clamp01(sum[1] / n), b = clamp01(sum[2] / n);
            const size_t idx = (size_t)row * p->width + x;
            argb_out[idx] = orc_pack_argb32(1.0f, r, g, b);
            if(rgba_out) {
                rgba_out[idx * 4 + 0] = r; rgba_out[idx * 4 + 1] = g;
                rgba_out[idx * 4 + 2] = b; rgba_out[idx * 4 + 3] = 1.0f;
            }
        }
    }
}

void orc_trace_paths(const orc_scene *scene, const orc_params *p,
                     uint32_t *argb_out, float *rgba_out, uint64_t *ray_count) {
    const view_t v = make_view(scene, p->width, p->height);
    uint64_t rays = 0;
    trace_rows(scene, p, &v, 0, p->row_count, argb_out, rgba_out, &rays);
    if(ray_count) *ray_count = rays;
}

/* ---------------------------------------------------------------- row-parallel drivers */

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

int orc_online_cpus(void) { return (int)sysconf(_SC_NPROCESSORS_ONLN); }

typedef struct {
    const orc_scene *scene;
    const orc_params *params;
    view_t view;
    uint32_t *argb;
    float *rgba;
    atomic_uint next_row;
    atomic_ullong rays;
    int primary;        /* 1: reference pixel kernel, 0: extended renderer */
} mt_job;

#define MT_ROWS_PER_GRAB 4u

static void *mt_worker(void *arg) {
    mt_job *job = (mt_job *)arg;
    const orc_params *p = job->params;
    uint64_t rays = 0;
    for(;;) {
        const uint32_t r0 = atomic_fetch_add(&job->next_row, MT_ROWS_PER_GRAB);
        if(r0 >= p->row_count) break;
        const uint32_t r1 = (r0 + MT_ROWS_PER_GRAB < p->row_count) ? r0 + MT_ROWS_PER_GRAB : p->row_count;
        if(job->primary) {
            orc_trace_primary(job->scene, p->width, p->height, p->row_begin + r0, r1 - r0,
                              job->argb + (size_t)r0 * p->width,
                              job->rgba ? job->rgba + (size_t)r0 * p->width * 4 : NULL);
        } else {
            /* trace_rows indexes outputs by band-relative row, so hand it the whole buffers */
            trace_rows(job->scene, p, &job->view, r0, r1, job->argb, job->rgba, &rays);
        }
    }
    atomic_fetch_add(&job->rays, rays);
    return NULL;
}

static double run_mt(mt_job *job, int threads) {
    if(threads < 1) threads = 1;
    pthread_t *tids = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    atomic_init(&job->next_row, 0u);
    atomic_init(&job->rays, 0ull);
    const double t0 = now_ms();
    for(int i = 1; i < threads; i++) pthread_create(&tids[i], NULL, mt_worker, job);
    mt_worker(job);
    for(int i = 1; i < threads; i++) pthread_join(tids[i], NULL);
    const double t1 = now_ms();
    free(tids);
    return t1 - t0;
}

double orc_trace_paths_mt(const orc_scene *scene, const orc_params *params, int threads,
                          uint32_t *argb_out, float *rgba_out, uint64_t *ray_count) {
    mt_job job;
    job.scene = scene; job.params = params;
    job.view = make_view(scene, params->width, params->height);
    job.argb = argb_out; job.rgba = rgba_out; job.primary = 0;
    const double ms = run_mt(&job, threads);
    if(ray_count) *ray_count = atomic_load(&job.rays);
    return ms;
}

double orc_trace_primary_mt(const orc_scene *scene, uint32_t width, uint32_t height, int threads,
                            uint32_t *argb_out) {
    orc_params p;
    memset(&p, 0, sizeof(p));
    p.width = width; p.height = height; p.row_begin = 0; p.row_count = height; p.spp = 1; p.max_depth = 1;
    mt_job job;
    job.scene = scene; job.params = &p;
    job.view = make_view(scene, width, height);
    job.argb = argb_out; job.rgba = NULL; job.primary = 1;
    return run_mt(&job, threads);
}
