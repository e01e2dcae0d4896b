// hip_raytracer_kernels.hip -- gfx950 kernels and the C ABI of include/hip_raytracer.h.
//
// Built with -ffp-contract=off: the compiler never fuses a multiply into an add on its own.
// "Exact" code paths therefore evaluate IEEE binary32 expressions in the order written, with
// correctly rounded division and square root, and reproduce the reference's CPU results bit for
// bit; "fast" code paths ask for fusion and hardware approximations explicitly
// (__builtin_fmaf, __builtin_amdgcn_rcpf / rsqf / sqrtf).
//
// Reference functions restated here (paths relative to the reference checkout):
//   viewport set-up            src/software_raytracer.cpp:68-91
//   primary ray                src/software_raytracer.cpp:218-221, src/ray.h:18-20
//   normalise                  src/vec.h:140-166
//   ray-sphere intersection    src/shapes.cpp:14-30
//   normal -> colour           src/aemath.h:34-42, src/software_raytracer.cpp:227-232
//   row gradient background    src/aemath.h:24-26, src/software_raytracer.cpp:210-215
//   colour -> 0xAARRGGBB       src/color.cpp:96-105
//   framebuffer layout         src/software_raytracer.cpp:94-108
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>       // types and prototypes only: librccl is opened with dlopen on first use

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>
#include <vector>

#include "hip_raytracer.h"
#include "rt_exact_math.h"

namespace {

// ------------------------------------------------------------------------------------------
// launch constants
// ------------------------------------------------------------------------------------------

constexpr int kTileW = 16;               // path pass: a 256-thread workgroup covers 16x16 pixels,
constexpr int kTileH = 16;               // each wave64 a 16x4 strip (coherent rays, 64-B row segments)
constexpr int kBlock = kTileW * kTileH;
constexpr int kLdsSpheres = 2048;        // shape-list chunk staged in LDS: 2048 x 16 B = 32 KiB
constexpr int kPrimaryBlock = 512;       // primary pass: 512 threads x 4 pixels each
constexpr float kTMin = 0.001f;
constexpr int kTraceParts = 5;           // aehip_backend_trace renders a large band in this many parts so that
                                         // the copy of one part to the host overlaps the render of the next;
constexpr uint32_t kTracePartEnd[kTraceParts] = {6, 14, 22, 28, 32};   // part k ends at this many 32nds of the band:
                                         // shrinking parts, because only the last part's copies are not hidden
constexpr uint32_t kFlagFoldScreen = 1u << 31;   // internal: screening rays may use folded per-pixel terms
// Shape lists at least this long use the streamed path pass.  Which pass wins depends on path
// lengths more than on the list: with long mixed-material paths the streamed pass is ahead from
// 16 spheres (1080p 16 spp: 0.98 ms per-lane / 0.73 streamed; 64 spheres 2.30 / 1.90), with short
// paths over 40 small spheres the per-lane pass is (0.45 / 0.73), as on a 5-sphere scene (1.9 / 2.6).
constexpr uint32_t kPoolMinSpheres = 48;

struct view_consts {
    float vw_half, vh_half;     // viewport_size * 0.5 (exact: power-of-two scale)
    float psx, psy;             // pixel size on the viewport plane
    float cam[3];
    float bg0[3];
    float bg1[3];
    uint32_t width, height;
    uint32_t row_begin, row_count;
    uint32_t tile_rows, tile_stride;    // 0, 0: contiguous band; else row tiles dealt with a stride
    uint32_t spp, max_depth;
    uint32_t seed_hash;         // pcg_hash(seed), hoisted to the host
    uint32_t flags;
    uint32_t n_spheres;
    float list_centre[3];       // origin of the coordinates the list prefilter works in (see prefilter_entry)
    uint32_t row_bits;          // a row of the pixel's sample grid has 2^row_bits cells (see sample_offset)
    float inv_spp;              // 1.0f / spp, rounded once on the host
};

struct f3 {
    float x, y, z;
};

// band row -> image row (aehip_params.tile_rows / tile_stride)
__host__ __device__ __forceinline__ uint32_t image_row(const view_consts &vc, uint32_t row) {
    if(vc.tile_rows == 0) return vc.row_begin + row;
    const uint32_t tile = row / vc.tile_rows;
    return vc.row_begin + tile * vc.tile_stride + (row - tile * vc.tile_rows);
}

// Rows of workgroups in the order a camera launch takes them, given the rows [lo, hi] in which the
// sphere can be hit (packed lo | hi << 16; hi < lo: plain order): lo, hi, lo + 1, hi - 1, ... inwards,
// then the rows outside.  The rows at the sphere's upper and lower edge hold its slowest strips
// (pixels of several classes, which walk their samples once per class) and the rows outside only
// background: the longest waves start with the launch and the shortest end it.  (Timeline of a 4K
// frame, tools/wave_timeline.py: taken top to bottom or centre outwards, the launch's last tenth
// was a few dozen waves of the edge rows on an otherwise empty chip.)
__device__ __forceinline__ uint32_t edges_first_row(uint32_t i, uint32_t packed) {
    const uint32_t lo = packed & 0xffffu, hi = packed >> 16;
    if(hi < lo) return i;
    const uint32_t inside = hi - lo + 1u;
    if(i < inside) return (i & 1u) ? hi - (i >> 1) : lo + (i >> 1);
    const uint32_t j = i - inside;
    return j < lo ? j : hi + 1u + (j - lo);
}

// ------------------------------------------------------------------------------------------
// arithmetic policy
// ------------------------------------------------------------------------------------------

template <bool EXACT>
__device__ __forceinline__ float rt_fma(float a, float b, float c) {
    if constexpr(EXACT) return a * b + c;            // two roundings (contraction is off)
    else return __builtin_fmaf(a, b, c);
}

template <bool EXACT>
__device__ __forceinline__ float rt_div(float a, float b) {
    if constexpr(EXACT) return a / b;                // correctly rounded (v_div_scale/fmas/fixup)
    else return a * __builtin_amdgcn_rcpf(b);
}

template <bool EXACT>
__device__ __forceinline__ float rt_sqrt(float a) {
    if constexpr(EXACT) {
        // hipcc's correctly rounded sequence.  (The shorter rt_exact::sqrt_tame needs a range test
        // and a branch here; measured on the camera pass that costs more than it saves: 8.6 vs 8.0 ms.)
        return __builtin_sqrtf(a);
    } else {
        return __builtin_amdgcn_sqrtf(a);
    }
}

template <bool EXACT>
__device__ __forceinline__ float dot3(f3 a, f3 b) {
    if constexpr(EXACT) return (a.x * b.x + a.y * b.y) + a.z * b.z;
    else return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x));
}

// src/vec.h:155-160: divide each lane by the length unless the length is zero
template <bool EXACT>
__device__ __forceinline__ f3 normalize3(f3 v) {
    if constexpr(EXACT) {
        const float m = (v.x * v.x + v.y * v.y) + v.z * v.z;
        if(rt_exact::tame3_bounded(v.x, v.y, v.z)) {     // |components| <= 2^40 by the scene bounds checked at setup
            // same bits as the plain operators below (rt_exact_math.h), one reciprocal shared
            // by the three divisions
            const float n = rt_exact::sqrt_tame(m);
            const float r = rt_exact::refined_rcp(n);
            return f3{rt_exact::div_with_rcp(v.x, n, r), rt_exact::div_with_rcp(v.y, n, r),
                      rt_exact::div_with_rcp(v.z, n, r)};
        }
        const float n = __builtin_sqrtf(m);
        if(n != 0.0f) {
            v.x = v.x / n;
            v.y = v.y / n;
            v.z = v.z / n;
        }
        return v;
    } else {
        const float m = dot3<false>(v, v);
        const float inv = m > 0.0f ? __builtin_amdgcn_rsqf(m) : 0.0f;
        return f3{v.x * inv, v.y * inv, v.z * inv};
    }
}

// src/aemath.h:24-26
template <bool EXACT>
__device__ __forceinline__ float lerp1(float t, float v0, float v1) {
    if constexpr(EXACT) return (1.0f - t) * v0 + (t * v1);
    else return __builtin_fmaf(1.0f - t, v0, t * v1);
}

// src/aemath.h:34-42 with ranges (-1,1) -> (0,1): 0 + (v + 1) * 1 / 2; the *1, /2 and 0+ are
// value-preserving, so (v + 1) * 0.5 is the same float
__device__ __forceinline__ float remap_unit(float v) { return (v + 1.0f) * 0.5f; }

// src/color.cpp:96-105: truncate towards zero, alpha is the constant 1.0f * 255.0f
__device__ __forceinline__ uint32_t pack_argb32(float r, float g, float b) {
    return (255u << 24) | (static_cast<uint32_t>(r * 255.0f) << 16) |
           (static_cast<uint32_t>(g * 255.0f) << 8) | (static_cast<uint32_t>(b * 255.0f) & 0xffu);
}

__device__ __forceinline__ float clamp01(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }

// src/software_raytracer.cpp:218-221 + src/ray.h:18-20
template <bool EXACT>
__device__ __forceinline__ f3 primary_direction(const view_consts &vc, float px, float py) {
    const float uvx = rt_fma<EXACT>(px, vc.psx, -vc.vw_half);
    const float uvy = rt_fma<EXACT>(py, vc.psy, -vc.vh_half);
    const f3 d = {uvx - vc.cam[0], uvy - vc.cam[1], 0.0f - vc.cam[2]};
    return normalize3<EXACT>(d);
}

// src/software_raytracer.cpp:210-215
template <bool EXACT>
__device__ __forceinline__ f3 row_background(const view_consts &vc, uint32_t y) {
    const float t = rt_div<EXACT>(static_cast<float>(y), static_cast<float>(vc.height));
    return f3{lerp1<EXACT>(t, vc.bg0[0], vc.bg1[0]), lerp1<EXACT>(t, vc.bg0[1], vc.bg1[1]),
              lerp1<EXACT>(t, vc.bg0[2], vc.bg1[2])};
}

// ------------------------------------------------------------------------------------------
// counter-based per-path random stream (replaces the stateful xorshift of src/random.cpp:18-29)
// ------------------------------------------------------------------------------------------

__host__ __device__ __forceinline__ uint32_t pcg_permute(uint32_t s) {
    const uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (w >> 22) ^ w;
}
__host__ __device__ __forceinline__ uint32_t pcg_hash(uint32_t v) { return pcg_permute(v * 747796405u + 2891336453u); }

// Key of path (pixel, sample) = pixel_term + sample_term: a pseudo-random rotation per pixel plus
// one pseudo-random word per sample index shared by all pixels (a Cranley-Patterson rotation of
// one sample pattern).  The sample terms are tabulated by the host (sample_terms[], read through
// the scalar cache), so a new sample costs the vector unit one integer add.
__host__ __device__ __forceinline__ uint32_t pixel_term(uint32_t seed_hash, uint32_t pixel) { return pcg_hash(pixel + seed_hash); }
__host__ __device__ __forceinline__ uint32_t sample_term(uint32_t seed_hash, uint32_t sample) {
    return pcg_hash(sample + (seed_hash ^ 0x9e3779b9u));
}
__device__ __forceinline__ uint32_t rng_next(uint32_t &state) {
    state = state * 747796405u + 2891336453u;
    return pcg_permute(state);
}

// Where the samples of a pixel sit: one per cell of a grid over the pixel, every cell's sample at
// the same place inside its cell -- the pixel's own offset (rx, ry), the halves of its pixel
// term.  Each sample is uniform over its cell and the cells tile the pixel, so the mean is an
// unbiased estimate of the pixel's box-filtered value with a stratified pattern's variance.
// Rows of nx = 2^floor(log2(spp) / 2) cells (32 x 32 for 1024 samples); sample s is cell i = s % nx
// of row r = s / nx; a last row that is not full holds its n_r cells at width 1 / n_r, and a row's
// height is its share n_r / spp of the samples:
//     jx = (i + rx) * (1 / n_r)            jy = (r nx + ry n_r) * (1 / spp)
// The cell index is the same in every lane of a wave that walks a pixel block's samples together,
// which is what the camera pass builds on: no hash, conversion or table read per sample.
struct pixel_offset {
    float rx, ry;
};
__device__ __forceinline__ pixel_offset offset_of_pixel(uint32_t pix_term) {
    return pixel_offset{static_cast<float>(pix_term >> 16) * (1.0f / 65536.0f), static_cast<float>(pix_term & 0xffffu) * (1.0f / 65536.0f)};
}
__device__ __forceinline__ uint32_t cells_in_row(const view_consts &vc, uint32_t row) {
    return min(1u << vc.row_bits, vc.spp - (row << vc.row_bits));
}
// Every block of AEHIP_SAMPLE_BLOCK samples is an even number of full rows of cells: the listed
// pixels of the fast build's camera pass then take 32 lanes a pixel row, a wave's two halves
// walking two rows of cells side by side (camera_body).
__host__ __device__ __forceinline__ bool blocks_are_pairs_of_rows(const view_consts &vc) {
    return (vc.flags & AEHIP_FLAG_JITTER) != 0 && vc.spp % AEHIP_SAMPLE_BLOCK == 0u && (2u << vc.row_bits) <= AEHIP_SAMPLE_BLOCK;
}
__device__ __forceinline__ void sample_offset(const view_consts &vc, uint32_t sample, uint32_t pix_term, float &jx, float &jy) {
    const uint32_t r = sample >> vc.row_bits, i = sample & ((1u << vc.row_bits) - 1u);
    const uint32_t n_r = cells_in_row(vc, r);
    const pixel_offset o = offset_of_pixel(pix_term);
    const float inv_row = 1.0f / static_cast<float>(n_r);          // correctly rounded (exact for a full row)
    jx = (static_cast<float>(i) + o.rx) * inv_row;
    jy = (static_cast<float>(r << vc.row_bits) + o.ry * static_cast<float>(n_r)) * vc.inv_spp;
}

// cos, sin of 2*pi*u, u in [0,1): octant reduction + Taylor polynomials on [0, pi/4], built from
// + and * only so the exact build reproduces the CPU checker's bits
template <bool EXACT>
__device__ __forceinline__ void cossin_2pi(float u, float &c_out, float &s_out) {
    const float k = u * 8.0f;
    int oct = static_cast<int>(k);
    oct = oct > 7 ? 7 : oct;
    const float f = k - static_cast<float>(oct);
    const float g = (oct & 1) ? (1.0f - f) : f;
    const float x = g * 0.78539816339744830962f;
    const float x2 = x * x;

    float p = -1.0f / 5040.0f;
    p = rt_fma<EXACT>(p, x2, 1.0f / 120.0f);
    p = rt_fma<EXACT>(p, x2, -1.0f / 6.0f);
    p = rt_fma<EXACT>(p, x2, 1.0f);
    const float sx = x * p;

    float q = 1.0f / 40320.0f;
    q = rt_fma<EXACT>(q, x2, -1.0f / 720.0f);
    q = rt_fma<EXACT>(q, x2, 1.0f / 24.0f);
    q = rt_fma<EXACT>(q, x2, -0.5f);
    q = rt_fma<EXACT>(q, x2, 1.0f);
    const float cx = q;

    const bool swap = ((oct + 1) >> 1) & 1;
    float c = swap ? sx : cx;
    float s = swap ? cx : sx;
    if(oct >= 2 && oct <= 5) c = -c;
    if(oct >= 4) s = -s;
    c_out = c;
    s_out = s;
}

template <bool EXACT>
__device__ __forceinline__ f3 unit_vector(uint32_t bits) {
    const float u1 = static_cast<float>(bits >> 16) * (1.0f / 65536.0f);
    const float u2 = static_cast<float>(bits & 0xffffu) * (1.0f / 65536.0f);
    const float z = 1.0f - 2.0f * u1;
    const float m = 1.0f - z * z;
    const float r = rt_sqrt<EXACT>(m > 0.0f ? m : 0.0f);
    float c, s;
    cossin_2pi<EXACT>(u2, c, s);
    return f3{r * c, r * s, z};
}

// ------------------------------------------------------------------------------------------
// constants of sphere 0 as seen from the camera (primary pass and single-sphere camera pass)
// ------------------------------------------------------------------------------------------

struct camera_consts {       // travels as a kernel argument (SGPRs)
    float oc[3];             // centre - camera (src/shapes.cpp:15)
    float c;                 // oc.oc - r*r     (src/shapes.cpp:18)
    float center[3];
    float magnitude;         // oc.oc + r*r: size of the terms behind c, for the screening margin
    // The hit discriminant of an unnormalised ray (u, v, dz) from the camera, h*h - dd*c with
    // h = (u, v, dz).oc and dd = u*u + v*v + dz*dz, written out as a quadratic in the ray's
    // viewport offsets: q_uu u*u + q_uv u*v + q_vv v*v + q_u u + q_v v + q_0
    float q_uu, q_uv, q_vv, q_u, q_v, q_0;
    // fast build shading of a hit on sphere 0: colour = draw * (shade_scale * t) + shade_offset
    float shade_scale;       // 0.5 / r
    float shade_offset[3];   // 0.5 - 0.5 * oc / r
    uint32_t near_root_only; // camera outside the sphere by more than t_min: a hit is the near root
    uint32_t short_exact_ok; // exact build: pixels all of whose samples hit may take the loop without tests (camera_body)
    float radius2;           // r*r
};

// Relative margin of the exact build's conservative miss tests (see screen_sphere below).
constexpr float kScreenMargin = 3e-5f;

// Where the fast build hands a camera ray to the exact arithmetic.  The reference's quadratic
// (src/shapes.cpp:15-23) forms its discriminant by cancellation, and the hit distance, hence the
// normal and the colour, moves by the rounding of that discriminant times M / (2 r sqrt(dd disc)),
// with disc = h*h - dd c and M = h*h + dd (oc.oc + r*r) for the unnormalised ray (h = draw.oc,
// dd = draw.draw).  With q = disc dd r*r / (M*M), any arithmetic other than the reference's own lands
// K / sqrt(q) away from the reference's colour; measured on MI355X over the reference's scene
// at 512x512 .. 3840x2160, off-axis cameras, far, small and grazing spheres: K <= 2.6e-8 for the
// primary pass, <= 4.9e-8 for the sampling pass (tools/fast_error_map.py).  Rays with q below this
// threshold, and rays within the screening margin of tangent, are computed in the reference's
// operation order instead: K / sqrt(kExactBelowQ) = 6.3e-6, inside the 1e-5 the jitter-free
// primary-hit comparison allows.  On the reference's view that is a ring about 0.9 % of the sphere's
// radius wide along the silhouette, 2.2 % of the pixels that hit; seen from far away
// (distance^2 > 65 r*r or so) every hit has q below the threshold.
constexpr float kExactBelowQ = 6e-5f;

// ------------------------------------------------------------------------------------------
// Rows of one colour around a box of workgroups.  Workgroup f of the fill takes segment
// f % kFillSegments of row f / kFillSegments of workgroups (rows_per_group band rows each), less
// the box's columns (first | count << 16, in tiles of kTileW pixels) where the row is one of the
// box's (lo | hi << 16); a wave takes every waves-th image row of it and writes the row's one
// colour over the span, 16 bytes a lane where the frame's width and address allow.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kFillSegments = 4;
template <typename RowColour>
__device__ __forceinline__ void fill_around_box(const view_consts &vc, uint32_t f, uint32_t rows_per_group, uint32_t box_rows,
                                                uint32_t box_cols, uint32_t *__restrict__ argb, float4 *__restrict__ rgba,
                                                RowColour row_colour) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, waves = blockDim.x >> 6;
    const uint32_t r = f / kFillSegments, j = f - r * kFillSegments;
    const uint32_t groups_x = (vc.width + kTileW - 1u) / kTileW;
    const uint32_t segment = ((groups_x + kFillSegments - 1u) / kFillSegments) * kTileW;
    const uint32_t x0 = min(j * segment, vc.width), x1 = min(x0 + segment, vc.width);
    // [x0, left_end) and [right_begin, x1): the segment without the box's columns
    uint32_t left_end = x1, right_begin = x1;
    if(r >= (box_rows & 0xffffu) && r <= (box_rows >> 16)) {
        const uint32_t c0 = (box_cols & 0xffffu) * kTileW, c1 = min(((box_cols & 0xffffu) + (box_cols >> 16)) * kTileW, vc.width);
        left_end = min(x1, max(x0, c0));
        right_begin = max(x0, min(x1, c1));
    }
    const bool wide = (vc.width & 3u) == 0u && (reinterpret_cast<uintptr_t>(argb) & 15u) == 0u;
    for(uint32_t k = wave; k < rows_per_group; k += waves) {
        const uint32_t row = r * rows_per_group + k;
        if(row >= vc.row_count) break;
        const float4 colour = row_colour(image_row(vc, row));
        const uint32_t word = pack_argb32(colour.x, colour.y, colour.z);
        const size_t base = static_cast<size_t>(row) * vc.width;
        for(int side = 0; side < 2; side++) {
            const uint32_t xa = side == 0 ? x0 : right_begin, xb = side == 0 ? left_end : x1;
            if(wide) {      // (the spans begin and end at multiples of the tile width or at the frame's edge)
                uint4 *line = reinterpret_cast<uint4 *>(argb + base);
                for(uint32_t q = (xa >> 2) + lane; q < (xb >> 2); q += 64u) line[q] = make_uint4(word, word, word, word);
            } else {
                for(uint32_t x = xa + lane; x < xb; x += 64u) argb[base + x] = word;
            }
            if(rgba) {
                for(uint32_t x = xa + lane; x < xb; x += 64u) rgba[base + x] = colour;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// primary pass: the reference's pixel kernel, 4 horizontally adjacent pixels per lane so the
// packed image leaves as one 16-byte store per lane (1 KiB per wave instruction)
// ------------------------------------------------------------------------------------------

// the reference's pixel, operation for operation: src/ray.h:18-20, src/shapes.cpp:14-30,
// src/software_raytracer.cpp:226-236
__device__ __forceinline__ f3 primary_pixel_reference(const f3 cam, const f3 oc, const f3 ctr, float c, const f3 draw, const f3 bg) {
    const f3 d = normalize3<true>(draw);                        // src/ray.h:18-20
    // src/shapes.cpp:14-20
    const float a = dot3<true>(d, d);
    const float b = -2.0f * dot3<true>(d, oc);
    const float disc = b * b - (4.0f * a) * c;
    if(!(disc >= 0.0f)) return bg;
    // src/shapes.cpp:23-25 : nearest root, no range test.  Square root and division through the
    // short sequences of rt_exact_math.h where the operands allow (same bits), else the compiler's.
    const float a2 = 2.0f * a;
    float t;
    if(disc >= rt_exact::kSqrtTameLo && disc <= rt_exact::kTameHi) {
        const float num = -b - rt_exact::sqrt_tame(disc);
        t = (rt_exact::tame(num) && rt_exact::tame(a2)) ? rt_exact::div_tame(num, a2) : num / a2;
    } else {
        t = (-b - rt_sqrt<true>(disc)) / a2;
    }
    const f3 p = {d.x * t + cam.x, d.y * t + cam.y, d.z * t + cam.z};
    const f3 n = normalize3<true>(f3{p.x - ctr.x, p.y - ctr.y, p.z - ctr.z});
    return f3{remap_unit(n.x), remap_unit(n.y), remap_unit(n.z)};
}

template <bool EXACT>
__device__ __forceinline__ f3 primary_pixel(const view_consts &vc, const camera_consts &cc, uint32_t x, uint32_t y, const f3 bg) {
    const f3 cam = {vc.cam[0], vc.cam[1], vc.cam[2]};
    const f3 oc = {cc.oc[0], cc.oc[1], cc.oc[2]};
    const f3 ctr = {cc.center[0], cc.center[1], cc.center[2]};
    // src/software_raytracer.cpp:218-221: viewport point of the pixel centre minus the camera
    const float uvx = rt_fma<EXACT>(static_cast<float>(x) + 0.5f, vc.psx, -vc.vw_half);
    const float uvy = rt_fma<EXACT>(static_cast<float>(y) + 0.5f, vc.psy, -vc.vh_half);
    const f3 draw = {uvx - cam.x, uvy - cam.y, 0.0f - cam.z};
    const float dd = __builtin_fmaf(draw.z, draw.z, __builtin_fmaf(draw.y, draw.y, draw.x * draw.x));
    const float h = __builtin_fmaf(draw.z, oc.z, __builtin_fmaf(draw.y, oc.y, draw.x * oc.x));
    const float q = h * h;

    // Conservative miss test on the unnormalised ray (fused arithmetic): a quarter of the
    // reference's discriminant has the sign of h*h - dd*c, and either evaluation is within
    // about a dozen unit roundoffs of M = h*h + dd*(oc.oc + r*r); below -kScreenMargin of that
    // magnitude the exact discriminant is negative too and the pixel is background.
    const float big = __builtin_fmaf(dd, cc.magnitude, q);
    const float disc = __builtin_fmaf(-dd, cc.c, q);                // unnormalised ray: t runs along `draw`
    if(!(__builtin_fmaf(kScreenMargin, big, disc) >= 0.0f)) return bg;

    if constexpr(EXACT) {
        return primary_pixel_reference(cam, oc, ctr, cc.c, draw, bg);
    } else {
        // ill-conditioned hits (see kExactBelowQ) and rays within the margin of tangent take the
        // reference's arithmetic, ray included (the fast build's ray above is rounded once per
        // coordinate where the reference rounds twice)
        if(!(disc * (dd * cc.radius2) >= kExactBelowQ * (big * big))) {
            const float ex = (static_cast<float>(x) + 0.5f) * vc.psx - vc.vw_half;
            const float ey = (static_cast<float>(y) + 0.5f) * vc.psy - vc.vh_half;
            return primary_pixel_reference(cam, oc, ctr, cc.c, f3{ex - cam.x, ey - cam.y, 0.0f - cam.z}, bg);
        }
        const float t = (h - __builtin_amdgcn_sqrtf(disc)) * __builtin_amdgcn_rcpf(dd);
        const f3 v = {__builtin_fmaf(draw.x, t, cam.x - ctr.x), __builtin_fmaf(draw.y, t, cam.y - ctr.y),
                      __builtin_fmaf(draw.z, t, cam.z - ctr.z)};
        const float half_inv = 0.5f * __builtin_amdgcn_rsqf(dot3<false>(v, v));
        return f3{__builtin_fmaf(v.x, half_inv, 0.5f), __builtin_fmaf(v.y, half_inv, 0.5f), __builtin_fmaf(v.z, half_inv, 0.5f)};
    }
}

// box_rows / box_cols: the box of cells (kPrimaryBoxRows band rows by kTileW columns) outside of
// which no pixel's ray comes within the screening margin of the sphere (sphere_box), columns
// [bx0, bx1) of band rows [by0, by1) -- the whole band when there is no such box.  The first
// compute_groups workgroups render the box's pixels; the rest write the rows' background around
// it (fill_around_box): on the reference's 4K frame 78 % of the pixels, which then cost stores only.
constexpr uint32_t kPrimaryBoxRows = 8;
template <bool EXACT, bool VEC4>
__global__ __launch_bounds__(kPrimaryBlock) void primary_kernel(view_consts vc, camera_consts cc,
                                                                 uint32_t *__restrict__ argb,
                                                                 float4 *__restrict__ rgba, uint32_t box_rows, uint32_t box_cols,
                                                                 uint32_t bx0, uint32_t bx1, uint32_t by0, uint32_t by1,
                                                                 uint32_t compute_groups) {
    constexpr uint32_t PX = VEC4 ? 4u : 1u;      // (2 or 1 a lane inside the box, for more and shorter waves: 7 % and 22 % slower)
    if(blockIdx.x >= compute_groups) {
        fill_around_box(vc, blockIdx.x - compute_groups, kPrimaryBoxRows, box_rows, box_cols, argb, rgba, [&](uint32_t y) {
            const f3 bg = row_background<EXACT>(vc, y);
            return make_float4(bg.x, bg.y, bg.z, 1.0f);
        });
        return;
    }
    const uint32_t groups_per_row = (bx1 - bx0) / PX;
    const uint32_t gid = blockIdx.x * kPrimaryBlock + threadIdx.x;
    if(gid >= groups_per_row * (by1 - by0)) return;

    const uint32_t row = by0 + gid / groups_per_row;
    const uint32_t x0 = bx0 + (gid % groups_per_row) * PX;
    const uint32_t y = image_row(vc, row);
    const f3 bg = row_background<EXACT>(vc, y);

    uint32_t packed[PX];
    f3 rgb[PX];
#pragma unroll
    for(uint32_t i = 0; i < PX; i++) {
        rgb[i] = primary_pixel<EXACT>(vc, cc, x0 + i, y, bg);
        packed[i] = pack_argb32(rgb[i].x, rgb[i].y, rgb[i].z);
    }

    const size_t base = static_cast<size_t>(row) * vc.width + x0;
    if constexpr(VEC4) {
        *reinterpret_cast<uint4 *>(argb + base) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
    } else {
        argb[base] = packed[0];
    }
    if(rgba) {
#pragma unroll
        for(uint32_t i = 0; i < PX; i++) rgba[base + i] = make_float4(rgb[i].x, rgb[i].y, rgb[i].z, 1.0f);
    }
}

// ------------------------------------------------------------------------------------------
// path pass
// ------------------------------------------------------------------------------------------

struct path_state {
    f3 o, d, thr;
    uint32_t rng;
    uint32_t bounce;
};

template <bool EXACT>
__device__ __forceinline__ void start_path(const view_consts &vc, uint32_t pix_term, const uint32_t *__restrict__ sample_terms,
                                           uint32_t x, uint32_t y, uint32_t sample, path_state &ps) {
    const uint32_t key = pix_term + sample_terms[sample];
    float jx = 0.5f, jy = 0.5f;
    if(vc.flags & AEHIP_FLAG_JITTER) sample_offset(vc, sample, pix_term, jx, jy);
    ps.rng = key;
    ps.bounce = 0;
    ps.o = f3{vc.cam[0], vc.cam[1], vc.cam[2]};
    ps.d = primary_direction<EXACT>(vc, static_cast<float>(x) + jx, static_cast<float>(y) + jy);
    ps.thr = f3{1.0f, 1.0f, 1.0f};
}

// Ray-sphere screening and resolution against an LDS chunk (sphere = centre xyz, w = r*r).
//
// The list is walked four spheres at a time: four LDS reads issued together, four screening
// values computed with no branch between them, ONE branch per group on their maximum, and the
// expensive part (square root, division, range tests) only for the spheres that passed.
//
// Screening value, >= 0 means "may be hit":
//   fast build:  the discriminant itself in half-b form with a = d.d taken as 1 (d is
//                normalised): h*h - c, h = d.oc, c = oc.oc - r*r, all fused.
//   exact build: the same fused value plus a margin of 3e-5 times at least h*h + oc.oc + r*r.  A
//                quarter of the exact discriminant b*b - 4ac (src/shapes.cpp:15-20) and this
//                fused h*h - c are two roundings of the same real number; each is off by at
//                most about a dozen unit roundoffs (6e-8) of the magnitudes that enter it,
//                h*h + oc.oc + r*r -- the sum before the cancellation, so the bound also holds
//                for rays that start on the sphere they test (c ~ 0).  The margin is ~20x that
//                bound, so a negative screening value implies a negative exact discriminant:
//                the sphere cannot change the result.  Spheres that pass are resolved with the
//                reference's operation order, which decides.
template <bool EXACT>
__device__ __forceinline__ float screen_sphere(const float4 g, const f3 o, const f3 d, float &h_out) {
    const f3 oc = {g.x - o.x, g.y - o.y, g.z - o.z};
    const float h = __builtin_fmaf(d.z, oc.z, __builtin_fmaf(d.y, oc.y, d.x * oc.x));
    const float c = __builtin_fmaf(oc.z, oc.z, __builtin_fmaf(oc.y, oc.y, __builtin_fmaf(oc.x, oc.x, -g.w)));
    h_out = h;
    if constexpr(EXACT) {
        // h*h - c + margin * (2 oc.oc + r*r), the last factor being at least the magnitude
        // h*h + oc.oc + r*r (the direction has unit length to rounding, so h*h <= oc.oc), written
        // with oc.oc = c + r*r:  h*h - (1 - 2 margin) c + 3 margin r*r
        const float u = __builtin_fmaf(c, -(1.0f - 2.0f * kScreenMargin), g.w * (3.0f * kScreenMargin));
        return __builtin_fmaf(h, h, u);
    } else {
        return __builtin_fmaf(h, h, -c);
    }
}

// Quadratic as src/shapes.cpp:14-23 with a (t_min, closest) range test and the far root added.
template <bool EXACT>
__device__ __forceinline__ void resolve_sphere(const float4 g, int index, float screen, float h, const f3 o, const f3 d,
                                               float a2, float a4, float &closest, int &hit) {
    if(!(screen >= 0.0f)) return;
    // near root if it lies in (t_min, closest), else the far root if it does.  A near root at or
    // beyond `closest` leaves nothing for the far one (t_far >= t_near), so one range test serves,
    // and the far root is only worked out when the near one is behind t_min.
    float t;
    if constexpr(EXACT) {
        // Cheap bounds on both roots before the exact arithmetic.  `screen` is at least a quarter
        // of the exact discriminant (that is what its margin is for) and h is d.oc to rounding, so
        // the exact roots lie in [h - s - slack, h + s + slack] with s = sqrt(screen); the slack
        // (1e-5 of the magnitudes involved, the radius bounded by 1 + r*r) is ten times what the
        // roundings of h, of the square root and of the division by 2a can add up to.  A sphere
        // whose nearest possible root is not nearer than the current hit, or whose farthest is
        // behind t_min, cannot change the result: most spheres a ray's line passes through.
        const float s = __builtin_amdgcn_sqrtf(screen);
        const float slack = 1e-5f * (__builtin_fabsf(h) + s + 1.0f + g.w);
        if(h - s - slack >= closest || h + s + slack <= kTMin) return;
        const f3 oc = {g.x - o.x, g.y - o.y, g.z - o.z};
        const float b = -2.0f * dot3<true>(d, oc);
        const float c = dot3<true>(oc, oc) - g.w;
        const float disc = b * b - a4 * c;
        if(!(disc >= 0.0f)) return;
        const float sq = rt_sqrt<true>(disc);
        t = (-b - sq) / a2;
        if(!(t > kTMin)) t = (-b + sq) / a2;
    } else {
        const float sq = __builtin_amdgcn_sqrtf(screen);
        t = h - sq;
        if(!(t > kTMin)) t = h + sq;
    }
    if((t > kTMin) && (t < closest)) {
        closest = t;
        hit = index;
    }
}

// Rays that start at the camera (the first segment of every path) meet per-sphere constants:
// oc = C - camera and the part of the screening value that does not depend on the direction
// (-c, with the exact build's margin folded in).  Set-up tabulates them as (oc.xyz, u); the
// screening value is then h*h + u with h = d.oc: 4 operations a sphere instead of 10.  h and,
// in the fast build, the value have the same bits as screen_sphere's (same operations in the same
// order, on the host for the tabulated part); the exact build's u is rounded once more, far
// inside the margin.  Spheres that pass are resolved from the sphere list as ever.
//
// The table of camera entries is stored four spheres at a time with the fields of two spheres side
// by side -- x0 x1 y0 y1 z0 z1 u0 u1 | x2 x3 y2 y3 z2 z3 u2 u3 -- and padded to whole groups with
// entries no ray can pass (u = -FLT_MAX).  A pair of fields is one 64-bit scalar operand of a packed
// instruction: v_pk_fma_f32 with a scalar-register pair as a source issues at the packed rate (4.25
// clocks for two lanes' worth, tools/valu_peak.hip), where v_fma_f32 with a scalar-register source
// issues at half rate (4.2 clocks for one).  The same operations on the same numbers as one
// sphere at a time would take, so the same bits, at half the issue slots: 8 packed instructions for
// four spheres instead of 16 half-rate ones (10,000 spheres, primary rays: 4.6 -> 3.9 ms).  The same
// packing of screen_sphere's ten operations for rays from anywhere measured no gain (10,000
// spheres, depth 4: 34.8 -> 33.9 ms fast, 38.9 -> 40.0 ms exact) and is not used.
typedef float float_pair __attribute__((ext_vector_type(2)));
struct camera_group {
    float_pair x01, y01, z01, u01, x23, y23, z23, u23;
};
__device__ __forceinline__ camera_group load_camera_group(const float4 *__restrict__ table, uint32_t group) {
    const float_pair *p = reinterpret_cast<const float_pair *>(table + group * 4u);
    return camera_group{p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7]};
}
template <bool EXACT>
__device__ __forceinline__ void test_camera_group(const camera_group &cg, const float4 *__restrict__ spheres, int index, const f3 o,
                                                  const f3 d, float a2, float a4, float &closest, int &hit) {
    const float_pair dx = {d.x, d.x}, dy = {d.y, d.y}, dz = {d.z, d.z};
    const float_pair h01 = __builtin_elementwise_fma(dz, cg.z01, __builtin_elementwise_fma(dy, cg.y01, dx * cg.x01));
    const float_pair h23 = __builtin_elementwise_fma(dz, cg.z23, __builtin_elementwise_fma(dy, cg.y23, dx * cg.x23));
    const float_pair k01 = __builtin_elementwise_fma(h01, h01, cg.u01);
    const float_pair k23 = __builtin_elementwise_fma(h23, h23, cg.u23);
    if(__builtin_fmaxf(__builtin_fmaxf(k01.x, k01.y), __builtin_fmaxf(k23.x, k23.y)) >= 0.0f) {
        // the spheres themselves are only fetched for the few that pass (exact build; the fast
        // build's roots come from the screening value and h alone)
        if(k01.x >= 0.0f) resolve_sphere<EXACT>(spheres[index], index, k01.x, h01.x, o, d, a2, a4, closest, hit);
        if(k01.y >= 0.0f) resolve_sphere<EXACT>(spheres[index + 1], index + 1, k01.y, h01.y, o, d, a2, a4, closest, hit);
        if(k23.x >= 0.0f) resolve_sphere<EXACT>(spheres[index + 2], index + 2, k23.x, h23.x, o, d, a2, a4, closest, hit);
        if(k23.y >= 0.0f) resolve_sphere<EXACT>(spheres[index + 3], index + 3, k23.y, h23.y, o, d, a2, a4, closest, hit);
    }
}
// The walk of intersect_chunk (below) over the packed table; `count` spheres, the table padded to groups of four.
template <bool EXACT>
__device__ __forceinline__ void intersect_camera_table(const float4 *__restrict__ table, uint32_t count, const f3 o, const f3 d,
                                                       float a2, float a4, float &closest, int &hit, const float4 *__restrict__ spheres) {
    const uint32_t groups = (count + 3u) / 4u;
    if(groups == 0u) return;
    uint32_t g = 0;
    camera_group A = load_camera_group(table, 0);
    while(g + 2 <= groups) {
        const camera_group B = load_camera_group(table, g + 1);
        test_camera_group<EXACT>(A, spheres, static_cast<int>(g * 4u), o, d, a2, a4, closest, hit);
        A = load_camera_group(table, min(g + 2, groups - 1));      // clamped: the last fetch may repeat a group
        test_camera_group<EXACT>(B, spheres, static_cast<int>((g + 1) * 4u), o, d, a2, a4, closest, hit);
        g += 2;
    }
    if(g < groups) test_camera_group<EXACT>(A, spheres, static_cast<int>(g * 4u), o, d, a2, a4, closest, hit);
}

struct sphere_group {
    float4 g0, g1, g2, g3;
};

__device__ __forceinline__ sphere_group load_group(const float4 *__restrict__ list, uint32_t group) {
    const float4 *p = list + group * 4u;
    return sphere_group{p[0], p[1], p[2], p[3]};
}

template <bool EXACT>
__device__ __forceinline__ void test_group(const sphere_group &sg, int index, const f3 o, const f3 d, float a2, float a4, float &closest,
                                           int &hit) {
    float h0, h1, h2, h3;
    const float k0 = screen_sphere<EXACT>(sg.g0, o, d, h0);
    const float k1 = screen_sphere<EXACT>(sg.g1, o, d, h1);
    const float k2 = screen_sphere<EXACT>(sg.g2, o, d, h2);
    const float k3 = screen_sphere<EXACT>(sg.g3, o, d, h3);
    if(__builtin_fmaxf(__builtin_fmaxf(k0, k1), __builtin_fmaxf(k2, k3)) >= 0.0f) {
        resolve_sphere<EXACT>(sg.g0, index, k0, h0, o, d, a2, a4, closest, hit);
        resolve_sphere<EXACT>(sg.g1, index + 1, k1, h1, o, d, a2, a4, closest, hit);
        resolve_sphere<EXACT>(sg.g2, index + 2, k2, h2, o, d, a2, a4, closest, hit);
        resolve_sphere<EXACT>(sg.g3, index + 3, k3, h3, o, d, a2, a4, closest, hit);
    }
}

// `list` is either the LDS chunk or (scalar mode) the list in global memory read with a
// wave-uniform index.  Groups are fetched one ahead of the group being tested, alternating two
// register sets so no copies are needed: the fetch of group g+1 is in flight while group g is
// screened.
template <bool EXACT>
__device__ __forceinline__ void intersect_chunk(const float4 *__restrict__ list, uint32_t count, uint32_t index_base,
                                                const f3 o, const f3 d, float a2, float a4, float &closest, int &hit) {
    const uint32_t groups = count / 4u;
    uint32_t g = 0;
    if(groups > 0) {
        sphere_group A = load_group(list, 0);
        while(g + 2 <= groups) {
            const sphere_group B = load_group(list, g + 1);
            test_group<EXACT>(A, static_cast<int>(index_base + g * 4u), o, d, a2, a4, closest, hit);
            A = load_group(list, min(g + 2, groups - 1));      // clamped: the last fetch may repeat a group
            test_group<EXACT>(B, static_cast<int>(index_base + (g + 1) * 4u), o, d, a2, a4, closest, hit);
            g += 2;
        }
        if(g < groups) {
            test_group<EXACT>(A, static_cast<int>(index_base + g * 4u), o, d, a2, a4, closest, hit);
            g++;
        }
    }
    for(uint32_t i = g * 4u; i < count; i++) {
        const float4 s = list[i];
        float h;
        const float k = screen_sphere<EXACT>(s, o, d, h);
        resolve_sphere<EXACT>(s, static_cast<int>(index_base + i), k, h, o, d, a2, a4, closest, hit);
    }
}

// List prefilter for rays from anywhere (8 operations a sphere; the screening value above takes
// 10 because it forms oc = C - O first).  With everything measured from a point G in the middle
// of the scene, Cc = C - G and Oc = O - G,
//     h = d.Cc - d.Oc          c = |Cc|^2 - 2 Cc.Oc + |Oc|^2 - r*r
// and what depends on the sphere alone (Cc and K = |Cc|^2 - r*r, tabulated at set-up) meets what
// depends on the ray alone (d, d.Oc, -2 Oc, |Oc|^2, worked out once per segment) in two chains of
// three fmas.  The expansion cancels where oc = C - O does not, so this value is only a filter:
// its rounding errors (about 30 unit roundoffs of (|Cc| + |Oc| + r)^2) and the screening value's
// own are covered five times over by a margin of kPrefilterMargin ((|Cc| + r)^2 + |Oc|^2), the
// sphere's share of it folded into K and the ray's into the |Oc|^2 term -- it grows only for
// spheres and rays far from G.  The exact build's screening margin is folded in the same way
// (factor 1 - 2m on c, 3 m r*r in K).  A negative value therefore implies a negative screening
// value; groups of four whose largest value is not negative go through the screening and
// resolution above, from the sphere list, unchanged: results keep their bits.
constexpr float kPrefilterMargin = 3.2e-5f;

struct prefilter_ray {
    f3 d;
    float neg_d_oc;     // -(d.Oc)
    f3 m2oc;            // -2 f Oc            (f = 1, exact build 1 - 2 kScreenMargin)
    float q;            // (f - kPrefilterMargin) |Oc|^2
};

template <bool EXACT>
__device__ __forceinline__ prefilter_ray make_prefilter_ray(const view_consts &vc, const f3 o, const f3 d) {
    constexpr float f = EXACT ? 1.0f - 2.0f * kScreenMargin : 1.0f;
    const f3 oc = {o.x - vc.list_centre[0], o.y - vc.list_centre[1], o.z - vc.list_centre[2]};
    prefilter_ray r;
    r.d = d;
    r.neg_d_oc = -__builtin_fmaf(d.z, oc.z, __builtin_fmaf(d.y, oc.y, d.x * oc.x));
    r.m2oc = f3{(-2.0f * f) * oc.x, (-2.0f * f) * oc.y, (-2.0f * f) * oc.z};
    r.q = (f - kPrefilterMargin) * __builtin_fmaf(oc.z, oc.z, __builtin_fmaf(oc.y, oc.y, oc.x * oc.x));
    return r;
}

// e = (Cc.xyz, K'), K' = f (|Cc|^2 - r*r) - [exact: 3 m r*r] - kPrefilterMargin (|Cc| + r)^2
__device__ __forceinline__ float prefilter_entry(const float4 e, const prefilter_ray &r) {
    const float h = __builtin_fmaf(r.d.z, e.z, __builtin_fmaf(r.d.y, e.y, __builtin_fmaf(r.d.x, e.x, r.neg_d_oc)));
    const float v = __builtin_fmaf(e.z, r.m2oc.z, __builtin_fmaf(e.y, r.m2oc.y, __builtin_fmaf(e.x, r.m2oc.x, r.q)));
    return __builtin_fmaf(h, h, -v) - e.w;
}

// The list walk of intersect_chunk over the prefilter table; `spheres` is the sphere list itself.
template <bool EXACT>
__device__ __forceinline__ void intersect_prefiltered(const view_consts &vc, const float4 *__restrict__ table,
                                                      const float4 *__restrict__ spheres, uint32_t count, const f3 o, const f3 d,
                                                      float a2, float a4, float &closest, int &hit) {
    const prefilter_ray ray = make_prefilter_ray<EXACT>(vc, o, d);
    auto test = [&](const sphere_group &e, uint32_t group) {
        const float v0 = prefilter_entry(e.g0, ray), v1 = prefilter_entry(e.g1, ray);
        const float v2 = prefilter_entry(e.g2, ray), v3 = prefilter_entry(e.g3, ray);
        if(__builtin_fmaxf(__builtin_fmaxf(v0, v1), __builtin_fmaxf(v2, v3)) >= 0.0f)
            test_group<EXACT>(load_group(spheres, group), static_cast<int>(group * 4u), o, d, a2, a4, closest, hit);
    };
    const uint32_t groups = count / 4u;
    uint32_t g = 0;
    if(groups > 0) {
        sphere_group A = load_group(table, 0);
        while(g + 2 <= groups) {
            const sphere_group B = load_group(table, g + 1);
            test(A, g);
            A = load_group(table, min(g + 2, groups - 1));      // clamped: the last fetch may repeat a group
            test(B, g + 1);
            g += 2;
        }
        if(g < groups) {
            test(A, g);
            g++;
        }
    }
    for(uint32_t i = g * 4u; i < count; i++) {
        if(prefilter_entry(table[i], ray) >= 0.0f) {
            const float4 s = spheres[i];
            float h;
            const float k = screen_sphere<EXACT>(s, o, d, h);
            resolve_sphere<EXACT>(s, static_cast<int>(i), k, h, o, d, a2, a4, closest, hit);
        }
    }
}

// Shades the hit (or miss) of the current segment.  Returns true when the path ended with
// radiance L; false when it continues with ps updated.
template <bool EXACT>
__device__ __forceinline__ bool shade_segment(const view_consts &vc, uint32_t y, int hit, float closest,
                                              const float4 *__restrict__ geom, const float4 *__restrict__ matl,
                                              const uint32_t *__restrict__ kind, path_state &ps, f3 &L) {
    L = f3{0.0f, 0.0f, 0.0f};
    if(hit < 0) {
        f3 sky;
        if(ps.bounce == 0) {
            sky = row_background<EXACT>(vc, y);
        } else {
            const float t = 0.5f * (ps.d.y + 1.0f);
            sky = f3{lerp1<EXACT>(t, vc.bg0[0], vc.bg1[0]), lerp1<EXACT>(t, vc.bg0[1], vc.bg1[1]),
                     lerp1<EXACT>(t, vc.bg0[2], vc.bg1[2])};
        }
        L = f3{ps.thr.x * sky.x, ps.thr.y * sky.y, ps.thr.z * sky.z};
        return true;
    }

    const float4 g = geom[hit];
    const uint32_t material = kind[hit];
    const f3 pt = {rt_fma<EXACT>(ps.d.x, closest, ps.o.x), rt_fma<EXACT>(ps.d.y, closest, ps.o.y),
                   rt_fma<EXACT>(ps.d.z, closest, ps.o.z)};
    const f3 n_out = normalize3<EXACT>(f3{pt.x - g.x, pt.y - g.y, pt.z - g.z});

    if(material == AEHIP_MAT_NORMAL) {
        L = f3{ps.thr.x * remap_unit(n_out.x), ps.thr.y * remap_unit(n_out.y), ps.thr.z * remap_unit(n_out.z)};
        return true;
    }

    const float4 m = matl[hit];
    const float dn = dot3<EXACT>(ps.d, n_out);
    const bool front = dn < 0.0f;
    const f3 n = front ? n_out : f3{-n_out.x, -n_out.y, -n_out.z};
    f3 nd;

    // the random unit vector of the diffuse and the fuzzy-metal branch, drawn once for both so a
    // wave with lanes on either material runs the polynomial once
    f3 u = {0.0f, 0.0f, 0.0f};
    if(material == AEHIP_MAT_LAMBERT || (material == AEHIP_MAT_METAL && m.w > 0.0f)) u = unit_vector<EXACT>(rng_next(ps.rng));

    if(material == AEHIP_MAT_LAMBERT) {
        nd = f3{n.x + u.x, n.y + u.y, n.z + u.z};
        if(__builtin_fabsf(nd.x) < 1e-8f && __builtin_fabsf(nd.y) < 1e-8f && __builtin_fabsf(nd.z) < 1e-8f) nd = n;
    } else if(material == AEHIP_MAT_METAL) {
        const float k = 2.0f * dot3<EXACT>(ps.d, n);
        nd = f3{rt_fma<EXACT>(-k, n.x, ps.d.x), rt_fma<EXACT>(-k, n.y, ps.d.y), rt_fma<EXACT>(-k, n.z, ps.d.z)};
        if(m.w > 0.0f) nd = f3{rt_fma<EXACT>(m.w, u.x, nd.x), rt_fma<EXACT>(m.w, u.y, nd.y), rt_fma<EXACT>(m.w, u.z, nd.z)};
        if(!(dot3<EXACT>(nd, n) > 0.0f)) return true;      // absorbed
    } else {
        const float ri = front ? rt_div<EXACT>(1.0f, m.w) : m.w;
        const f3 md = {-ps.d.x, -ps.d.y, -ps.d.z};
        const float cd = dot3<EXACT>(md, n);
        const float cos_t = cd < 1.0f ? cd : 1.0f;
        const float s2 = 1.0f - cos_t * cos_t;
        const float sin_t = rt_sqrt<EXACT>(s2 > 0.0f ? s2 : 0.0f);
        float r0 = rt_div<EXACT>(1.0f - ri, 1.0f + ri);
        r0 = r0 * r0;
        const float w = 1.0f - cos_t;
        const float w2 = w * w;
        const float reflectance = rt_fma<EXACT>(1.0f - r0, (w2 * w2) * w, r0);
        const float draw = static_cast<float>(rng_next(ps.rng) >> 8) * (1.0f / 16777216.0f);
        if(ri * sin_t > 1.0f || reflectance > draw) {
            const float k = 2.0f * dot3<EXACT>(ps.d, n);
            nd = f3{rt_fma<EXACT>(-k, n.x, ps.d.x), rt_fma<EXACT>(-k, n.y, ps.d.y), rt_fma<EXACT>(-k, n.z, ps.d.z)};
        } else {
            const f3 perp = {(rt_fma<EXACT>(n.x, cos_t, ps.d.x)) * ri, (rt_fma<EXACT>(n.y, cos_t, ps.d.y)) * ri,
                             (rt_fma<EXACT>(n.z, cos_t, ps.d.z)) * ri};
            const float par = -rt_sqrt<EXACT>(__builtin_fabsf(1.0f - dot3<EXACT>(perp, perp)));
            nd = f3{rt_fma<EXACT>(n.x, par, perp.x), rt_fma<EXACT>(n.y, par, perp.y), rt_fma<EXACT>(n.z, par, perp.z)};
        }
    }

    ps.thr = f3{ps.thr.x * m.x, ps.thr.y * m.y, ps.thr.z * m.z};
    ps.o = pt;
    ps.d = normalize3<EXACT>(nd);
    ps.bounce += 1;
    if(ps.bounce >= vc.max_depth) return true;              // depth exhausted: no light
    return false;
}

// One lane owns one pixel and walks its samples in index order; the moment a path ends the lane
// starts the pixel's next sample (path regeneration), so a wave only idles lanes whose pixel is
// complete.  The sum over samples is therefore sequential per pixel whatever the divergence.
//
// Where the shape list is read from (LIST):
//   kListLdsOnce    the whole list fits the LDS buffer and is staged once; no barrier inside the
//                   segment loop, each wave leaves when its own lanes are done;
//   kListLdsChunked the list streams through the LDS buffer in chunks and the workgroup
//                   iterates segments together (a barrier pair per chunk);
//   kListScalar     no LDS: the sphere index is wave-uniform, so the list is read straight from
//                   global memory through the scalar cache into SGPRs (s_load_dwordx4..x16);
//                   waves stay independent for any list length and occupancy is not capped by
//                   the LDS allocation.
constexpr int kListLdsOnce = 0, kListLdsChunked = 1, kListScalar = 2;

template <bool EXACT, int LIST>
__global__ __launch_bounds__(kBlock) void path_kernel(view_consts vc, const float4 *__restrict__ geom,
                                                      const float4 *__restrict__ matl,
                                                      const uint32_t *__restrict__ kind,
                                                      const uint32_t *__restrict__ sample_terms, uint32_t *__restrict__ argb,
                                                      float4 *__restrict__ rgba, unsigned long long *ray_counter,
                                                      const float4 *__restrict__ camera_list) {
    constexpr bool SINGLE_CHUNK = LIST != kListLdsChunked;      // no workgroup barrier in the loop
    __shared__ float4 s_geom[LIST == kListScalar ? 1 : kLdsSpheres];

    const uint32_t tid = threadIdx.x;
    const uint32_t x = blockIdx.x * kTileW + (tid & (kTileW - 1));
    const uint32_t row = blockIdx.y * kTileH + (tid / kTileW);
    const uint32_t y = image_row(vc, row);
    const bool valid = (x < vc.width) && (row < vc.row_count);

    if constexpr(LIST == kListLdsOnce) {
        for(uint32_t i = tid; i < vc.n_spheres; i += kBlock) s_geom[i] = geom[i];
        __syncthreads();
    }

    f3 sum = {0.0f, 0.0f, 0.0f};      // block sums in block order
    f3 block = {0.0f, 0.0f, 0.0f};    // the current block's samples in sample order
    uint32_t sample = 0;
    uint32_t nrays = 0;
    uint32_t iterations = 0;        // segment iterations this wave executed (same value in every lane)
    bool alive = valid;             // the pixel still has samples to finish
    bool tracing = valid;           // the lane is following a path right now
    const bool regenerate = !(vc.flags & AEHIP_FLAG_NO_REGEN);
    path_state ps;
    ps.o = ps.d = ps.thr = f3{0.0f, 0.0f, 0.0f};
    ps.rng = 0;
    ps.bounce = 0;
    const uint32_t pix_term = pixel_term(vc.seed_hash, y * vc.width + x);
    if(alive) start_path<EXACT>(vc, pix_term, sample_terms, x, y, 0, ps);

    for(;;) {
        if constexpr(SINGLE_CHUNK) {
            if(!__any(alive)) break;
        } else {
            if(!__syncthreads_or(alive)) break;
        }
        // without regeneration a finished lane idles until its whole wave has finished, then
        // the wave starts the next sample together
        if(!regenerate && !__any(tracing)) {
            if(alive) {
                start_path<EXACT>(vc, pix_term, sample_terms, x, y, sample, ps);
                tracing = true;
            }
        }
        iterations++;

        const float a = dot3<EXACT>(ps.d, ps.d);
        const float a4 = 4.0f * a;
        const float a2 = 2.0f * a;
        float closest = 3.402823466e+38f;
        int hit = -1;

        if constexpr(LIST == kListLdsOnce) {
            if(tracing) intersect_chunk<EXACT>(s_geom, vc.n_spheres, 0, ps.o, ps.d, a2, a4, closest, hit);
        } else if constexpr(LIST == kListScalar) {
            // a wave whose rays all start at the camera (every wave of a one-segment render, the
            // first iteration otherwise) walks the table of camera entries instead
            if(!__any(tracing && ps.bounce != 0)) {
                if(tracing) intersect_camera_table<EXACT>(camera_list, vc.n_spheres, ps.o, ps.d, a2, a4, closest, hit, geom);
            } else {
                if(tracing) intersect_chunk<EXACT>(geom, vc.n_spheres, 0, ps.o, ps.d, a2, a4, closest, hit);
            }
        } else {
            for(uint32_t base = 0; base < vc.n_spheres; base += kLdsSpheres) {
                const uint32_t count = min(static_cast<uint32_t>(kLdsSpheres), vc.n_spheres - base);
                __syncthreads();
                for(uint32_t i = tid; i < count; i += kBlock) s_geom[i] = geom[base + i];
                __syncthreads();
                if(tracing) intersect_chunk<EXACT>(s_geom, count, base, ps.o, ps.d, a2, a4, closest, hit);
            }
        }

        if(tracing) {
            nrays++;
            f3 L;
            if(shade_segment<EXACT>(vc, y, hit, closest, geom, matl, kind, ps, L)) {
                block.x += L.x;
                block.y += L.y;
                block.z += L.z;
                sample++;
                if((sample & (AEHIP_SAMPLE_BLOCK - 1u)) == 0u || sample >= vc.spp) {    // block complete
                    sum.x += block.x;
                    sum.y += block.y;
                    sum.z += block.z;
                    block = f3{0.0f, 0.0f, 0.0f};
                }
                if(sample >= vc.spp) {
                    alive = false;
                    tracing = false;
                } else if(regenerate) {
                    start_path<EXACT>(vc, pix_term, sample_terms, x, y, sample, ps);
                } else {
                    tracing = false;
                }
            }
        }
    }

    if(valid) {
        const float n = static_cast<float>(vc.spp);
        // the mean is a true division in both builds (once per pixel): an approximate reciprocal
        // would turn a row of exact 1.0 samples into 0.99999994 and the truncating pack into 254
        const float r = clamp01(sum.x / n);
        const float g = clamp01(sum.y / n);
        const float b = clamp01(sum.z / n);
        const size_t idx = static_cast<size_t>(row) * vc.width + x;
        argb[idx] = pack_argb32(r, g, b);
        if(rgba) rgba[idx] = make_float4(r, g, b, 1.0f);
    }

    if(vc.flags & AEHIP_FLAG_COUNT_RAYS) {
        // wave-level sums, gathered per workgroup in LDS: one pair of global atomics per
        // workgroup (every wave of a launch adding to the same two words serialises)
        __shared__ uint32_t s_counts[2];
        if(tid < 2) s_counts[tid] = 0;
        __syncthreads();
        uint32_t v = nrays;
#pragma unroll
        for(int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if((tid & 63u) == 0) {
            atomicAdd(&s_counts[0], v);
            atomicAdd(&s_counts[1], iterations);
        }
        __syncthreads();
        if(tid == 0) {
            atomicAdd(ray_counter, static_cast<unsigned long long>(s_counts[0]));
            atomicAdd(ray_counter + 1, 64ull * s_counts[1]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// streamed path pass: the wave, not the lane, owns pixels, and persistent waves take work units
// from a queue.
//
// A unit is a 16x4 pixel strip times a chunk of up to kChunkSamples samples: a pool of 64*chunk
// work items; item i is sample i/64 of pixel i%64, so an item row is the strip's 64 coherent
// pixels.  Whenever lanes finish their paths, a ballot counts them and the next items of the pool
// are dealt out in lane order (prefix popcount of the ballot; the pool cursor is wave-uniform),
// so lanes stay busy whatever the spread of path lengths -- with one lane per pixel (path_kernel)
// a lane whose pixel is finished idles until its wave's slowest pixel is.
//
// A wave does not end with its unit: that would idle lanes while each pool drains (its last,
// longest paths) and, per launch, while the last and heaviest waves finish on an emptying chip.
// A wave takes the next units from a global counter when its
// current pool is dealt out, and deals the new unit's items to lanes as they come free while
// other lanes are still finishing paths of earlier units.  A wave leaves when the queue is empty
// and its own paths are done -- it never waits for another wave -- so lanes idle only in a
// wave's very last paths, and the launch needs no more waves than the chip holds.  Chunks keep
// a strip over a glass sphere from pinning one wave for the whole launch: its samples are spread
// over several waves.  Measured on the depth-50 mixed-material scene: 93-99 % of lane slots trace
// a ray (76 % with one strip per wave, 57 % with per-lane regeneration, 24 % without).
//
// Where the radiance goes: samples must be added in the specified order (exact build), and the
// image should not depend on which wave finished what first (fast build), so finished paths are
// not accumulated as they come.  Each path's radiance is stored under its (pixel, sample) in a
// device buffer, one block of up to AEHIP_SAMPLE_BLOCK samples per launch, and a second kernel
// adds the block's samples in sample order and the block sums in block order.  The tracing
// kernel therefore keeps no per-strip state beyond the dealing cursor.  The buffer costs 12 bytes
// per path of HBM traffic each way -- nothing beside the tracing -- and 768 bytes per pixel of
// memory.
// ------------------------------------------------------------------------------------------

// The first-hit pre-pass pays when camera rays are a large share of all rays and a list walk costs
// far more than the extra shading rounds: 2,000 spheres, sparse (1.3 rays a path): 16.2 -> 13.4 ms
// fast, 20.2 -> 15.7 exact; 10,000 (2.5 rays a path): 43.0 -> 36.9, 51.3 -> 46.9; 1,200 spheres in
// a dense cloud (7.9 rays a path): 26.0 -> 25.8, 35.6 -> 37.3; 300 spheres: 8.6 -> 9.4 (not used).
constexpr uint32_t kFirstHitsMinSpheres = 1024;
constexpr uint32_t kDealRounds = 4;          // lane utilisation at 2,000 spheres: 0.55 / 0.75 / 0.82 / 0.86 with 1 / 2 / 3 / 4
constexpr uint32_t kChunkSamples = 4;        // samples of a strip that one queue unit covers (256 items; measured best of 1..16)

struct radiance3 {
    float r, g, b;
};

// The strips one launch of the streamed pass works on, and where their paths' radiance goes: the
// band's strips [begin, begin + count), sample `s` of the current block of pixel `p` of window
// strip `w` at radiance[(w * stride + s) * 64 + p] (stride = samples per block the buffer holds
// per strip).  A band with more strips than the scratch holds is rendered window after window.
struct strip_window {
    uint32_t begin, count, stride;
};

// First segments of the streamed pass.  Every path starts at the camera, so its first ray-list
// walk can use the table of camera entries (4 operations a test instead of 10) if whole waves
// walk it together: one wave per work unit, lane = pixel, the unit's samples one after the other.
// The nearest hit of each path is left in the path's radiance slot as (t, sphere index); the
// tracing kernel picks it up when it deals the path to a lane and goes straight to shading.
template <bool EXACT>
__global__ __launch_bounds__(kBlock) void first_hit_kernel(view_consts vc, const float4 *__restrict__ geom,
                                                           const float4 *__restrict__ camera_list,
                                                           const uint32_t *__restrict__ sample_terms,
                                                           radiance3 *__restrict__ radiance, uint32_t sample_begin,
                                                           uint32_t sample_count, unsigned long long *ray_counter, strip_window win) {
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u;
    const uint32_t strips_x = (vc.width + kTileW - 1) / kTileW;
    const uint32_t chunks = (sample_count + kChunkSamples - 1u) / kChunkSamples;
    const uint32_t n_strips = win.count;
    const uint32_t unit = blockIdx.x * (kBlock / 64) + wave;
    if(unit >= n_strips * chunks) return;
    const uint32_t chunk = unit / n_strips;
    const uint32_t wstrip = unit - chunk * n_strips;       // strip of the window
    const uint32_t strip = win.begin + wstrip;            // strip of the band
    const uint32_t sy = strip / strips_x;
    const uint32_t x = (strip - sy * strips_x) * kTileW + (lane & 15u);
    const uint32_t row = sy * 4u + (lane >> 4);
    const bool valid = x < vc.width && row < vc.row_count;
    const uint32_t y = image_row(vc, row);
    const uint32_t term = pixel_term(vc.seed_hash, y * vc.width + x);
    const uint32_t begin = chunk * kChunkSamples;
    const uint32_t count = min(kChunkSamples, sample_count - begin);
    for(uint32_t s = 0; s < count; s++) {
        const uint32_t local = begin + s;
        path_state ps;
        start_path<EXACT>(vc, term, sample_terms, x, y, sample_begin + local, ps);
        float a2 = 2.0f, a4 = 4.0f;
        if constexpr(EXACT) {
            const float a = dot3<true>(ps.d, ps.d);
            a2 = 2.0f * a;
            a4 = 4.0f * a;
        }
        float closest = 3.402823466e+38f;
        int hit = -1;
        intersect_camera_table<EXACT>(camera_list, vc.n_spheres, ps.o, ps.d, a2, a4, closest, hit, geom);
        if(valid) radiance[(static_cast<size_t>(wstrip) * win.stride + local) * 64u + lane] = radiance3{closest, __int_as_float(hit), 0.0f};
    }
    if((vc.flags & AEHIP_FLAG_COUNT_RAYS) && lane == 0) atomicAdd(ray_counter + 1, 64ull * count);
}

template <bool EXACT, bool PREFILTER>
__global__ __launch_bounds__(kBlock) void path_stream_kernel(view_consts vc, const float4 *__restrict__ geom,
                                                                    const float4 *__restrict__ prefilter_list,
                                                             const float4 *__restrict__ matl,
                                                             const uint32_t *__restrict__ kind,
                                                             const uint32_t *__restrict__ sample_terms,
                                                             radiance3 *__restrict__ radiance, uint32_t sample_begin,
                                                             uint32_t sample_count, unsigned long long *ray_counter,
                                                             uint32_t *unit_queue, uint32_t unit_batch, strip_window win) {
    constexpr int kWaves = kBlock / 64;
    __shared__ uint32_t s_term[kWaves][64];          // per-pixel key terms of the unit being dealt

    const uint32_t tid = threadIdx.x;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u;
    const uint32_t strips_x = (vc.width + kTileW - 1) / kTileW;
    const uint32_t chunks = (sample_count + kChunkSamples - 1u) / kChunkSamples;     // units per strip
    const uint32_t n_strips = win.count;
    const uint32_t n_units = n_strips * chunks;

    // wave-uniform state
    uint32_t pool = 0, cursor = 0;                   // items of the current unit, next undealt one (cursor == pool: none left)
    bool queue_empty = false;
    uint32_t batch_next = 0, batch_end = 0;          // units taken from the queue and not yet started
    uint32_t strip = 0xffffffffu, strip_x0 = 0, strip_row0 = 0, chunk_begin = 0;      // no strip yet

    // lane state
    uint32_t nrays = 0, iterations = 0;
    uint32_t y = 0;
    size_t slot = 0;                                 // where this lane's path stores its radiance
    bool tracing = false;
    path_state ps;
    ps.o = ps.d = ps.thr = f3{0.0f, 0.0f, 0.0f};
    ps.rng = 0;
    ps.bounce = 0;

    for(;;) {
        const unsigned long long want = __ballot(!tracing);
        if(want != 0ull) {
            if(cursor >= pool && !queue_empty) {
                // Units are numbered chunk-major and taken kUnitBatch at a time: every wave of the
                // launch adds to the same counter, and atomics on one address go through one at a
                // time (130,000 single takes alone were 1.7 ms of a 1080p 16-spp launch).  The
                // units of a batch are neighbouring strips at the same samples, so a heavy strip's
                // chunks still go to different waves.
                if(batch_next >= batch_end) {
                    uint32_t first = 0;
                    if(lane == 0) first = atomicAdd(unit_queue, unit_batch);
                    batch_next = __builtin_amdgcn_readfirstlane(first);
                    batch_end = min(batch_next + unit_batch, n_units);
                }
                if(batch_next >= n_units) {
                    queue_empty = true;
                } else {
                    const uint32_t id = batch_next++;
                    const uint32_t chunk = id / n_strips;
                    const uint32_t new_strip = id - chunk * n_strips;       // strip of the window
                    chunk_begin = chunk * kChunkSamples;
                    pool = 64u * min(kChunkSamples, sample_count - chunk_begin);
                    cursor = 0;
                    if(new_strip != strip) {
                        strip = new_strip;
                        const uint32_t band_strip = win.begin + strip;
                        const uint32_t sy = band_strip / strips_x;
                        strip_x0 = (band_strip - sy * strips_x) * kTileW;
                        strip_row0 = sy * 4u;
                        __builtin_amdgcn_wave_barrier();
                        s_term[wave][lane] = pixel_term(vc.seed_hash, image_row(vc, strip_row0 + (lane >> 4)) * vc.width + strip_x0 + (lane & 15u));
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
            if(cursor < pool) {
                const uint32_t before = __builtin_popcountll(want & ((1ull << lane) - 1ull));
                const uint32_t item = cursor + before;
                cursor += min(static_cast<uint32_t>(__builtin_popcountll(want)), pool - cursor);
                if(!tracing && item < pool) {
                    const uint32_t pix = item & 63u;
                    const uint32_t local = chunk_begin + (item >> 6);           // sample within the block
                    const uint32_t x = strip_x0 + (pix & 15u);
                    const uint32_t row = strip_row0 + (pix >> 4);
                    if(x < vc.width && row < vc.row_count) {
                        y = image_row(vc, row);
                        slot = (static_cast<size_t>(strip) * win.stride + local) * 64u + pix;
                        start_path<EXACT>(vc, s_term[wave][pix], sample_terms, x, y, sample_begin + local, ps);
                        tracing = true;
                    }
                    // an item of a pixel outside the image is simply dropped
                }
            }
        }

        if(!__any(tracing)) {
            if(queue_empty && cursor >= pool) break;
            continue;               // only out-of-image items were dealt, or the next unit is due
        }
        iterations++;
        if(tracing) {
            float a2 = 2.0f, a4 = 4.0f;
            if constexpr(EXACT) {
                const float a = dot3<true>(ps.d, ps.d);     // ~1, not assumed 1 (src/shapes.cpp:16)
                a2 = 2.0f * a;
                a4 = 4.0f * a;
            }
            float closest = 3.402823466e+38f;
            int hit = -1;
            if constexpr(PREFILTER) intersect_prefiltered<EXACT>(vc, prefilter_list, geom, vc.n_spheres, ps.o, ps.d, a2, a4, closest, hit);
            else intersect_chunk<EXACT>(geom, vc.n_spheres, 0, ps.o, ps.d, a2, a4, closest, hit);
            nrays++;
            f3 L;
            if(shade_segment<EXACT>(vc, y, hit, closest, geom, matl, kind, ps, L)) {
                radiance[slot] = radiance3{L.x, L.y, L.z};
                tracing = false;
            }
        }
    }

    if(vc.flags & AEHIP_FLAG_COUNT_RAYS) {
        uint32_t v = nrays;
#pragma unroll
        for(int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if(lane == 0) {
            atomicAdd(ray_counter, static_cast<unsigned long long>(v));
            atomicAdd(ray_counter + 1, 64ull * iterations);
        }
    }
}

// path_stream_kernel after first_hit_kernel: a path dealt to a lane brings its first hit and is
// shaded before the lane's first list walk.  An iteration is therefore "shade, deal" repeated up
// to `deal_rounds` times -- so that lanes whose path ends at its first hit (most camera rays that
// see the sky) get another path before the walk instead of idling through it -- and then the
// walk for every lane whose ray has no hit yet.
template <bool EXACT, bool PREFILTER>
__global__ __launch_bounds__(kBlock) void path_stream_first_hits_kernel(view_consts vc, const float4 *__restrict__ geom,
                                                                                          const float4 *__restrict__ prefilter_list,
                                                             const float4 *__restrict__ matl,
                                                             const uint32_t *__restrict__ kind,
                                                             const uint32_t *__restrict__ sample_terms,
                                                             radiance3 *__restrict__ radiance, uint32_t sample_begin,
                                                             uint32_t sample_count, unsigned long long *ray_counter,
                                                             uint32_t *unit_queue, uint32_t unit_batch, strip_window win, uint32_t deal_rounds) {
    constexpr int kWaves = kBlock / 64;
    __shared__ uint32_t s_term[kWaves][64];          // per-pixel key terms of the unit being dealt

    const uint32_t tid = threadIdx.x;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u;
    const uint32_t strips_x = (vc.width + kTileW - 1) / kTileW;
    const uint32_t chunks = (sample_count + kChunkSamples - 1u) / kChunkSamples;     // units per strip
    const uint32_t n_strips = win.count;
    const uint32_t n_units = n_strips * chunks;

    // wave-uniform state
    uint32_t pool = 0, cursor = 0;                   // items of the current unit, next undealt one (cursor == pool: none left)
    bool queue_empty = false;
    uint32_t batch_next = 0, batch_end = 0;          // units taken from the queue and not yet started
    uint32_t strip = 0xffffffffu, strip_x0 = 0, strip_row0 = 0, chunk_begin = 0;      // no strip yet

    // lane state
    uint32_t nrays = 0, iterations = 0;
    uint32_t y = 0;
    size_t slot = 0;                                 // where this lane's path stores its radiance
    bool tracing = false;
    path_state ps;
    ps.o = ps.d = ps.thr = f3{0.0f, 0.0f, 0.0f};
    ps.rng = 0;
    ps.bounce = 0;

    bool pending = false;                            // the lane's ray has its nearest hit and waits to be shaded
    float closest = 3.402823466e+38f;
    int hit = -1;

    // gives the lanes without a path the next items of the pool, taking the next unit when the
    // pool is dealt out
    auto deal = [&]() {
        const unsigned long long want = __ballot(!tracing);
        if(want == 0ull) return;
        if(cursor >= pool && !queue_empty) {
            // Units are numbered chunk-major and taken kUnitBatch at a time: every wave of the
            // launch adds to the same counter, and atomics on one address go through one at a
            // time (130,000 single takes alone were 1.7 ms of a 1080p 16-spp launch).  The
            // units of a batch are neighbouring strips at the same samples, so a heavy strip's
            // chunks still go to different waves.
            if(batch_next >= batch_end) {
                uint32_t first = 0;
                if(lane == 0) first = atomicAdd(unit_queue, unit_batch);
                batch_next = __builtin_amdgcn_readfirstlane(first);
                batch_end = min(batch_next + unit_batch, n_units);
            }
            if(batch_next >= n_units) {
                queue_empty = true;
            } else {
                const uint32_t id = batch_next++;
                const uint32_t chunk = id / n_strips;
                const uint32_t new_strip = id - chunk * n_strips;       // strip of the window
                chunk_begin = chunk * kChunkSamples;
                pool = 64u * min(kChunkSamples, sample_count - chunk_begin);
                cursor = 0;
                if(new_strip != strip) {
                    strip = new_strip;
                    const uint32_t band_strip = win.begin + strip;
                    const uint32_t sy = band_strip / strips_x;
                    strip_x0 = (band_strip - sy * strips_x) * kTileW;
                    strip_row0 = sy * 4u;
                    __builtin_amdgcn_wave_barrier();
                    s_term[wave][lane] = pixel_term(vc.seed_hash, image_row(vc, strip_row0 + (lane >> 4)) * vc.width + strip_x0 + (lane & 15u));
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        if(cursor < pool) {
            const uint32_t before = __builtin_popcountll(want & ((1ull << lane) - 1ull));
            const uint32_t item = cursor + before;
            cursor += min(static_cast<uint32_t>(__builtin_popcountll(want)), pool - cursor);
            if(!tracing && item < pool) {
                const uint32_t pix = item & 63u;
                const uint32_t local = chunk_begin + (item >> 6);           // sample within the block
                const uint32_t x = strip_x0 + (pix & 15u);
                const uint32_t row = strip_row0 + (pix >> 4);
                if(x < vc.width && row < vc.row_count) {
                    y = image_row(vc, row);
                    slot = (static_cast<size_t>(strip) * win.stride + local) * 64u + pix;
                    start_path<EXACT>(vc, s_term[wave][pix], sample_terms, x, y, sample_begin + local, ps);
                    tracing = true;
                    const radiance3 first = radiance[slot];
                    closest = first.r;
                    hit = __float_as_int(first.g);
                    pending = true;
                    nrays++;
                }
                // an item of a pixel outside the image is simply dropped
            }
        }
    };
    // shades the lane's hit; a path that ends stores its radiance and frees the lane
    auto shade = [&](int hit_index, float hit_t) {
        f3 L;
        if(shade_segment<EXACT>(vc, y, hit_index, hit_t, geom, matl, kind, ps, L)) {
            radiance[slot] = radiance3{L.x, L.y, L.z};
            tracing = false;
        }
    };
    auto walk_list = [&](int &hit_index, float &hit_t) {
        float a2 = 2.0f, a4 = 4.0f;
        if constexpr(EXACT) {
            const float a = dot3<true>(ps.d, ps.d);     // ~1, not assumed 1 (src/shapes.cpp:16)
            a2 = 2.0f * a;
            a4 = 4.0f * a;
        }
        hit_t = 3.402823466e+38f;
        hit_index = -1;
        if constexpr(PREFILTER) intersect_prefiltered<EXACT>(vc, prefilter_list, geom, vc.n_spheres, ps.o, ps.d, a2, a4, hit_t, hit_index);
        else intersect_chunk<EXACT>(geom, vc.n_spheres, 0, ps.o, ps.d, a2, a4, hit_t, hit_index);
        nrays++;
    };

    for(;;) {
        // shade what has a hit, deal new paths to the lanes that came free; the new paths have a
        // hit to shade too, so go round again (a path dealt in the last round waits: its lane
        // sits out this iteration's list walk)
        for(uint32_t round = 0;; round++) {
            if(__any(pending)) {
                if(pending) {
                    pending = false;
                    shade(hit, closest);
                }
            }
            deal();
            if(round + 1 >= deal_rounds || !__any(pending)) break;
        }

        if(!__any(tracing)) {
            if(queue_empty && cursor >= pool) break;
            continue;               // only out-of-image items were dealt, or the next unit is due
        }
        iterations++;
        if(tracing && !pending) {
            walk_list(hit, closest);
            pending = true;
        }
    }

    if(vc.flags & AEHIP_FLAG_COUNT_RAYS) {
        uint32_t v = nrays;
#pragma unroll
        for(int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if(lane == 0) {
            atomicAdd(ray_counter, static_cast<unsigned long long>(v));
            atomicAdd(ray_counter + 1, 64ull * iterations);
        }
    }
}

// Adds one block's samples in sample order, then the block sum to the pixel's total; the last
// block's launch takes the mean and packs.  One wave per strip, lane l owns pixel l: the 64
// lanes read 768 contiguous bytes per sample.
__global__ __launch_bounds__(kBlock) void resolve_block_kernel(view_consts vc, const radiance3 *__restrict__ radiance,
                                                               radiance3 *__restrict__ totals, uint32_t sample_count,
                                                               uint32_t first_block, uint32_t last_block,
                                                               uint32_t *__restrict__ argb, float4 *__restrict__ rgba, strip_window win) {
    const uint32_t strips_x = (vc.width + kTileW - 1) / kTileW;
    const uint32_t wstrip = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);       // strip of the window
    const uint32_t lane = threadIdx.x & 63u;
    if(wstrip >= win.count) return;
    const uint32_t strip = win.begin + wstrip;
    const uint32_t sy = strip / strips_x;
    const uint32_t x = (strip - sy * strips_x) * kTileW + (lane & 15u);
    const uint32_t row = sy * 4u + (lane >> 4);
    if(x >= vc.width || row >= vc.row_count) return;

    const radiance3 *p = radiance + static_cast<size_t>(wstrip) * win.stride * 64u + lane;
    f3 block = {0.0f, 0.0f, 0.0f};
    for(uint32_t s = 0; s < sample_count; s++) {
        const radiance3 v = p[static_cast<size_t>(s) * 64u];
        block.x += v.r;
        block.y += v.g;
        block.z += v.b;
    }
    const size_t idx = static_cast<size_t>(row) * vc.width + x;
    const size_t tidx = static_cast<size_t>(wstrip) * 64u + lane;        // totals: per pixel of the window
    f3 sum = {0.0f, 0.0f, 0.0f};
    if(!first_block) {
        const radiance3 t = totals[tidx];
        sum = f3{t.r, t.g, t.b};
    }
    sum.x += block.x;
    sum.y += block.y;
    sum.z += block.z;
    if(last_block) {
        const float n = static_cast<float>(vc.spp);
        const float r = clamp01(sum.x / n);
        const float g = clamp01(sum.y / n);
        const float b = clamp01(sum.z / n);
        argb[idx] = pack_argb32(r, g, b);
        if(rgba) rgba[idx] = make_float4(r, g, b, 1.0f);
    } else {
        totals[tidx] = radiance3{sum.x, sum.y, sum.z};
    }
}

// ------------------------------------------------------------------------------------------
// Pixels of the single-sphere camera pass that the fast build must not render with its own
// arithmetic: those in which some sample position gives a ray within the screening margin of
// tangent, or a hit with conditioning value q below kExactBelowQ.  Decided per pixel from bounds
// over the pixel's whole jitter range, so the decision holds for every sample.  Set-up lists
// them (ring_pixels_kernel; entries x | band row << 16, sorted by the host); the fast build's
// launch renders the listed pixels with the exact build's code and leaves them out of its own.
// ------------------------------------------------------------------------------------------

// What the bounds over a pixel's sample positions say about its samples (single-sphere camera pass).
struct pixel_class {
    bool may_hit;               // false: every sample misses, by more than the screening margin
    bool well_conditioned;      // every sample hits, with conditioning value q >= kExactBelowQ
    bool faces_sphere;          // h = draw.oc > 0 at the pixel (h keeps its sign over a pixel whose samples all hit)
    bool newton_ok;             // the hit distance may be carried from cell to cell by one Newton step (kNewtonTolerance)
    bool newton_twice_ok;       // ... by two steps a cell
    bool newton_flat_ok;        // ... by one step with 1 / sqrt(disc) constant along the row
};

// The fast build's hit loop carries the hit distance t of the unnormalised ray along a row of
// cells: from the previous cell's t one Newton step on dd t*t - 2 h t + c = 0, whose derivative
// at the near root is -2 sqrt(disc); 1 / sqrt(disc) is expanded to first order about the middle of
// the row.  No square root and no reciprocal per sample (quarter-rate instructions that cost a mix of
// full-rate ones 13-15 clocks each, tools/valu_peak.hip).  With dt the distance's change from
// one cell to the next and eps the relative error of the expanded 1 / sqrt(disc), a step leaves
// t within eps dt + dt*dt dd / (2 sqrt(disc)) of the root.  classify_pixel bounds that over the
// pixel from the discriminant's gradient and admits the pixel when the colour moves by less
// than this (other pixels take the loop that computes every root from scratch).
constexpr float kNewtonTolerance = 1e-6f;

__device__ __forceinline__ pixel_class classify_pixel(const view_consts &vc, const camera_consts &cc, uint32_t x, uint32_t y) {
    const bool jitter = (vc.flags & AEHIP_FLAG_JITTER) != 0;
    // the sample positions of the pixel span [u0, u0 + eu] x [v0, v0 + ev] on the viewport,
    // measured from the camera (a single point without jitter)
    const float xb = static_cast<float>(x) + (jitter ? 0.0f : 0.5f), yb = static_cast<float>(y) + (jitter ? 0.0f : 0.5f);
    const float u0 = __builtin_fmaf(xb, vc.psx, -(vc.vw_half + vc.cam[0]));
    const float v0 = __builtin_fmaf(yb, vc.psy, -(vc.vh_half + vc.cam[1]));
    const float eu = jitter ? vc.psx : 0.0f, ev = jitter ? vc.psy : 0.0f;
    const float dz = 0.0f - vc.cam[2];
    float disc_min = 3.402823466e+38f, disc_max = -3.402823466e+38f, big_max = 0.0f, h_corner = 0.0f;
    float h_min = 3.402823466e+38f, dd_max = 0.0f, slope_max = 0.0f;
#pragma unroll
    for(int k = 0; k < 4; k++) {
        const float u = u0 + ((k & 1) ? eu : 0.0f), v = v0 + ((k & 2) ? ev : 0.0f);
        const float dd = __builtin_fmaf(v, v, __builtin_fmaf(u, u, dz * dz));
        const float h = __builtin_fmaf(v, cc.oc[1], __builtin_fmaf(u, cc.oc[0], dz * cc.oc[2]));
        const float disc = __builtin_fmaf(-dd, cc.c, h * h);
        disc_min = fminf(disc_min, disc);
        disc_max = fmaxf(disc_max, disc);
        big_max = fmaxf(big_max, __builtin_fmaf(dd, cc.magnitude, h * h));
        if(k == 0) h_corner = h;
        h_min = fminf(h_min, h);
        dd_max = fmaxf(dd_max, dd);
        // d disc / d u, linear in (u, v): largest at a corner
        slope_max = fmaxf(slope_max, fabsf(__builtin_fmaf(2.0f * cc.q_uu, u, __builtin_fmaf(cc.q_uv, v, cc.q_u))));
        // (and d disc / d v, in units of the step along u: the pixel is psy tall and psx wide)
        slope_max = fmaxf(slope_max, fabsf(__builtin_fmaf(2.0f * cc.q_vv, v, __builtin_fmaf(cc.q_uv, u, cc.q_v))) * (vc.psy * __builtin_amdgcn_rcpf(vc.psx)));
    }
    // smallest dd over the rectangle: each coordinate at its smallest magnitude
    const float ua = (u0 <= 0.0f && u0 + eu >= 0.0f) ? 0.0f : fminf(fabsf(u0), fabsf(u0 + eu));
    const float va = (v0 <= 0.0f && v0 + ev >= 0.0f) ? 0.0f : fminf(fabsf(v0), fabsf(v0 + ev));
    const float dd_min = __builtin_fmaf(va, va, __builtin_fmaf(ua, ua, dz * dz));
    // the discriminant is a quadratic in (u, v): inside the rectangle it leaves the range of
    // the corner values by at most its second-order terms; plus the evaluations' own rounding
    const float pad = fabsf(cc.q_uu) * eu * eu + fabsf(cc.q_uv) * eu * ev + fabsf(cc.q_vv) * ev * ev + 1e-5f * big_max;
    pixel_class pc;
    // (the magnitude h*h + dd*(oc.oc + r*r) is a sum of squares of linear forms: over the rectangle it
    // is largest at a corner, so big_max bounds it for every sample)
    pc.may_hit = disc_max + pad >= -(kScreenMargin * 1.01f) * big_max;
    // q >= kExactBelowQ for every sample of the pixel, with 5 % to spare; false for NaN / inf: such pixels are listed
    pc.well_conditioned = (disc_min - pad) * (dd_min * cc.radius2) >= (kExactBelowQ * 1.05f) * (big_max * big_max);
    pc.faces_sphere = h_corner > 0.0f;
    // Newton's step along a row of cells (see kNewtonTolerance): the widest cell of the pixel's grid
    // is a ragged last row's
    const uint32_t last_row_cells = vc.spp - (((vc.spp - 1u) >> vc.row_bits) << vc.row_bits);
    // (These are bounds with a factor to spare, compared with a tolerance: the hardware's reciprocal, reciprocal square
    // root and square root -- an ulp off -- do for them what a dozen divisions and two square roots in full did, at a
    // tenth of the instructions; s_min = 0, h_min <= 0, dd_min = 0 still end in inf or NaN and so in "no".)
    const float du = jitter ? vc.psx * __builtin_amdgcn_rcpf(static_cast<float>(min(1u << vc.row_bits, last_row_cells))) : 0.0f;
    const float inv_s_min = __builtin_amdgcn_rsqf(fmaxf(disc_min - pad, 0.0f));          // 1 / sqrt of the smallest discriminant
    const float ds = slope_max * du * (0.5f * inv_s_min);               // change of sqrt(disc) over a cell
    const float t_max = cc.c * __builtin_amdgcn_rcpf(h_min);             // t = c / (h + sqrt(disc))
    const float u_abs = fmaxf(fabsf(u0), fabsf(u0 + eu));
    const float dt = (fabsf(cc.oc[0]) * du + ds + 2.0f * t_max * u_abs * du) * __builtin_amdgcn_rcpf(dd_min);
    // 1 / sqrt(disc) comes from its first-order expansion about the pixel's centre.  With
    // g = the relative change of disc across the pixel (2 cells ds / s_min along either axis;
    // slope_max covers both) the second-order term left out is 3/8 (g/2)^2 of it at an edge's
    // middle, twice that in a corner: under g*g / 4.  A row's first T comes from such an expansion
    // too (left out: about (t / (h + sqrt(disc))) (cells ds)^2 / (8 sqrt(disc)) an axis, under
    // g cells dt / 4), then takes a step like the others.
    const float cells = static_cast<float>(1u << vc.row_bits);
    const float pixel_change = 2.0f * cells * ds * inv_s_min;
    const float eps = 0.25f * pixel_change * pixel_change;
    const float start = dt + 0.25f * pixel_change * cells * dt;
    const float curvature = dd_max * (0.5f * inv_s_min);                // dd / (2 sqrt(disc)): the step's second-order term is this times dt*dt
    const float t_error = 2.0f * (eps * start + start * start * curvature);
    // colour = draw * (shade_scale * t) + constant; false for NaN (s_min = 0, h_min <= 0 give inf or a negative bound)
    const float colour_per_t = cc.shade_scale * __builtin_amdgcn_sqrtf(dd_max);
    const bool sane = h_min > 0.0f && cc.c > 0.0f && eps <= 0.05f;
    pc.newton_ok = sane && colour_per_t * t_error <= kNewtonTolerance;
    // a second step starts where the first ended (t_error carries a factor 2 to spare already)
    const float t_error_twice = 2.0f * (eps * t_error + t_error * t_error * curvature);
    pc.newton_twice_ok = sane && t_error <= 0.25f * start && colour_per_t * t_error_twice <= kNewtonTolerance;
    // with 1 / sqrt(disc) taken as constant along a row of cells (its value in the pixel's middle
    // column): off by its relative change across the pixel on top of the expansion's own error
    const float eps_flat = eps + cells * ds * inv_s_min;
    pc.newton_flat_ok = pc.newton_ok && colour_per_t * 2.0f * (eps_flat * start + start * start * curvature) <= kNewtonTolerance;
    return pc;
}

// True when no sample of the pixels [x0, x1) x [y0, y1) (image rows) can hit: the corner bound of
// classify_pixel over the whole rectangle, with twice its padding -- so every pixel inside is
// one classify_pixel would call a miss as well, and none of them is an ill-conditioned one.
__device__ __forceinline__ bool rectangle_misses(const view_consts &vc, const camera_consts &cc, uint32_t x0, uint32_t x1, uint32_t y0,
                                                 uint32_t y1) {
    const bool jitter = (vc.flags & AEHIP_FLAG_JITTER) != 0;
    const float shift = jitter ? 0.0f : 0.5f;
    const float u0 = __builtin_fmaf(static_cast<float>(x0) + shift, vc.psx, -(vc.vw_half + vc.cam[0]));
    const float v0 = __builtin_fmaf(static_cast<float>(y0) + shift, vc.psy, -(vc.vh_half + vc.cam[1]));
    const float eu = static_cast<float>(x1 - x0 - (jitter ? 0u : 1u)) * vc.psx, ev = static_cast<float>(y1 - y0 - (jitter ? 0u : 1u)) * vc.psy;
    const float dz = 0.0f - vc.cam[2];
    float disc_max = -3.402823466e+38f, big_max = 0.0f;
#pragma unroll
    for(int k = 0; k < 4; k++) {
        const float u = u0 + ((k & 1) ? eu : 0.0f), v = v0 + ((k & 2) ? ev : 0.0f);
        const float dd = __builtin_fmaf(v, v, __builtin_fmaf(u, u, dz * dz));
        const float h = __builtin_fmaf(v, cc.oc[1], __builtin_fmaf(u, cc.oc[0], dz * cc.oc[2]));
        disc_max = fmaxf(disc_max, __builtin_fmaf(-dd, cc.c, h * h));
        big_max = fmaxf(big_max, __builtin_fmaf(dd, cc.magnitude, h * h));
    }
    const float pad = fabsf(cc.q_uu) * eu * eu + fabsf(cc.q_uv) * eu * ev + fabsf(cc.q_vv) * ev * ev + 1e-5f * big_max;
    return disc_max + 2.0f * pad < -(kScreenMargin * 1.01f) * big_max;       // false for NaN
}

__device__ __forceinline__ bool pixel_needs_reference(const view_consts &vc, const camera_consts &cc, uint32_t x, uint32_t y) {
    const pixel_class pc = classify_pixel(vc, cc, x, y);
    return pc.may_hit && !pc.well_conditioned;
}

__global__ __launch_bounds__(256) void ring_pixels_kernel(view_consts vc, camera_consts cc, uint32_t *__restrict__ count,
                                                          uint32_t *__restrict__ list, uint32_t capacity) {
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    bool ring = false;
    uint32_t x = 0, row = 0;
    if(gid < vc.width * vc.row_count) {
        row = gid / vc.width;
        x = gid - row * vc.width;
        ring = pixel_needs_reference(vc, cc, x, image_row(vc, row));
    }
    // one atomic per wave
    const unsigned long long mask = __ballot(ring);
    if(mask != 0ull) {
        const uint32_t lane = threadIdx.x & 63u;
        uint32_t base = 0;
        if(lane == static_cast<uint32_t>(__builtin_ctzll(mask))) base = atomicAdd(count, static_cast<uint32_t>(__builtin_popcountll(mask)));
        base = __shfl(base, __builtin_ctzll(mask), 64);
        if(ring) {
            const uint32_t slot = base + static_cast<uint32_t>(__builtin_popcountll(mask & ((1ull << lane) - 1ull)));
            if(list && slot < capacity) list[slot] = x | (row << 16);
        }
    }
}

// Pixels of the band that the single-sphere camera pass walks through a sample loop: all but those
// camera_body culls from the same classification (every sample misses; fast build: also a sphere
// that only the far side of the camera faces).  Listed pixels are walked, in the reference's
// arithmetic.
__global__ __launch_bounds__(256) void walked_pixels_kernel(view_consts vc, camera_consts cc, uint32_t fast, uint32_t *__restrict__ count) {
    // a few hundred workgroups stride over the band: atomics on one word go through one at a time, so one a workgroup
    __shared__ uint32_t s_count;
    if(threadIdx.x == 0u) s_count = 0u;
    __syncthreads();
    const unsigned long long pixels = static_cast<unsigned long long>(vc.width) * vc.row_count;
    uint32_t mine = 0;
    for(unsigned long long gid = blockIdx.x * 256u + threadIdx.x; gid < pixels; gid += gridDim.x * 256u) {
        const uint32_t row = static_cast<uint32_t>(gid / vc.width), x = static_cast<uint32_t>(gid - static_cast<unsigned long long>(row) * vc.width);
        const pixel_class pc = classify_pixel(vc, cc, x, image_row(vc, row));
        bool culled = !pc.may_hit;
        if(fast != 0u && cc.near_root_only != 0u && pc.may_hit && pc.well_conditioned && !pc.faces_sphere) culled = true;
        mine += culled ? 0u : 1u;
    }
#pragma unroll
    for(int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
    if((threadIdx.x & 63u) == 0u && mine != 0u) atomicAdd(&s_count, mine);
    __syncthreads();
    if(threadIdx.x == 0u && s_count != 0u) atomicAdd(count, s_count);
}

#ifdef AEHIP_WAVE_TIMELINE
// Measurement build (make timeline; tools/wave_timeline.py): every wave of the fast build's
// single-sphere launch leaves its start and end on the constant-rate clock, the shader clocks between
// the two and whether it rendered listed pixels, indexed by workgroup and wave (no atomics: they would serialise).
constexpr uint32_t kTimelineWaves = 1u << 18;
__device__ unsigned long long g_timeline[kTimelineWaves * 3];
__device__ unsigned int g_timeline_waves;
struct timeline_stamp {
    unsigned long long t0;
    uint32_t kind;
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    __device__ ~timeline_stamp() {
        if((threadIdx.x & 63u) == 0u) {
            const uint32_t i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
            if(i < kTimelineWaves) {
                g_timeline[3u * i] = t0;
                g_timeline[3u * i + 1u] = __builtin_amdgcn_s_memrealtime();
                g_timeline[3u * i + 2u] = kind | ((__builtin_amdgcn_s_memtime() - c0) << 8);
            }
            if(i == 0u) g_timeline_waves = gridDim.x * (blockDim.x >> 6);
        }
    }
};
#endif

// camera_body's split_and_tile argument: waves per strip (log2) in the low byte and
constexpr uint32_t kCameraTiled = 0x100u;      // a workgroup owns a 16x16 tile and renders its strips in rounds
// Listed pixels of the fast build: the reference's arithmetic beyond the discriminant too.  The
// fast tail (hardware square root and reciprocals) leaves the hit distance t off by a few 1e-7
// of itself, and the hit point minus the centre, of length r, takes that error times t: harmless
// at t / r of a few (the reference's scene: 2 to 4), a few 1e-5 of the colour at 250.  Set where
// the sphere is more than kFastTailMaxDistance radii from the camera.
constexpr uint32_t kCameraExactTail = 0x200u;
constexpr float kFastTailMaxDistance = 10.0f;

// ------------------------------------------------------------------------------------------
// camera pass: the path pass specialised for scenes in which every surface ends the path
// (all materials NORMAL -- the reference's scene).  Every path is one segment from the camera,
// so the sample loop is wave-uniform, the per-sphere terms that only depend on the ray origin
// (oc = C - camera, c = oc.oc - r*r) are computed once per workgroup, and the sample key is the
// per-pixel term plus the tabulated per-sample term, fetched through the scalar cache one
// iteration ahead.  Results are the same bits as path_kernel's (exact build) for such scenes.
//
// Work split: a wave renders a 16x4 pixel strip, one pixel per lane.  The samples of a pixel are
// summed in blocks of AEHIP_SAMPLE_BLOCK, and blocks are independent, so 2^split_log2 waves of
// the workgroup share a strip and take its blocks in turn.  Waves are that much shorter and that
// many more, which shortens the launch's tail (the time the last waves run on a chip that is
// emptying) in proportion -- a third of the run time of a 4K band on one of 8 GPUs otherwise.
// The exact build adds block sums in block order: the waves of a strip meet after every round of
// blocks and each adds the round's sums to its copy of the total.  The fast build keeps one sum
// per wave and adds them once at the end.
// ------------------------------------------------------------------------------------------

//
// Pixel list: with `pixel_list` a workgroup renders listed pixels (x | band row << 16) instead of
// a rectangle, 64 consecutive entries taking a strip's place, group_x counting the groups.
// SKIP_LISTED: pixels that pixel_needs_reference() are left out (the same launch renders them
// from the list with the exact build's arithmetic, see camera_hybrid_kernel).
// a pixel's mean colour from the sum of its samples (alpha 1)
__device__ __forceinline__ float4 mean_colour(const view_consts &vc, f3 total) {
    const float n = static_cast<float>(vc.spp);
    // the mean is a true division in both builds (once per pixel): an approximate reciprocal
    // would turn a row of exact 1.0 samples into 0.99999994 and the truncating pack into 254
    return make_float4(clamp01(total.x / n), clamp01(total.y / n), clamp01(total.z / n), 1.0f);
}

// fast build: the sum of a pixel all of whose samples miss -- the row background times the sample count
__device__ __forceinline__ f3 background_total(const view_consts &vc, uint32_t py) {
    const f3 bg_row = row_background<false>(vc, py);
    const float misses = static_cast<float>(vc.spp);
    return f3{misses * bg_row.x, misses * bg_row.y, misses * bg_row.z};
}

// Fast build, one sphere: the background outside the box of workgroups in which the sphere can be
// hit (sphere_box, host side), written by the last workgroups of the launch instead of by one
// workgroup a tile (fill_around_box).  (A 4K frame's background: 100,000 waves of 64 pixels each,
// 50 us of a launch bound by how fast waves start, against 540 workgroups that store at the
// memory's rate.)
__device__ __forceinline__ void fill_background(const view_consts &vc, uint32_t f, uint32_t rows_per_group, uint32_t row_order,
                                                uint32_t box_cols, uint32_t *__restrict__ argb, float4 *__restrict__ rgba) {
    fill_around_box(vc, f, rows_per_group, row_order, box_cols, argb, rgba,
                    [&](uint32_t y) { return mean_colour(vc, background_total(vc, y)); });
}

template <bool EXACT, bool ONE, bool FOLD, bool AXIAL, bool SKIP_LISTED, bool FAST_AFTER_DISC = false>
__device__ __forceinline__ void camera_body(const view_consts &vc, const camera_consts &cc, uint32_t split_and_tile,
                                            const float4 *__restrict__ geom, const uint32_t *__restrict__ sample_terms,
                                            uint32_t *__restrict__ argb, float4 *__restrict__ rgba, uint32_t group_x,
                                            uint32_t group_y, const uint32_t *__restrict__ pixel_list, uint32_t list_count,
                                            uint32_t list_row_bias, const float4 *__restrict__ row_miss) {
    constexpr uint32_t kWaves = kBlock / 64;
    __shared__ float4 s_occ[ONE ? 1 : kLdsSpheres];     // oc.xyz, c
    __shared__ float4 s_ctr[ONE ? 1 : kLdsSpheres];     // centre.xyz, oc.oc + r*r
    __shared__ float s_part[EXACT ? 2 : 1][kWaves][3][64];      // per-wave sums handed to the strip's other waves
    __shared__ uint32_t s_hits[EXACT ? 1 : kWaves][64];

    const uint32_t tid = threadIdx.x;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u;
    // A pixel's sample blocks are summed in `shares` interleaved shares (block b belongs to share
    // b mod shares), and the shares are added in order.  Usually a strip's `shares` waves render
    // one share each.  One sphere, rectangle launch, kCameraTiled: every wave renders a strip of
    // its own, share after share -- same sums, same bits -- so the workgroup covers a 16x16 tile and
    // its waves share nothing: no classes through LDS, no barrier, and a tile around which
    // nothing is hit costs one short workgroup of busy waves instead of `shares` workgroups with
    // one busy wave each.  (The exact build adds block sums in block order whatever the split.)
    const uint32_t shares = 1u << (split_and_tile & 0xffu);
    const bool tiled = ONE && pixel_list == nullptr && (split_and_tile & kCameraTiled) != 0u;
    const uint32_t split_log2 = tiled ? 0u : (split_and_tile & 0xffu);
    const uint32_t split = 1u << split_log2;                    // waves per strip
    const uint32_t part = wave & (split - 1u);                  // this wave's place among the strip's waves
    const uint32_t strip = wave >> split_log2;                  // strip of the workgroup
    const uint32_t strips_per_group = kWaves >> split_log2;
    const f3 cam = {vc.cam[0], vc.cam[1], vc.cam[2]};
    const uint32_t n_spheres = ONE ? 1u : vc.n_spheres;
    const uint32_t nblocks = (vc.spp + AEHIP_SAMPLE_BLOCK - 1u) / AEHIP_SAMPLE_BLOCK;

    auto store_pixel = [&](uint32_t px, uint32_t prow, f3 total) {
        const float4 mean = mean_colour(vc, total);
        const size_t idx = static_cast<size_t>(prow) * vc.width + px;
        argb[idx] = pack_argb32(mean.x, mean.y, mean.z);
        if(rgba) rgba[idx] = mean;
    };
    // a pixel all of whose samples miss: the row background summed spp times -- in the fast build a
    // product, in the exact build the sum in sample order that set-up tabulated per image row
    auto all_miss_total = [&](uint32_t py) {
        if constexpr(EXACT) {
            const float4 t = row_miss[py];
            return f3{t.x, t.y, t.z};
        } else {
            return background_total(vc, py);
        }
    };

    // One sphere: bounds over the pixel's sample positions (classify_pixel) say of all but the
    // pixels along the silhouette that every sample misses or that every sample hits.
    //  - every sample misses: the pixel stays out of the sample loop (all_miss_total; without
    //    set-up's table the exact build culls nothing).
    //  - every sample hits, well conditioned, near root (fast build): a hit loop, which neither
    //    tests nor branches.
    // Everything else goes through the general loop.
    // A strip is classified by its first wave (a class costs about a hundred vector instructions),
    // which writes the strip at once when none of its pixels needs the sample loop; the strip's
    // other waves get the classes through LDS.  Before that every wave tests the workgroup's whole
    // rectangle: around most of the frame nothing can be hit, and the workgroup ends without
    // classes or a barrier.
    constexpr uint32_t kCulled = 1u, kSureHit = 2u, kListed = 4u, kNewton = 8u, kNewtonTwice = 16u, kNewtonFlat = 32u;
    __shared__ uint32_t s_class[ONE ? kWaves : 1][64];
    __shared__ uint32_t s_strip_work[kWaves];
    uint32_t x = group_x * kTileW + (lane & 15u);
    uint32_t row = (group_y * strips_per_group + strip) * 4u + (lane >> 4);
    bool valid = (x < vc.width) && (row < vc.row_count);
    // Listed pixels, blocks of whole pairs of rows: a pixel has two lanes, 32 apart, which walk the
    // two rows of a pair at the same time and add their sums -- half as many steps a wave, twice
    // as many waves.  (The listed pixels' waves are the longest of the launch, 20,000 dependent
    // instructions in the reference's arithmetic at 1024 samples; in an eighth of a 4K frame the
    // other waves were done after 62 us and these after 84.)
    const bool halves = FAST_AFTER_DISC && pixel_list != nullptr && blocks_are_pairs_of_rows(vc);
    const bool fast_tail = FAST_AFTER_DISC && (split_and_tile & kCameraExactTail) == 0u;       // the same in every lane
    const uint32_t half = halves ? lane >> 5 : 0u;
    if(pixel_list) {
        const uint32_t entry = halves ? (group_x * strips_per_group + strip) * 32u + (lane & 31u) : (group_x * strips_per_group + strip) * 64u + lane;
        valid = entry < list_count;
        const uint32_t packed = valid ? pixel_list[entry] : 0u;
        x = packed & 0xffffu;
        row = (packed >> 16) - list_row_bias;      // entries hold rows of the whole band; a launch may render part of it
    }
    const uint32_t y = image_row(vc, row);
    bool culled = false, sure_hit = false, newton = false, newton_twice = false, newton_flat = false;
    if constexpr(ONE) {
        if(pixel_list == nullptr && (!EXACT || row_miss != nullptr)) {
            const uint32_t grow0 = group_y * strips_per_group * 4u;
            const uint32_t grow1 = min(grow0 + strips_per_group * 4u, vc.row_count);       // the grid has no workgroup past the last row
            if(!(vc.flags & AEHIP_FLAG_WALK_ALL) &&
               rectangle_misses(vc, cc, group_x * kTileW, min(group_x * kTileW + kTileW, vc.width), image_row(vc, grow0), image_row(vc, grow1 - 1u) + 1u)) {
                if(valid && part == 0u) store_pixel(x, row, all_miss_total(y));
                return;
            }
            uint32_t cls = 0;
            bool strip_work = false;
            if(part == 0u) {
                const pixel_class pc = classify_pixel(vc, cc, x, y);
                if(SKIP_LISTED && pc.may_hit && !pc.well_conditioned) cls |= kListed;
                if(!pc.may_hit) cls |= kCulled;
                if constexpr(!EXACT) {
                    if(cc.near_root_only != 0u && pc.may_hit && pc.well_conditioned)
                        cls |= pc.faces_sphere ? (pc.newton_flat_ok ? (kNewton | kNewtonFlat) : pc.newton_ok ? kNewton : pc.newton_twice_ok ? (kNewton | kNewtonTwice) : kSureHit)
                                               : kCulled;    // behind the camera: no root counts
                }
                if constexpr(EXACT) {
                    if(cc.short_exact_ok != 0u && pc.may_hit && pc.well_conditioned && pc.faces_sphere) cls |= kSureHit;
                }
                if(vc.flags & AEHIP_FLAG_WALK_ALL) cls &= kListed;      // every pixel through the general loop
                strip_work = __ballot(valid && (cls & (kCulled | kListed)) == 0u) != 0ull;
                if(!strip_work && valid && (cls & kListed) == 0u) store_pixel(x, row, all_miss_total(y));
            }
            if(split == 1u) {
                if(!strip_work) return;         // (no barrier follows in a workgroup whose waves share nothing)
            } else {
                if(part == 0u) {
                    s_class[strip][lane] = cls;
                    if(lane == 0u) s_strip_work[strip] = strip_work ? 1u : 0u;
                }
                __syncthreads();
                cls = s_class[strip][lane];
                uint32_t any = 0;
                for(uint32_t i = 0; i < strips_per_group; i++) any |= s_strip_work[i];
                if(any == 0u) return;           // every strip of the workgroup has been written
                strip_work = s_strip_work[strip] != 0u;
            }
            valid = valid && (cls & kListed) == 0u && strip_work;
            culled = (cls & kCulled) != 0u;
            sure_hit = (cls & kSureHit) != 0u;
            newton = (cls & kNewton) != 0u;
            newton_twice = (cls & kNewtonTwice) != 0u;
            newton_flat = (cls & kNewtonFlat) != 0u;
        }
    }
    const bool work = valid && !culled;

    if constexpr(!ONE) {
        for(uint32_t i = tid; i < n_spheres; i += kBlock) {
            const float4 g = geom[i];
            const f3 oc = {g.x - cam.x, g.y - cam.y, g.z - cam.z};
            const float occ = dot3<EXACT>(oc, oc);
            s_occ[i] = make_float4(oc.x, oc.y, oc.z, occ - g.w);
            s_ctr[i] = make_float4(g.x, g.y, g.z, occ + g.w);
        }
        __syncthreads();
    }

    (void)sample_terms;     // every path of this pass ends at its first hit: no path ever draws a random number
    const f3 bg = row_background<EXACT>(vc, y);
    const uint32_t pix_term = pixel_term(vc.seed_hash, y * vc.width + x);
    // Sample positions (see sample_offset): the pixel's samples sit on a grid, the cell index is
    // the same in every lane, and a row of cells shares its vertical position.  So the sample loop
    // runs row by row: what depends on the row (the ray's v coordinate and everything built from
    // it) is worked out once per row, and going to the next cell is one vector add.
    const bool jitter = (vc.flags & AEHIP_FLAG_JITTER) != 0;
    const pixel_offset off = offset_of_pixel(pix_term);
    const float xf = static_cast<float>(x), yf = static_cast<float>(y);
    const float dz = 0.0f - cam.z;
    const float dz2 = dz * dz;
    const uint32_t row_cells = 1u << vc.row_bits;
    // the ray's viewport offsets of a sample at (jx, jy) in the pixel, folded into one fma per coordinate:
    // u = ux0 + psx * jx, v = uy0 + psy * jy
    const float ux0 = __builtin_fmaf(xf, vc.psx, -(vc.vw_half + cam.x));
    const float uy0 = __builtin_fmaf(yf, vc.psy, -(vc.vh_half + cam.y));

    // One sphere: the discriminant h*h - dd*c of the unnormalised ray (u, v, dz) is a quadratic in
    // (u, v) with the coefficients of camera_consts; along a row of cells v is fixed and
    //   disc(u) = u * (q_uu * u + B) + C,   B = q_uv v + q_u,   C = v (q_vv v + q_v) + q_0:
    // two fmas per sample.  Every term of the expansion is at most a few times
    // M = h*h + dd*(oc.oc + r*r), so this evaluation is, like the direct one, within a few dozen
    // unit roundoffs of M of the real value.  The exact build screens with it: its constant term
    // takes the margin times an upper bound of M over the pixel (all terms taken positive at the
    // corner of the pixel farthest from the axes).
    float screen_margin = 0.0f;
    if constexpr(ONE && EXACT) {
        const float ua = fmaxf(fabsf(ux0), fabsf(ux0 + vc.psx));
        const float va = fmaxf(fabsf(uy0), fabsf(uy0 + vc.psy));
        const float hmax = __builtin_fmaf(ua, fabsf(cc.oc[0]), __builtin_fmaf(va, fabsf(cc.oc[1]), fabsf(dz * cc.oc[2])));
        const float ddmax = __builtin_fmaf(ua, ua, __builtin_fmaf(va, va, dz2));
        screen_margin = (kScreenMargin * 1.001f) * __builtin_fmaf(ddmax, cc.magnitude, hmax * hmax);
    }

    // a row of cells inside the sample range [s, s_end): which cells, and where they sit
    struct cell_row {
        uint32_t count;         // cells of the row inside the range
        uint32_t first_cell;    // index of the first of them
        uint32_t cells;         // cells in the whole row
        float first;            // first_cell as a float
        float inv_cells;        // 1 / cells (exact for a full row)
        float jy;               // vertical position of the row's samples in the pixel
    };
    // (the integers are the same in every lane; saying so keeps the sample loops' counters and
    // branches in the scalar unit wherever the loops sit)
    auto row_at = [&](uint32_t s, uint32_t s_end, uint32_t rows_down = 0u) {     // rows_down: this lane's row is that many below
        cell_row cr;
        if(!jitter) {           // every sample at the pixel centre: one "row" for the whole range
            cr.count = __builtin_amdgcn_readfirstlane(s_end - s);
            cr.first_cell = 0u;
            cr.cells = 1u;
            cr.first = 0.0f;
            cr.inv_cells = 0.0f;
            cr.jy = 0.5f;
            return cr;
        }
        const uint32_t r = __builtin_amdgcn_readfirstlane(s >> vc.row_bits);
        const uint32_t n_r = __builtin_amdgcn_readfirstlane(cells_in_row(vc, r));
        cr.count = __builtin_amdgcn_readfirstlane(min(s_end, (r << vc.row_bits) + n_r) - s);
        cr.first_cell = __builtin_amdgcn_readfirstlane(s & (row_cells - 1u));
        cr.cells = n_r;
        cr.first = static_cast<float>(cr.first_cell);
        // (a power of two in all rows but a ragged last one: its reciprocal comes from the exponent,
        // in the scalar unit, instead of a division)
        cr.inv_cells = (n_r & (n_r - 1u)) == 0u ? __uint_as_float((127u - static_cast<uint32_t>(__builtin_ctz(n_r))) << 23)
                                                : __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(1.0f / static_cast<float>(n_r))));
        cr.jy = (static_cast<float>((r + rows_down) << vc.row_bits) + off.ry * static_cast<float>(n_r)) * vc.inv_spp;
        return cr;
    };

    f3 sum = {0.0f, 0.0f, 0.0f};

    auto write_pixel = [&]() {
        if(valid && part == 0u && half == 0u) {
            if constexpr(EXACT && ONE) {
                if(culled) sum = all_miss_total(y);
            }
            store_pixel(x, row, sum);
        }
    };

    if constexpr(EXACT) {
        // Exact build, one sphere, a wave all of whose pixels are hit by every sample, well conditioned,
        // at the near root beyond t_min (classify_pixel; short_exact_ok): the reference's sample without
        // the screening, the sign and range tests and the choice between hit and background --
        // they are decided -- and with the square root and the division through the short sequences
        // of rt_exact_math.h, whose operands are in range here.  The same operations on the same
        // values as the general loop below, so the same bits; 15 of its 150 instructions a sample fewer.
        bool wave_sure = false;
        if constexpr(ONE && !FAST_AFTER_DISC) wave_sure = pixel_list == nullptr && __ballot(valid && !sure_hit) == 0ull && __ballot(valid) != 0ull;
        for(uint32_t first = 0; first < nblocks; first += split) {      // a round: `split` consecutive blocks
            const uint32_t blk = first + part;
            f3 block = {0.0f, 0.0f, 0.0f};
            if(blk < nblocks && wave_sure) {
                const f3 oc = {cc.oc[0], cc.oc[1], cc.oc[2]};
                const f3 ctr = {cc.center[0], cc.center[1], cc.center[2]};
                const bool straight_ahead = cc.oc[0] == 0.0f && cc.oc[1] == 0.0f && dz != 0.0f;     // (short_exact_ok: |oc| >= 2^-30, so oc.z != 0)
                const uint32_t s0 = blk * AEHIP_SAMPLE_BLOCK;
                const uint32_t s1 = min(vc.spp, s0 + AEHIP_SAMPLE_BLOCK);
                for(uint32_t s = s0; s < s1;) {
                    const cell_row cr = row_at(s, s1);
                    s += cr.count;
                    const float py = yf + cr.jy;
                    const float draw_y = (py * vc.psy - vc.vh_half) - cam.y;      // src/software_raytracer.cpp:218-221
                    float cell = cr.first;
                    for(uint32_t k = 0; k < cr.count; k++, cell += 1.0f) {
                        const float jx = jitter ? (cell + off.rx) * cr.inv_cells : 0.5f;
                        const float px = xf + jx;
                        const f3 d = normalize3<true>(f3{(px * vc.psx - vc.vw_half) - cam.x, draw_y, dz});
                        const float a = dot3<true>(d, d);                   // src/shapes.cpp:14-25
                        // (centre straight ahead of the camera: (d.x*0 + d.y*0) + d.z*oc.z is d.z*oc.z, a zero added to a non-zero)
                        const float b = -2.0f * (straight_ahead ? d.z * oc.z : dot3<true>(d, oc));
                        const float disc = b * b - (4.0f * a) * cc.c;
                        const float t = rt_exact::div_tame(-b - rt_exact::sqrt_tame(disc), 2.0f * a);
                        const f3 pt = {d.x * t + cam.x, d.y * t + cam.y, d.z * t + cam.z};
                        const f3 n = normalize3<true>(f3{pt.x - ctr.x, pt.y - ctr.y, pt.z - ctr.z});
                        block.x += remap_unit(n.x);
                        block.y += remap_unit(n.y);
                        block.z += remap_unit(n.z);
                    }
                }
            } else if(blk < nblocks && work) {
                const uint32_t s0 = blk * AEHIP_SAMPLE_BLOCK;
                const uint32_t s1 = min(vc.spp, s0 + AEHIP_SAMPLE_BLOCK);
                for(uint32_t s = s0; s < s1;) {
                    const cell_row cr = row_at(s, s1, half);
                    s += halves ? 2u * cr.count : cr.count;
                    // the row's part of the sample's ray exactly as the reference builds it
                    // (src/software_raytracer.cpp:218-221), and of the folded screening ray
                    const float py = yf + cr.jy;
                    const float draw_y = (py * vc.psy - vc.vh_half) - cam.y;
                    const float v = __builtin_fmaf(cr.jy, vc.psy, uy0);
                    const float du = vc.psx * cr.inv_cells;
                    const float u_first = __builtin_fmaf(jitter ? off.rx * cr.inv_cells : 0.5f, vc.psx, ux0);
                    float qb = 0.0f, qc = 0.0f;
                    if constexpr(ONE && FOLD) {
                        qb = __builtin_fmaf(cc.q_uv, v, cc.q_u);
                        qc = __builtin_fmaf(v, __builtin_fmaf(cc.q_vv, v, cc.q_v), cc.q_0) + screen_margin;
                    }
                    float cell = cr.first;
                    for(uint32_t k = 0; k < cr.count; k++, cell += 1.0f) {
                        // Conservative miss test on the unnormalised ray in fused arithmetic.  With
                        // h = draw.oc and dd = draw.draw the reference's discriminant has the sign of
                        // h*h - dd*c.  Both evaluations are within about a dozen unit roundoffs of
                        // h*h + dd*(oc.oc + r*r), so a value below -kScreenMargin of that magnitude is
                        // negative in the exact evaluation as well, and the sample takes the row
                        // background whatever its direction is.  Everything else runs the exact
                        // arithmetic below.  Where the camera is not close to the viewport plane (host
                        // check, kFlagFoldScreen) the screening ray comes from fmas instead of the
                        // reference's operations: it differs from the exact ray by a few ulps of the
                        // viewport extent, which moves the screening value by under 6e-6 of the
                        // magnitude -- inside the margin with room to spare.
                        auto exact_draw = [&]() {
                            const float jx = jitter ? (cell + off.rx) * cr.inv_cells : 0.5f;
                            const float px = xf + jx;
                            return f3{(px * vc.psx - vc.vw_half) - cam.x, draw_y, dz};
                        };
                        f3 draw = {0.0f, 0.0f, dz};
                        bool maybe = false;
                        if constexpr(ONE && FOLD) {
                            const float u = __builtin_fmaf(cell, du, u_first);
                            maybe = __builtin_fmaf(u, __builtin_fmaf(cc.q_uu, u, qb), qc) >= 0.0f;
                        } else {
                            float rx, ry;
                            if constexpr(FOLD) {
                                rx = __builtin_fmaf(cell, du, u_first);
                                ry = v;
                            } else {
                                draw = exact_draw();
                                rx = draw.x;
                                ry = draw.y;
                            }
                            const float dd = __builtin_fmaf(ry, ry, __builtin_fmaf(rx, rx, dz2));
                            for(uint32_t i = 0; i < n_spheres; i++) {
                                const float4 kk = ONE ? make_float4(cc.oc[0], cc.oc[1], cc.oc[2], cc.c) : s_occ[i];
                                const float mag = ONE ? cc.magnitude : s_ctr[i].w;
                                const float h = __builtin_fmaf(ry, kk.y, __builtin_fmaf(rx, kk.x, dz * kk.z));
                                const float q = h * h;
                                const float screen = __builtin_fmaf(kScreenMargin, __builtin_fmaf(dd, mag, q), __builtin_fmaf(-dd, kk.w, q));
                                maybe = maybe || (screen >= 0.0f);
                            }
                        }
                        if constexpr(FOLD) {
                            if(maybe) draw = exact_draw();
                        }

                        f3 L = bg;
                        if(maybe) {
                            const f3 d = normalize3<true>(draw);
                            const float a = dot3<true>(d, d);
                            const float a4 = 4.0f * a;
                            const float a2 = 2.0f * a;
                            float closest = 3.402823466e+38f;
                            int hit = -1;
                            for(uint32_t i = 0; i < n_spheres; i++) {
                                const float4 kk = ONE ? make_float4(cc.oc[0], cc.oc[1], cc.oc[2], cc.c) : s_occ[i];
                                const float b = -2.0f * dot3<true>(d, f3{kk.x, kk.y, kk.z});
                                const float disc = b * b - a4 * kk.w;
                                if(disc >= 0.0f) {
                                    // FAST_AFTER_DISC: up to here the sample is the reference's, bit for
                                    // bit (ray, normalisation, discriminant -- where its rounding is
                                    // amplified); from here on only t / r amplifies, and for a sphere
                                    // within kFastTailMaxDistance radii the hardware's approximate
                                    // square root and reciprocals stay within 2e-6 of it (fast_tail)
                                    float sq, inv_a2 = 0.0f, t;
                                    if(fast_tail) {
                                        sq = __builtin_amdgcn_sqrtf(disc);
                                        inv_a2 = __builtin_amdgcn_rcpf(a2);
                                        t = (-b - sq) * inv_a2;
                                    } else {
                                        sq = rt_sqrt<true>(disc);
                                        t = (-b - sq) / a2;
                                    }
                                    bool ok = (t > kTMin) && (t < closest);
                                    if(!ok) {
                                        // rare (camera inside a sphere): keep it a branch -- computed
                                        // unconditionally, the second IEEE division costs every hit 13 issue slots
                                        asm volatile("");
                                        t = fast_tail ? (-b + sq) * inv_a2 : (-b + sq) / a2;
                                        ok = (t > kTMin) && (t < closest);
                                    }
                                    if(ok) {
                                        closest = t;
                                        hit = static_cast<int>(i);
                                    }
                                }
                            }
                            if(hit >= 0) {
                                const f3 ctr = ONE ? f3{cc.center[0], cc.center[1], cc.center[2]}
                                                   : f3{s_ctr[hit].x, s_ctr[hit].y, s_ctr[hit].z};
                                if(fast_tail) {
                                    const f3 w = {__builtin_fmaf(d.x, closest, cam.x - ctr.x), __builtin_fmaf(d.y, closest, cam.y - ctr.y),
                                                  __builtin_fmaf(d.z, closest, cam.z - ctr.z)};
                                    const float half_inv = 0.5f * __builtin_amdgcn_rsqf(dot3<false>(w, w));
                                    L = f3{__builtin_fmaf(w.x, half_inv, 0.5f), __builtin_fmaf(w.y, half_inv, 0.5f), __builtin_fmaf(w.z, half_inv, 0.5f)};
                                } else {
                                    const f3 pt = {d.x * closest + cam.x, d.y * closest + cam.y, d.z * closest + cam.z};
                                    const f3 n = normalize3<true>(f3{pt.x - ctr.x, pt.y - ctr.y, pt.z - ctr.z});
                                    L = f3{remap_unit(n.x), remap_unit(n.y), remap_unit(n.z)};      // throughput is 1
                                }
                            }
                        }
                        block.x += L.x;
                        block.y += L.y;
                        block.z += L.z;
                    }
                }
            }
            if(halves) {        // the pair's other row (the same sum in both lanes: a + b = b + a)
                block.x += __shfl_xor(block.x, 32, 64);
                block.y += __shfl_xor(block.y, 32, 64);
                block.z += __shfl_xor(block.z, 32, 64);
            }
            if(split == 1u) {
                sum.x += block.x;
                sum.y += block.y;
                sum.z += block.z;
            } else {
                // hand the block sum to the strip's other waves; every wave adds the round's sums in
                // block order to its own copy of the total.  Two buffers alternate, so the barrier
                // of the next round also orders this round's reads before the buffer's next writes.
                const uint32_t buf = (first >> split_log2) & 1u;
                s_part[buf][wave][0][lane] = block.x;
                s_part[buf][wave][1][lane] = block.y;
                s_part[buf][wave][2][lane] = block.z;
                __syncthreads();
                const uint32_t live = min(split, nblocks - first);
                const uint32_t w0 = wave - part;       // the strip's first wave
                for(uint32_t p = 0; p < live; p++) {
                    sum.x += s_part[buf][w0 + p][0][lane];
                    sum.y += s_part[buf][w0 + p][1][lane];
                    sum.z += s_part[buf][w0 + p][2][lane];
                }
            }
        }
    } else {
        // fast build: the ray stays unnormalised (the hit test is scale invariant, t runs
        // along `draw`); only hits add to the sum, misses are counted and added as
        // (misses * background); the order of additions is free
        uint32_t hits = 0;
        f3 hit = {0.0f, 0.0f, 0.0f};        // the hit loop's sums, before the factor c * shade_scale
        // the hit loop's two per-sample constants, in vector registers: an instruction with a
        // scalar-register operand issues at half the rate of one without (measured, tools/valu_peak.hip)
        float q_uu_v = 0.0f, oc_x_v = 0.0f, c4_v = 0.0f;
        if constexpr(ONE) {
            asm volatile("v_mov_b32 %0, %1" : "=v"(q_uu_v) : "s"(cc.q_uu));
            asm volatile("v_mov_b32 %0, %1" : "=v"(oc_x_v) : "s"(cc.oc[0]));
            asm volatile("v_mov_b32 %0, %1" : "=v"(c4_v) : "s"(0.25f * cc.c));
        }
        // The Newton hit loop's view of the pixel: T = t / 2 = c / (2 (h + sqrt(disc))) and
        // rho = 1 / sqrt(disc) at the centre of the pixel with their gradients over the viewport.
        // rho is used as that first-order expansion throughout the pixel; T's expansion only
        // starts the rows (d disc / du = 2 (h oc.x - c u), likewise in v).
        struct newton_pixel {
            float u_c, v_c, rho_c, rho_u, rho_v, T_c, T_u, T_v;
        };
        newton_pixel np = {};
        if constexpr(ONE) {
            if(newton) {
                np.u_c = __builtin_fmaf(0.5f, vc.psx, ux0);
                np.v_c = __builtin_fmaf(0.5f, vc.psy, uy0);
                const float h_c = __builtin_fmaf(np.u_c, cc.oc[0], __builtin_fmaf(np.v_c, cc.oc[1], dz * cc.oc[2]));
                const float dd_c = __builtin_fmaf(np.u_c, np.u_c, __builtin_fmaf(np.v_c, np.v_c, dz2));
                const float disc_c = __builtin_fmaf(-dd_c, cc.c, h_c * h_c);
                np.rho_c = __builtin_amdgcn_rsqf(disc_c);
                const float inv_w = __builtin_amdgcn_rcpf(__builtin_fmaf(disc_c, np.rho_c, h_c));
                np.T_c = (0.5f * cc.c) * inv_w;
                const float half_disc_u = __builtin_fmaf(h_c, cc.oc[0], -(cc.c * np.u_c));
                const float half_disc_v = __builtin_fmaf(h_c, cc.oc[1], -(cc.c * np.v_c));
                const float rho3 = -(np.rho_c * np.rho_c) * np.rho_c;
                np.rho_u = rho3 * half_disc_u;
                np.rho_v = rho3 * half_disc_v;
                const float t_w = -np.T_c * inv_w;
                np.T_u = t_w * __builtin_fmaf(np.rho_c, half_disc_u, cc.oc[0]);
                np.T_v = t_w * __builtin_fmaf(np.rho_c, half_disc_v, cc.oc[1]);
            }
        }
        // this wave's shares of the pixel's blocks (one, or all of them one after the other), each summed apart
        f3 shares_sum = {0.0f, 0.0f, 0.0f};
        uint32_t shares_hits = 0;
        for(uint32_t share = part; share < shares; share += split) {
            sum = f3{0.0f, 0.0f, 0.0f};
            hit = f3{0.0f, 0.0f, 0.0f};
            hits = 0;
            if constexpr(ONE) {
                if(newton) {
                    // The hit loop without a square root or a reciprocal (see kNewtonTolerance):
                    // T = t / 2 of the previous cell takes one Newton step on
                    //   f / 4 = dd T*T - h T + c / 4,   (f / 4)' = 2 dd T - h = -sqrt(disc) at the near root,
                    // with rho ~ 1 / sqrt(disc) from its first-order expansion about the pixel's centre
                    // (one add a cell).  Nine full-rate vector instructions a sample for a row on its own,
                    // eight where h is a row's constant and seven where rho is too (SIDEWAYS and FLAT below).
                    // Rows of equal length are walked side by side -- four at a time for flat pixels where the
                    // share has them, else two: they share u and u*u and the sum over the rows of T that
                    // multiplies u, and independent chains keep the vector unit fed -- measured 21 clocks a
                    // sample for two rows against 24 for one row at a time; four rows take 7.5 instructions
                    // a sample where two take 8.  The rows are the share's, block after block: the order in
                    // which a share's samples are added is free in this build, as long as it is the same
                    // whatever else is in the pixel's wave (see where the rows are grouped, below).
                    //
                    // Where 1 / sqrt(disc) is a row's constant and h is too (FLAT, not SIDEWAYS) the step is
                    // a polynomial in T with one coefficient that changes along the row:
                    //   T + rho (T (dd T - h) + c / 4) = T ((rho dd) T + (1 - rho h)) + rho c / 4,
                    // rho dd = rho u*u + rho dd_row one fused operation a cell, the other two coefficients the
                    // row's: three operations a step instead of four (dd, two for the residual, T), 5.5
                    // instructions a sample with four rows side by side.  Its roundings are of the size of
                    // the other form's (each is a few unit roundoffs of c / (4 sqrt(disc))).  A pixel takes
                    // this form when its own class says so, whatever its neighbours in the wave take: a
                    // wave that holds both kinds walks the rows once for each (set-up's order of workgroups
                    // counts such a wave at the sum and starts it early).
                    struct newton_row {
                        float v, h_row, dd_row, rho, T, C;      // (polynomial form: h_row holds 1 - rho h, dd_row holds rho dd_row, C rho c / 4)
                    };
                    // the share's rows of cells in sample order, one a call (false: none left)
                    uint32_t next_blk = 0, next_s = 0, next_end = 0;
                    auto first_row = [&]() {
                        next_blk = share;
                        next_s = share * AEHIP_SAMPLE_BLOCK;
                        next_end = min(vc.spp, next_s + AEHIP_SAMPLE_BLOCK);
                    };
                    auto take_row = [&](cell_row &cr) {
                        if(next_blk >= nblocks) return false;
                        cr = row_at(next_s, next_end);
                        next_s += cr.count;
                        if(next_s >= next_end) {
                            next_blk += shares;
                            next_s = next_blk * AEHIP_SAMPLE_BLOCK;
                            next_end = min(vc.spp, next_s + AEHIP_SAMPLE_BLOCK);
                        }
                        return true;
                    };
                    auto same_cells = [](const cell_row &a, const cell_row &b) {
                        return a.count == b.count && a.first_cell == b.first_cell && a.cells == b.cells;
                    };
                    // (TWICE: some pixel of the strip needs a second step a cell -- newton_twice; the
                    // others keep their first step's value, so a pixel's arithmetic does not depend on its neighbours)
                    // (SIDEWAYS: the sphere's centre is off the camera's axis horizontally, so h = draw.oc
                    // changes along a row; in the reference's scene it does not and is the row's constant)
                    // (FLAT: every pixel of the strip may take 1 / sqrt(disc) as constant along a row --
                    // newton_flat; such a pixel among others steps by zero, the same arithmetic)
                    auto walk_rows = [&](auto twice_tag, auto sideways_tag, auto flat_tag) {
                        constexpr bool TWICE = decltype(twice_tag)::value, SIDEWAYS = decltype(sideways_tag)::value, FLAT = decltype(flat_tag)::value;
                        constexpr bool POLY = FLAT && !SIDEWAYS && !TWICE;
                        // where the rows of one shape start along u, and a row's own start: 1 / sqrt(disc) one cell
                        // before its first cell, and T at the first cell (the first step refines it), both from
                        // their expansions about the pixel's centre (newton_pixel)
                        float du, u_first, drho;
                        auto start_rows = [&](const cell_row &cr) {
                            du = jitter ? vc.psx * cr.inv_cells : 0.0f;
                            u_first = __builtin_fmaf(jitter ? (cr.first + off.rx) * cr.inv_cells : 0.5f, vc.psx, ux0);
                            drho = newton_flat ? 0.0f : np.rho_u * du;
                        };
                        auto start_row = [&](const cell_row &cr) {
                            newton_row nr;
                            nr.v = __builtin_fmaf(cr.jy, vc.psy, uy0);
                            nr.h_row = __builtin_fmaf(nr.v, cc.oc[1], dz * cc.oc[2]);
                            nr.dd_row = __builtin_fmaf(nr.v, nr.v, dz2);
                            const float off_u = u_first - np.u_c, off_v = nr.v - np.v_c;
                            const float rho_column = __builtin_fmaf(np.rho_v, off_v, np.rho_c);        // in the pixel's middle column
                            nr.rho = newton_flat ? rho_column : __builtin_fmaf(np.rho_u, off_u - du, rho_column);
                            nr.T = __builtin_fmaf(np.T_u, off_u, __builtin_fmaf(np.T_v, off_v, np.T_c));
                            nr.C = 0.0f;
                            if constexpr(POLY) {
                                nr.C = nr.rho * c4_v;
                                nr.h_row = __builtin_fmaf(-nr.rho, nr.h_row, 1.0f);
                                nr.dd_row = nr.rho * nr.dd_row;
                            }
                            return nr;
                        };
                        // a cell of a row: the step (two with TWICE) on the row's T
                        auto step = [&](newton_row &r, float u, float uu) {
                            if constexpr(POLY) {
                                r.T = __builtin_fmaf(r.T, __builtin_fmaf(__builtin_fmaf(r.rho, uu, r.dd_row), r.T, r.h_row), r.C);
                                return;
                            }
                            if constexpr(!FLAT) r.rho += drho;
                            const float dd = uu + r.dd_row;
                            const float h = SIDEWAYS ? __builtin_fmaf(u, oc_x_v, r.h_row) : r.h_row;
                            const float f4 = __builtin_fmaf(r.T, __builtin_fmaf(dd, r.T, -h), c4_v);
                            r.T = __builtin_fmaf(f4, r.rho, r.T);
                            if constexpr(TWICE) {
                                const float g4 = __builtin_fmaf(r.T, __builtin_fmaf(dd, r.T, -h), c4_v);
                                const float again = __builtin_fmaf(g4, r.rho, r.T);
                                r.T = newton_twice ? again : r.T;
                            }
                        };
                        auto walk_one = [&](const cell_row &cr) {
                            start_rows(cr);
                            newton_row ra = start_row(cr);
                            float u = u_first - du, row_scale = 0.0f;
#pragma unroll 4
                            for(uint32_t k = 0; k < cr.count; k++) {
                                u += du;
                                if constexpr(POLY) {
                                    step(ra, u, u * u);
                                    hit.x = __builtin_fmaf(u, ra.T, hit.x);
                                    row_scale += ra.T;
                                    continue;
                                }
                                if constexpr(!FLAT) ra.rho += drho;
                                const float dd = __builtin_fmaf(u, u, ra.dd_row);
                                const float h = SIDEWAYS ? __builtin_fmaf(u, oc_x_v, ra.h_row) : ra.h_row;
                                const float f4 = __builtin_fmaf(ra.T, __builtin_fmaf(dd, ra.T, -h), c4_v);
                                ra.T = __builtin_fmaf(f4, ra.rho, ra.T);
                                if constexpr(TWICE) {
                                    const float g4 = __builtin_fmaf(ra.T, __builtin_fmaf(dd, ra.T, -h), c4_v);
                                    const float again = __builtin_fmaf(g4, ra.rho, ra.T);
                                    ra.T = newton_twice ? again : ra.T;
                                }
                                hit.x = __builtin_fmaf(u, ra.T, hit.x);
                                row_scale += ra.T;
                            }
                            hit.y = __builtin_fmaf(ra.v, row_scale, hit.y);
                            hit.z += row_scale;
                            hits += cr.count;
                        };
                        auto walk_two = [&](const cell_row &cr, const cell_row &cr2) {
                            start_rows(cr);
                            newton_row ra = start_row(cr), rb = start_row(cr2);
                            float u = u_first - du, scale_a = 0.0f, scale_b = 0.0f;
#pragma unroll 2
                            for(uint32_t k = 0; k < cr.count; k++) {
                                u += du;
                                const float uu = u * u;
                                step(ra, u, uu);
                                step(rb, u, uu);
                                scale_a += ra.T;
                                scale_b += rb.T;
                                hit.x = __builtin_fmaf(u, ra.T + rb.T, hit.x);
                            }
                            hit.y = __builtin_fmaf(ra.v, scale_a, __builtin_fmaf(rb.v, scale_b, hit.y));
                            hit.z += scale_a + scale_b;
                            hits += 2u * cr.count;
                        };
                        auto walk_four = [&](const cell_row &cr, const cell_row &cr2, const cell_row &cr3, const cell_row &cr4) {
                            start_rows(cr);
                            newton_row ra = start_row(cr), rb = start_row(cr2), rc = start_row(cr3), rd = start_row(cr4);
                            float u = u_first - du, scale_a = 0.0f, scale_b = 0.0f, scale_c = 0.0f, scale_d = 0.0f;
                            for(uint32_t k = 0; k < cr.count; k++) {
                                u += du;
                                const float uu = u * u;
                                step(ra, u, uu);
                                step(rb, u, uu);
                                step(rc, u, uu);
                                step(rd, u, uu);
                                scale_a += ra.T;
                                scale_b += rb.T;
                                scale_c += rc.T;
                                scale_d += rd.T;
                                hit.x = __builtin_fmaf(u, (ra.T + rb.T) + (rc.T + rd.T), hit.x);
                            }
                            hit.y = __builtin_fmaf(ra.v, scale_a, __builtin_fmaf(rb.v, scale_b, __builtin_fmaf(rc.v, scale_c, __builtin_fmaf(rd.v, scale_d, hit.y))));
                            hit.z += (scale_a + scale_b) + (scale_c + scale_d);
                            hits += 4u * cr.count;
                        };
                        cell_row c0, c1, c2, c3, ahead;
                        first_row();
                        bool have = take_row(c0);
                        while(have) {
                            // up to three more rows of the same shape
                            uint32_t n = 1u;
                            bool more = take_row(ahead);
                            if(more && same_cells(ahead, c0)) {
                                c1 = ahead;
                                n = 2u;
                                more = take_row(ahead);
                                if(FLAT && more && same_cells(ahead, c0)) {
                                    c2 = ahead;
                                    n = 3u;
                                    more = take_row(ahead);
                                    if(more && same_cells(ahead, c0)) {
                                        c3 = ahead;
                                        n = 4u;
                                        more = take_row(ahead);
                                    }
                                }
                            }
                            // (Four rows at a time for flat pixels, two for the others, whether these take a second
                            // step or not: the order in which a pixel's samples are added follows from its own class,
                            // not from what else is in its wave.  With a second step four rows take 82 registers.)
                            if constexpr(FLAT) {
                                if(n == 4u) walk_four(c0, c1, c2, c3);
                            }
                            if(n == 2u || n == 3u) walk_two(c0, c1);
                            if(n == 1u) walk_one(c0);
                            if(n == 3u) walk_one(c2);
                            c0 = ahead;
                            have = more;
                        }
                    };
                    const std::true_type yes{};
                    const std::false_type no{};
                    // flat pixels in a walk of their own, the others in theirs (with the second step if one of them takes it)
                    const bool sideways = cc.oc[0] != 0.0f;
                    if(newton_flat) {
                        if(sideways) walk_rows(no, yes, yes);
                        else walk_rows(no, no, yes);
                    } else {
                        const bool twice = __ballot(newton_twice) != 0ull;
                        if(sideways) {
                            if(twice) walk_rows(yes, yes, no);
                            else walk_rows(no, yes, no);
                        } else {
                            if(twice) walk_rows(yes, no, no);
                            else walk_rows(no, no, no);
                        }
                    }
                }
            }
            for(uint32_t blk = share; blk < nblocks; blk += shares) {
                const uint32_t s0 = blk * AEHIP_SAMPLE_BLOCK;
                const uint32_t s1 = min(vc.spp, s0 + AEHIP_SAMPLE_BLOCK);
                if constexpr(ONE) {
                    if(sure_hit) {
                        // The hit loop: every sample of the pixel hits at the near root (see the
                        // general loop below for the algebra).  Nine vector instructions a sample,
                        // no test and no branch; the factor c * shade_scale is applied to the sums.
                        for(uint32_t s = s0; s < s1;) {
                            const cell_row cr = row_at(s, s1);
                            s += cr.count;
                            const float v = __builtin_fmaf(cr.jy, vc.psy, uy0);
                            const float du = jitter ? vc.psx * cr.inv_cells : 0.0f;
                            float u = __builtin_fmaf(jitter ? (cr.first + off.rx) * cr.inv_cells : 0.5f, vc.psx, ux0) - du;
                            const float qb = __builtin_fmaf(cc.q_uv, v, cc.q_u);
                            const float qc = __builtin_fmaf(v, __builtin_fmaf(cc.q_vv, v, cc.q_v), cc.q_0);
                            const float h_row = __builtin_fmaf(v, cc.oc[1], dz * cc.oc[2]);
                            float row_scale = 0.0f;
#pragma unroll 4
                            for(uint32_t k = 0; k < cr.count; k++) {
                                u += du;
                                const float disc = __builtin_fmaf(u, __builtin_fmaf(q_uu_v, u, qb), qc);
                                const float h = __builtin_fmaf(u, oc_x_v, h_row);
                                // the near root over dd, (h - sqrt(disc)) / dd, is c / (h + sqrt(disc)):
                                // (h - s)(h + s) = h*h - disc = dd * c.  No cancellation, and dd is not needed.
                                const float sc = __builtin_amdgcn_rcpf(h + __builtin_amdgcn_sqrtf(disc));
                                hit.x = __builtin_fmaf(u, sc, hit.x);
                                row_scale += sc;
                            }
                            hit.y = __builtin_fmaf(v, row_scale, hit.y);
                            hit.z += row_scale;
                            hits += cr.count;
                        }
                    }
                }
                if(!work || sure_hit || newton) continue;
                for(uint32_t s = s0; s < s1;) {
                    const cell_row cr = row_at(s, s1);
                    s += cr.count;
                    const float v = __builtin_fmaf(cr.jy, vc.psy, uy0);
                    const float du = jitter ? vc.psx * cr.inv_cells : 0.0f;
                    // u of the cell before the row's first: the loop steps before it tests
                    float u = __builtin_fmaf(jitter ? (cr.first + off.rx) * cr.inv_cells : 0.5f, vc.psx, ux0) - du;
                    if constexpr(ONE) {
                        const float qb = __builtin_fmaf(cc.q_uv, v, cc.q_u);
                        const float qc = __builtin_fmaf(v, __builtin_fmaf(cc.q_vv, v, cc.q_v), cc.q_0);
                        const float h_row = __builtin_fmaf(v, cc.oc[1], dz * cc.oc[2]);       // h = u oc.x + h_row
                        const float dd_row = __builtin_fmaf(v, v, dz2);                      // dd = u u + dd_row
                        float row_scale = 0.0f;
                        // a missing sample is four vector instructions: step, two fmas, compare
#pragma unroll 4
                        for(uint32_t k = 0; k < cr.count; k++) {
                            u += du;
                            const float disc = __builtin_fmaf(u, __builtin_fmaf(cc.q_uu, u, qb), qc);
                            if(disc >= 0.0f) {
                                const float dd = __builtin_fmaf(u, u, dd_row);
                                const float h = __builtin_fmaf(u, cc.oc[0], h_row);
                                // The hit is at draw * t, t = root / dd, root = h -+ sqrt(disc); it lies
                                // beyond t_min when root > 0 and root*root > t_min^2 * dd.  With the
                                // camera outside the sphere by more than t_min (near_root_only) the two
                                // roots have the sign of h and the near one always passes.
                                // A point of the sphere is a radius away from the centre, so the
                                // normal is (draw*t - oc) / r without a normalisation, and its colour
                                // 0.5*n + 0.5 = draw * (0.5 t / r) + (0.5 - 0.5 oc / r).  The sum over
                                // the pixel's hits needs only sum(u*s), sum(v*s), sum(s) with
                                // s = 0.5 t / r (v and dz leave the sums of a row and of the pixel); the
                                // constant parts are added once, below.
                                const float sq = __builtin_amdgcn_sqrtf(disc);
                                float root = h - sq;
                                bool ok;
                                if(cc.near_root_only != 0u) {
                                    ok = h > 0.0f;
                                } else {
                                    const float far_enough = (kTMin * kTMin) * dd;
                                    if(!(root > 0.0f && root * root > far_enough)) root = h + sq;
                                    ok = root > 0.0f && root * root > far_enough;
                                }
                                if(ok) {
                                    const float sc = (root * cc.shade_scale) * __builtin_amdgcn_rcpf(dd);
                                    sum.x = __builtin_fmaf(u, sc, sum.x);
                                    row_scale += sc;
                                    hits++;
                                }
                            }
                        }
                        sum.y = __builtin_fmaf(v, row_scale, sum.y);
                        sum.z += row_scale;
                    } else {
                        const float dd_row = __builtin_fmaf(v, v, dz2);
                        for(uint32_t k = 0; k < cr.count; k++) {
                            u += du;
                            const float dd = __builtin_fmaf(u, u, dd_row);
                            float closest = 3.402823466e+38f;
                            int hit = -1;
                            for(uint32_t i = 0; i < n_spheres; i++) {
                                const float4 kk = s_occ[i];
                                const float h = __builtin_fmaf(v, kk.y, __builtin_fmaf(u, kk.x, dz * kk.z));
                                const float disc = __builtin_fmaf(-dd, kk.w, h * h);
                                if(disc >= 0.0f) {
                                    const float sq = __builtin_amdgcn_sqrtf(disc);
                                    const float inv = __builtin_amdgcn_rcpf(dd);
                                    const float tmin = kTMin * __builtin_amdgcn_rsqf(dd);       // t_min in units of |draw|
                                    float t = (h - sq) * inv;
                                    bool ok = (t > tmin) && (t < closest);
                                    if(!ok) {
                                        t = (h + sq) * inv;
                                        ok = (t > tmin) && (t < closest);
                                    }
                                    if(ok) {
                                        closest = t;
                                        hit = static_cast<int>(i);
                                    }
                                }
                            }
                            if(hit >= 0) {
                                const f3 ctr = {s_ctr[hit].x, s_ctr[hit].y, s_ctr[hit].z};
                                const f3 w = {__builtin_fmaf(u, closest, cam.x - ctr.x), __builtin_fmaf(v, closest, cam.y - ctr.y),
                                              __builtin_fmaf(dz, closest, cam.z - ctr.z)};
                                const float half_inv = 0.5f * __builtin_amdgcn_rsqf(dot3<false>(w, w));
                                sum.x += __builtin_fmaf(w.x, half_inv, 0.5f);
                                sum.y += __builtin_fmaf(w.y, half_inv, 0.5f);
                                sum.z += __builtin_fmaf(w.z, half_inv, 0.5f);
                                hits++;
                            }
                        }
                    }
                }
            }
            if constexpr(ONE) {
                const float scale = (newton ? 2.0f : cc.c) * cc.shade_scale;        // the Newton loop sums t / 2, the closed-form loop t / c
                sum.x = __builtin_fmaf(hit.x, scale, sum.x);
                sum.y = __builtin_fmaf(hit.y, scale, sum.y);
                sum.z = __builtin_fmaf(hit.z, scale, sum.z);
            }
            // (the first share as it is, the others added to it in order: what the strip's first wave does below with its partners' sums)
            if(share == part) {
                shares_sum = sum;
                shares_hits = hits;
            } else {
                shares_sum.x += sum.x;
                shares_sum.y += sum.y;
                shares_sum.z += sum.z;
                shares_hits += hits;
            }
        }
        sum = shares_sum;
        hits = shares_hits;
        if(split > 1u) {
            s_part[0][wave][0][lane] = sum.x;
            s_part[0][wave][1][lane] = sum.y;
            s_part[0][wave][2][lane] = sum.z;
            s_hits[wave][lane] = hits;
            __syncthreads();
            if(part == 0u) {
                for(uint32_t p = 1; p < split; p++) {
                    sum.x += s_part[0][wave + p][0][lane];
                    sum.y += s_part[0][wave + p][1][lane];
                    sum.z += s_part[0][wave + p][2][lane];
                    hits += s_hits[wave + p][lane];
                }
            }
        }
        if constexpr(ONE) {
            // constant parts of the hits' colours (see the sample loop)
            const float hf = static_cast<float>(hits);
            sum.x = __builtin_fmaf(hf, cc.shade_offset[0], sum.x);
            sum.y = __builtin_fmaf(hf, cc.shade_offset[1], sum.y);
            sum.z = __builtin_fmaf(hf, cc.shade_offset[2], dz * sum.z);
        }
        const float misses = static_cast<float>(vc.spp - hits);
        sum.x = __builtin_fmaf(misses, bg.x, sum.x);
        sum.y = __builtin_fmaf(misses, bg.y, sum.y);
        sum.z = __builtin_fmaf(misses, bg.z, sum.z);
    }

    write_pixel();

    // (Ray and lane-slot counts of this pass follow from the image size -- one ray per sample, a
    // strip's 64 lanes counted for each of its samples, walked or decided by the pixel's bounds --
    // and are filled in by the host.)
}

template <bool EXACT, bool ONE, bool FOLD, bool AXIAL, bool ORDERED = false>
__global__ __launch_bounds__(kBlock) void camera_kernel(view_consts vc, camera_consts cc, uint32_t split_log2, uint32_t row_order,
                                                        const float4 *__restrict__ geom,
                                                        const uint32_t *__restrict__ sample_terms,
                                                        uint32_t *__restrict__ argb, float4 *__restrict__ rgba,
                                                        const float4 *__restrict__ row_miss, const uint32_t *__restrict__ tile_order) {
    // (ORDERED: the grid's workgroups in set-up's order, tile_order: column | row << 16 -- build_tile_order)
    uint32_t gx = blockIdx.x, gy = 0u;
    if constexpr(ORDERED) {
        const uint32_t t = __builtin_amdgcn_readfirstlane(tile_order[blockIdx.y * gridDim.x + blockIdx.x]);
        gx = t & 0xffffu;
        gy = t >> 16;
    } else {
        gy = edges_first_row(blockIdx.y, row_order);
    }
    camera_body<EXACT, ONE, FOLD, AXIAL, false>(vc, cc, split_log2, geom, sample_terms, argb, rgba, gx, gy, nullptr, 0u, 0u, row_miss);
}

// The exact build's value of a pixel whose samples all miss, before the division by spp: the row's
// background added spp times in the camera pass's order (blocks of AEHIP_SAMPLE_BLOCK samples, then
// the block sums).  One entry per image row; one thread per row.
__global__ __launch_bounds__(64) void row_miss_sums_kernel(view_consts vc, uint32_t image_rows, float4 *__restrict__ out) {
    const uint32_t y = blockIdx.x * 64u + threadIdx.x;
    if(y >= image_rows) return;
    const f3 bg = row_background<true>(vc, y);
    f3 sum = {0.0f, 0.0f, 0.0f};
    for(uint32_t s0 = 0; s0 < vc.spp; s0 += AEHIP_SAMPLE_BLOCK) {
        const uint32_t s1 = min(vc.spp, s0 + AEHIP_SAMPLE_BLOCK);
        f3 block = {0.0f, 0.0f, 0.0f};
        for(uint32_t s = s0; s < s1; s++) {
            block.x += bg.x;
            block.y += bg.y;
            block.z += bg.z;
        }
        sum.x += block.x;
        sum.y += block.y;
        sum.z += block.z;
    }
    out[y] = make_float4(sum.x, sum.y, sum.z, 0.0f);
}

// One sphere: what each workgroup of a rectangle of the launch's grid (the sphere's box, or all of it) will cost,
// for the order in which the launch takes them (build_tile_order).  A wave's time does not depend on how many of
// its pixels walk a loop, only on which loops it walks.  Fast build: the Newton loop in its polynomial form, with
// one step a cell, with two, the closed-form loop (weights: vector-unit clocks a sample, roughly).  Exact build:
// the loop for waves of sure hits, or the general loop.  A workgroup lasts as long as its longest wave.  One
// workgroup here for one of the launch's, in the rectangle's row-major order.
__global__ __launch_bounds__(kBlock) void tile_cost_kernel(view_consts vc, camera_consts cc, uint32_t rows_per_group, uint32_t first_col,
                                                           uint32_t first_row, uint32_t width, uint32_t exact, uint32_t *__restrict__ cost) {
    __shared__ uint32_t s_cost;
    if(threadIdx.x == 0) s_cost = 0u;
    __syncthreads();
    const uint32_t line = blockIdx.x / width;
    const uint32_t gx = first_col + (blockIdx.x - line * width), gy = first_row + line;
    const uint32_t strip = threadIdx.x >> 6, lane = threadIdx.x & 63u, strips = rows_per_group / 4u;
    const uint32_t x = gx * kTileW + (lane & 15u), row = (gy * strips + strip) * 4u + (lane >> 4);
    if(strip < strips) {
        bool flat = false, once = false, twice = false, closed = false, sure = false, general = false;
        if(x < vc.width && row < vc.row_count) {
            const pixel_class pc = classify_pixel(vc, cc, x, image_row(vc, row));
            if(exact) {
                // (camera_body, exact build: a wave all of whose pixels are sure hits takes the short loop)
                sure = pc.may_hit && cc.short_exact_ok != 0u && pc.well_conditioned && pc.faces_sphere;
                general = pc.may_hit && !sure;
            } else if(pc.may_hit && pc.well_conditioned) {         // (ill-conditioned pixels are the list's, not the box's)
                if(cc.near_root_only == 0u) closed = true;  // the general loop: a square root and a reciprocal a hit, like the closed form
                else if(pc.faces_sphere) {
                    flat = pc.newton_flat_ok;
                    once = !flat && pc.newton_ok;
                    twice = !flat && !once && pc.newton_twice_ok;
                    closed = !flat && !once && !twice;
                }
            }
        }
        uint32_t c = 0;
        if(exact) {
            if(__ballot(general) != 0ull) c = 150u;
            else if(__ballot(sure) != 0ull) c = 135u;
        } else {        // (flat pixels in a walk of their own: the polynomial step where the sphere's centre is in the camera's vertical plane)
            if(__ballot(twice) != 0ull) c += 24u;
            else if(__ballot(once) != 0ull) c += 16u;
            if(__ballot(flat) != 0ull) c += cc.oc[0] != 0.0f ? 15u : 11u;
        }
        if(__ballot(closed) != 0ull) c += 40u;
        if(lane == 0u) atomicMax(&s_cost, c);
    }
    __syncthreads();
    if(threadIdx.x == 0) cost[blockIdx.x] = s_cost;
}

// The fast build's single-sphere launch when the band has ill-conditioned pixels: the first
// list_groups workgroups render the listed pixels with the exact build's code up to the
// discriminant (the reference's ray, normalisation and hit decision, bit for bit) and the fast
// arithmetic after it, the others the rectangle in the fast arithmetic minus those pixels.  One launch, so the few hundred list workgroups run beside the bulk instead of
// on an empty chip (as a launch of their own they took 0.3 ms of a 2.2 ms 4K frame).
// (No occupancy bound: held to the 64 vector registers of eight waves a SIMD the rectangle's body
// spilled eight of them and rebuilt constants inside its row loop; with the 77 to 79 it has taken since, six
// waves, the 4K 1024-sample frame went from 0.433 to 0.408 ms.)
// (WALK: the launch walks every sample of every pixel, AEHIP_FLAG_WALK_ALL -- the same code under a
// bound of eight waves a SIMD, which the general loop's square roots and reciprocals want: 1.42
// against 1.58 ms a 4K frame.)
template <bool FOLD, bool AXIAL, bool WALK>
__global__ __launch_bounds__(kBlock, WALK ? 8 : 1) void camera_hybrid_kernel(view_consts vc, camera_consts cc, uint32_t split_log2, uint32_t row_order,
                                                               const uint32_t *__restrict__ sample_terms,
                                                               uint32_t *__restrict__ argb, float4 *__restrict__ rgba,
                                                               const uint32_t *__restrict__ pixel_list, uint32_t list_count,
                                                               uint32_t list_row_bias, uint32_t list_groups, uint32_t box_cols,
                                                               uint32_t box_groups, uint32_t rows_per_group,
                                                               const uint32_t *__restrict__ tile_order) {
#ifdef AEHIP_WAVE_TIMELINE
    timeline_stamp stamp{__builtin_amdgcn_s_memrealtime(), blockIdx.x < list_groups ? 1u : blockIdx.x - list_groups < box_groups ? 0u : 2u};
#endif
    if(blockIdx.x < list_groups) {
        camera_body<true, true, FOLD, AXIAL, false, true>(vc, cc, split_log2, nullptr, sample_terms, argb, rgba, blockIdx.x, 0u, pixel_list,
                                                          list_count, list_row_bias, nullptr);
        return;
    }
    // the box's workgroups (the whole grid when there is no box) -- in set-up's order, the ones that last
    // longest first (tile_order: column | row << 16), or else row by row in edges_first_row's order --
    // then the workgroups that fill the background around the box
    const uint32_t id = blockIdx.x - list_groups;
    if(id < box_groups) {
        const uint32_t width = box_cols >> 16;
        const uint32_t line = id / width;
        uint32_t gx = (box_cols & 0xffffu) + (id - line * width), gy = edges_first_row(line, row_order);
        if(tile_order) {
            const uint32_t t = __builtin_amdgcn_readfirstlane(tile_order[id]);
            gx = t & 0xffffu;
            gy = t >> 16;
        }
        camera_body<false, true, false, AXIAL, true>(vc, cc, split_log2, nullptr, sample_terms, argb, rgba, gx, gy, nullptr, 0u, 0u, nullptr);
    } else {
        fill_background(vc, id - box_groups, rows_per_group, row_order, box_cols, argb, rgba);
    }
}

// ------------------------------------------------------------------------------------------
// host side: error reporting, backend object, C ABI
// ------------------------------------------------------------------------------------------

void release_frame_state();        // multi-device frame state, defined with aehip_trace_frame

thread_local char g_error[512] = "";
bool g_initialised = false;
int g_device_count = 0;

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    return code;
}

#define AEHIP_CHECK(expr)                                                                              \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if(e_ != hipSuccess) return fail(AEHIP_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));  \
    } while(0)

}  // namespace

namespace {
// Scratch of the streamed pass and of the ray counters: three 64-bit words (ray segments, lane
// slots, the unit queue's cursor), the radiance block of a window of strips, the per-pixel sums of
// the window's finished blocks.  A backend's own (allocated by set-up, at most kScratchBudget) or,
// for aehip_backend_trace_device_static, the fixed array below, one copy per GPU.
// Radiance + totals a backend allocates at most.  The GPU has 288 GB: a 3840x2160 frame at 64 or more samples per pixel
// needs 6.5 GB for all its strips at once, and every window more is a launch pair more with a tail of its own (the
// depth-50 scene at 3840x2160, 16 samples: 33.0 ms in the four windows 448 MiB gave it, 27.7 ms in one; the 10,000-sphere
// scene 126 -> 121 ms).  AEHIP_SCRATCH_MIB in the environment sets another limit (tests of the window loop).
constexpr size_t kScratchBudget = static_cast<size_t>(8192) << 20;
inline size_t scratch_budget() {
    const char *env = std::getenv("AEHIP_SCRATCH_MIB");
    if(env && *env) {
        char *end = nullptr;
        const unsigned long long mib = std::strtoull(env, &end, 10);
        if(end && *end == '\0' && mib >= 1ull && mib <= (1ull << 20)) return static_cast<size_t>(mib) << 20;
    }
    return kScratchBudget;
}
constexpr size_t kStaticScratchBytes = AEHIP_STATIC_SCRATCH_BYTES;
constexpr size_t kScratchHeader = 256;                                  // counters, and alignment of what follows
constexpr uint32_t kMaxStaticWindows = 4096;                           // windows a call through the static scratch may take
static __device__ __align__(256) unsigned char g_static_scratch[kStaticScratchBytes];
// Device address of the array on each GPU (every GPU holds its own copy at its own address), looked
// up once per device with that device current; host code never uses the array's name as a pointer.
constexpr int kMaxDevices = 64;
unsigned char *g_static_scratch_ptr[kMaxDevices] = {};
// End of the last launch that used the array on each GPU, whichever backend made it: two backends on one
// GPU that call aehip_backend_trace_device_static on different streams are ordered by it.  Kept until the
// process ends, like the array.
hipEvent_t g_static_scratch_event[kMaxDevices] = {};

hipError_t resolve_static_scratch(int device) {        // `device` must be current
    if(device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    if(g_static_scratch_ptr[device]) return hipSuccess;
    void *ptr = nullptr;
    hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_static_scratch));
    if(e == hipSuccess && !g_static_scratch_event[device]) e = hipEventCreateWithFlags(&g_static_scratch_event[device], hipEventDisableTiming);
    if(e == hipSuccess) g_static_scratch_ptr[device] = static_cast<unsigned char *>(ptr);
    return e;
}

struct stream_scratch {
    unsigned long long *counters = nullptr;      // [0] ray segments, [1] lane slots, [2] unit queue
    radiance3 *radiance = nullptr;
    radiance3 *totals = nullptr;
    uint32_t window_strips = 0;                  // strips the radiance / totals hold
    uint32_t stride = 0;                         // samples per strip the radiance holds
};

// bytes of radiance + totals per strip for a request of `spp` samples per pixel
inline uint32_t scratch_stride(uint32_t spp) { return spp < AEHIP_SAMPLE_BLOCK ? spp : AEHIP_SAMPLE_BLOCK; }
inline size_t scratch_bytes_per_strip(uint32_t spp) { return (static_cast<size_t>(scratch_stride(spp)) * 64u + 64u) * sizeof(radiance3); }

}  // namespace

struct aehip_backend {
    int device = 0;
    int cu_count = 1;                        // compute units of the device
    uint32_t *framebuffer = nullptr;        // borrowed
    bool ready = false;

    aehip_params params{};
    view_consts vc{};
    bool all_terminal = false;               // every material ends the path: camera_kernel applies
    camera_consts cc{};                      // sphere 0 relative to the camera (camera_kernel, ONE)

    float4 *d_geom = nullptr;                // n x (cx, cy, cz, r*r)
    float4 *d_camera = nullptr;              // camera entries (oc.xyz, u) of the spheres, packed four at a time: see load_camera_group
    float4 *d_prefilter = nullptr;           // n x (Cc.xyz, K'): prefilter entries, see prefilter_entry
    float last_trace_render_ms = 0.0f;       // render time of the last aehip_backend_trace of the current request
    bool prefilter_pays = false;             // few enough rays pass it for it to be worth its 8 operations a sphere
    float4 *d_matl = nullptr;                // n x (albedo rgb, param)
    uint32_t *d_kind = nullptr;              // n x material id
    uint32_t scene_capacity = 0;

    uint32_t *d_argb = nullptr;              // own output band (aehip_backend_trace)
    size_t argb_capacity = 0;                // pixels
    uint32_t *h_staging = nullptr;           // pinned
    size_t staging_capacity = 0;
    unsigned long long *d_rays = nullptr;    // ray segments, lane slots, unit queue of the streamed pass
    uint32_t *d_sample_terms = nullptr;      // spp + 1 words: per-sample key terms (+1 spare for prefetch)
    float4 *d_row_miss = nullptr;            // per image row: an all-miss pixel's sum, exact build (row_miss_sums_kernel)
    uint32_t row_miss_capacity = 0;
    uint32_t sample_terms_capacity = 0;
    uint32_t *d_ring = nullptr;              // fast single-sphere camera pass: pixels the exact kernel redoes, sorted by band row
    uint32_t ring_capacity = 0, ring_count = 0;
    std::vector<uint32_t> ring_row_offset;   // entries before band row r (size rows + 1)
    // fast single-sphere camera pass: the box's workgroups, longest first (build_tile_order), and the launch they are for
    uint32_t *d_tile_order = nullptr;
    uint32_t tile_order_capacity = 0, tile_order_count = 0;
    uint32_t tile_order_box_rows = 0, tile_order_box_cols = 0, tile_order_rows_per_group = 0, tile_order_row_begin = 0, tile_order_row_count = 0;
    radiance3 *d_radiance = nullptr;         // streamed pass: a window's block of per-path radiance, then its
    size_t radiance_capacity = 0;            //   per-pixel sums of finished blocks; one allocation, size in bytes
    uint32_t window_strips = 0;              // strips of the band the allocation holds at the request's samples per pixel
    bool wants_streamed_pass = false;        // the request goes through the streamed pass (decided by set-up)
    hipEvent_t ev_scratch = nullptr;         // end of the last launch that used the scratch or the counters
    unsigned long long *last_counters = nullptr;     // where the last launch counted rays (own or static scratch)
    uint64_t setup_digest = 0;               // aehip_trace_frame: digest of the scene and request of the last set-up

    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;       // device -> host copies of finished parts of a frame (aehip_backend_trace)
    hipEvent_t ev_begin = nullptr, ev_end = nullptr, ev_copy = nullptr;
    hipEvent_t ev_part_done[kTraceParts] = {}, ev_part_copied[kTraceParts] = {};
    bool timed_on_own_stream = false;
    const void *attr_func = nullptr;         // kernel whose resource figures are in stats
    aehip_stats stats{};
    bool counts_known = false;               // the last launch's ray / lane-slot counts follow from its shape
    uint64_t known_rays = 0, known_lane_slots = 0;
    uint64_t walked_pixels = 0;              // single-sphere camera pass: pixels of the band that go through a sample loop (set-up counts them)
    bool culls_pixels = false;               // ... and whether that pass decides any pixel without walking it
    const void *stream_func = nullptr;       // streamed pass: the tracing kernel set-up chose, and how many of its
    int stream_groups_per_cu = 0;            //   workgroups a compute unit holds at once
};

namespace {

// Does a launch of this backend's request go through the streamed path pass?
bool takes_streamed_pass(const aehip_backend *b) {
    const view_consts &vc = b->vc;
    if(b->params.pass == AEHIP_PASS_PRIMARY) return false;
    if(b->all_terminal && vc.n_spheres <= static_cast<uint32_t>(kLdsSpheres)) return false;
    // The streamed pass pays a few LDS operations per path; it wins once a segment costs more
    // than that, i.e. for all but the smallest shape lists.
    // (Paths of one segment all end together: nothing to stream, the per-lane pass is leaner.)
    return b->params.list_source != 1u && !(vc.flags & AEHIP_FLAG_NO_REGEN) && vc.n_spheres >= kPoolMinSpheres &&
           vc.max_depth > 1;
}

// the backend's own scratch, as set-up sized it
stream_scratch own_scratch(const aehip_backend *b) {
    stream_scratch sc;
    sc.counters = b->d_rays;
    sc.stride = scratch_stride(b->vc.spp);
    sc.window_strips = b->window_strips;
    sc.radiance = b->d_radiance;
    sc.totals = b->d_radiance ? b->d_radiance + static_cast<size_t>(sc.window_strips) * sc.stride * 64u : nullptr;
    return sc;
}

// Does the camera pass run its single-sphere variant (sphere constants in scalar registers, hits
// shaded through 1/r) for this backend's scene?  A sphere of (nearly) no radius has no usable
// 1/r and goes through the list variant.
bool single_sphere_camera_pass(const aehip_backend *b) {
    return b->all_terminal && b->vc.n_spheres == 1 && std::isfinite(b->cc.shade_scale) && b->cc.shade_scale < 1e18f;
}

// Lists the band's ill-conditioned pixels (ring_pixels_kernel): count, size the list, fill, sort
// by row on the host so that a launch over part of the band takes a contiguous piece of it.
int build_ring_list(aehip_backend *b) {
    const view_consts &vc = b->vc;
    const uint64_t pixels = static_cast<uint64_t>(vc.width) * vc.row_count;
    const dim3 grid(static_cast<uint32_t>((pixels + 255u) / 256u));
    uint32_t *d_count = reinterpret_cast<uint32_t *>(b->d_rays + 2);          // the streamed pass's queue word: idle during set-up
    uint32_t count = 0;
    AEHIP_CHECK(hipMemsetAsync(d_count, 0, sizeof(uint32_t), b->stream));
    hipLaunchKernelGGL(ring_pixels_kernel, grid, dim3(256), 0, b->stream, vc, b->cc, d_count, static_cast<uint32_t *>(nullptr), 0u);
    AEHIP_CHECK(hipMemcpyAsync(&count, d_count, sizeof(uint32_t), hipMemcpyDeviceToHost, b->stream));
    AEHIP_CHECK(hipStreamSynchronize(b->stream));
    b->ring_row_offset.assign(static_cast<size_t>(vc.row_count) + 1, 0u);
    if(count == 0) return AEHIP_OK;
    if(count > b->ring_capacity) {
        if(b->d_ring) (void)hipFree(b->d_ring);
        b->d_ring = nullptr; b->ring_capacity = 0;
        AEHIP_CHECK(hipMalloc(reinterpret_cast<void **>(&b->d_ring), sizeof(uint32_t) * count));
        b->ring_capacity = count;
    }
    AEHIP_CHECK(hipMemsetAsync(d_count, 0, sizeof(uint32_t), b->stream));
    hipLaunchKernelGGL(ring_pixels_kernel, grid, dim3(256), 0, b->stream, vc, b->cc, d_count, b->d_ring, count);
    std::vector<uint32_t> list(count);
    AEHIP_CHECK(hipMemcpyAsync(list.data(), b->d_ring, sizeof(uint32_t) * count, hipMemcpyDeviceToHost, b->stream));
    AEHIP_CHECK(hipStreamSynchronize(b->stream));
    // row-major order (the row is the high half of an entry)
    std::sort(list.begin(), list.end());
    for(uint32_t e : list) b->ring_row_offset[static_cast<size_t>(e >> 16) + 1]++;
    for(size_t r = 0; r < vc.row_count; r++) b->ring_row_offset[r + 1] += b->ring_row_offset[r];
    AEHIP_CHECK(hipMemcpyAsync(b->d_ring, list.data(), sizeof(uint32_t) * count, hipMemcpyHostToDevice, b->stream));
    AEHIP_CHECK(hipStreamSynchronize(b->stream));
    b->ring_count = count;
    return AEHIP_OK;
}

// the band row whose image row is nearest to y (inverse of image_row, clamped to the band)
uint32_t band_row_near(const view_consts &vc, double y) {
    const double rel = y - static_cast<double>(vc.row_begin);
    if(!(rel > 0.0)) return 0u;
    const uint64_t r = rel >= 4.0e9 ? 0xffffffffull : static_cast<uint64_t>(rel);
    uint64_t row = r;
    if(vc.tile_rows != 0) {
        const uint64_t tile = r / vc.tile_stride, in_tile = r - tile * vc.tile_stride;
        row = tile * vc.tile_rows + std::min<uint64_t>(in_tile, vc.tile_rows - 1u);
    }
    return static_cast<uint32_t>(std::min<uint64_t>(row, vc.row_count - 1u));
}

// Rows of workgroups (rows_per_group band rows each) between which sphere 0 can be hit, packed
// lo | hi << 16 for edges_first_row; 1 (hi < lo: plain order) when there is no such range to
// speak of.  For a ray (u, v, dz) the discriminant is a quadratic in (u, v) (camera_consts); its
// largest value along u is non-negative between two values of v, the sphere's upper and lower
// edge on the viewport.
uint32_t sphere_group_rows(const aehip_backend *b, const view_consts &vc, uint32_t rows_per_group, uint32_t groups) {
    const camera_consts &cc = b->cc;
    const double quu = cc.q_uu, quv = cc.q_uv, qvv = cc.q_vv, qu = cc.q_u, qv = cc.q_v, q0 = cc.q_0;
    if(!(quu < 0.0)) return 1u;                             // (the camera is not outside the sphere)
    const double a2 = quv * quv - 4.0 * quu * qvv, a1 = 2.0 * quv * qu - 4.0 * quu * qv, a0 = qu * qu - 4.0 * quu * q0;
    const double d = a1 * a1 - 4.0 * a2 * a0;
    if(!(a2 < 0.0) || !(d > 0.0)) return 1u;
    const double root = std::sqrt(d);
    const double v_lo = (-a1 + root) / (2.0 * a2), v_hi = (-a1 - root) / (2.0 * a2);       // a2 < 0: v_lo < v_hi
    const double to_row = 1.0 / static_cast<double>(vc.psy), shift = static_cast<double>(vc.vh_half) + static_cast<double>(vc.cam[1]);
    const uint32_t lo = band_row_near(vc, (v_lo + shift) * to_row - 1.0) / rows_per_group;
    const uint32_t hi = std::min(band_row_near(vc, (v_hi + shift) * to_row + 1.0) / rows_per_group, groups - 1u);
    return hi < lo ? 1u : lo | (hi << 16);
}

// The box of workgroups (rows_per_group band rows by kTileW columns each) outside of which
// classify_pixel calls every pixel a miss: rows packed lo | hi << 16 like sphere_group_rows,
// columns first | count << 16.  False when there is no such box.  A pixel may be hit (or be an
// ill-conditioned one) only if at one of its corners disc >= -(1.01 kScreenMargin + 1e-5) big
// less the discriminant's second-order terms over a pixel (classify_pixel), big = h*h +
// dd*(oc.oc + r*r) at most its value there but for the pixel's width.  Both are quadratics in the
// ray's (u, v); with twice that margin their sum is one, and where its form is negative definite
// the region is an ellipse with the extent worked out here, two pixels added all round.
bool sphere_box(const aehip_backend *b, const view_consts &vc, uint32_t rows_per_group, uint32_t groups_x, uint32_t groups_y,
                uint32_t *rows, uint32_t *cols) {
    const camera_consts &cc = b->cc;
    const double ox = cc.oc[0], oy = cc.oc[1], oz = cc.oc[2], dz = 0.0 - static_cast<double>(vc.cam[2]);
    const double m = 2.0 * (1.01 * static_cast<double>(kScreenMargin) + 1e-5), big = cc.magnitude, c = cc.c;
    // h*h (1 + m) + dd (m big - c), h = u ox + v oy + dz oz, dd = u*u + v*v + dz*dz
    const double hh = 1.0 + m, k = m * big - c;
    const double quu = hh * ox * ox + k, qvv = hh * oy * oy + k, quv = 2.0 * hh * ox * oy;
    const double qu = 2.0 * hh * ox * dz * oz, qv = 2.0 * hh * oy * dz * oz;
    const double psx = vc.psx, psy = vc.psy;
    const double q0 = hh * dz * oz * dz * oz + k * dz * dz + std::fabs(quu) * psx * psx + std::fabs(quv) * psx * psy + std::fabs(qvv) * psy * psy;
    if(!(quu < 0.0) || !(qvv < 0.0)) return false;
    const double a2 = quv * quv - 4.0 * quu * qvv;
    if(!(a2 < 0.0)) return false;
    // rows: the largest value along u is non-negative between two values of v; columns likewise
    const double r1 = 2.0 * quv * qu - 4.0 * quu * qv, r0 = qu * qu - 4.0 * quu * q0;
    const double c1 = 2.0 * quv * qv - 4.0 * qvv * qu, c0 = qv * qv - 4.0 * qvv * q0;
    const double dr = r1 * r1 - 4.0 * a2 * r0, dc = c1 * c1 - 4.0 * a2 * c0;
    if(!(dr > 0.0) || !(dc > 0.0)) return false;
    const double v_lo = (-r1 + std::sqrt(dr)) / (2.0 * a2), v_hi = (-r1 - std::sqrt(dr)) / (2.0 * a2);         // a2 < 0: lo < hi
    const double u_lo = (-c1 + std::sqrt(dc)) / (2.0 * a2), u_hi = (-c1 - std::sqrt(dc)) / (2.0 * a2);
    const double y_lo = (v_lo + static_cast<double>(vc.vh_half) + static_cast<double>(vc.cam[1])) / psy - 2.0;
    const double y_hi = (v_hi + static_cast<double>(vc.vh_half) + static_cast<double>(vc.cam[1])) / psy + 2.0;
    const double x_lo = (u_lo + static_cast<double>(vc.vw_half) + static_cast<double>(vc.cam[0])) / psx - 2.0;
    const double x_hi = (u_hi + static_cast<double>(vc.vw_half) + static_cast<double>(vc.cam[0])) / psx + 2.0;
    if(!(y_lo <= y_hi) || !(x_lo <= x_hi) || !(x_hi >= 0.0) || !(x_lo < static_cast<double>(vc.width))) return false;
    const uint32_t row_lo = band_row_near(vc, y_lo) / rows_per_group;
    const uint32_t row_hi = std::min(band_row_near(vc, y_hi) / rows_per_group, groups_y - 1u);
    const uint32_t col_lo = x_lo > 0.0 ? static_cast<uint32_t>(x_lo) / static_cast<uint32_t>(kTileW) : 0u;
    const uint32_t col_hi = std::min(x_hi < 4.0e9 ? static_cast<uint32_t>(x_hi) / static_cast<uint32_t>(kTileW) : 0xffffffffu, groups_x - 1u);
    if(row_hi < row_lo || col_hi < col_lo || row_hi > 0xffffu || col_hi > 0xffffu) return false;
    *rows = row_lo | (row_hi << 16);
    *cols = col_lo | ((col_hi - col_lo + 1u) << 16);
    return true;
}

// How a camera-pass launch over the band `vc` is laid out: waves a strip, rows a workgroup, the grid, the order of its rows.
struct camera_layout {
    uint32_t split_log2, rows_per_group, row_order, split_arg;
    bool one, tiled, fold;
    dim3 grid;
};
camera_layout camera_pass_layout(const aehip_backend *b, const view_consts &vc) {
    camera_layout lay;
    const bool exact = !(vc.flags & AEHIP_FLAG_FAST);
    // waves per strip: as many as divide the pixel's sample blocks evenly (or nearly so)
    const uint32_t nblocks = (vc.spp + AEHIP_SAMPLE_BLOCK - 1u) / AEHIP_SAMPLE_BLOCK;
    lay.split_log2 = 0;
    for(uint32_t l = 2; l >= 1; l--) {
        const uint32_t waves = 1u << l;
        if(nblocks % waves == 0 || nblocks >= 4u * waves) {
            lay.split_log2 = l;
            break;
        }
    }
    // a sphere of (nearly) no radius has no usable 1/r: it goes through the list variant
    lay.one = single_sphere_camera_pass(b);
    // one sphere: a workgroup may cover a 16x16 tile, every wave a strip of its own (camera_body)
    // (measured on the 4K 1024-sample frame, fast build: the whole frame 0.55 -> 0.49 ms and half of it,
    // 4.1 M pixels, 0.316 -> 0.293 ms with tiles, a quarter 0.170 -> 0.174 ms, an eighth 0.101 -> 0.125 ms
    // -- four times fewer, four times longer waves leave a short launch's tail emptier -- and the
    // exact build, whose waves are long as it is, loses 7 % on the whole frame; with the workgroups in
    // build_tile_order's order, cold clocks: whole frame 0.401 against 0.443 ms, half 0.220 / 0.232, a quarter
    // 0.128 / 0.128, an eighth 0.083 / 0.076)
    lay.tiled = lay.one && !exact && static_cast<uint64_t>(vc.width) * vc.row_count >= 4000000ull;
    lay.rows_per_group = lay.tiled ? static_cast<uint32_t>(kTileH) : static_cast<uint32_t>(kTileH) >> lay.split_log2;
    lay.grid = dim3((vc.width + kTileW - 1) / kTileW, (vc.row_count + lay.rows_per_group - 1) / lay.rows_per_group);
    // the rows of workgroups in which the sphere can be hit, for the fast build's order of rows
    // (edges_first_row; the exact build, whose strips cost much the same wherever they are, keeps plain order)
    lay.row_order = lay.one && !exact && lay.grid.y <= 0xffffu ? sphere_group_rows(b, vc, lay.rows_per_group, lay.grid.y) : 1u;
    // (oc.oc = c + r*r: the sphere's distance from the camera against its radius)
    const bool exact_tail = lay.one && b->cc.c + b->cc.radius2 > kFastTailMaxDistance * kFastTailMaxDistance * b->cc.radius2;
    lay.split_arg = lay.split_log2 | (lay.tiled ? kCameraTiled : 0u) | (exact_tail ? kCameraExactTail : 0u);
    lay.fold = (vc.flags & kFlagFoldScreen) != 0;
    return lay;
}

// One sphere: the order in which the launch over the whole band takes its workgroups -- the ones that last longest
// first (tile_cost_kernel; ties in row-major order), those without a sample to walk last.  Fast build, a band with
// ill-conditioned pixels: the workgroups of the sphere's box.  Exact build where it culls (the folded screening
// ray): all of the grid.  A launch ends when its last workgroup does, and workgroups last from nothing (every pixel
// decided by its bounds) to 70 us (fast build, pixels of several classes): taken row by row, the 4K frame's last
// workgroups were full-length ones on an emptying chip and the short ones sat between the long ones all through the
// launch, each costing the dispatcher a slot (fast: 0.343 -> 0.324 ms a 1024-sample 4K frame; exact: 4.29 -> 4.11 ms).
// Launches over part of the band (aehip_backend_trace's) keep the row-by-row order.
int build_tile_order(aehip_backend *b) {
    const view_consts &vc = b->vc;
    b->tile_order_count = 0;
    if(b->params.pass != AEHIP_PASS_PATH || (vc.flags & AEHIP_FLAG_WALK_ALL) || !single_sphere_camera_pass(b)) return AEHIP_OK;
    const bool exact = !(vc.flags & AEHIP_FLAG_FAST);
    const camera_layout lay = camera_pass_layout(b, vc);
    if(lay.grid.x > 0xffffu || lay.grid.y > 0xffffu) return AEHIP_OK;
    uint32_t box_rows = 0, box_cols = 0;
    if(exact) {
        if(!lay.fold) return AEHIP_OK;                      // (no table of all-miss sums: the pass culls nothing)
        box_rows = (lay.grid.y - 1u) << 16;
        box_cols = lay.grid.x << 16;
    } else if(b->ring_row_offset.empty() || !sphere_box(b, vc, lay.rows_per_group, lay.grid.x, lay.grid.y, &box_rows, &box_cols)) {
        return AEHIP_OK;
    }
    const uint32_t width = box_cols >> 16, groups = width * ((box_rows >> 16) - (box_rows & 0xffffu) + 1u);
    if(groups > (1u << 22)) return AEHIP_OK;                // (16 MB of order and a sort of four million entries: beyond that, row by row)
    if(groups > b->tile_order_capacity) {
        if(b->d_tile_order) (void)hipFree(b->d_tile_order);
        b->d_tile_order = nullptr; b->tile_order_capacity = 0;
        if(hipMalloc(reinterpret_cast<void **>(&b->d_tile_order), sizeof(uint32_t) * groups) != hipSuccess) {
            (void)hipGetLastError();        // (the order is an optimisation: without it the launch goes row by row)
            b->d_tile_order = nullptr;
            return AEHIP_OK;
        }
        b->tile_order_capacity = groups;
    }
    hipLaunchKernelGGL(tile_cost_kernel, dim3(groups), dim3(kBlock), 0, b->stream, vc, b->cc, lay.rows_per_group, box_cols & 0xffffu,
                       box_rows & 0xffffu, width, exact ? 1u : 0u, b->d_tile_order);
    AEHIP_CHECK(hipGetLastError());
    std::vector<uint32_t> cost(groups), order(groups);
    AEHIP_CHECK(hipMemcpyAsync(cost.data(), b->d_tile_order, sizeof(uint32_t) * groups, hipMemcpyDeviceToHost, b->stream));
    AEHIP_CHECK(hipStreamSynchronize(b->stream));
    for(uint32_t i = 0; i < groups; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t l, uint32_t r) { return cost[l] > cost[r]; });
    for(uint32_t i = 0; i < groups; i++) {
        const uint32_t id = order[i], line = id / width;
        cost[i] = ((box_cols & 0xffffu) + (id - line * width)) | (((box_rows & 0xffffu) + line) << 16);     // (cost[] is done with: reused for the packed order)
    }
    AEHIP_CHECK(hipMemcpyAsync(b->d_tile_order, cost.data(), sizeof(uint32_t) * groups, hipMemcpyHostToDevice, b->stream));
    AEHIP_CHECK(hipStreamSynchronize(b->stream));
    b->tile_order_count = groups;
    b->tile_order_box_rows = box_rows;
    b->tile_order_box_cols = box_cols;
    b->tile_order_rows_per_group = lay.rows_per_group;
    b->tile_order_row_begin = vc.row_begin;
    b->tile_order_row_count = vc.row_count;
    return AEHIP_OK;
}

// Clears the ray counters and the unit queue.  A kernel rather than hipMemsetAsync: under stream capture
// the memset nodes of these 8- and 16-byte clears did not take effect on replay (the counters came back
// with the array's old contents), while kernel nodes replay as launched.
__global__ __launch_bounds__(64) void clear_words_kernel(unsigned long long *__restrict__ words, uint32_t count) {
    if(threadIdx.x < count) words[threadIdx.x] = 0ull;
}

// The tracing kernel of the streamed pass for this backend's request (decided by the build, the list's
// length and set-up's estimate of what the list prefilter saves).
const void *streamed_kernel(const aehip_backend *b) {
    const bool exact = !(b->vc.flags & AEHIP_FLAG_FAST);
    // the list prefilter (8 operations a sphere) pays where the screening value costs 12 (exact
    // build; the fast build's 10 are not enough to cover the groups that pass) and few groups
    // of four pass it in any lane (set-up's estimate)
    const bool prefilter = exact && b->prefilter_pays;
    if(b->vc.n_spheres >= kFirstHitsMinSpheres) {
        if(exact) return prefilter ? reinterpret_cast<const void *>(path_stream_first_hits_kernel<true, true>) : reinterpret_cast<const void *>(path_stream_first_hits_kernel<true, false>);
        return prefilter ? reinterpret_cast<const void *>(path_stream_first_hits_kernel<false, true>) : reinterpret_cast<const void *>(path_stream_first_hits_kernel<false, false>);
    }
    if(exact) return prefilter ? reinterpret_cast<const void *>(path_stream_kernel<true, true>) : reinterpret_cast<const void *>(path_stream_kernel<true, false>);
    return prefilter ? reinterpret_cast<const void *>(path_stream_kernel<false, true>) : reinterpret_cast<const void *>(path_stream_kernel<false, false>);
}

// Is `stream` being captured into a HIP graph?  (false for the null stream and on any error: a stream that cannot be asked is not capturing)
bool stream_is_capturing(hipStream_t stream) {
    if(stream == nullptr) return false;
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    if(hipStreamIsCapturing(stream, &status) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return status != hipStreamCaptureStatusNone;
}

int launch(aehip_backend *b, uint32_t *argb_dev, float *rgba_dev, hipStream_t stream, const stream_scratch &scratch) {
    const view_consts &vc = b->vc;
    // Launches of one backend share one scratch: a launch that uses it (the streamed pass, the ray
    // counters) is ordered after the last one that did, whatever streams the two were given.
    // (Under stream capture the launch leaves events alone: a graph's replays are ordered by the streams
    // they are launched on.)
    const bool uses_scratch = (vc.flags & AEHIP_FLAG_COUNT_RAYS) || takes_streamed_pass(b);
    const bool capturing = uses_scratch && stream_is_capturing(stream);        // (asked only where the answer matters: a launch without scratch has no event to leave alone)
    // the fixed array of aehip_backend_trace_device_static is shared by every backend on the GPU: its users are
    // ordered by the GPU's event as well as by the backend's own
    hipEvent_t shared_event = nullptr;
    if(uses_scratch && b->device >= 0 && b->device < kMaxDevices && g_static_scratch_ptr[b->device] != nullptr &&
       reinterpret_cast<unsigned char *>(scratch.counters) == g_static_scratch_ptr[b->device])
        shared_event = g_static_scratch_event[b->device];
    if(uses_scratch && !capturing) {
        AEHIP_CHECK(hipStreamWaitEvent(stream, b->ev_scratch, 0));
        if(shared_event) AEHIP_CHECK(hipStreamWaitEvent(stream, shared_event, 0));
    }
    const bool exact = !(vc.flags & AEHIP_FLAG_FAST);
    float4 *rgba = reinterpret_cast<float4 *>(rgba_dev);
    const void *func = nullptr;

    if(vc.flags & AEHIP_FLAG_COUNT_RAYS) hipLaunchKernelGGL(clear_words_kernel, dim3(1), dim3(64), 0, stream, scratch.counters, 2u);
    b->last_counters = scratch.counters;
    b->counts_known = false;

    if(b->params.pass == AEHIP_PASS_PRIMARY) {
        const bool vec4 = (vc.width % 4u) == 0 && (reinterpret_cast<uintptr_t>(argb_dev) % 16u) == 0;
        // the sphere's box of cells (kPrimaryBoxRows x kTileW pixels), or the whole band; around a box
        // the launch's last workgroups write the rows' background
        const uint32_t cells_x = (vc.width + kTileW - 1u) / kTileW, cells_y = (vc.row_count + kPrimaryBoxRows - 1u) / kPrimaryBoxRows;
        uint32_t box_rows = 0, box_cols = 0, fill_groups = 0, bx0 = 0, bx1 = vc.width, by0 = 0, by1 = vc.row_count;
        if(cells_x <= 0xffffu && cells_y <= 0xffffu && vc.n_spheres >= 1u && sphere_box(b, vc, kPrimaryBoxRows, cells_x, cells_y, &box_rows, &box_cols)) {
            fill_groups = cells_y * kFillSegments;
            bx0 = (box_cols & 0xffffu) * kTileW;
            bx1 = std::min(((box_cols & 0xffffu) + (box_cols >> 16)) * static_cast<uint32_t>(kTileW), vc.width);
            by0 = (box_rows & 0xffffu) * kPrimaryBoxRows;
            by1 = std::min(((box_rows >> 16) + 1u) * kPrimaryBoxRows, vc.row_count);
        }
        const uint64_t box_w = bx1 - bx0, box_h = by1 - by0;
        const uint64_t groups = box_w / (vec4 ? 4u : 1u) * box_h;
        const uint32_t compute_groups = static_cast<uint32_t>((groups + kPrimaryBlock - 1) / kPrimaryBlock);
        const dim3 grid(compute_groups + fill_groups);
        if(exact && vec4) { func = reinterpret_cast<const void *>(primary_kernel<true, true>); hipLaunchKernelGGL((primary_kernel<true, true>), grid, dim3(kPrimaryBlock), 0, stream, vc, b->cc, argb_dev, rgba, box_rows, box_cols, bx0, bx1, by0, by1, compute_groups); }
        else if(exact) { func = reinterpret_cast<const void *>(primary_kernel<true, false>); hipLaunchKernelGGL((primary_kernel<true, false>), grid, dim3(kPrimaryBlock), 0, stream, vc, b->cc, argb_dev, rgba, box_rows, box_cols, bx0, bx1, by0, by1, compute_groups); }
        else if(vec4) { func = reinterpret_cast<const void *>(primary_kernel<false, true>); hipLaunchKernelGGL((primary_kernel<false, true>), grid, dim3(kPrimaryBlock), 0, stream, vc, b->cc, argb_dev, rgba, box_rows, box_cols, bx0, bx1, by0, by1, compute_groups); }
        else { func = reinterpret_cast<const void *>(primary_kernel<false, false>); hipLaunchKernelGGL((primary_kernel<false, false>), grid, dim3(kPrimaryBlock), 0, stream, vc, b->cc, argb_dev, rgba, box_rows, box_cols, bx0, bx1, by0, by1, compute_groups); }
    } else if(b->all_terminal && vc.n_spheres <= static_cast<uint32_t>(kLdsSpheres)) {
        const camera_layout lay = camera_pass_layout(b, vc);
        const uint32_t split_log2 = lay.split_log2, rows_per_group = lay.rows_per_group, row_order = lay.row_order, split_arg = lay.split_arg;
        const bool one = lay.one, fold = lay.fold;
        const dim3 grid = lay.grid;
#define AEHIP_LAUNCH_CAMERA(E, O, F, A)                                                                                \
    do {                                                                                                               \
        if((E) && (O) && (F) && grid_order) {                                                                          \
            func = reinterpret_cast<const void *>(camera_kernel<E, O, F, A, (E) && (O) && (F)>);                       \
            hipLaunchKernelGGL((camera_kernel<E, O, F, A, (E) && (O) && (F)>), grid, dim3(kBlock), 0, stream, vc, b->cc, split_arg, row_order, \
                               b->d_geom, b->d_sample_terms, argb_dev, rgba, b->d_row_miss, grid_order);               \
        } else {                                                                                                       \
            func = reinterpret_cast<const void *>(camera_kernel<E, O, F, A>);                                          \
            hipLaunchKernelGGL((camera_kernel<E, O, F, A>), grid, dim3(kBlock), 0, stream, vc, b->cc, split_arg, row_order, \
                               b->d_geom, b->d_sample_terms, argb_dev, rgba, (E) && (O) && (F) && !(vc.flags & AEHIP_FLAG_WALK_ALL) ? b->d_row_miss : nullptr, \
                               static_cast<const uint32_t *>(nullptr));                                                \
        }                                                                                                              \
    } while(0)
        // set-up's order of the whole grid's workgroups (exact build), if this launch is the one it was made for
        const uint32_t *grid_order = exact && !(vc.flags & AEHIP_FLAG_WALK_ALL) && b->tile_order_count == grid.x * grid.y &&
                                             b->tile_order_box_cols == (grid.x << 16) && b->tile_order_box_rows == ((grid.y - 1u) << 16) &&
                                             b->tile_order_rows_per_group == rows_per_group && b->tile_order_row_begin == vc.row_begin &&
                                             b->tile_order_row_count == vc.row_count
                                         ? b->d_tile_order : nullptr;
        // one ray per sample, whether the sample loop walks it or the pixel's bounds decide it; a
        // strip's 64 lanes are counted for each of its samples
        b->counts_known = true;
        b->known_rays = static_cast<uint64_t>(vc.width) * vc.row_count * vc.spp;
        b->known_lane_slots = static_cast<uint64_t>((vc.width + kTileW - 1) / kTileW) * ((vc.row_count + 3u) / 4u) * 64u * vc.spp;
        // the quadratic's cross term vanishes when the sphere's centre is on the camera axis
        const bool axial = one && b->cc.q_uv == 0.0f;
        if(exact) {
            if(one && fold && axial) AEHIP_LAUNCH_CAMERA(true, true, true, true);
            else if(one && fold) AEHIP_LAUNCH_CAMERA(true, true, true, false);
            else if(one) AEHIP_LAUNCH_CAMERA(true, true, false, false);
            else if(fold) AEHIP_LAUNCH_CAMERA(true, false, true, false);
            else AEHIP_LAUNCH_CAMERA(true, false, false, false);
        } else {
            if(one && !b->ring_row_offset.empty()) {
                // the band's ill-conditioned pixels (if it has any), those of this launch's rows, come
                // first in the grid, from the list, in the reference's arithmetic
                const uint32_t row0 = vc.tile_rows == 0 ? vc.row_begin - b->params.row_begin : 0u;
                const uint32_t first = b->ring_row_offset[row0], last = b->ring_row_offset[row0 + vc.row_count];
                const uint32_t per_group = (blocks_are_pairs_of_rows(vc) ? 32u : 64u) * (static_cast<uint32_t>(kBlock / 64) >> split_log2);
                const uint32_t list_groups = (last - first + per_group - 1u) / per_group;
                // where the sphere's box of workgroups is known, only its workgroups render tiles and
                // fill_background's write the rest of the band (not when every sample is to be walked)
                uint32_t box_rows = row_order, box_cols = grid.x << 16, box_groups = grid.x * grid.y, fill_groups = 0;
                if(grid.x <= 0xffffu && grid.y <= 0xffffu && !(vc.flags & AEHIP_FLAG_WALK_ALL) &&
                   sphere_box(b, vc, rows_per_group, grid.x, grid.y, &box_rows, &box_cols)) {
                    box_groups = (box_cols >> 16) * ((box_rows >> 16) - (box_rows & 0xffffu) + 1u);
                    fill_groups = grid.y * kFillSegments;
                }
                const dim3 line(list_groups + box_groups + fill_groups);
                // set-up's order of the box's workgroups, if this launch is the one it was made for
                const uint32_t *tile_order = fill_groups != 0 && b->tile_order_count == box_groups && b->tile_order_box_rows == box_rows &&
                                                     b->tile_order_box_cols == box_cols && b->tile_order_rows_per_group == rows_per_group &&
                                                     b->tile_order_row_begin == vc.row_begin && b->tile_order_row_count == vc.row_count
                                                 ? b->d_tile_order : nullptr;
#define AEHIP_LAUNCH_HYBRID(F, A, W)                                                                                   \
    do {                                                                                                               \
        func = reinterpret_cast<const void *>(camera_hybrid_kernel<F, A, W>);                                          \
        hipLaunchKernelGGL((camera_hybrid_kernel<F, A, W>), line, dim3(kBlock), 0, stream, vc, b->cc, split_arg, box_rows, \
                           b->d_sample_terms, argb_dev, rgba, b->d_ring + first, last - first, row0, list_groups, box_cols, \
                           box_groups, rows_per_group, tile_order);                                                    \
    } while(0)
                const bool walk = (vc.flags & AEHIP_FLAG_WALK_ALL) != 0;
                if(fold && axial) { if(walk) AEHIP_LAUNCH_HYBRID(true, true, true); else AEHIP_LAUNCH_HYBRID(true, true, false); }
                else if(fold) { if(walk) AEHIP_LAUNCH_HYBRID(true, false, true); else AEHIP_LAUNCH_HYBRID(true, false, false); }
                else { if(walk) AEHIP_LAUNCH_HYBRID(false, false, true); else AEHIP_LAUNCH_HYBRID(false, false, false); }
#undef AEHIP_LAUNCH_HYBRID
            }
            else if(one && axial) AEHIP_LAUNCH_CAMERA(false, true, false, true);
            else if(one) AEHIP_LAUNCH_CAMERA(false, true, false, false);
            else AEHIP_LAUNCH_CAMERA(false, false, false, false);
        }
#undef AEHIP_LAUNCH_CAMERA
    } else {
        const dim3 grid((vc.width + kTileW - 1) / kTileW, (vc.row_count + kTileH - 1) / kTileH);
        // List source.  Measured on MI355X the scalar-cache stream is as fast as or faster than LDS
        // staging at every list length (no barrier, 8 waves/SIMD instead of 5), so it is the
        // default; list_source == 1 in the request selects LDS staging.
        int list = kListScalar;
        if(b->params.list_source == 1u) list = vc.n_spheres <= static_cast<uint32_t>(kLdsSpheres) ? kListLdsOnce : kListLdsChunked;
#define AEHIP_LAUNCH_PATH(E, L)                                                                                          \
    do {                                                                                                               \
        func = reinterpret_cast<const void *>(path_kernel<E, L>);                                                      \
        hipLaunchKernelGGL((path_kernel<E, L>), grid, dim3(kBlock), 0, stream, vc, b->d_geom, b->d_matl, b->d_kind,    \
                           b->d_sample_terms, argb_dev, rgba, scratch.counters, b->d_camera);                          \
    } while(0)
        if(takes_streamed_pass(b)) {
            // persistent waves: no more workgroups than the chip holds at once (a larger grid
            // would be just as correct -- late waves would find the queue empty); one launch
            // pair per block of samples.  The kernel and its residency are set-up's (streamed_kernel).
            const bool first_hits = vc.n_spheres >= kFirstHitsMinSpheres;
            const uint32_t deal_rounds = kDealRounds;
            func = b->stream_func;
            const int per_cu = b->stream_groups_per_cu;
            if(!func) return fail(AEHIP_E_INVALID, "streamed pass without a kernel (set-up did not choose one for this request)");
            const uint32_t strips = ((vc.width + kTileW - 1) / kTileW) * ((vc.row_count + 3u) / 4u);
            const uint32_t resident = static_cast<uint32_t>(per_cu > 0 ? per_cu : 1) * static_cast<uint32_t>(b->cu_count);
            const uint32_t nblocks = (vc.spp + AEHIP_SAMPLE_BLOCK - 1u) / AEHIP_SAMPLE_BLOCK;
            // Units a wave takes per atomic.  Every wave of the launch adds to the same counter and
            // atomics on one address go through one at a time (13 ns each measured: 130,000 single
            // takes are 1.7 ms of a 1080p 16-spp launch), so cheap units are taken several at a
            // time; costly ones singly, for the finest balance at the end of the launch.
            const uint32_t unit_batch = std::min(16u, std::max(1u, 128u / vc.n_spheres));
            if(!scratch.radiance || !scratch.totals || !scratch.counters || scratch.window_strips == 0 || scratch.stride < scratch_stride(vc.spp))
                return fail(AEHIP_E_INVALID, "streamed pass without scratch (set-up did not size it for this request)");
            uint32_t *queue = reinterpret_cast<uint32_t *>(scratch.counters + 2);
            // window after window of strips, each through all its sample blocks
            for(uint32_t first = 0; first < strips; first += scratch.window_strips) {
                const strip_window win{first, std::min(scratch.window_strips, strips - first), scratch.stride};
                for(uint32_t blk = 0; blk < nblocks; blk++) {
                    const uint32_t begin = blk * AEHIP_SAMPLE_BLOCK;
                    const uint32_t count = std::min(AEHIP_SAMPLE_BLOCK, vc.spp - begin);
                    const uint64_t units = static_cast<uint64_t>(win.count) * ((count + kChunkSamples - 1u) / kChunkSamples);
                    const uint32_t groups = static_cast<uint32_t>(std::min<uint64_t>(resident, (units + 3u) / 4u));
                    hipLaunchKernelGGL(clear_words_kernel, dim3(1), dim3(64), 0, stream, scratch.counters + 2, 1u);
                    if(first_hits) {
                        const dim3 first_grid(static_cast<uint32_t>((units + 3u) / 4u));
                        if(exact) hipLaunchKernelGGL(first_hit_kernel<true>, first_grid, dim3(kBlock), 0, stream, vc, b->d_geom, b->d_camera, b->d_sample_terms, scratch.radiance, begin, count, scratch.counters, win);
                        else hipLaunchKernelGGL(first_hit_kernel<false>, first_grid, dim3(kBlock), 0, stream, vc, b->d_geom, b->d_camera, b->d_sample_terms, scratch.radiance, begin, count, scratch.counters, win);
                    }
                    radiance3 *radiance = scratch.radiance;
                    unsigned long long *counters = scratch.counters;
                    void *args[] = {const_cast<view_consts *>(&vc), &b->d_geom, &b->d_prefilter, &b->d_matl, &b->d_kind, &b->d_sample_terms,
                                    &radiance, const_cast<uint32_t *>(&begin), const_cast<uint32_t *>(&count), &counters, &queue,
                                    const_cast<uint32_t *>(&unit_batch), const_cast<strip_window *>(&win), const_cast<uint32_t *>(&deal_rounds)};
                    // (the kernel without first hits has no last parameter and ignores the extra entry)
                    AEHIP_CHECK(hipLaunchKernel(func, dim3(groups), dim3(kBlock), args, 0, stream));
                    hipLaunchKernelGGL(resolve_block_kernel, dim3((win.count + 3u) / 4u), dim3(kBlock), 0, stream, vc, scratch.radiance, scratch.totals,
                                       count, blk == 0 ? 1u : 0u, blk + 1 == nblocks ? 1u : 0u, argb_dev, rgba, win);
                }
            }
        } else if(exact) {
            if(list == kListLdsOnce) AEHIP_LAUNCH_PATH(true, kListLdsOnce);
            else if(list == kListLdsChunked) AEHIP_LAUNCH_PATH(true, kListLdsChunked);
            else AEHIP_LAUNCH_PATH(true, kListScalar);
        } else {
            if(list == kListLdsOnce) AEHIP_LAUNCH_PATH(false, kListLdsOnce);
            else if(list == kListLdsChunked) AEHIP_LAUNCH_PATH(false, kListLdsChunked);
            else AEHIP_LAUNCH_PATH(false, kListScalar);
        }
#undef AEHIP_LAUNCH_PATH
    }
    AEHIP_CHECK(hipGetLastError());
    if(uses_scratch && !capturing) {
        AEHIP_CHECK(hipEventRecord(b->ev_scratch, stream));
        if(shared_event) AEHIP_CHECK(hipEventRecord(shared_event, stream));
    }

    // resource figures of the kernel just launched; queried once per kernel and backend, not per launch
    // (and not while a graph is being captured: the figures then stay those of the last launch outside capture)
    if(func != b->attr_func && !stream_is_capturing(stream)) {
        hipFuncAttributes attr;
        if(hipFuncGetAttributes(&attr, func) == hipSuccess) {
            b->stats.vgprs = static_cast<uint32_t>(attr.numRegs);
            b->stats.lds_bytes = static_cast<uint32_t>(attr.sharedSizeBytes);
        }
        b->attr_func = func;
    }
    b->stats.paths = static_cast<uint64_t>(vc.width) * vc.row_count * (b->params.pass == AEHIP_PASS_PRIMARY ? 1u : vc.spp);
    return AEHIP_OK;
}

}  // namespace

extern "C" {

int aehip_init(void) {
    if(g_initialised) return g_device_count > 0 ? AEHIP_OK : AEHIP_E_NO_DEVICE;
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if(e != hipSuccess || n <= 0) {
        g_device_count = 0;
        return fail(AEHIP_E_NO_DEVICE, "no HIP device: %s", e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    }
    g_device_count = n;
    g_initialised = true;
    return AEHIP_OK;
}

void aehip_terminate(void) {
    release_frame_state();
    g_initialised = false;
    g_device_count = 0;
}

int aehip_device_count(void) { return g_initialised ? g_device_count : 0; }

const char *aehip_last_error(void) { return g_error; }

int aehip_backend_create(int device, uint32_t *framebuffer, aehip_backend **out) {
    if(!out) return fail(AEHIP_E_INVALID, "aehip_backend_create: out is null");
    *out = nullptr;
    if(!g_initialised) {
        const int rc = aehip_init();
        if(rc != AEHIP_OK) return rc;
    }
    if(device < 0 || device >= g_device_count) return fail(AEHIP_E_INVALID, "device %d out of range (0..%d)", device, g_device_count - 1);

    aehip_backend *b = new(std::nothrow) aehip_backend();
    if(!b) return fail(AEHIP_E_INVALID, "out of host memory");
    b->device = device;
    b->framebuffer = framebuffer;

    hipError_t e = hipSetDevice(device);
    if(e == hipSuccess) e = resolve_static_scratch(device);
    if(e == hipSuccess) e = hipDeviceGetAttribute(&b->cu_count, hipDeviceAttributeMultiprocessorCount, device);
    // a blocking stream: ordered after whatever the caller queued on the device's null stream
    // (a buffer it just cleared, say) before handing us that buffer
    if(e == hipSuccess) e = hipStreamCreateWithFlags(&b->stream, hipStreamDefault);
    if(e == hipSuccess) e = hipEventCreate(&b->ev_begin);
    if(e == hipSuccess) e = hipEventCreate(&b->ev_end);
    if(e == hipSuccess) e = hipEventCreate(&b->ev_copy);
    if(e == hipSuccess) e = hipEventCreateWithFlags(&b->ev_scratch, hipEventDisableTiming);
    if(e == hipSuccess) e = hipStreamCreateWithFlags(&b->copy_stream, hipStreamNonBlocking);
    for(int k = 0; k < kTraceParts && e == hipSuccess; k++) {
        e = hipEventCreateWithFlags(&b->ev_part_done[k], hipEventDisableTiming);
        if(e == hipSuccess) e = hipEventCreateWithFlags(&b->ev_part_copied[k], hipEventDisableTiming);
    }
    if(e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_rays), 3 * sizeof(unsigned long long));
    if(e != hipSuccess) {
        const int rc = fail(AEHIP_E_HIP, "backend create on device %d: %s", device, hipGetErrorString(e));
        aehip_backend_destroy(b);
        return rc;
    }
    *out = b;
    return AEHIP_OK;
}

void aehip_backend_destroy(aehip_backend *b) {
    if(!b) return;
    (void)hipSetDevice(b->device);
    if(b->stream) (void)hipStreamSynchronize(b->stream);
    if(b->d_geom) (void)hipFree(b->d_geom);
    if(b->d_matl) (void)hipFree(b->d_matl);
    if(b->d_kind) (void)hipFree(b->d_kind);
    if(b->d_camera) (void)hipFree(b->d_camera);
    if(b->d_prefilter) (void)hipFree(b->d_prefilter);
    if(b->d_ring) (void)hipFree(b->d_ring);
    if(b->d_tile_order) (void)hipFree(b->d_tile_order);
    if(b->d_argb) (void)hipFree(b->d_argb);
    if(b->d_rays) (void)hipFree(b->d_rays);
    if(b->d_sample_terms) (void)hipFree(b->d_sample_terms);
    if(b->d_row_miss) (void)hipFree(b->d_row_miss);
    if(b->d_radiance) (void)hipFree(b->d_radiance);
    if(b->ev_scratch) (void)hipEventDestroy(b->ev_scratch);
    if(b->h_staging) (void)hipHostFree(b->h_staging);
    if(b->ev_begin) (void)hipEventDestroy(b->ev_begin);
    if(b->ev_end) (void)hipEventDestroy(b->ev_end);
    if(b->ev_copy) (void)hipEventDestroy(b->ev_copy);
    for(int k = 0; k < kTraceParts; k++) {
        if(b->ev_part_done[k]) (void)hipEventDestroy(b->ev_part_done[k]);
        if(b->ev_part_copied[k]) (void)hipEventDestroy(b->ev_part_copied[k]);
    }
    if(b->copy_stream) (void)hipStreamDestroy(b->copy_stream);
    if(b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
}

int aehip_backend_setup(aehip_backend *b, const aehip_scene *scene, const aehip_params *params) {
    if(!b || !scene || !params) return fail(AEHIP_E_INVALID, "aehip_backend_setup: null argument");
    b->ready = false;
    aehip_params p = *params;
    if(p.width == 0 || p.height == 0) return fail(AEHIP_E_INVALID, "resolution %ux%u is empty", p.width, p.height);
    if(p.width > 65535u || p.height > 65535u) return fail(AEHIP_E_TOO_LARGE, "resolution %ux%u exceeds 65535 per side", p.width, p.height);
    if(p.row_begin >= p.height) return fail(AEHIP_E_INVALID, "row_begin %u outside image of %u rows", p.row_begin, p.height);
    if(p.tile_rows == 0) {
        p.tile_stride = 0;
        if(p.row_count == 0) p.row_count = p.height - p.row_begin;
        if(p.row_begin + static_cast<uint64_t>(p.row_count) > p.height) return fail(AEHIP_E_INVALID, "rows [%u,%u) outside image of %u rows", p.row_begin, p.row_begin + p.row_count, p.height);
    } else {
        if(p.tile_stride < p.tile_rows) return fail(AEHIP_E_INVALID, "tile_stride %u is less than tile_rows %u", p.tile_stride, p.tile_rows);
        // rows the band's tiles have below the image's last row
        const uint32_t span = p.height - p.row_begin;
        const uint32_t whole = span / p.tile_stride, rest = span % p.tile_stride;
        const uint64_t avail = static_cast<uint64_t>(whole) * p.tile_rows + (rest < p.tile_rows ? rest : p.tile_rows);
        if(p.row_count == 0) p.row_count = static_cast<uint32_t>(avail);
        if(p.row_count > avail) return fail(AEHIP_E_INVALID, "%u rows in tiles of %u every %u rows from row %u leave the image of %u rows", p.row_count, p.tile_rows, p.tile_stride, p.row_begin, p.height);
    }
    if(p.pass != AEHIP_PASS_PRIMARY && p.pass != AEHIP_PASS_PATH) return fail(AEHIP_E_INVALID, "unknown pass %u", p.pass);
    if(p.spp == 0 || p.max_depth == 0) return fail(AEHIP_E_INVALID, "spp and max_depth must be >= 1");
    if(scene->sphere_count == 0 || !scene->spheres) return fail(AEHIP_E_INVALID, "scene has no spheres");
    // Coordinates must be finite and moderate: the kernels' screening tests assume that squared
    // distances stay far from the fp32 overflow range.
    auto moderate = [](float v) { return std::isfinite(v) && std::fabs(v) <= 1e11f; };
    for(int k = 0; k < 3; k++) {
        if(!moderate(scene->camera_pos[k])) return fail(AEHIP_E_INVALID, "camera position is not finite or beyond 1e11");
    }
    for(uint32_t i = 0; i < scene->sphere_count; i++) {
        const aehip_sphere &sp = scene->spheres[i];
        if(sp.material > AEHIP_MAT_DIELECTRIC) return fail(AEHIP_E_INVALID, "sphere %u: unknown material %u", i, sp.material);
        if(!moderate(sp.center[0]) || !moderate(sp.center[1]) || !moderate(sp.center[2]) || !moderate(sp.radius))
            return fail(AEHIP_E_INVALID, "sphere %u: centre or radius is not finite or beyond 1e11", i);
    }

    AEHIP_CHECK(hipSetDevice(b->device));

    // viewport: src/software_raytracer.cpp:68-91
    float vw, vh;
    if(p.width > p.height) {
        vw = static_cast<float>(p.width) / static_cast<float>(p.height);
        vh = 1.0f;
    } else {
        vw = 1.0f;
        vh = static_cast<float>(p.height) / static_cast<float>(p.width);
    }
    view_consts &vc = b->vc;
    vc.vw_half = vw * 0.5f;
    vc.vh_half = vh * 0.5f;
    vc.psx = vw / static_cast<float>(p.width);
    vc.psy = vh / static_cast<float>(p.height);
    for(int i = 0; i < 3; i++) {
        vc.cam[i] = scene->camera_pos[i];
        vc.bg0[i] = scene->background0[i];
        vc.bg1[i] = scene->background1[i];
    }
    vc.width = p.width;
    vc.height = p.height;
    vc.row_begin = p.row_begin;
    vc.row_count = p.row_count;
    vc.tile_rows = p.tile_rows;
    vc.tile_stride = p.tile_stride;
    vc.spp = p.spp;
    vc.max_depth = p.max_depth;
    vc.seed_hash = pcg_hash(p.seed);
    {
        uint32_t bits = 0;
        while((2u << bits) <= p.spp && bits < 31) bits++;       // floor(log2(spp))
        vc.row_bits = bits / 2;
        vc.inv_spp = 1.0f / static_cast<float>(p.spp);
    }
    vc.flags = p.flags & ~kFlagFoldScreen;
    {
        // The folded screening ray is off from the exact one by a few ulps of the viewport extent
        // (camera offset included); relative to the ray length, at least |camera.z|, that stays far
        // inside the screening margin when the camera is not close to the viewport plane.
        const float extent = std::fmax(vc.vw_half, vc.vh_half) + std::fmax(std::fabs(vc.cam[0]), std::fabs(vc.cam[1]));
        if(std::fabs(vc.cam[2]) >= 0.25f * extent) vc.flags |= kFlagFoldScreen;
    }
    vc.n_spheres = scene->sphere_count;

    // scene upload (replicated on every GPU that renders a band)
    const uint32_t n = scene->sphere_count;
    if(n > b->scene_capacity) {
        if(b->d_geom) (void)hipFree(b->d_geom);
        if(b->d_matl) (void)hipFree(b->d_matl);
        if(b->d_kind) (void)hipFree(b->d_kind);
        if(b->d_camera) (void)hipFree(b->d_camera);
            if(b->d_prefilter) (void)hipFree(b->d_prefilter);
        b->d_geom = nullptr; b->d_matl = nullptr; b->d_kind = nullptr; b->d_camera = nullptr; b->d_prefilter = nullptr; b->scene_capacity = 0;
        AEHIP_CHECK(hipMalloc(reinterpret_cast<void **>(&b->d_geom), sizeof(float4) * n));
        AEHIP_CHECK(hipMalloc(reinterpret_cast<void **>(&b->d_camera), sizeof(float4) * ((n + 3u) / 4u * 4u)));      // whole groups of four
        AEHIP_CHECK(hipMalloc(reinterpret_cast<void **>(&b->d_prefilter), sizeof(float4) * n));
        AEHIP_CHECK(hipMalloc(reinterpret_cast<void **>(&b->d_matl), sizeof(float4) * n));
        AEHIP_CHECK(hipMalloc(reinterpret_cast<void **>(&b->d_kind), sizeof(uint32_t) * n));
        b->scene_capacity = n;
    }
    {
        float4 *geom = static_cast<float4 *>(std::malloc(sizeof(float4) * n));
        float4 *matl = static_cast<float4 *>(std::malloc(sizeof(float4) * n));
        uint32_t *kind = static_cast<uint32_t *>(std::malloc(sizeof(uint32_t) * n));
        const uint32_t n_padded = (n + 3u) / 4u * 4u;
        float4 *camera = static_cast<float4 *>(std::malloc(sizeof(float4) * n_padded));
        float4 *prefilter = static_cast<float4 *>(std::malloc(sizeof(float4) * n));
        if(!geom || !matl || !kind || !camera || !prefilter) {
            std::free(geom); std::free(matl); std::free(kind); std::free(camera); std::free(prefilter);
            return fail(AEHIP_E_INVALID, "out of host memory staging %u spheres", n);
        }
        for(uint32_t i = 0; i < n; i++) {
            const aehip_sphere &s = scene->spheres[i];
            const float r = std::fabs(s.radius);                      // src/shapes.cpp:10-12
            geom[i] = make_float4(s.center[0], s.center[1], s.center[2], r * r);
            matl[i] = make_float4(s.albedo[0], s.albedo[1], s.albedo[2], s.param);
            kind[i] = s.material;
            // camera entry: screen_sphere's oc and c for a ray from the camera, operation for operation
            const float ocx = geom[i].x - vc.cam[0], ocy = geom[i].y - vc.cam[1], ocz = geom[i].z - vc.cam[2];
            const float c = std::fmaf(ocz, ocz, std::fmaf(ocy, ocy, std::fmaf(ocx, ocx, -geom[i].w)));
            const float u = (p.flags & AEHIP_FLAG_FAST) ? -c : std::fmaf(c, -(1.0f - 2.0f * kScreenMargin), geom[i].w * (3.0f * kScreenMargin));
            camera[i] = make_float4(ocx, ocy, ocz, u);
        }
        {
            // prefilter entries: centres measured from the component-wise median of the centres (a
            // far ground sphere does not drag it away from where the small spheres are), K' in double
            std::vector<float> axis(n);
            for(int k = 0; k < 3; k++) {
                for(uint32_t i = 0; i < n; i++) axis[i] = k == 0 ? geom[i].x : k == 1 ? geom[i].y : geom[i].z;
                std::nth_element(axis.begin(), axis.begin() + n / 2, axis.end());
                vc.list_centre[k] = std::isfinite(axis[n / 2]) ? axis[n / 2] : 0.0f;
            }
            const double f = (p.flags & AEHIP_FLAG_FAST) ? 1.0 : 1.0 - 2.0 * static_cast<double>(kScreenMargin);
            const double m3 = (p.flags & AEHIP_FLAG_FAST) ? 0.0 : 3.0 * static_cast<double>(kScreenMargin);
            std::vector<double> distance2(n), section(n);
            for(uint32_t i = 0; i < n; i++) {
                const float cx = geom[i].x - vc.list_centre[0], cy = geom[i].y - vc.list_centre[1], cz = geom[i].z - vc.list_centre[2];
                const double cc = static_cast<double>(cx) * cx + static_cast<double>(cy) * cy + static_cast<double>(cz) * cz;
                const double r2 = geom[i].w, reach = std::sqrt(cc) + std::sqrt(r2);
                const double k = f * (cc - r2) - m3 * r2 - static_cast<double>(kPrefilterMargin) * reach * reach;
                // rounded down: K' enters the filter's value with a minus sign
                float kf = static_cast<float>(k);
                if(static_cast<double>(kf) > k) kf = std::nextafterf(kf, -INFINITY);
                prefilter[i] = make_float4(cx, cy, cz, kf);
                distance2[i] = cc;
                section[i] = r2 + static_cast<double>(kPrefilterMargin) * reach * reach;     // the filter's cross-section / pi
            }
            // Does the filter pay?  A group of four goes on to the full screening when any of the
            // wave's 64 rays passes any of its spheres, so it must be rare per ray and sphere: a
            // line through a scene of radius R passes a sphere with probability ~ section / R^2
            // (R: the 90th percentile of the centres' distances, so that a far ground sphere counts
            // as "always passes" and nothing more).  Measured: 10,000 small spheres in a 40-unit
            // slab (2 % of the groups) 46.9 -> 39.7 ms; 1,200 in a dense cloud (half of them) 37.2 -> 42.1.
            std::vector<double> sorted(distance2);
            std::nth_element(sorted.begin(), sorted.begin() + (n - 1) * 9 / 10, sorted.end());
            const double scene_r2 = std::max(sorted[(n - 1) * 9 / 10], 1e-12);
            double passing = 0.0;                   // expected number of groups that pass in some lane
            for(uint32_t i = 0; i < n; i += 4) {
                double q = 0.0;
                for(uint32_t j = i; j < std::min(i + 4u, n); j++) q += section[j] / scene_r2;
                passing += std::min(1.0, 64.0 * q);
            }
            b->prefilter_pays = passing <= 0.1 * ((n + 3u) / 4u);
        }
        b->all_terminal = true;
        for(uint32_t i = 0; i < n; i++) b->all_terminal = b->all_terminal && kind[i] == AEHIP_MAT_NORMAL;
        {
            // sphere 0 as seen from the camera, in the reference's operation order
            // (src/shapes.cpp:15,18): oc = C - O, c = oc.oc - r*r
            const float ocx = geom[0].x - vc.cam[0], ocy = geom[0].y - vc.cam[1], ocz = geom[0].z - vc.cam[2];
            b->cc.oc[0] = ocx; b->cc.oc[1] = ocy; b->cc.oc[2] = ocz;
            const float occ = (ocx * ocx + ocy * ocy) + ocz * ocz;
            b->cc.c = occ - geom[0].w;
            b->cc.center[0] = geom[0].x; b->cc.center[1] = geom[0].y; b->cc.center[2] = geom[0].z;
            b->cc.magnitude = occ + geom[0].w;
            const double kx = ocx, ky = ocy, c = b->cc.c;
            const double dz = 0.0f - vc.cam[2], hz = dz * static_cast<double>(ocz);
            b->cc.q_uu = static_cast<float>(kx * kx - c);
            b->cc.q_uv = static_cast<float>(2.0 * kx * ky);
            b->cc.q_vv = static_cast<float>(ky * ky - c);
            b->cc.q_u = static_cast<float>(2.0 * kx * hz);
            b->cc.q_v = static_cast<float>(2.0 * ky * hz);
            b->cc.q_0 = static_cast<float>(hz * hz - c * dz * dz);
            const double r0 = std::sqrt(static_cast<double>(geom[0].w));
            b->cc.shade_scale = static_cast<float>(0.5 / r0);               // inf for a point sphere: see launch()
            b->cc.shade_offset[0] = static_cast<float>(0.5 - 0.5 * kx / r0);
            b->cc.shade_offset[1] = static_cast<float>(0.5 - 0.5 * ky / r0);
            b->cc.shade_offset[2] = static_cast<float>(0.5 - 0.5 * static_cast<double>(ocz) / r0);
            b->cc.near_root_only = std::sqrt(static_cast<double>(occ)) - r0 > 1.001 * static_cast<double>(kTMin) ? 1u : 0u;
            // the near root is beyond t_min with a factor to spare whatever the roundings, and the operands
            // of the hit's square root and division are inside the ranges of rt_exact_math.h: a
            // well-conditioned hit has discriminant >= 1e-3 oc.oc (classify_pixel's bound with
            // big >= dd (oc.oc + r*r) >= 2 dd |oc| r), and -b - sqrt(disc) lies between twice the near root and 4 |oc|
            const double distance = std::sqrt(static_cast<double>(occ));
            b->cc.short_exact_ok = (distance - r0 > 4.0 * static_cast<double>(kTMin) && distance >= 0x1p-30 && distance <= 0x1p+36) ? 1u : 0u;
            b->cc.radius2 = geom[0].w;
        }
        // the table of camera entries goes up in the packed layout of load_camera_group: whole
        // groups of four (the padding can never pass: a screening value of about -1e30), the
        // fields of two spheres side by side
        for(uint32_t i = n; i < n_padded; i++) camera[i] = make_float4(0.0f, 0.0f, 0.0f, -1e30f);
        for(uint32_t i = 0; i < n_padded; i += 4u) {
            const float4 e0 = camera[i], e1 = camera[i + 1u], e2 = camera[i + 2u], e3 = camera[i + 3u];
            camera[i] = make_float4(e0.x, e1.x, e0.y, e1.y);
            camera[i + 1u] = make_float4(e0.z, e1.z, e0.w, e1.w);
            camera[i + 2u] = make_float4(e2.x, e3.x, e2.y, e3.y);
            camera[i + 3u] = make_float4(e2.z, e3.z, e2.w, e3.w);
        }
        hipError_t e = hipMemcpyAsync(b->d_geom, geom, sizeof(float4) * n, hipMemcpyHostToDevice, b->stream);
        if(e == hipSuccess) e = hipMemcpyAsync(b->d_matl, matl, sizeof(float4) * n, hipMemcpyHostToDevice, b->stream);
        if(e == hipSuccess) e = hipMemcpyAsync(b->d_kind, kind, sizeof(uint32_t) * n, hipMemcpyHostToDevice, b->stream);
        if(e == hipSuccess) e = hipMemcpyAsync(b->d_camera, camera, sizeof(float4) * n_padded, hipMemcpyHostToDevice, b->stream);
        if(e == hipSuccess) e = hipMemcpyAsync(b->d_prefilter, prefilter, sizeof(float4) * n, hipMemcpyHostToDevice, b->stream);
        if(e == hipSuccess) e = hipStreamSynchronize(b->stream);
        std::free(geom); std::free(matl); std::free(kind); std::free(camera); std::free(prefilter);
        if(e != hipSuccess) return fail(AEHIP_E_HIP, "scene upload: %s", hipGetErrorString(e));
    }

    // scratch of the streamed pass (see launch()): the radiance block and block totals of a window
    // of strips, the whole band if it fits scratch_budget()
    {
        b->params = p;                  // takes_streamed_pass() reads the request
        const bool wanted = takes_streamed_pass(b);
        b->wants_streamed_pass = wanted;
        b->stream_func = nullptr;
        b->stream_groups_per_cu = 0;
        if(wanted) {
            // the tracing kernel and its residency, here rather than per launch: a launch may be under stream capture
            b->stream_func = streamed_kernel(b);
            AEHIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&b->stream_groups_per_cu, b->stream_func, kBlock, 0));
            const size_t strips = static_cast<size_t>((p.width + kTileW - 1) / kTileW) * ((p.row_count + 3u) / 4u);
            const size_t per_strip = scratch_bytes_per_strip(p.spp);
            size_t window = std::min(strips, std::max<size_t>(1, scratch_budget() / per_strip));
            if(window * per_strip > b->radiance_capacity) {
                if(b->d_radiance) (void)hipFree(b->d_radiance);
                b->d_radiance = nullptr; b->radiance_capacity = 0;
                // A GPU shared with other work may not have the gigabytes a whole 4K frame's window takes: halve the
                // window until the allocation fits (down to what 64 MiB hold) -- more windows, the same image
                hipError_t e = hipErrorOutOfMemory;
                for(;;) {
                    e = hipMalloc(reinterpret_cast<void **>(&b->d_radiance), window * per_strip);
                    if(e != hipErrorOutOfMemory || window * per_strip <= (static_cast<size_t>(64) << 20) || window == 1) break;
                    (void)hipGetLastError();
                    b->d_radiance = nullptr;
                    window = (window + 1) / 2;
                }
                if(e != hipSuccess) {
                    b->d_radiance = nullptr;
                    return fail(AEHIP_E_HIP, "scratch of the streamed pass (%zu bytes): %s", window * per_strip, hipGetErrorString(e));
                }
                b->radiance_capacity = window * per_strip;
            }
            b->window_strips = static_cast<uint32_t>(window);
        } else {
            b->window_strips = 0;
        }
    }

    // fast build, single-sphere camera pass: the pixels the exact kernel renders (ring_pixels_kernel)
    b->ring_count = 0;
    b->ring_row_offset.clear();
    b->tile_order_count = 0;
    if((p.flags & AEHIP_FLAG_FAST) && p.pass == AEHIP_PASS_PATH && single_sphere_camera_pass(b)) {
        const int rc = build_ring_list(b);
        if(rc != AEHIP_OK) return rc;
    }

    // single-sphere camera pass: the pixels that go through a sample loop (aehip_stats.samples_walked).  The
    // pass classifies pixels in the fast build always and in the exact build where it screens with the folded
    // ray (launch() hands it the table of all-miss sums only then), unless every sample is to be walked.
    b->walked_pixels = static_cast<uint64_t>(p.width) * p.row_count;
    b->culls_pixels = p.pass == AEHIP_PASS_PATH && single_sphere_camera_pass(b) && !(p.flags & AEHIP_FLAG_WALK_ALL) &&
                      ((p.flags & AEHIP_FLAG_FAST) || (vc.flags & kFlagFoldScreen));
    if(b->culls_pixels) {
        uint32_t *d_count = reinterpret_cast<uint32_t *>(b->d_rays + 2);          // the streamed pass's queue word: idle during set-up
        uint32_t count = 0;
        AEHIP_CHECK(hipMemsetAsync(d_count, 0, sizeof(uint32_t), b->stream));
        const uint64_t pixels = static_cast<uint64_t>(p.width) * p.row_count;
        hipLaunchKernelGGL(walked_pixels_kernel, dim3(static_cast<uint32_t>(std::min<uint64_t>((pixels + 255u) / 256u, 2048u))), dim3(256), 0, b->stream, vc, b->cc,
                           (p.flags & AEHIP_FLAG_FAST) ? 1u : 0u, d_count);
        AEHIP_CHECK(hipGetLastError());
        AEHIP_CHECK(hipMemcpyAsync(&count, d_count, sizeof(uint32_t), hipMemcpyDeviceToHost, b->stream));
        AEHIP_CHECK(hipStreamSynchronize(b->stream));
        b->walked_pixels = count;
    }

    // exact build, single-sphere camera pass: the value of the pixels whose samples all miss
    if(!(p.flags & AEHIP_FLAG_FAST) && p.pass == AEHIP_PASS_PATH && single_sphere_camera_pass(b)) {
        if(p.height > b->row_miss_capacity) {
            if(b->d_row_miss) (void)hipFree(b->d_row_miss);
            b->d_row_miss = nullptr; b->row_miss_capacity = 0;
            AEHIP_CHECK(hipMalloc(reinterpret_cast<void **>(&b->d_row_miss), sizeof(float4) * p.height));
            b->row_miss_capacity = p.height;
        }
        hipLaunchKernelGGL(row_miss_sums_kernel, dim3((p.height + 63u) / 64u), dim3(64), 0, b->stream, vc, p.height, b->d_row_miss);
        AEHIP_CHECK(hipGetLastError());
    }

    // single-sphere camera pass: the order in which a launch over the whole band takes its workgroups
    {
        const int rc = build_tile_order(b);
        if(rc != AEHIP_OK) return rc;
    }

    // per-sample key terms
    {
        const uint32_t count = p.spp + 1;
        if(count > b->sample_terms_capacity) {
            if(b->d_sample_terms) (void)hipFree(b->d_sample_terms);
            b->d_sample_terms = nullptr; b->sample_terms_capacity = 0;
            AEHIP_CHECK(hipMalloc(reinterpret_cast<void **>(&b->d_sample_terms), sizeof(uint32_t) * count));
            b->sample_terms_capacity = count;
        }
        uint32_t *terms = static_cast<uint32_t *>(std::malloc(sizeof(uint32_t) * count));
        if(!terms) return fail(AEHIP_E_INVALID, "out of host memory for %u sample terms", count);
        for(uint32_t i = 0; i < count; i++) terms[i] = sample_term(vc.seed_hash, i);
        hipError_t e = hipMemcpyAsync(b->d_sample_terms, terms, sizeof(uint32_t) * count, hipMemcpyHostToDevice, b->stream);
        if(e == hipSuccess) e = hipStreamSynchronize(b->stream);
        std::free(terms);
        if(e != hipSuccess) return fail(AEHIP_E_HIP, "sample term upload: %s", hipGetErrorString(e));
    }

    b->params = p;
    b->last_trace_render_ms = 0.0f;
    b->ready = true;
    return AEHIP_OK;
}

int aehip_backend_trace_device(aehip_backend *b, uint32_t *argb_dev, float *rgba_dev, void *hip_stream) {
    if(!b) return fail(AEHIP_E_INVALID, "aehip_backend_trace_device: backend is null");
    if(!b->ready) return fail(AEHIP_E_NOT_SETUP, "trace before a successful setup");
    if(!argb_dev) return fail(AEHIP_E_INVALID, "argb_dev is null");
    AEHIP_CHECK(hipSetDevice(b->device));
    const bool own = hip_stream == nullptr;
    hipStream_t stream = own ? b->stream : static_cast<hipStream_t>(hip_stream);
    const bool timed = (b->vc.flags & AEHIP_FLAG_NO_TIMING) == 0u;
    if(timed && stream_is_capturing(stream)) return fail(AEHIP_E_INVALID, "the stream is being captured: timing events cannot go into a graph, set AEHIP_FLAG_NO_TIMING");
    if(timed) AEHIP_CHECK(hipEventRecord(b->ev_begin, stream));
    const int rc = launch(b, argb_dev, rgba_dev, stream, own_scratch(b));
    if(rc != AEHIP_OK) return rc;
    if(timed) AEHIP_CHECK(hipEventRecord(b->ev_end, stream));
    b->timed_on_own_stream = own;
    if(own) AEHIP_CHECK(hipStreamSynchronize(stream));
    return AEHIP_OK;
}

int aehip_backend_trace_device_static(aehip_backend *b, uint32_t *argb_dev, float *rgba_dev, void *hip_stream) {
    if(!b) return fail(AEHIP_E_INVALID, "aehip_backend_trace_device_static: backend is null");
    if(!b->ready) return fail(AEHIP_E_NOT_SETUP, "trace before a successful setup");
    if(!argb_dev) return fail(AEHIP_E_INVALID, "argb_dev is null");
    AEHIP_CHECK(hipSetDevice(b->device));
    unsigned char *base = (b->device >= 0 && b->device < kMaxDevices) ? g_static_scratch_ptr[b->device] : nullptr;
    if(!base) return fail(AEHIP_E_HIP, "the static scratch has no address on device %d", b->device);
    // layout: counters | radiance of a window of strips | the window's block totals
    stream_scratch sc;
    sc.counters = reinterpret_cast<unsigned long long *>(base);
    sc.stride = scratch_stride(b->vc.spp);
    if(takes_streamed_pass(b)) {
        const size_t strips = static_cast<size_t>((b->vc.width + kTileW - 1) / kTileW) * ((b->vc.row_count + 3u) / 4u);
        const size_t window = (kStaticScratchBytes - kScratchHeader) / scratch_bytes_per_strip(b->vc.spp);
        if(window == 0 || strips > window * kMaxStaticWindows)
            return fail(AEHIP_E_TOO_LARGE, "a band of %zu strips of 16x4 pixels at %u samples per pixel needs more than %u passes through the %zu MiB static scratch",
                        strips, b->vc.spp, kMaxStaticWindows, kStaticScratchBytes >> 20);
        sc.window_strips = static_cast<uint32_t>(std::min(window, strips));
        sc.radiance = reinterpret_cast<radiance3 *>(base + kScratchHeader);
        sc.totals = sc.radiance + static_cast<size_t>(sc.window_strips) * sc.stride * 64u;
    }
    const bool own = hip_stream == nullptr;
    hipStream_t stream = own ? b->stream : static_cast<hipStream_t>(hip_stream);
    const bool timed = (b->vc.flags & AEHIP_FLAG_NO_TIMING) == 0u;
    if(timed && stream_is_capturing(stream)) return fail(AEHIP_E_INVALID, "the stream is being captured: timing events cannot go into a graph, set AEHIP_FLAG_NO_TIMING");
    if(timed) AEHIP_CHECK(hipEventRecord(b->ev_begin, stream));
    const int rc = launch(b, argb_dev, rgba_dev, stream, sc);
    if(rc != AEHIP_OK) return rc;
    if(timed) AEHIP_CHECK(hipEventRecord(b->ev_end, stream));
    b->timed_on_own_stream = own;
    if(own) AEHIP_CHECK(hipStreamSynchronize(stream));
    return AEHIP_OK;
}

int aehip_backend_trace(aehip_backend *b) {
    if(!b) return fail(AEHIP_E_INVALID, "aehip_backend_trace: backend is null");
    if(!b->ready) return fail(AEHIP_E_NOT_SETUP, "trace before a successful setup");
    if(!b->framebuffer) return fail(AEHIP_E_INVALID, "backend was created without a framebuffer");
    AEHIP_CHECK(hipSetDevice(b->device));

    const size_t pixels = static_cast<size_t>(b->vc.width) * b->vc.row_count;
    if(pixels > b->argb_capacity) {
        if(b->d_argb) (void)hipFree(b->d_argb);
        b->d_argb = nullptr; b->argb_capacity = 0;
        AEHIP_CHECK(hipMalloc(reinterpret_cast<void **>(&b->d_argb), pixels * sizeof(uint32_t)));
        b->argb_capacity = pixels;
    }
    if(pixels > b->staging_capacity) {
        if(b->h_staging) (void)hipHostFree(b->h_staging);
        b->h_staging = nullptr; b->staging_capacity = 0;
        AEHIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&b->h_staging), pixels * sizeof(uint32_t), hipHostMallocDefault));
        b->staging_capacity = pixels;
    }

    // The borrowed framebuffer is usually a MAP_SHARED file mapping (src/output_linux.cpp:24-29):
    // the image goes device -> pinned staging at full PCIe rate, then by memcpy into the mapping.
    // A large contiguous band is rendered in kTraceParts launches: while one part renders, the
    // part before it travels to the host, and the one before that is copied into the caller's
    // buffer.  (Not with the ray counters, which a launch resets, nor with dealt row tiles, nor
    // with the streamed pass, which already is several launches a frame and gains nothing.)
    // Measured for a 3840x2160 frame of 1024 samples: 2.5 ms a call instead of 4.0 (fast build).
    const uint32_t width = b->vc.width, rows = b->vc.row_count;
    const bool in_parts = b->vc.tile_rows == 0 && !(b->vc.flags & AEHIP_FLAG_COUNT_RAYS) && rows >= 64u && pixels >= (1u << 20) &&
                          !takes_streamed_pass(b);
    // In a render much longer than the copies (known from this request's previous call) a part's
    // copies fit beside a following part a third its size, and every extra launch costs a round
    // of waves: three parts, each about a third of the one before.
    const double copies_ms = static_cast<double>(pixels) * sizeof(uint32_t) / 1.5e7;       // ~15 GB/s, both copies in a row
    const bool long_render = b->last_trace_render_ms > 2.0 * copies_ms;
    const int parts = !in_parts ? 1 : long_render ? 3 : kTraceParts;
    const view_consts whole = b->vc;
    // part k is band rows [part_begin(k), part_begin(k + 1)), cut on multiples of 16 rows (the tallest workgroup tile)
    auto part_begin = [&](int k) -> uint32_t {
        if(k <= 0) return 0u;
        if(k >= parts) return rows;
        const uint32_t end = long_render ? (k == 1 ? 22u : 29u) : kTracePartEnd[k - 1];
        return static_cast<uint32_t>(static_cast<uint64_t>(rows) * end / 32u) / 16u * 16u;
    };
    uint64_t paths = 0;
    bool used[kTraceParts] = {};
    AEHIP_CHECK(hipEventRecord(b->ev_begin, b->stream));
    for(int k = 0; k < parts; k++) {
        const uint32_t r0 = part_begin(k), r1 = part_begin(k + 1);
        if(r1 == r0) continue;
        const size_t offset = static_cast<size_t>(r0) * width, count = static_cast<size_t>(r1 - r0) * width;
        b->vc.row_begin = whole.row_begin + r0;
        b->vc.row_count = r1 - r0;
        const int rc = launch(b, b->d_argb + offset, nullptr, b->stream, own_scratch(b));
        b->vc = whole;
        if(rc != AEHIP_OK) {
            (void)hipStreamSynchronize(b->copy_stream);     // earlier parts' copies still write the staging buffer
            return rc;
        }
        paths += b->stats.paths;
        used[k] = true;
        AEHIP_CHECK(hipEventRecord(b->ev_part_done[k], b->stream));
        AEHIP_CHECK(hipStreamWaitEvent(b->copy_stream, b->ev_part_done[k], 0));
        AEHIP_CHECK(hipMemcpyAsync(b->h_staging + offset, b->d_argb + offset, count * sizeof(uint32_t), hipMemcpyDeviceToHost, b->copy_stream));
        AEHIP_CHECK(hipEventRecord(b->ev_part_copied[k], b->copy_stream));
    }
    AEHIP_CHECK(hipEventRecord(b->ev_end, b->stream));
    AEHIP_CHECK(hipEventRecord(b->ev_copy, b->copy_stream));
    b->stats.paths = paths;
    for(int k = 0; k < parts; k++) {
        if(!used[k]) continue;
        const uint32_t r0 = part_begin(k), r1 = part_begin(k + 1);
        const size_t offset = static_cast<size_t>(r0) * width, count = static_cast<size_t>(r1 - r0) * width;
        AEHIP_CHECK(hipEventSynchronize(b->ev_part_copied[k]));
        std::memcpy(b->framebuffer + offset, b->h_staging + offset, count * sizeof(uint32_t));
    }
    AEHIP_CHECK(hipStreamSynchronize(b->stream));
    b->timed_on_own_stream = true;
    float ms = 0.0f;
    if(hipEventElapsedTime(&ms, b->ev_end, b->ev_copy) == hipSuccess) b->stats.copy_ms = ms;
    if(hipEventElapsedTime(&ms, b->ev_begin, b->ev_end) == hipSuccess) b->last_trace_render_ms = ms;
    return AEHIP_OK;
}

int aehip_backend_background_columns(aehip_backend *b, uint32_t *col_begin, uint32_t *col_end) {
    if(!b || !col_begin || !col_end) return fail(AEHIP_E_INVALID, "aehip_backend_background_columns: null argument");
    if(!b->ready) return fail(AEHIP_E_NOT_SETUP, "aehip_backend_background_columns before a successful setup");
    // the whole frame's rows, whatever the band: every band of one frame gets the same answer
    view_consts vc = b->vc;
    vc.row_begin = 0;
    vc.row_count = vc.height;
    vc.tile_rows = 0;
    vc.tile_stride = 0;
    *col_begin = 0;
    *col_end = vc.width;
    // one sphere: the box of 16-pixel columns outside of which no sample position of any pixel comes within the
    // screening margin of the sphere (the box the kernels themselves leave to their background writers)
    const uint32_t groups_x = (vc.width + kTileW - 1) / kTileW, groups_y = (vc.row_count + 3u) / 4u;
    uint32_t rows = 0, cols = 0;
    if(vc.n_spheres == 1u && groups_x <= 0xffffu && groups_y <= 0xffffu && groups_y >= 1u && sphere_box(b, vc, 4u, groups_x, groups_y, &rows, &cols)) {
        *col_begin = (cols & 0xffffu) * static_cast<uint32_t>(kTileW);
        *col_end = std::min(vc.width, ((cols & 0xffffu) + (cols >> 16)) * static_cast<uint32_t>(kTileW));
    }
    return AEHIP_OK;
}

int aehip_backend_stats(aehip_backend *b, aehip_stats *out) {
    if(!b || !out) return fail(AEHIP_E_INVALID, "aehip_backend_stats: null argument");
    AEHIP_CHECK(hipSetDevice(b->device));
    float ms = 0.0f;
    if(b->vc.flags & AEHIP_FLAG_NO_TIMING) {
        // no events to wait on: the device's work is the only ordering there is (the counters below are read after it)
        AEHIP_CHECK(hipDeviceSynchronize());
    } else {
        AEHIP_CHECK(hipEventSynchronize(b->ev_end));
        AEHIP_CHECK(hipEventElapsedTime(&ms, b->ev_begin, b->ev_end));
    }
    b->stats.kernel_ms = ms;
    if((b->vc.flags & AEHIP_FLAG_COUNT_RAYS) && b->counts_known) {
        b->stats.rays = b->known_rays;
        b->stats.lane_slots = b->known_lane_slots;
    } else if(b->vc.flags & AEHIP_FLAG_COUNT_RAYS) {
        unsigned long long counters[2] = {0, 0};
        AEHIP_CHECK(hipMemcpy(counters, b->last_counters ? b->last_counters : b->d_rays, sizeof(counters), hipMemcpyDeviceToHost));
        b->stats.rays = counters[0];
        b->stats.lane_slots = counters[1];
    } else {
        b->stats.rays = 0;
        b->stats.lane_slots = 0;
    }
    // samples that went through a sample loop: set-up's count for the pass that culls pixels (whole band), else every path
    b->stats.samples_walked = b->culls_pixels ? b->walked_pixels * b->vc.spp : b->stats.paths;
    *out = b->stats;
    return AEHIP_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// one frame on several devices from one host thread
// ------------------------------------------------------------------------------------------

namespace {

struct rccl_api {
    void *lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

struct frame_state {
    std::vector<aehip_backend *> bands;      // one backend per logical device
    std::vector<ncclComm_t> comms;           // valid when the devices are distinct GPUs
    rccl_api rccl;
    // gather of kept columns (device 0): the bands' packed rows stacked in band order, the frame in image
    // order, and for every frame row the stacked row it comes from
    uint32_t *d_stack = nullptr, *d_image = nullptr;
    long long *d_row_source = nullptr;
    size_t stack_capacity = 0, image_capacity = 0, row_source_capacity = 0;    // words, pixels, rows
    uint64_t row_source_key = 0;             // (rows, tile height, bands) the row map was built for
    std::vector<uint32_t *> d_packed;        // per band: its packed rows on its own device (peers only)
    std::vector<size_t> packed_capacity;     // words
    hipEvent_t ev_received = nullptr;        // device 0: the peers' rows have arrived
    int device0 = 0;
};
frame_state g_frame;

void release_frame_state() {
    for(ncclComm_t c : g_frame.comms)
        if(c && g_frame.rccl.CommDestroy) (void)g_frame.rccl.CommDestroy(c);
    g_frame.comms.clear();
    for(size_t i = 0; i < g_frame.d_packed.size(); i++) {
        if(g_frame.d_packed[i] && i < g_frame.bands.size() && hipSetDevice(g_frame.bands[i]->device) == hipSuccess) (void)hipFree(g_frame.d_packed[i]);
    }
    g_frame.d_packed.clear();
    g_frame.packed_capacity.clear();
    if(!g_frame.bands.empty() && hipSetDevice(g_frame.device0) == hipSuccess) {
        if(g_frame.d_stack) (void)hipFree(g_frame.d_stack);
        if(g_frame.d_image) (void)hipFree(g_frame.d_image);
        if(g_frame.d_row_source) (void)hipFree(g_frame.d_row_source);
        if(g_frame.ev_received) (void)hipEventDestroy(g_frame.ev_received);
    }
    g_frame.d_stack = g_frame.d_image = nullptr;
    g_frame.d_row_source = nullptr;
    g_frame.stack_capacity = g_frame.image_capacity = g_frame.row_source_capacity = 0;
    g_frame.row_source_key = 0;
    g_frame.ev_received = nullptr;
    for(aehip_backend *b : g_frame.bands) aehip_backend_destroy(b);
    g_frame.bands.clear();
    if(g_frame.rccl.lib) dlclose(g_frame.rccl.lib);
    g_frame.rccl = rccl_api{};
}

int load_rccl(rccl_api &api) {
    if(api.lib) return AEHIP_OK;
    void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if(!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if(!lib) return fail(AEHIP_E_HIP, "cannot load librccl: %s", dlerror());
    api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(dlsym(lib, "ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
    api.Send = reinterpret_cast<decltype(api.Send)>(dlsym(lib, "ncclSend"));
    api.Recv = reinterpret_cast<decltype(api.Recv)>(dlsym(lib, "ncclRecv"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    if(!api.CommInitAll || !api.CommDestroy || !api.GroupStart || !api.GroupEnd || !api.Send || !api.Recv || !api.GetErrorString) {
        dlclose(lib);
        return fail(AEHIP_E_HIP, "librccl lacks a required entry point");
    }
    api.lib = lib;
    return AEHIP_OK;
}

#define AEHIP_RCCL_CHECK(api, expr)                                                                          \
    do {                                                                                                     \
        ncclResult_t r_ = (expr);                                                                            \
        if(r_ != ncclSuccess) return fail(AEHIP_E_HIP, "%s failed: %s", #expr, (api).GetErrorString(r_));    \
    } while(0)

}  // namespace

// Band i of n when the frame's rows are cut into tiles of tile_rows rows dealt round-robin: which
// rows it has and where its rows start among the densely stored bands (band order).
#ifdef AEHIP_WAVE_TIMELINE
// out: 3 words a wave (start, end in 10 ns ticks; low byte 0 = rendered a tile, 1 = listed pixels, 2 = filled background, above it the shader clocks between start and end), at most `capacity` waves; returns the count
extern "C" int aehip_wave_timeline(unsigned long long *out, unsigned int capacity) {
    unsigned int n = 0;
    if(hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_timeline_waves), sizeof(n)) != hipSuccess) return -1;
    n = std::min(std::min(n, capacity), kTimelineWaves);
    if(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_timeline), sizeof(unsigned long long) * 3u * n) != hipSuccess) return -1;
    return static_cast<int>(n);
}
#endif

// ------------------------------------------------------------------------------------------
// bands between GPUs: the kept columns of a band behind one background pixel a row, and back
// ------------------------------------------------------------------------------------------

namespace {

// out row r = { band[r][bg_col], band[r][col_begin .. col_end) }: one thread a word
__global__ __launch_bounds__(256) void pack_columns_kernel(const uint32_t *__restrict__ band, uint32_t rows, uint32_t width,
                                                           uint32_t col_begin, uint32_t kept, uint32_t bg_col, uint32_t *__restrict__ out) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x, r = blockIdx.y;
    if(k >= kept || r >= rows) return;
    const uint32_t x = k == 0u ? bg_col : col_begin + k - 1u;
    out[static_cast<size_t>(r) * kept + k] = band[static_cast<size_t>(r) * width + x];
}

// image row y from packed row row_source[y]: the row's own words inside [col_begin, col_end), its
// background word outside; a thread writes four pixels (one 16-byte store where the width allows)
template <bool VEC4>
__global__ __launch_bounds__(256) void expand_columns_kernel(const uint32_t *__restrict__ packed, unsigned long long packed_rows,
                                                             const long long *__restrict__ row_source, uint32_t height, uint32_t width,
                                                             uint32_t col_begin, uint32_t col_end, uint32_t kept, uint32_t *__restrict__ image) {
    const uint32_t x0 = (blockIdx.x * 256u + threadIdx.x) * 4u, y = blockIdx.y;
    if(x0 >= width || y >= height) return;
    const long long source = row_source[y];
    if(source < 0 || static_cast<unsigned long long>(source) >= packed_rows) return;       // a row no band holds: left as it is
    const uint32_t *src = packed + static_cast<size_t>(source) * kept;
    uint32_t w[4];
#pragma unroll
    for(uint32_t i = 0; i < 4u; i++) {
        const uint32_t x = x0 + i;
        w[i] = (x >= col_begin && x < col_end) ? src[1u + x - col_begin] : src[0];
    }
    uint32_t *dst = image + static_cast<size_t>(y) * width + x0;
    if constexpr(VEC4) {
        *reinterpret_cast<uint4 *>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
#pragma unroll
        for(uint32_t i = 0; i < 4u; i++)
            if(x0 + i < width) dst[i] = w[i];
    }
}

}  // namespace

extern "C" int aehip_band_pack_columns(const uint32_t *band_dev, uint32_t rows, uint32_t width, uint32_t col_begin, uint32_t col_end,
                                       uint32_t bg_col, uint32_t *packed_dev, void *hip_stream) {
    if(!band_dev || !packed_dev) return fail(AEHIP_E_INVALID, "aehip_band_pack_columns: null buffer");
    if(rows == 0 || rows > 65535u || width == 0 || !(col_begin < col_end) || col_end > width || bg_col >= width)
        return fail(AEHIP_E_INVALID, "aehip_band_pack_columns: %u rows of %u, columns %u..%u, background column %u", rows, width, col_begin, col_end, bg_col);
    if(bg_col >= col_begin && bg_col < col_end)
        return fail(AEHIP_E_INVALID, "aehip_band_pack_columns: background column %u lies inside the kept columns %u..%u", bg_col, col_begin, col_end);
    const uint32_t kept = col_end - col_begin + 1u;
    hipLaunchKernelGGL(pack_columns_kernel, dim3((kept + 255u) / 256u, rows), dim3(256), 0, static_cast<hipStream_t>(hip_stream), band_dev, rows,
                       width, col_begin, kept, bg_col, packed_dev);
    AEHIP_CHECK(hipGetLastError());
    return AEHIP_OK;
}

extern "C" int aehip_band_expand_columns(const uint32_t *packed_dev, uint64_t packed_rows, const long long *row_source_dev, uint32_t height,
                                         uint32_t width, uint32_t col_begin, uint32_t col_end, uint32_t *image_dev, void *hip_stream) {
    if(!packed_dev || !row_source_dev || !image_dev) return fail(AEHIP_E_INVALID, "aehip_band_expand_columns: null buffer");
    if(height == 0 || height > 65535u || width == 0 || !(col_begin < col_end) || col_end > width || packed_rows == 0)
        return fail(AEHIP_E_INVALID, "aehip_band_expand_columns: %u rows of %u, columns %u..%u, %llu packed rows", height, width, col_begin, col_end,
                    static_cast<unsigned long long>(packed_rows));
    const uint32_t kept = col_end - col_begin + 1u;
    const dim3 grid(((width + 3u) / 4u + 255u) / 256u, height);
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    if(width % 4u == 0 && reinterpret_cast<uintptr_t>(image_dev) % 16u == 0)
        hipLaunchKernelGGL(expand_columns_kernel<true>, grid, dim3(256), 0, stream, packed_dev, static_cast<unsigned long long>(packed_rows), row_source_dev, height, width, col_begin, col_end, kept, image_dev);
    else
        hipLaunchKernelGGL(expand_columns_kernel<false>, grid, dim3(256), 0, stream, packed_dev, static_cast<unsigned long long>(packed_rows), row_source_dev, height, width, col_begin, col_end, kept, image_dev);
    AEHIP_CHECK(hipGetLastError());
    return AEHIP_OK;
}

extern "C" int aehip_frame_band(uint32_t row_count, uint32_t tile_rows, int n_bands, int band, uint32_t *first_row,
                                uint32_t *rows, uint32_t *rows_before) {
    if(n_bands < 1 || band < 0 || band >= n_bands || !first_row || !rows || !rows_before) return fail(AEHIP_E_INVALID, "aehip_frame_band: bad argument");
    const uint32_t T = tile_rows ? tile_rows : 4u;
    const uint32_t tiles = (row_count + T - 1) / T;
    if(static_cast<uint32_t>(n_bands) > tiles) return fail(AEHIP_E_INVALID, "%d devices for %u tiles of %u rows", n_bands, tiles, T);
    const uint32_t stride = static_cast<uint32_t>(n_bands) * T;
    uint32_t before = 0;
    for(int i = 0; i <= band; i++) {
        const uint32_t span = row_count - static_cast<uint32_t>(i) * T;
        const uint32_t whole = span / stride, rest = span % stride;
        const uint32_t mine = whole * T + (rest < T ? rest : T);
        if(i == band) {
            *first_row = static_cast<uint32_t>(i) * T;
            *rows = mine;
            *rows_before = before;
        }
        before += mine;
    }
    return AEHIP_OK;
}

extern "C" int aehip_trace_frame(int n_devices, const aehip_scene *scene, const aehip_params *params,
                                 uint32_t *framebuffer, aehip_stats *stats_out) {
    if(!scene || !params || !framebuffer) return fail(AEHIP_E_INVALID, "aehip_trace_frame: null argument");
    if(n_devices < 1 || n_devices > 64) return fail(AEHIP_E_INVALID, "n_devices %d out of range (1..64)", n_devices);
    if(!g_initialised) {
        const int rc = aehip_init();
        if(rc != AEHIP_OK) return rc;
    }
    const int n = n_devices;
    const int physical = g_device_count;
    const bool distinct = n <= physical;                 // every band has a GPU of its own

    aehip_params frame = *params;
    if(frame.height == 0 || frame.width == 0) return fail(AEHIP_E_INVALID, "resolution %ux%u is empty", frame.width, frame.height);
    if(frame.row_count == 0) {
        if(frame.row_begin >= frame.height) return fail(AEHIP_E_INVALID, "row_begin %u outside image of %u rows", frame.row_begin, frame.height);
        frame.row_count = frame.height - frame.row_begin;
    }
    if(frame.row_begin + static_cast<uint64_t>(frame.row_count) > frame.height) return fail(AEHIP_E_INVALID, "rows outside the image");
    // tiles of T rows dealt round-robin: band i = tiles i, i+n, i+2n, ... of the frame's rows, so
    // that every device gets the same mix of cheap and costly rows
    const uint32_t T = frame.tile_rows ? frame.tile_rows : 4u;
    const uint32_t tiles = (frame.row_count + T - 1) / T;
    if(static_cast<uint32_t>(n) > tiles) return fail(AEHIP_E_INVALID, "%d devices for %u tiles of %u rows", n, tiles, T);

    // backends, one per logical device, kept for later frames
    while(static_cast<int>(g_frame.bands.size()) < n) {
        aehip_backend *b = nullptr;
        const int rc = aehip_backend_create(static_cast<int>(g_frame.bands.size()) % physical, nullptr, &b);
        if(rc != AEHIP_OK) return rc;
        g_frame.bands.push_back(b);
    }
    if(distinct && n > 1 && static_cast<int>(g_frame.comms.size()) != n) {
        const int rc = load_rccl(g_frame.rccl);
        if(rc != AEHIP_OK) return rc;
        for(ncclComm_t c : g_frame.comms)
            if(c) (void)g_frame.rccl.CommDestroy(c);
        g_frame.comms.clear();
        // into a local vector first: a failed initialisation must not leave null communicators behind
        std::vector<ncclComm_t> fresh(static_cast<size_t>(n), nullptr);
        std::vector<int> devs(static_cast<size_t>(n));
        for(int i = 0; i < n; i++) devs[static_cast<size_t>(i)] = i;
        AEHIP_RCCL_CHECK(g_frame.rccl, g_frame.rccl.CommInitAll(fresh.data(), n, devs.data()));
        g_frame.comms.swap(fresh);
    }

    std::vector<uint32_t> band_rows(static_cast<size_t>(n)), offset(static_cast<size_t>(n) + 1, 0u);   // in rows
    const uint32_t stride = static_cast<uint32_t>(n) * T;
    for(int i = 0; i < n; i++) {
        uint32_t first_row = 0;
        const int rc = aehip_frame_band(frame.row_count, T, n, i, &first_row, &band_rows[static_cast<size_t>(i)], &offset[static_cast<size_t>(i)]);
        if(rc != AEHIP_OK) return rc;
        offset[static_cast<size_t>(i) + 1] = offset[static_cast<size_t>(i)] + band_rows[static_cast<size_t>(i)];
    }

    // Set-up (scene upload, tables, a synchronisation each) only when the scene or the request
    // changed since the band's last frame: a caller that renders frame after frame pays it once.
    uint64_t digest = 1469598103934665603ull;
    auto mix = [&digest](const void *data, size_t bytes) {
        const unsigned char *p = static_cast<const unsigned char *>(data);
        for(size_t k = 0; k < bytes; k++) digest = (digest ^ p[k]) * 1099511628211ull;
    };
    mix(&frame, sizeof(frame));
    mix(scene->background0, sizeof(scene->background0));
    mix(scene->background1, sizeof(scene->background1));
    mix(scene->camera_pos, sizeof(scene->camera_pos));
    mix(&scene->sphere_count, sizeof(scene->sphere_count));
    if(scene->spheres) mix(scene->spheres, sizeof(aehip_sphere) * scene->sphere_count);
    mix(&n, sizeof(n));
    for(int i = 0; i < n; i++) {
        aehip_backend *b = g_frame.bands[static_cast<size_t>(i)];
        if(b->ready && b->setup_digest == digest) continue;
        aehip_params p = frame;
        p.row_begin = frame.row_begin + static_cast<uint32_t>(i) * T;
        p.row_count = band_rows[static_cast<size_t>(i)];
        p.tile_rows = n > 1 ? T : 0u;
        p.tile_stride = n > 1 ? stride : 0u;
        b->setup_digest = 0;
        const int rc = aehip_backend_setup(b, scene, &p);
        if(rc != AEHIP_OK) return rc;
        b->setup_digest = digest;
    }

    // device 0 holds every band, densely, one after the other, and renders its own band straight
    // into place; peers render into band buffers of their own
    aehip_backend *b0 = g_frame.bands[0];
    g_frame.device0 = b0->device;
    const size_t frame_pixels = static_cast<size_t>(frame.width) * frame.row_count;
    for(int i = 0; i < n; i++) {
        aehip_backend *b = g_frame.bands[static_cast<size_t>(i)];
        const size_t need = i == 0 ? frame_pixels : static_cast<size_t>(frame.width) * b->vc.row_count;
        AEHIP_CHECK(hipSetDevice(b->device));
        if(need > b->argb_capacity) {
            if(b->d_argb) (void)hipFree(b->d_argb);
            b->d_argb = nullptr; b->argb_capacity = 0;
            AEHIP_CHECK(hipMalloc(reinterpret_cast<void **>(&b->d_argb), need * sizeof(uint32_t)));
            b->argb_capacity = need;
        }
    }
    AEHIP_CHECK(hipSetDevice(b0->device));
    if(frame_pixels > b0->staging_capacity) {
        if(b0->h_staging) (void)hipHostFree(b0->h_staging);
        b0->h_staging = nullptr; b0->staging_capacity = 0;
        AEHIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&b0->h_staging), frame_pixels * sizeof(uint32_t), hipHostMallocDefault));
        b0->staging_capacity = frame_pixels;
    }
    if(!g_frame.ev_received) AEHIP_CHECK(hipEventCreateWithFlags(&g_frame.ev_received, hipEventDisableTiming));

    // What a peer sends.  Outside the columns aehip_backend_background_columns reports, every pixel of a
    // band row holds the row's background: a peer packs those columns behind one background pixel a row
    // (pack_columns_kernel), and device 0 rebuilds the frame's rows, in image order, from the stacked packed
    // rows of all bands, its own included (expand_columns_kernel).  A peer's band crosses one xGMI link: on the
    // reference's scene the kept columns are 40 % of a row.  Scenes for which no such columns are known
    // (several shapes), or columns that save less than a quarter, go as whole rows, as before.
    uint32_t col_begin = 0, col_end = frame.width;
    if(n > 1) {
        const int rc = aehip_backend_background_columns(b0, &col_begin, &col_end);
        if(rc != AEHIP_OK) return rc;
    }
    const uint32_t kept = col_end - col_begin + 1u;
    const bool narrow = n > 1 && col_end > col_begin && static_cast<uint64_t>(kept) * 4u <= static_cast<uint64_t>(frame.width) * 3u;
    const uint32_t bg_col = col_begin > 0u ? 0u : frame.width - 1u;         // a column outside the kept ones
    if(narrow) {
        g_frame.d_packed.resize(static_cast<size_t>(n), nullptr);
        g_frame.packed_capacity.resize(static_cast<size_t>(n), 0);
        for(int i = 1; i < n; i++) {
            aehip_backend *b = g_frame.bands[static_cast<size_t>(i)];
            const size_t need = static_cast<size_t>(kept) * b->vc.row_count;
            if(need > g_frame.packed_capacity[static_cast<size_t>(i)]) {
                AEHIP_CHECK(hipSetDevice(b->device));
                if(g_frame.d_packed[static_cast<size_t>(i)]) (void)hipFree(g_frame.d_packed[static_cast<size_t>(i)]);
                g_frame.d_packed[static_cast<size_t>(i)] = nullptr; g_frame.packed_capacity[static_cast<size_t>(i)] = 0;
                AEHIP_CHECK(hipMalloc(reinterpret_cast<void **>(&g_frame.d_packed[static_cast<size_t>(i)]), need * sizeof(uint32_t)));
                g_frame.packed_capacity[static_cast<size_t>(i)] = need;
            }
        }
        AEHIP_CHECK(hipSetDevice(b0->device));
        const size_t stack_words = static_cast<size_t>(kept) * frame.row_count;
        if(stack_words > g_frame.stack_capacity) {
            if(g_frame.d_stack) (void)hipFree(g_frame.d_stack);
            g_frame.d_stack = nullptr; g_frame.stack_capacity = 0;
            AEHIP_CHECK(hipMalloc(reinterpret_cast<void **>(&g_frame.d_stack), stack_words * sizeof(uint32_t)));
            g_frame.stack_capacity = stack_words;
        }
        if(frame_pixels > g_frame.image_capacity) {
            if(g_frame.d_image) (void)hipFree(g_frame.d_image);
            g_frame.d_image = nullptr; g_frame.image_capacity = 0;
            AEHIP_CHECK(hipMalloc(reinterpret_cast<void **>(&g_frame.d_image), frame_pixels * sizeof(uint32_t)));
            g_frame.image_capacity = frame_pixels;
        }
        if(frame.row_count > g_frame.row_source_capacity) {
            if(g_frame.d_row_source) (void)hipFree(g_frame.d_row_source);
            g_frame.d_row_source = nullptr; g_frame.row_source_capacity = 0; g_frame.row_source_key = 0;
            AEHIP_CHECK(hipMalloc(reinterpret_cast<void **>(&g_frame.d_row_source), frame.row_count * sizeof(long long)));
            g_frame.row_source_capacity = frame.row_count;
        }
        // frame row y is row (y / T) / n * T + y % T of band (y / T) % n; the map changes with the rows, the tile height or the bands only
        const uint64_t key = (static_cast<uint64_t>(frame.row_count) << 32) ^ (static_cast<uint64_t>(T) << 8) ^ static_cast<uint64_t>(n);
        if(key != g_frame.row_source_key) {
            std::vector<long long> source(frame.row_count);
            for(uint32_t y = 0; y < frame.row_count; y++) {
                const uint32_t tile = y / T, band = tile % static_cast<uint32_t>(n);
                source[y] = static_cast<long long>(offset[band]) + (tile / static_cast<uint32_t>(n)) * T + (y - tile * T);
            }
            AEHIP_CHECK(hipMemcpy(g_frame.d_row_source, source.data(), source.size() * sizeof(long long), hipMemcpyHostToDevice));
            g_frame.row_source_key = key;
        }
    }

    // From the first launch on, an error must not return while other devices still work on buffers
    // a later call may free: wait for every band's stream, then report.
    auto settle = [&](int rc) {
        for(int i = 0; i < n; i++) {
            aehip_backend *b = g_frame.bands[static_cast<size_t>(i)];
            if(hipSetDevice(b->device) == hipSuccess) (void)hipStreamSynchronize(b->stream);
        }
        if(hipSetDevice(b0->device) == hipSuccess) (void)hipStreamSynchronize(b0->copy_stream);
        return rc;
    };
#define AEHIP_FRAME_CHECK(expr)                                                                                     \
    do {                                                                                                            \
        hipError_t e_ = (expr);                                                                                     \
        if(e_ != hipSuccess) return settle(fail(AEHIP_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)));       \
    } while(0)
#define AEHIP_FRAME_RCCL_CHECK(api, expr)                                                                           \
    do {                                                                                                            \
        ncclResult_t r_ = (expr);                                                                                   \
        if(r_ != ncclSuccess) return settle(fail(AEHIP_E_HIP, "%s failed: %s", #expr, (api).GetErrorString(r_)));   \
    } while(0)

    // render: every device at once, each on its own stream; a peer packs its kept columns behind its render
    for(int i = 0; i < n; i++) {
        aehip_backend *b = g_frame.bands[static_cast<size_t>(i)];
        AEHIP_FRAME_CHECK(hipSetDevice(b->device));
        AEHIP_FRAME_CHECK(hipEventRecord(b->ev_begin, b->stream));
        const int rc = launch(b, b->d_argb, nullptr, b->stream, own_scratch(b));
        if(rc != AEHIP_OK) return settle(rc);
        AEHIP_FRAME_CHECK(hipEventRecord(b->ev_end, b->stream));
        if(narrow && i > 0) {
            hipLaunchKernelGGL(pack_columns_kernel, dim3((kept + 255u) / 256u, b->vc.row_count), dim3(256), 0, b->stream, b->d_argb, b->vc.row_count,
                               frame.width, col_begin, kept, bg_col, g_frame.d_packed[static_cast<size_t>(i)]);
            AEHIP_FRAME_CHECK(hipGetLastError());
        }
    }

    // Gather the peers' rows on device 0.  Nothing here waits on the host, and nothing waits for device
    // 0's own render: the RCCL receives go to device 0's second stream and run beside that render (a
    // device-to-device copy, for bands that share device 0's GPU, ends in an event); device 0's render
    // stream waits for them only before it puts the frame together.
    const uint32_t row_words = narrow ? kept : frame.width;
    uint32_t *gathered = narrow ? g_frame.d_stack : b0->d_argb;
    if(n > 1 && distinct) {
        rccl_api &api = g_frame.rccl;
        AEHIP_FRAME_RCCL_CHECK(api, api.GroupStart());
        for(int i = 1; i < n; i++) {
            aehip_backend *b = g_frame.bands[static_cast<size_t>(i)];
            const size_t count = static_cast<size_t>(row_words) * b->vc.row_count;
            AEHIP_FRAME_CHECK(hipSetDevice(b->device));
            AEHIP_FRAME_RCCL_CHECK(api, api.Send(narrow ? g_frame.d_packed[static_cast<size_t>(i)] : b->d_argb, count, ncclUint32, 0,
                                                 g_frame.comms[static_cast<size_t>(i)], b->stream));
            AEHIP_FRAME_CHECK(hipSetDevice(b0->device));
            AEHIP_FRAME_RCCL_CHECK(api, api.Recv(gathered + static_cast<size_t>(row_words) * offset[static_cast<size_t>(i)], count, ncclUint32, i,
                                                 g_frame.comms[0], b0->copy_stream));
        }
        AEHIP_FRAME_RCCL_CHECK(api, api.GroupEnd());
        AEHIP_FRAME_CHECK(hipSetDevice(b0->device));
        AEHIP_FRAME_CHECK(hipEventRecord(g_frame.ev_received, b0->copy_stream));
        AEHIP_FRAME_CHECK(hipStreamWaitEvent(b0->stream, g_frame.ev_received, 0));
    } else if(n > 1) {
        for(int i = 1; i < n; i++) {
            aehip_backend *b = g_frame.bands[static_cast<size_t>(i)];
            const size_t bytes = static_cast<size_t>(row_words) * b->vc.row_count * sizeof(uint32_t);
            AEHIP_FRAME_CHECK(hipSetDevice(b->device));
            AEHIP_FRAME_CHECK(hipMemcpyPeerAsync(gathered + static_cast<size_t>(row_words) * offset[static_cast<size_t>(i)], b0->device,
                                                 narrow ? g_frame.d_packed[static_cast<size_t>(i)] : b->d_argb, b->device, bytes, b->stream));
            AEHIP_FRAME_CHECK(hipEventRecord(b->ev_copy, b->stream));
            AEHIP_FRAME_CHECK(hipSetDevice(b0->device));
            AEHIP_FRAME_CHECK(hipStreamWaitEvent(b0->stream, b->ev_copy, 0));
        }
    }

    AEHIP_FRAME_CHECK(hipSetDevice(b0->device));
    if(narrow) {
        // device 0's own band joins the stack (its rows come first), then the frame's rows in image order
        hipLaunchKernelGGL(pack_columns_kernel, dim3((kept + 255u) / 256u, b0->vc.row_count), dim3(256), 0, b0->stream, b0->d_argb, b0->vc.row_count,
                           frame.width, col_begin, kept, bg_col, g_frame.d_stack);
        const dim3 grid(((frame.width + 3u) / 4u + 255u) / 256u, frame.row_count);
        if(frame.width % 4u == 0)
            hipLaunchKernelGGL(expand_columns_kernel<true>, grid, dim3(256), 0, b0->stream, g_frame.d_stack, static_cast<unsigned long long>(frame.row_count),
                               g_frame.d_row_source, frame.row_count, frame.width, col_begin, col_end, kept, g_frame.d_image);
        else
            hipLaunchKernelGGL(expand_columns_kernel<false>, grid, dim3(256), 0, b0->stream, g_frame.d_stack, static_cast<unsigned long long>(frame.row_count),
                               g_frame.d_row_source, frame.row_count, frame.width, col_begin, col_end, kept, g_frame.d_image);
        AEHIP_FRAME_CHECK(hipGetLastError());
    }
    // frame -> pinned staging
    AEHIP_FRAME_CHECK(hipMemcpyAsync(b0->h_staging, narrow ? g_frame.d_image : b0->d_argb, frame_pixels * sizeof(uint32_t), hipMemcpyDeviceToHost, b0->stream));
    AEHIP_FRAME_CHECK(hipEventRecord(b0->ev_copy, b0->stream));
    AEHIP_FRAME_CHECK(hipStreamSynchronize(b0->stream));
#undef AEHIP_FRAME_CHECK
#undef AEHIP_FRAME_RCCL_CHECK
    if(narrow) {
        // the frame arrived in image order
        std::memcpy(framebuffer, b0->h_staging, frame_pixels * sizeof(uint32_t));
    } else {
        // pinned staging (band order) -> caller's buffer (image order), one copy per tile
        for(int i = 0; i < n; i++) {
            const uint32_t rows = band_rows[static_cast<size_t>(i)];
            for(uint32_t j = 0; j < rows; j += T) {
                const uint32_t tile = static_cast<uint32_t>(i) + (j / T) * static_cast<uint32_t>(n);
                const uint32_t count = rows - j < T ? rows - j : T;
                std::memcpy(framebuffer + static_cast<size_t>(frame.width) * tile * T,
                            b0->h_staging + static_cast<size_t>(frame.width) * (offset[static_cast<size_t>(i)] + j),
                            static_cast<size_t>(frame.width) * count * sizeof(uint32_t));
            }
        }
    }

    if(stats_out) {
        aehip_stats total{};
        for(int i = 0; i < n; i++) {
            aehip_stats st{};
            const int rc = aehip_backend_stats(g_frame.bands[static_cast<size_t>(i)], &st);
            if(rc != AEHIP_OK) return rc;
            total.kernel_ms = st.kernel_ms > total.kernel_ms ? st.kernel_ms : total.kernel_ms;
            total.rays += st.rays;
            total.paths += st.paths;
            total.lane_slots += st.lane_slots;
            total.samples_walked += st.samples_walked;
            total.vgprs = st.vgprs;
            total.lds_bytes = st.lds_bytes;
        }
        float ms = 0.0f;
        if(hipEventElapsedTime(&ms, b0->ev_end, b0->ev_copy) == hipSuccess) total.copy_ms = ms;   // gather + copy, as seen by device 0
        *stats_out = total;
    }
    return AEHIP_OK;
}
