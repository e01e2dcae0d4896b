/*
 * hip_raytracer.h -- C ABI of the MI355X (gfx950) rendering backend.
 *
 * This is the boundary a host program binds: plain C, POD arguments, no C++ or torch types.
 * The reference's backends are C++ classes behind `ae::raytracer`
 * (src/raytracer.h:11-33: ctor(u32 *framebuffer), `bool setup()`, `void trace()`), with the GPU
 * one adding static `init()` / `terminate()` (src/vulkan_raytracer.h:13-20).  Each entry point
 * below names the member it stands behind; `ae::hip_raytracer` (raytracer_amd/host/) is the thin
 * C++ class that forwards to them.
 *
 * Conventions: every function returning int returns AEHIP_OK (0) or a negative AEHIP_E_* code
 * and never throws; aehip_last_error() gives the message for the calling thread's last failure.
 * A backend object is used from one thread at a time (the reference calls setup()/trace() once
 * from its main thread, src/main.cpp:27-59).
 */
#ifndef AE_HIP_RAYTRACER_H
#define AE_HIP_RAYTRACER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AEHIP_OK              0
#define AEHIP_E_NO_DEVICE    -1   /* no usable HIP device / runtime failure at init        */
#define AEHIP_E_INVALID      -2   /* bad argument (null pointer, zero size, bad band, ...) */
#define AEHIP_E_HIP          -3   /* a HIP runtime call failed; see aehip_last_error()     */
#define AEHIP_E_NOT_SETUP    -4   /* trace before a successful setup                        */
#define AEHIP_E_TOO_LARGE    -5   /* size beyond what an entry point supports               */

/* Surface response of a sphere. */
enum {
    AEHIP_MAT_NORMAL     = 0,  /* the reference's shading: colour = remap(normal), path ends
                                  (src/software_raytracer.cpp:227-234)                      */
    AEHIP_MAT_LAMBERT    = 1,  /* diffuse, albedo                                            */
    AEHIP_MAT_METAL      = 2,  /* mirror + param = fuzz radius                               */
    AEHIP_MAT_DIELECTRIC = 3   /* param = index of refraction, Schlick reflectance           */
};

enum {
    AEHIP_FLAG_JITTER     = 1u,  /* sample position drawn inside the pixel; off = pixel centre
                                    (src/software_raytracer.cpp:218)                         */
    AEHIP_FLAG_FAST       = 2u,  /* fused multiply-add and hardware reciprocal / rsqrt; off =
                                    IEEE fp32 in the reference's operation order             */
    AEHIP_FLAG_COUNT_RAYS = 4u,  /* count ray segments and lane slots on the device
                                    (aehip_stats.rays, aehip_stats.lane_slots)               */
    AEHIP_FLAG_NO_REGEN   = 8u,  /* measurement aid: a lane whose path ended waits until every
                                    lane of its wave is done before starting its next sample
                                    (the image is unchanged; only lane utilisation differs)  */
    AEHIP_FLAG_WALK_ALL   = 16u, /* measurement aid: the single-sphere camera pass walks every
                                    sample of every pixel instead of deciding, from bounds over
                                    the pixel's sample positions, the pixels whose samples all
                                    miss (and, fast build, all hit).  Exact build: the image is
                                    unchanged bit for bit; fast build: within its tolerance   */
    AEHIP_FLAG_NO_TIMING  = 32u  /* aehip_backend_trace_device[_static] record no timing events
                                    around the launch (two event records cost the host 7 us and
                                    the queue two barriers: as much as a short frame's launch);
                                    aehip_stats.kernel_ms then reads 0.  For callers that time
                                    frames themselves.  The image is unchanged                */
};

enum {
    AEHIP_PASS_PRIMARY = 0,  /* the reference's pass: sphere 0 only, nearest root, no range
                                test, one ray at the pixel centre (src/shapes.cpp:14-30)     */
    AEHIP_PASS_PATH    = 1   /* shape list + closest hit, materials, spp, bounce loop        */
};

/* 48 bytes.  center/radius mirror ae::sphere (src/shapes.h:10-19); radius is used as |radius|
 * (src/shapes.cpp:10-12). */
typedef struct aehip_sphere {
    float    center[3];
    float    radius;
    float    albedo[3];
    float    param;
    uint32_t material;
    uint32_t reserved[3];
} aehip_sphere;

/* The four scene globals of ae::raytracer (src/raytracer.h:16-19, src/raytracer.cpp:10-13),
 * with the single sphere generalised to a list. */
typedef struct aehip_scene {
    float    background0[4];   /* r,g,b,a : colour at image row 0    */
    float    background1[4];   /* r,g,b,a : colour towards row H     */
    float    camera_pos[4];    /* x,y,z,0                            */
    uint32_t sphere_count;
    uint32_t reserved;
    const aehip_sphere *spheres;   /* host memory, copied by setup */
} aehip_scene;

#define AEHIP_SAMPLE_BLOCK 64u

typedef struct aehip_params {
    uint32_t width, height;         /* what raytracer::get_resolution() returns (src/raytracer.cpp:18-34) */
    uint32_t row_begin, row_count;  /* band of image rows this backend renders; row_count 0 = every row
                                       of the band below `height`.  Outputs hold the band's row_count
                                       rows densely, band row 0 = image row row_begin. */
    uint32_t spp;                   /* samples per pixel, >= 1.  A pixel's samples are summed in blocks
                                       of AEHIP_SAMPLE_BLOCK: sample order inside a block, block sums
                                       in block order (blocks are independent work for the GPU). */
    uint32_t max_depth;             /* ray segments per path, >= 1 */
    uint32_t seed;
    uint32_t flags;                 /* AEHIP_FLAG_* */
    uint32_t pass;                  /* AEHIP_PASS_* */
    uint32_t list_source;           /* where the path pass reads the shape list from: 0 = library's
                                       choice, 1 = staged in LDS per workgroup, 2 = streamed through
                                       the scalar cache.  Results are identical; only speed differs. */
    uint32_t tile_rows;             /* 0: the band is the contiguous rows row_begin .. row_begin+row_count-1.
                                       T > 0: the band is row tiles of T rows starting at image rows
                                       row_begin, row_begin + tile_stride, row_begin + 2*tile_stride, ...
                                       (band row j = image row row_begin + (j / T) * tile_stride + j % T);
                                       tiles dealt round-robin to N GPUs (row_begin = g*T, tile_stride =
                                       N*T) give every GPU the same mix of cheap and costly rows.  The
                                       reference schedules tiles of 4 rows (src/raytracer.h:28). */
    uint32_t tile_stride;           /* image rows between the first rows of consecutive tiles, >= tile_rows */
} aehip_params;

typedef struct aehip_stats {
    float    kernel_ms;     /* device time of the last trace's kernel(s), from HIP events
                               (0 with AEHIP_FLAG_NO_TIMING)                                */
    float    copy_ms;       /* device->host copy of the last aehip_backend_trace           */
    uint64_t rays;          /* ray segments (only with AEHIP_FLAG_COUNT_RAYS)              */
    uint64_t paths;         /* width * rows * spp of the last trace                        */
    uint64_t lane_slots;    /* 64 x (segment iterations executed by all waves): rays / lane_slots
                               is the fraction of lanes doing useful work in the bounce loop
                               (only with AEHIP_FLAG_COUNT_RAYS)                            */
    uint32_t vgprs, sgprs, lds_bytes, reserved;   /* resource use of the kernel last launched */
    uint64_t samples_walked; /* camera samples of the last trace that went through a sample loop.  The
                               single-sphere camera pass decides pixels whose samples all miss from
                               bounds over the pixel's sample positions and never walks them: this is
                               spp x the pixels it did walk, counted on the device at set-up with the
                               kernels' own classification (AEHIP_FLAG_WALK_ALL: every pixel).  Every
                               other pass walks every sample (= paths).  `rays` keeps counting one
                               segment per sample of the image, walked or decided.              */
} aehip_stats;

typedef struct aehip_backend aehip_backend;

/* stands behind vulkan_raytracer::init() (src/vulkan_raytracer.h:13): loads the runtime, counts
 * gfx950 devices.  Returns AEHIP_OK when at least one device is usable. */
int aehip_init(void);
/* stands behind vulkan_raytracer::terminate() (src/vulkan_raytracer.h:14; called unconditionally
 * at exit, src/main.cpp:21): safe without a prior init. */
void aehip_terminate(void);
int aehip_device_count(void);
const char *aehip_last_error(void);

/* stands behind the backend constructor `raytracer(u32 *buffer)` (src/raytracer.cpp:15-16).
 * `framebuffer` is borrowed host memory for width*row_count packed 0xAARRGGBB words; it may be
 * NULL for a backend that is only ever traced into device memory. */
int aehip_backend_create(int device, uint32_t *framebuffer, aehip_backend **out);
void aehip_backend_destroy(aehip_backend *backend);

/* stands behind `bool setup()` (src/raytracer.h:25; CPU version src/software_raytracer.cpp:68-91):
 * validates the request, derives the viewport, uploads the scene, sizes device buffers. */
int aehip_backend_setup(aehip_backend *backend, const aehip_scene *scene, const aehip_params *params);

/* stands behind `void trace()` (src/raytracer.h:26; called at src/main.cpp:55): renders and
 * copies the band into the borrowed framebuffer; the image is complete when it returns. */
int aehip_backend_trace(aehip_backend *backend);

/* Same render, outputs left in device memory (for callers that keep the image on the GPU or
 * gather bands across GPUs): argb_dev = width*row_count u32; rgba_dev (nullable) =
 * width*row_count float4 holding the averaged colour before quantisation.  Asynchronous on
 * `hip_stream` (a hipStream_t; NULL = the backend's own stream, which is ordered after work already
 * queued on the device's null stream and which the call then waits on). */
int aehip_backend_trace_device(aehip_backend *backend, uint32_t *argb_dev, float *rgba_dev, void *hip_stream);

/* aehip_backend_trace_device for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the
 * reference.  Same arguments, same image bit for bit; the call makes no hipMalloc / hipFree and
 * calls no allocating library routine: the scratch of the streamed path pass (per-path radiance
 * of one block of samples, per-pixel block totals, the work-unit queue) and the ray counters live
 * in a fixed array of AEHIP_STATIC_SCRATCH_BYTES that every GPU the library runs on holds until
 * the process ends.  A band whose strips do not fit the array at once is rendered in several
 * passes over it; AEHIP_E_TOO_LARGE beyond 4096 passes (about 5.5 million 16x4-pixel strips at
 * 64 or more samples per pixel).  The array keeps the last call's contents (the call clears
 * what must start at zero); every call on a GPU shares that GPU's array.  Outside stream capture the
 * library orders the calls that use it after one another (an event per GPU, whichever backends and
 * streams the calls come from), so they do not race; they do not run side by side either.
 * Passes that use no scratch (the reference's pass, the camera pass, the per-lane path pass
 * without AEHIP_FLAG_COUNT_RAYS) run exactly as through aehip_backend_trace_device.
 * Under stream capture (hip_stream is capturing into a HIP graph): the request must carry
 * AEHIP_FLAG_NO_TIMING (timing events cannot be read back from a graph; AEHIP_E_INVALID otherwise),
 * and the call records nothing but kernel and memset nodes -- the events that order scratch users
 * outside capture are left alone, so replays of the graph are ordered by the streams they are
 * launched on and must not overlap other calls that use the scratch on that GPU. */
#define AEHIP_STATIC_SCRATCH_BYTES (64u << 20)
int aehip_backend_trace_device_static(aehip_backend *backend, uint32_t *argb_dev, float *rgba_dev, void *hip_stream);

int aehip_backend_stats(aehip_backend *backend, aehip_stats *out);

/* Columns [*col_begin, *col_end) of the band outside of which every pixel is its row's background:
 * no sample position of such a pixel sends a ray within reach of a shape, so in every band row all
 * of them hold one and the same colour (the reference's row gradient, src/software_raytracer.cpp:
 * 210-215, which is also what its tiles outside the sphere hold).  Known for scenes of one sphere
 * (the reference's), from the same bound the kernels use to leave those pixels to their
 * background writers; (0, width) otherwise.  The columns follow from the scene, the camera and the
 * frame's size -- not from the band, the seed, the sample count or AEHIP_FLAG_* -- so the GPUs that
 * render the bands of one frame report the same columns.  For callers that move bands between
 * GPUs: those columns and one pixel of background a row are all a peer needs to send.  No
 * counterpart in the reference, whose frame never leaves one address space. */
int aehip_backend_background_columns(aehip_backend *backend, uint32_t *col_begin, uint32_t *col_end);

/* The two ends of moving a band's kept columns (no counterpart in the reference either; device
 * pointers, asynchronous on `hip_stream`, on the current device).
 * pack:   packed row r = { band[r][bg_col], band[r][col_begin .. col_end) }, rows of
 *         col_end - col_begin + 1 words; bg_col is a column outside [col_begin, col_end).
 * expand: image row y (height rows of width words) from packed row row_source[y] (64-bit indices
 *         into the `packed_rows` packed rows of every band, stacked as they were received): its own
 *         words inside [col_begin, col_end), the row's background word outside.  An index outside
 *         [0, packed_rows) leaves its image row unwritten (no read happens through it).
 * Both launch on the device that is current in the calling thread; the buffers and the stream must
 * belong to it. */
int aehip_band_pack_columns(const uint32_t *band_dev, uint32_t rows, uint32_t width, uint32_t col_begin, uint32_t col_end,
                            uint32_t bg_col, uint32_t *packed_dev, void *hip_stream);
int aehip_band_expand_columns(const uint32_t *packed_dev, uint64_t packed_rows, const long long *row_source_dev, uint32_t height,
                              uint32_t width, uint32_t col_begin, uint32_t col_end, uint32_t *image_dev, void *hip_stream);

/* One frame on `n_devices` GPUs of this node, driven from the calling thread; the multi-GPU form
 * of setup() + trace() (src/raytracer.h:25-26) for a host program that is one process, as the
 * reference is (src/main.cpp:11-25).  The rows params->row_begin .. +row_count (0 = all) are cut
 * into tiles of params->tile_rows rows (0 = 4, the reference's tile height, src/raytracer.h:28)
 * dealt round-robin to n_devices bands, band i is rendered on device i (scene replicated), the
 * bands are gathered on device 0 with one RCCL group of ncclSend / ncclRecv (librccl is loaded on
 * first use), and the rows are copied to their places in `framebuffer` (width * row_count words,
 * host memory, image order).  params->tile_stride is ignored.  The result does not depend on
 * n_devices or the tile height.
 * n_devices may exceed the number of GPUs; bands that share device 0's GPU are then moved with
 * device-to-device copies instead of RCCL.  `stats` is optional: kernel_ms is the slowest band's
 * kernel time, rays / paths / lane_slots are summed over bands. */
int aehip_trace_frame(int n_devices, const aehip_scene *scene, const aehip_params *params, uint32_t *framebuffer,
                      aehip_stats *stats);

/* The band arithmetic of aehip_trace_frame on its own (no GPU needed): when `row_count` rows are cut
 * into tiles of `tile_rows` rows (0 = 4, src/raytracer.h:28) dealt round-robin to `n_bands` bands,
 * band `band` starts at row *first_row (relative to the frame's first row), holds *rows rows in
 * tiles every n_bands * tile_rows rows, and its rows follow *rows_before rows of earlier bands in
 * the densely stored gather buffer.  The reference's scheduler walks the same tiles one by one
 * (src/software_raytracer.cpp:242-261). */
int aehip_frame_band(uint32_t row_count, uint32_t tile_rows, int n_bands, int band, uint32_t *first_row, uint32_t *rows,
                     uint32_t *rows_before);

#ifdef __cplusplus
}
#endif
#endif /* AE_HIP_RAYTRACER_H */
