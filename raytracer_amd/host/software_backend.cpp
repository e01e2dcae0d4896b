#include "software_backend.h"

#include "cli_options.h"

#include <atomic>
#include <chrono>
#include <cmath>
#include <thread>
#include <vector>

// Everything below is IEEE binary32 evaluated in the order written: this file is compiled with
// -ffp-contract=off and without FMA code generation, like the reference (build.sh:7-31).

namespace {

struct vec3 {
    f32 x = 0.0f, y = 0.0f, z = 0.0f;
};

inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }

// src/vec.h:136-138
inline f32 dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

// src/vec.h:140-166: length over four lanes (the fourth is 0 and adds +0), true division per lane
inline vec3 unit(vec3 v) {
    const f32 len = std::sqrt(((v.x * v.x + v.y * v.y) + v.z * v.z) + 0.0f * 0.0f);
    if(len != 0.0f) {
        v.x /= len;
        v.y /= len;
        v.z /= len;
    }
    return v;
}

// src/ray.h:25
inline vec3 along(vec3 origin, vec3 dir, f32 t) { return {origin.x + dir.x * t, origin.y + dir.y * t, origin.z + dir.z * t}; }

// src/aemath.h:24-26
inline f32 mix(f32 t, f32 v0, f32 v1) { return (1.0f - t) * v0 + (t * v1); }

// src/aemath.h:34-42 called with (-1, 1) -> (0, 1), src/software_raytracer.cpp:227-232
inline f32 to_unit_range(f32 v) { return 0.0f + (v - -1.0f) * (1.0f - 0.0f) / (1.0f - -1.0f); }

// src/color.cpp:96-105: truncation, only blue masked
inline u32 pack(f32 a, f32 r, f32 g, f32 b) {
    return (static_cast<u32>(a * 255.0f) << 24) | (static_cast<u32>(r * 255.0f) << 16) | (static_cast<u32>(g * 255.0f) << 8)
           | (static_cast<u32>(b * 255.0f) & 0xffu);
}

inline f32 saturate(f32 v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }

// ---- sample streams: PCG RXS-M-XS 32 as hash and as per-path generator (the HIP backend's)

inline u32 permute(u32 s) {
    const u32 w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (w >> 22) ^ w;
}
inline u32 step(u32 s) { return s * 747796405u + 2891336453u; }
inline u32 hash(u32 v) { return permute(step(v)); }

struct stream {
    u32 state;
    u32 next() {
        state = step(state);
        return permute(state);
    }
};

// cos, sin of 2 pi u from + and * only: octant reduction, Taylor polynomials on [0, pi/4]
inline void turn(f32 u, f32 &c, f32 &s) {
    const f32 k = u * 8.0f;
    int octant = static_cast<int>(k);
    if(octant > 7) octant = 7;
    const f32 f = k - static_cast<f32>(octant);
    const f32 x = ((octant & 1) ? (1.0f - f) : f) * 0.78539816339744830962f;
    const f32 x2 = x * x;
    f32 sp = -1.0f / 5040.0f;
    sp = sp * x2 + (1.0f / 120.0f);
    sp = sp * x2 + (-1.0f / 6.0f);
    sp = sp * x2 + 1.0f;
    const f32 sx = x * sp;
    f32 cp = 1.0f / 40320.0f;
    cp = cp * x2 + (-1.0f / 720.0f);
    cp = cp * x2 + (1.0f / 24.0f);
    cp = cp * x2 + (-0.5f);
    cp = cp * x2 + 1.0f;
    const bool swapped = ((octant + 1) >> 1) & 1;
    c = swapped ? sx : cp;
    s = swapped ? cp : sx;
    if(octant >= 2 && octant <= 5) c = -c;
    if(octant >= 4) s = -s;
}

// uniform direction from one 32-bit draw: height from the high half, azimuth from the low half
inline vec3 direction_on_sphere(u32 bits) {
    const f32 u1 = static_cast<f32>(bits >> 16) * (1.0f / 65536.0f);
    const f32 u2 = static_cast<f32>(bits & 0xffffu) * (1.0f / 65536.0f);
    const f32 z = 1.0f - 2.0f * u1;
    const f32 r = std::sqrt(std::fmax(0.0f, 1.0f - z * z));
    f32 c, s;
    turn(u2, c, s);
    return {r * c, r * s, z};
}

constexpr f32 nearest_accepted = 0.001f;

struct view {
    f32 vw, vh, psx, psy;
    vec3 eye, top, bottom;      // camera, background at row 0, background towards the last row
    u32 width, height;
    u32 pixel_rotation, sample_rotation;     // hash(seed), hash(seed) ^ golden ratio
    bool jitter;
    u32 depth;
    u32 spp, row_bits;          // samples per pixel; a row of the sample grid has 2^row_bits cells
};

// src/software_raytracer.cpp:218-221 + src/ray.h:18-20
inline vec3 camera_ray(const view &v, f32 px, f32 py) {
    const f32 uvx = px * v.psx - v.vw * 0.5f;
    const f32 uvy = py * v.psy - v.vh * 0.5f;
    return unit(vec3{uvx - v.eye.x, uvy - v.eye.y, 0.0f - v.eye.z});
}

// src/software_raytracer.cpp:210-215
inline vec3 row_gradient(const view &v, u32 y) {
    const f32 t = static_cast<f32>(y) / static_cast<f32>(v.height);
    return {mix(t, v.top.x, v.bottom.x), mix(t, v.top.y, v.bottom.y), mix(t, v.top.z, v.bottom.z)};
}

// One path of the extended model; returns its radiance and counts its segments.
vec3 follow_path(const view &v, const std::vector<aehip_sphere> &spheres, u32 x, u32 y, u32 sample, u64 &rays) {
    const u32 key = hash(y * v.width + x + v.pixel_rotation) + hash(sample + v.sample_rotation);
    stream draws{key};
    f32 jx = 0.5f, jy = 0.5f;
    if(v.jitter) {
        // One sample per cell of a grid over the pixel (rows of 2^row_bits cells, a shorter last
        // row at its own cell width, row heights in proportion to their sample counts), every
        // sample at the pixel's own offset inside its cell: the halves of the pixel's hash.
        const u32 pixel_word = hash(y * v.width + x + v.pixel_rotation);
        const u32 row_cells = 1u << v.row_bits;
        const u32 row = sample >> v.row_bits, cell = sample & (row_cells - 1u);
        const u32 left = v.spp - (row << v.row_bits);
        const u32 in_row = left < row_cells ? left : row_cells;
        const f32 ox = static_cast<f32>(pixel_word >> 16) * (1.0f / 65536.0f);
        const f32 oy = static_cast<f32>(pixel_word & 0xffffu) * (1.0f / 65536.0f);
        jx = (static_cast<f32>(cell) + ox) * (1.0f / static_cast<f32>(in_row));
        jy = (static_cast<f32>(row << v.row_bits) + oy * static_cast<f32>(in_row)) * (1.0f / static_cast<f32>(v.spp));
    }
    vec3 origin = v.eye;
    vec3 dir = camera_ray(v, static_cast<f32>(x) + jx, static_cast<f32>(y) + jy);
    vec3 carried{1.0f, 1.0f, 1.0f};

    for(u32 segment = 0; segment < v.depth; segment++) {
        rays++;
        // nearest accepted root over the list; per sphere the quadratic of src/shapes.cpp:14-23,
        // near root first, then the far one, each inside (nearest_accepted, best so far)
        const f32 a = dot(dir, dir);
        const f32 a4 = 4 * a, a2 = 2.0f * a;
        f32 best = 3.402823466e+38f;
        const aehip_sphere *struck = nullptr;
        for(const aehip_sphere &s : spheres) {
            const f32 radius = std::fabs(s.radius);          // src/shapes.cpp:10-12
            const vec3 oc{s.center[0] - origin.x, s.center[1] - origin.y, s.center[2] - origin.z};
            const f32 b = -2.0f * dot(dir, oc);
            const f32 c = dot(oc, oc) - radius * radius;
            const f32 disc = b * b - a4 * c;
            if(!(disc >= 0)) continue;
            const f32 root = std::sqrt(disc);
            f32 t = (-b - root) / a2;
            if(!(t > nearest_accepted && t < best)) {
                t = (-b + root) / a2;
                if(!(t > nearest_accepted && t < best)) continue;
            }
            best = t;
            struck = &s;
        }

        if(!struck) {
            vec3 sky;
            if(segment == 0) {
                sky = row_gradient(v, y);
            } else {
                const f32 t = 0.5f * (dir.y + 1.0f);
                sky = {mix(t, v.top.x, v.bottom.x), mix(t, v.top.y, v.bottom.y), mix(t, v.top.z, v.bottom.z)};
            }
            return {carried.x * sky.x, carried.y * sky.y, carried.z * sky.z};
        }

        const vec3 centre{struck->center[0], struck->center[1], struck->center[2]};
        const vec3 point = along(origin, dir, best);
        const vec3 outward = unit(point - centre);           // src/shapes.cpp:25

        if(struck->material == AEHIP_MAT_NORMAL) {          // src/software_raytracer.cpp:227-234
            return {carried.x * to_unit_range(outward.x), carried.y * to_unit_range(outward.y), carried.z * to_unit_range(outward.z)};
        }

        const bool from_outside = dot(dir, outward) < 0.0f;
        const vec3 n = from_outside ? outward : -outward;
        vec3 next;
        if(struck->material == AEHIP_MAT_LAMBERT) {
            const vec3 u = direction_on_sphere(draws.next());
            next = {n.x + u.x, n.y + u.y, n.z + u.z};
            if(std::fabs(next.x) < 1e-8f && std::fabs(next.y) < 1e-8f && std::fabs(next.z) < 1e-8f) next = n;
        } else if(struck->material == AEHIP_MAT_METAL) {
            const f32 k = 2.0f * dot(dir, n);
            next = {dir.x - k * n.x, dir.y - k * n.y, dir.z - k * n.z};
            if(struck->param > 0.0f) {
                const vec3 u = direction_on_sphere(draws.next());
                next = {next.x + struck->param * u.x, next.y + struck->param * u.y, next.z + struck->param * u.z};
            }
            if(!(dot(next, n) > 0.0f)) return {};           // absorbed
        } else {
            const f32 ratio = from_outside ? (1.0f / struck->param) : struck->param;
            const f32 cosine = std::fmin(dot(-dir, n), 1.0f);
            const f32 sine = std::sqrt(std::fmax(0.0f, 1.0f - cosine * cosine));
            f32 r0 = (1.0f - ratio) / (1.0f + ratio);
            r0 = r0 * r0;
            const f32 w = 1.0f - cosine, w2 = w * w;
            const f32 reflectance = r0 + (1.0f - r0) * ((w2 * w2) * w);
            const f32 u = static_cast<f32>(draws.next() >> 8) * (1.0f / 16777216.0f);
            if(ratio * sine > 1.0f || reflectance > u) {
                const f32 k = 2.0f * dot(dir, n);
                next = {dir.x - k * n.x, dir.y - k * n.y, dir.z - k * n.z};
            } else {
                const vec3 across{(dir.x + n.x * cosine) * ratio, (dir.y + n.y * cosine) * ratio, (dir.z + n.z * cosine) * ratio};
                const f32 inward = -std::sqrt(std::fabs(1.0f - dot(across, across)));
                next = {across.x + n.x * inward, across.y + n.y * inward, across.z + n.z * inward};
            }
        }
        carried = {carried.x * struck->albedo[0], carried.y * struck->albedo[1], carried.z * struck->albedo[2]};
        origin = point;
        dir = unit(next);
    }
    return {};      // depth exhausted
}

}  // namespace

namespace ae {

software_raytracer::software_raytracer(u32 *buffer)
    : raytracer(buffer) {}

bool software_raytracer::setup() {
    if(!framebuffer_) {
        error_ = "no framebuffer (output file could not be mapped)";
        return false;
    }
    if(raytracer::spheres.empty()) {
        error_ = "scene has no spheres";
        return false;
    }
    const command_handler &options = command_handler::get();
    const auto [width, height] = raytracer::get_resolution();
    width_ = width;
    height_ = height;

    // src/software_raytracer.cpp:68-91
    if(width > height) {
        viewport_w_ = static_cast<f32>(width) / static_cast<f32>(height);
        viewport_h_ = 1.0f;
    } else {
        viewport_w_ = 1.0f;
        viewport_h_ = static_cast<f32>(height) / static_cast<f32>(width);
    }
    pixel_w_ = viewport_w_ / static_cast<f32>(width);
    pixel_h_ = viewport_h_ / static_cast<f32>(height);

    spp_ = options.number("spp", 1);
    depth_ = options.number("depth", 1);
    seed_ = options.number("seed", 1);
    jitter_ = options.enabled("jitter");
    if(spp_ == 0 || depth_ == 0) {
        error_ = "spp and depth must be >= 1";
        return false;
    }
    // the reference's own request (built-in scene, one centred sample, one segment): its pass has
    // no range test on the root (src/shapes.cpp:23-25)
    reference_pass_ = spp_ == 1 && depth_ == 1 && !jitter_ && !options.has("scene");

    // The reference starts one worker per online processor (src/software_raytracer.cpp:145-148).
    // Capped: on a shared machine the processor count is the whole machine's.
    u32 threads = options.number("threads", 0);
    if(threads == 0) threads = std::thread::hardware_concurrency();
    threads_ = threads < 1 ? 1 : (threads > 64 ? 64 : threads);
    return true;
}

void software_raytracer::trace_rows(u32 row_begin, u32 row_end, u64 &rays) const {
    view v{};
    v.vw = viewport_w_; v.vh = viewport_h_; v.psx = pixel_w_; v.psy = pixel_h_;
    v.eye = {raytracer::camera_pos.x_, raytracer::camera_pos.y_, raytracer::camera_pos.z_};
    v.top = {raytracer::background0.r_, raytracer::background0.g_, raytracer::background0.b_};
    v.bottom = {raytracer::background1.r_, raytracer::background1.g_, raytracer::background1.b_};
    v.width = width_; v.height = height_;
    v.pixel_rotation = hash(seed_);
    v.sample_rotation = hash(seed_) ^ 0x9e3779b9u;
    v.jitter = jitter_;
    v.depth = depth_;
    v.spp = spp_;
    u32 bits = 0;
    while((2u << bits) <= spp_ && bits < 31) bits++;
    v.row_bits = bits / 2;      // 2^floor(log2(spp) / 2) cells a row: 32 x 32 for 1024 samples

    for(u32 y = row_begin; y < row_end; y++) {
        u32 *out = framebuffer_ + static_cast<size_t>(y) * width_;
        if(reference_pass_) {
            // src/software_raytracer.cpp:205-240 with src/shapes.cpp:14-30: first sphere only,
            // near root, no range test
            const aehip_sphere &s = raytracer::spheres.front();
            const vec3 centre{s.center[0], s.center[1], s.center[2]};
            const f32 radius = std::fabs(s.radius);
            const vec3 bg = row_gradient(v, y);
            const u32 bg_word = pack(1.0f, bg.x, bg.y, bg.z);
            for(u32 x = 0; x < width_; x++) {
                const vec3 dir = camera_ray(v, static_cast<f32>(x) + 0.5f, static_cast<f32>(y) + 0.5f);
                const vec3 oc = centre - v.eye;
                const f32 a = dot(dir, dir);
                const f32 b = -2.0f * dot(dir, oc);
                const f32 c = dot(oc, oc) - radius * radius;
                const f32 disc = b * b - 4 * a * c;
                u32 word = bg_word;
                if(disc >= 0) {
                    const f32 t = (-b - std::sqrt(disc)) / (2.0f * a);
                    const vec3 n = unit(along(v.eye, dir, t) - centre);
                    word = pack(1.0f, to_unit_range(n.x), to_unit_range(n.y), to_unit_range(n.z));
                }
                out[x] = word;
            }
            rays += width_;
            continue;
        }
        for(u32 x = 0; x < width_; x++) {
            // samples in blocks of AEHIP_SAMPLE_BLOCK: sample order inside a block, block sums in block order
            vec3 total;
            for(u32 first = 0; first < spp_; first += AEHIP_SAMPLE_BLOCK) {
                const u32 last = spp_ - first < AEHIP_SAMPLE_BLOCK ? spp_ : first + AEHIP_SAMPLE_BLOCK;
                vec3 block;
                for(u32 s = first; s < last; s++) {
                    const vec3 L = follow_path(v, raytracer::spheres, x, y, s, rays);
                    block.x += L.x; block.y += L.y; block.z += L.z;
                }
                total.x += block.x; total.y += block.y; total.z += block.z;
            }
            const f32 n = static_cast<f32>(spp_);
            out[x] = pack(1.0f, saturate(total.x / n), saturate(total.y / n), saturate(total.z / n));
        }
    }
}

void software_raytracer::trace() {
    const auto t0 = std::chrono::steady_clock::now();
    std::atomic<u32> next_row{0};
    std::atomic<u64> rays{0};
    auto worker = [&]() {
        u64 mine = 0;
        for(;;) {
            const u32 begin = next_row.fetch_add(tile_size);
            if(begin >= height_) break;
            trace_rows(begin, begin + tile_size < height_ ? begin + tile_size : height_, mine);
        }
        rays.fetch_add(mine);
    };
    const u32 helpers = threads_ > 1 ? threads_ - 1 : 0;
    std::vector<std::thread> pool;
    pool.reserve(helpers);
    for(u32 i = 0; i < helpers; i++) pool.emplace_back(worker);
    worker();
    for(std::thread &t : pool) t.join();

    stats_.paths = static_cast<u64>(width_) * height_ * (reference_pass_ ? 1u : spp_);
    stats_.rays = rays.load();
    stats_.threads = threads_;
    stats_.trace_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

}
