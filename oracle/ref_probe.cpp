// TEST INFRASTRUCTURE ONLY.  oracle/_ref/ref_probe: calls the reference's own functions, linked
// from the objects compiled out of /root/reference/src, one pixel or one 4x4 tile at a time on a
// single thread.  It bypasses the reference's multithreaded tile queue (whose `finished_`
// handshake is racy, src/software_raytracer.cpp:173-177,259) so fixtures are deterministic.
//
//   ref_probe image  W H out.bin            raw u32 ARGB image, W*H words, produced by
//                                           software_raytracer::trace_tile on every tile
//   ref_probe floats W H out.bin x y [x y…] one 20-float record per listed pixel (layout below)
//
// W and H pass through the reference's command parser, so they are aligned up to 4 like any
// reference run (src/commands.cpp:83-92); the aligned values are printed on stdout.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <queue>
#include <span>
#include <string>
#include <utility>
#include <vector>

#include "aemath.h"
#include "color.h"
#include "commands.h"
#include "ray.h"
#include "raytracer.h"
#include "shapes.h"
#include "vec.h"

// trace_tile/setup state are private members; open them for the harness only.  Member access
// control does not take part in name mangling, so this links against the unmodified object.
#define private public
#include "software_raytracer.h"
#undef private

struct probe_record {
    float x, y;             // pixel coordinates
    float dir[3];           // normalised primary-ray direction (ae::ray ctor, src/ray.h:18-20)
    float hit;              // 1.0 = sphere::intersects returned true
    float t;                // hit_info.t_
    float point[3];         // hit_info.point_
    float normal[3];        // hit_info.normal_
    float rgb[3];           // colour handed to get_argb32 (remap of normal, or lerp background)
    float argb_as_float;    // bit pattern of the packed u32, stored through memcpy
    float pad[3];
};
static_assert(sizeof(probe_record) == 20 * sizeof(float));

int main(int argc, char **argv) {
    if(argc < 5) {
        std::fprintf(stderr, "usage: ref_probe image|floats W H out.bin [x y ...]\n");
        return 2;
    }

    const std::string mode = argv[1];
    std::string ws = argv[2], hs = argv[3];
    char a0[] = "ref_probe", a1[] = "--width", a3[] = "--height";
    char *args[] = {a0, a1, ws.data(), a3, hs.data()};
    ae::command_handler::create(std::span<char *>(args, 5));

    auto [w, h] = ae::raytracer::get_resolution();
    std::printf("width=%u height=%u\n", w, h);

    ae::software_raytracer rt(nullptr);
    rt.setup();

    FILE *f = std::fopen(argv[4], "wb");
    if(!f) return 3;

    if(mode == "image") {
        std::vector<u32> image(static_cast<size_t>(w) * h);
        ae::tile_data tile;
        // get_next_tile walks tiles x-fastest (src/software_raytracer.cpp:242-261); tile.row
        // indexes x and tile.col indexes y (src/software_raytracer.cpp:95-99,206-207).
        while(rt.get_next_tile(tile)) {
            rt.trace_tile(tile);
            const u32 ys = tile.col * ae::raytracer::tile_size;
            const u32 xs = tile.row * ae::raytracer::tile_size;
            u32 k = 0;
            for(u32 y = ys; y < ys + ae::raytracer::tile_size; y++)
                for(u32 x = xs; x < xs + ae::raytracer::tile_size; x++)
                    image[static_cast<size_t>(y) * w + x] = tile.pixels[k++];
        }
        std::fwrite(image.data(), sizeof(u32), image.size(), f);
    } else if(mode == "floats") {
        for(int i = 5; i + 1 < argc; i += 2) {
            const u32 px = static_cast<u32>(std::strtoul(argv[i], nullptr, 10));
            const u32 py = static_cast<u32>(std::strtoul(argv[i + 1], nullptr, 10));

            // Same expressions the reference evaluates per pixel, on the reference's own types
            // (src/software_raytracer.cpp:210-236), so every float below is the reference's value.
            const f32 yf = static_cast<f32>(py);
            const f32 t = yf / static_cast<f32>(rt.height_);
            const ae::vec4f uv = (ae::vec4f(static_cast<f32>(px) + 0.5f, yf + 0.5f, 0.0f) * rt.pixel_size_)
                - (rt.viewport_size_ * ae::vec4f(0.5f, 0.5f, 1.0f));
            const ae::ray ray(ae::raytracer::camera_pos, uv - ae::raytracer::camera_pos);
            ae::ray_hit_info hit;

            probe_record rec;
            std::memset(&rec, 0, sizeof(rec));
            rec.x = static_cast<float>(px);
            rec.y = static_cast<float>(py);
            rec.dir[0] = ray.direction().x_;
            rec.dir[1] = ray.direction().y_;
            rec.dir[2] = ray.direction().z_;

            ae::color c;
            if(ae::raytracer::sphere.intersects(ray, hit)) {
                const std::pair<f32, f32> in{-1.0f, 1.0f}, out{0.0f, 1.0f};
                c = ae::color(ae::remap(hit.normal_.x_, in, out),
                              ae::remap(hit.normal_.y_, in, out),
                              ae::remap(hit.normal_.z_, in, out));
                rec.hit = 1.0f;
                rec.t = hit.t_;
                rec.point[0] = hit.point_.x_; rec.point[1] = hit.point_.y_; rec.point[2] = hit.point_.z_;
                rec.normal[0] = hit.normal_.x_; rec.normal[1] = hit.normal_.y_; rec.normal[2] = hit.normal_.z_;
            } else {
                c = ae::color(ae::lerp(t, ae::raytracer::background0.r_, ae::raytracer::background1.r_),
                              ae::lerp(t, ae::raytracer::background0.g_, ae::raytracer::background1.g_),
                              ae::lerp(t, ae::raytracer::background0.b_, ae::raytracer::background1.b_));
            }
            rec.rgb[0] = c.r_; rec.rgb[1] = c.g_; rec.rgb[2] = c.b_;
            const u32 packed = c.get_argb32();
            std::memcpy(&rec.argb_as_float, &packed, sizeof(packed));
            std::fwrite(&rec, sizeof(rec), 1, f);
        }
    } else {
        std::fclose(f);
        return 2;
    }

    std::fclose(f);
    ae::command_handler::destroy();
    return 0;
}
