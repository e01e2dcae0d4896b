#include "backend_interface.h"

#include "cli_options.h"

namespace {
    u32 cached_width = 0;
    u32 cached_height = 0;

    aehip_sphere reference_sphere() {
        // src/raytracer.cpp:13: unit sphere at (0,0,-2); the reference shades it by its normal
        aehip_sphere s{};
        s.center[2] = -2.0f;
        s.radius = 1.0f;
        s.albedo[0] = s.albedo[1] = s.albedo[2] = 1.0f;
        s.material = AEHIP_MAT_NORMAL;
        return s;
    }
}

namespace ae {

color raytracer::background0 = color::from_rgb(1.0f, 1.0f, 1.0f);      // src/raytracer.cpp:10
color raytracer::background1 = color::from_argb(0xff4da6f0u);           // src/raytracer.cpp:11
vec4f raytracer::camera_pos{0.0f, 0.0f, 1.0f, 0.0f};                    // src/raytracer.cpp:12
std::vector<aehip_sphere> raytracer::spheres{reference_sphere()};

std::pair<u32, u32> raytracer::get_resolution() {
    if(cached_width == 0 || cached_height == 0) {
        const command_handler &options = command_handler::get();
        cached_width = options.number("width", 0);
        cached_height = options.number("height", 0);
    }
    return {cached_width, cached_height};
}

}
