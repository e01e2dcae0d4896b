// The interface every rendering backend of the host program implements, shaped like the
// reference's ae::raytracer (src/raytracer.h:11-33): a backend borrows the output framebuffer at
// construction, setup() says whether it can run, trace() fills the framebuffer before it
// returns.  As in the reference the scene lives in public statics (src/raytracer.h:16-19); the
// single sphere is generalised to a list that --scene can replace.
#pragma once

#include "host_types.h"
#include "hip_raytracer.h"

#include <utility>
#include <vector>

namespace ae {
    // a,r,g,b floats, the layout of ae::color (src/color.h:26-29)
    struct color {
        f32 a_ = 0.0f, r_ = 0.0f, g_ = 0.0f, b_ = 0.0f;

        static color from_rgb(f32 r, f32 g, f32 b) { return color{1.0f, r, g, b}; }
        // 0xAARRGGBB -> floats, each byte / 255.0f (src/color.cpp:32-36)
        static color from_argb(u32 argb) {
            auto ch = [argb](u32 shift) { return static_cast<f32>((argb >> shift) & 0xffu) / 255.0f; };
            return color{ch(24), ch(16), ch(8), ch(0)};
        }
    };

    struct vec4f {
        f32 x_ = 0.0f, y_ = 0.0f, z_ = 0.0f, w_ = 0.0f;
    };

    class raytracer {
    public:
        static constexpr u32 tile_size = 4;      // src/raytracer.h:13: image sides are multiples of it

        // scene globals; defaults are the reference's test scene (src/raytracer.cpp:10-13)
        static color background0;
        static color background1;
        static vec4f camera_pos;
        static std::vector<aehip_sphere> spheres;

        // width/height from the command line, cached (src/raytracer.cpp:18-34)
        static std::pair<u32, u32> get_resolution();

        virtual ~raytracer() = default;
        virtual bool setup() = 0;
        virtual void trace() = 0;

    protected:
        explicit raytracer(u32 *buffer) : framebuffer_(buffer) {}
        u32 *framebuffer_ = nullptr;
    };
}
