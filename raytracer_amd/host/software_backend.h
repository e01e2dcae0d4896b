// ae::software_raytracer: the CPU backend of the host program, the peer of ae::hip_raytracer
// behind the same interface -- what runs when --hip is absent and what the program falls back to
// when the GPU backend cannot initialise or set up (the reference's policy, src/main.cpp:40-52).
//
// It renders what the reference's CPU backend renders (src/software_raytracer.cpp:68-91 viewport,
// :205-240 pixel kernel, src/shapes.cpp:14-30, src/color.cpp:96-105) with the same IEEE fp32
// operations in the same order, so its image is the reference's byte for byte, and the same
// extended model as the HIP backend (shape list, materials, samples per pixel, bounce depth,
// counter-based sample streams), so every command line gives the same image with or without a GPU.
//
// Scheduling differs from the reference on purpose: its per-tile queue with three lock
// operations per 16 pixels and a `finished_` flag read under another mutex than it is written
// under (src/software_raytracer.cpp:173-177,259) can return before the last tiles are copied.
// Here rows are dealt in tiles of raytracer::tile_size by one atomic counter to joinable threads
// that write straight into the framebuffer; trace() returns after joining them.
#pragma once

#include "backend_interface.h"

#include <string>

namespace ae {
    class software_raytracer final : public raytracer {
    public:
        explicit software_raytracer(u32 *buffer);

        bool setup() override;
        void trace() override;

        struct statistics {
            u64 paths = 0, rays = 0;
            double trace_ms = 0.0;
            u32 threads = 0;
        };
        const statistics &stats() const { return stats_; }
        const std::string &error() const { return error_; }

    private:
        // src/software_raytracer.h:36-43: viewport and pixel size on the z = 0 plane
        f32 viewport_w_ = 1.0f, viewport_h_ = 1.0f, pixel_w_ = 0.0f, pixel_h_ = 0.0f;
        u32 width_ = 0, height_ = 0;
        u32 spp_ = 1, depth_ = 1, seed_ = 1, threads_ = 1;
        bool jitter_ = false, reference_pass_ = true;
        statistics stats_;
        std::string error_;

        void trace_rows(u32 row_begin, u32 row_end, u64 &rays) const;
    };
}
