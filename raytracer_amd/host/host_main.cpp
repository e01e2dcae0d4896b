// raytracer_hip: the host program around the HIP backend.  Same flow as the reference's entry
// point (src/main.cpp:11-59): parse the command line, map the output file, construct a backend
// on the mapped framebuffer, setup(), trace(), tear down.  The backend here is always
// ae::hip_raytracer; where the reference falls back silently when a GPU backend cannot set up
// (src/main.cpp:47-49), this program says why on stderr and exits non-zero, because it has no
// CPU renderer to fall back to.
#include "backend_interface.h"
#include "cli_options.h"
#include "hip_backend.h"
#include "scene_file.h"
#include "tga_output.h"

#include <cstdio>
#include <memory>

int main(int argc, char *argv[]) {
    ae::command_handler::create(std::span<char *>(argv, static_cast<size_t>(argc)));
    const ae::command_handler &options = ae::command_handler::get();

    int status = 0;
    {
        std::string error;
        if(options.has("scene") && !ae::load_scene_file(options.text("scene", ""), error)) {
            std::fprintf(stderr, "raytracer_hip: %s\n", error.c_str());
            status = 2;
        }

        const auto [width, height] = ae::raytracer::get_resolution();
        ae::output output(options.text("output", "output.tga"), width, height, options.enabled("tight-file"));

        if(status == 0) {
            if(!ae::hip_raytracer::init()) {
                std::fprintf(stderr, "raytracer_hip: %s\n", aehip_last_error());
                status = 1;
            } else {
                ae::hip_raytracer backend(static_cast<u32 *>(output.get_buffer()));
                if(backend.setup()) backend.trace();
                if(!backend.traced()) {
                    std::fprintf(stderr, "raytracer_hip: %s\n", backend.error().c_str());
                    status = 1;
                } else if(options.enabled("stats")) {
                    const aehip_stats st = backend.stats();
                    std::fprintf(stderr,
                                 "{\"width\": %u, \"height\": %u, \"paths\": %llu, \"rays\": %llu, \"lane_slots\": %llu, \"kernel_ms\": %.4f, "
                                 "\"copy_ms\": %.4f, \"mpaths_per_s\": %.1f, \"vgprs\": %u, \"lds_bytes\": %u}\n",
                                 width, height, static_cast<unsigned long long>(st.paths),
                                 static_cast<unsigned long long>(st.rays),
                                 static_cast<unsigned long long>(st.lane_slots), st.kernel_ms, st.copy_ms,
                                 st.kernel_ms > 0 ? static_cast<double>(st.paths) / (st.kernel_ms * 1e3) : 0.0, st.vgprs,
                                 st.lds_bytes);
                }
            }
        }
    }

    ae::hip_raytracer::terminate();
    ae::command_handler::destroy();
    return status;
}
