// raytracer_hip: the host program around the rendering backends.  Same flow as the reference's
// entry point (src/main.cpp:11-59): parse the command line, map the output file, construct a
// backend on the mapped framebuffer, setup(), trace(), tear down.
//
// Backend choice follows src/main.cpp:40-52 with --hip in --compute's place: the flag together
// with a runtime that initialises selects ae::hip_raytracer; if its setup() fails the program
// falls back to ae::software_raytracer, which is also what runs without the flag.  The reference
// falls back without a word; this program adds one line on stderr saying why, and still exits 0
// with the image written.
#include "backend_interface.h"
#include "cli_options.h"
#include "hip_backend.h"
#include "scene_file.h"
#include "software_backend.h"
#include "tga_output.h"

#include <cstdio>
#include <memory>

namespace {

// returns the process exit status
int run_raytracer(void *buffer) {
    const ae::command_handler &options = ae::command_handler::get();
    const auto [width, height] = ae::raytracer::get_resolution();
    u32 *pixels = static_cast<u32 *>(buffer);

    if(options.enabled("hip")) {
        if(ae::hip_raytracer::init()) {
            ae::hip_raytracer gpu(pixels);
            if(gpu.setup()) gpu.trace();
            if(gpu.traced()) {
                if(options.enabled("stats")) {
                    const aehip_stats st = gpu.stats();
                    std::fprintf(stderr,
                                 "{\"backend\": \"hip\", \"width\": %u, \"height\": %u, \"paths\": %llu, \"samples_walked\": %llu, \"rays\": %llu, "
                                 "\"lane_slots\": %llu, \"kernel_ms\": %.4f, \"copy_ms\": %.4f, \"mpaths_per_s\": %.1f, \"vgprs\": %u, \"lds_bytes\": %u}\n",
                                 width, height, static_cast<unsigned long long>(st.paths), static_cast<unsigned long long>(st.samples_walked),
                                 static_cast<unsigned long long>(st.rays), static_cast<unsigned long long>(st.lane_slots), st.kernel_ms, st.copy_ms,
                                 st.kernel_ms > 0 ? static_cast<double>(st.paths) / (st.kernel_ms * 1e3) : 0.0, st.vgprs, st.lds_bytes);
                }
                return 0;
            }
            std::fprintf(stderr, "raytracer_hip: HIP backend failed (%s); using the software backend\n", gpu.error().c_str());
        } else {
            std::fprintf(stderr, "raytracer_hip: HIP backend unavailable (%s); using the software backend\n", aehip_last_error());
        }
    }

    ae::software_raytracer cpu(pixels);
    if(!cpu.setup()) {
        std::fprintf(stderr, "raytracer_hip: %s\n", cpu.error().c_str());
        return 1;
    }
    cpu.trace();
    if(options.enabled("stats")) {
        const ae::software_raytracer::statistics &st = cpu.stats();
        std::fprintf(stderr,
                     "{\"backend\": \"software\", \"width\": %u, \"height\": %u, \"paths\": %llu, \"rays\": %llu, \"trace_ms\": %.4f, "
                     "\"mpaths_per_s\": %.1f, \"threads\": %u}\n",
                     width, height, static_cast<unsigned long long>(st.paths), static_cast<unsigned long long>(st.rays), st.trace_ms,
                     st.trace_ms > 0 ? static_cast<double>(st.paths) / (st.trace_ms * 1e3) : 0.0, st.threads);
    }
    return 0;
}

}

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

        if(status == 0) status = run_raytracer(output.get_buffer());
    }

    ae::hip_raytracer::terminate();
    ae::command_handler::destroy();
    return status;
}
