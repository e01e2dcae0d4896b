// TEST INFRASTRUCTURE ONLY.  Entry point for oracle/_ref/raytracer_ref: the reference's own CPU
// software backend, compiled from the sources under /root/reference/src, driven the way
// src/main.cpp:11-59 drives it but without the Vulkan branch (its headers are an un-vendored
// submodule, so src/main.cpp and src/vulkan_raytracer.cpp cannot be compiled here).
//
// Extra, harness-only behaviour: `--repeat N` (default 1) re-runs setup()+trace() on fresh
// backend objects and prints the best wall time of trace() to stdout as
//   ref_trace_ms=<best> nproc=<online cpus> width=<W> height=<H>
// so the GPU-box bench can time the real reference.  The rendered file is output.tga in the
// current directory, exactly as the reference writes it.
#include "commands.h"
#include "output.h"
#include "raytracer.h"
#include "software_raytracer.h"
#include "system.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <time.h>
#include <unistd.h>

static double now_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

int main(int argc, char *argv[]) {
    ae::system_init();
    ae::command_handler::create(std::span(argv, static_cast<size_t>(argc)));

    int repeat = 1;
    bool timing = false;
    for(int i = 1; i + 1 < argc; i++) {
        if(std::strcmp(argv[i], "--repeat") == 0) {
            repeat = std::atoi(argv[i + 1]);
            timing = true;
        }
    }
    if(repeat < 1) repeat = 1;

    double best = 1e300;
    {
        std::unique_ptr<ae::output> output = std::make_unique<ae::output>(std::string_view("output.tga"));
        void *buffer = output->get_buffer();
        for(int r = 0; r < repeat; r++) {
            std::unique_ptr<ae::raytracer> raytracer =
                std::make_unique<ae::software_raytracer>(reinterpret_cast<u32 *>(buffer));
            if(raytracer->setup()) {
                const double t0 = now_ms();
                raytracer->trace();
                const double t1 = now_ms();
                if(t1 - t0 < best) best = t1 - t0;
            }
            // The reference's detached worker threads may still be returning from thread_func
            // (src/software_raytracer.cpp:263-278) when trace() comes back; let them drain before
            // the backend object they point at is destroyed.
            usleep(20000);
        }
    }

    if(timing) {
        auto [w, h] = ae::raytracer::get_resolution();
        std::printf("ref_trace_ms=%.4f nproc=%ld width=%u height=%u\n", best,
                    sysconf(_SC_NPROCESSORS_ONLN), w, h);
    }

    ae::command_handler::destroy();
    return 0;
}
