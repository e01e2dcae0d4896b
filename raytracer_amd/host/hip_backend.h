// ae::hip_raytracer: the MI355X backend behind the ae::raytracer interface, a thin C++ shell
// over the C ABI of include/hip_raytracer.h.  Its shape follows the reference's other GPU
// backend (src/vulkan_raytracer.h:8-20): static init()/terminate() for the runtime, then the
// usual constructor(framebuffer) / setup() / trace().
#pragma once

#include "backend_interface.h"

#include <string>

namespace ae {
    class hip_raytracer final : public raytracer {
    public:
        static bool init();
        static void terminate();

        explicit hip_raytracer(u32 *buffer);
        ~hip_raytracer() override;

        bool setup() override;
        void trace() override;

        bool traced() const { return traced_; }
        const std::string &error() const { return error_; }
        aehip_stats stats() const;

    private:
        aehip_backend *backend_ = nullptr;       // single-device render
        aehip_scene scene_{};                    // kept for the multi-device frame call
        aehip_params params_{};
        aehip_stats frame_stats_{};
        u32 gpus_ = 1;
        std::string error_;
        bool traced_ = false;
    };
}
