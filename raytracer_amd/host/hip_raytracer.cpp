#include "hip_backend.h"

#include "cli_options.h"

namespace ae {

bool hip_raytracer::init() { return aehip_init() == AEHIP_OK; }

void hip_raytracer::terminate() { aehip_terminate(); }

hip_raytracer::hip_raytracer(u32 *buffer)
    : raytracer(buffer) {}

hip_raytracer::~hip_raytracer() { aehip_backend_destroy(backend_); }

bool hip_raytracer::setup() {
    if(!framebuffer_) {
        error_ = "no framebuffer (output file could not be mapped)";
        return false;
    }

    const command_handler &options = command_handler::get();
    const auto [width, height] = raytracer::get_resolution();

    gpus_ = options.number("gpus", 1);
    if(gpus_ < 1) gpus_ = 1;
    if(gpus_ > static_cast<u32>(aehip_device_count())) {
        error_ = "--gpus " + std::to_string(gpus_) + " but only " + std::to_string(aehip_device_count()) + " device(s) present";
        return false;
    }

    if(gpus_ == 1 && !backend_ && aehip_backend_create(0, framebuffer_, &backend_) != AEHIP_OK) {
        error_ = aehip_last_error();
        return false;
    }

    aehip_scene &scene = scene_;
    scene = aehip_scene{};
    const color &b0 = raytracer::background0, &b1 = raytracer::background1;
    const float bg0[4] = {b0.r_, b0.g_, b0.b_, b0.a_}, bg1[4] = {b1.r_, b1.g_, b1.b_, b1.a_};
    const float cam[4] = {raytracer::camera_pos.x_, raytracer::camera_pos.y_, raytracer::camera_pos.z_, 0.0f};
    for(int i = 0; i < 4; i++) {
        scene.background0[i] = bg0[i];
        scene.background1[i] = bg1[i];
        scene.camera_pos[i] = cam[i];
    }
    scene.sphere_count = static_cast<u32>(raytracer::spheres.size());
    scene.spheres = raytracer::spheres.data();

    aehip_params &params = params_;
    params = aehip_params{};
    params.width = width;
    params.height = height;
    params.spp = options.number("spp", 1);
    params.max_depth = options.number("depth", 1);
    params.seed = options.number("seed", 1);
    params.flags = (options.enabled("jitter") ? AEHIP_FLAG_JITTER : 0u) | (options.enabled("fast") ? AEHIP_FLAG_FAST : 0u)
                 | (options.enabled("stats") ? AEHIP_FLAG_COUNT_RAYS : 0u);
    // With the reference's defaults (built-in scene, one centred sample, depth 1) run the
    // reference's own pass; anything else needs the general one.
    const bool reference_request = params.spp == 1 && params.max_depth == 1 && !options.enabled("jitter")
                                   && !options.has("scene");
    params.pass = reference_request ? AEHIP_PASS_PRIMARY : AEHIP_PASS_PATH;

    // several GPUs: the frame call cuts the image into row bands itself and validates there
    if(gpus_ > 1) return true;

    if(aehip_backend_setup(backend_, &scene, &params) != AEHIP_OK) {
        error_ = aehip_last_error();
        return false;
    }
    return true;
}

void hip_raytracer::trace() {
    if(gpus_ > 1) {
        traced_ = aehip_trace_frame(static_cast<int>(gpus_), &scene_, &params_, framebuffer_, &frame_stats_) == AEHIP_OK;
    } else {
        traced_ = aehip_backend_trace(backend_) == AEHIP_OK;
    }
    if(!traced_) error_ = aehip_last_error();
}

aehip_stats hip_raytracer::stats() const {
    if(gpus_ > 1) return frame_stats_;
    aehip_stats st{};
    if(backend_) aehip_backend_stats(backend_, &st);
    return st;
}

}
