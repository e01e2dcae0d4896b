// Text scene files for --scene (the reference has no scene input; its scene is compiled in,
// src/raytracer.cpp:10-13).  One directive per line, '#' starts a comment:
//   camera x y z
//   background0 r g b          colour at image row 0
//   background1 r g b          colour towards the last row
//   sphere cx cy cz radius material r g b param
// with material one of: normal lambert metal dielectric (param = fuzz / index of refraction).
#pragma once

#include <string>

namespace ae {
    // Replaces the scene statics of ae::raytracer.  Returns false and fills `error` on failure.
    bool load_scene_file(const std::string &path, std::string &error);
}
