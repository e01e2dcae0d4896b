#include "scene_file.h"

#include "backend_interface.h"

#include <fstream>
#include <sstream>
#include <vector>

namespace ae {

bool load_scene_file(const std::string &path, std::string &error) {
    std::ifstream in(path);
    if(!in) {
        error = "cannot open scene file " + path;
        return false;
    }

    std::vector<aehip_sphere> spheres;
    color bg0 = raytracer::background0, bg1 = raytracer::background1;
    vec4f cam = raytracer::camera_pos;

    std::string line;
    size_t line_no = 0;
    while(std::getline(in, line)) {
        line_no++;
        const size_t hash = line.find('#');
        if(hash != std::string::npos) line.erase(hash);
        std::istringstream fields(line);
        std::string what;
        if(!(fields >> what)) continue;

        bool ok = true;
        if(what == "camera") {
            ok = static_cast<bool>(fields >> cam.x_ >> cam.y_ >> cam.z_);
        } else if(what == "background0" || what == "background1") {
            f32 r, g, b;
            ok = static_cast<bool>(fields >> r >> g >> b);
            if(ok) (what == "background0" ? bg0 : bg1) = color::from_rgb(r, g, b);
        } else if(what == "sphere") {
            aehip_sphere s{};
            std::string material;
            ok = static_cast<bool>(fields >> s.center[0] >> s.center[1] >> s.center[2] >> s.radius >> material
                                          >> s.albedo[0] >> s.albedo[1] >> s.albedo[2] >> s.param);
            if(material == "normal") s.material = AEHIP_MAT_NORMAL;
            else if(material == "lambert") s.material = AEHIP_MAT_LAMBERT;
            else if(material == "metal") s.material = AEHIP_MAT_METAL;
            else if(material == "dielectric") s.material = AEHIP_MAT_DIELECTRIC;
            else ok = false;
            if(ok) spheres.push_back(s);
        } else {
            ok = false;
        }

        if(!ok) {
            error = path + ":" + std::to_string(line_no) + ": cannot parse '" + what + "' directive";
            return false;
        }
    }

    if(spheres.empty()) {
        error = path + ": no spheres";
        return false;
    }

    raytracer::spheres = std::move(spheres);
    raytracer::background0 = bg0;
    raytracer::background1 = bg1;
    raytracer::camera_pos = cam;
    return true;
}

}
