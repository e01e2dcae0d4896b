#include "cli_options.h"

#include "backend_interface.h"

#include <bitset>
#include <cassert>
#include <cerrno>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <memory>

namespace {

using variant = ae::command_handler::variant;

enum class kind { number, toggle, text };

struct flag_spec {
    const char *flag;       // as typed, e.g. "--width"
    const char *name;       // lookup key, e.g. "width"
    kind type;
    variant fallback;       // monostate = no default (flag simply absent)
};

const flag_spec specs[] = {
    // the reference's flags (src/commands.cpp:46-51)
    {"--width", "width", kind::number, 512u},
    {"--height", "height", kind::number, 512u},
    {"--compute", "compute", kind::toggle, false},
    {"--output", "output", kind::text, std::monostate{}},
    // backend choice: --hip selects the MI355X backend as --compute selects the reference's Vulkan one
    // (src/main.cpp:40-42); without it, or when it cannot start, the software backend renders
    {"--hip", "hip", kind::toggle, false},
    {"--gpus", "gpus", kind::number, 1u},
    {"--spp", "spp", kind::number, 1u},
    {"--depth", "depth", kind::number, 1u},
    {"--seed", "seed", kind::number, 1u},
    {"--scene", "scene", kind::text, std::monostate{}},
    {"--jitter", "jitter", kind::toggle, false},
    {"--fast", "fast", kind::toggle, false},
    {"--stats", "stats", kind::toggle, false},
    {"--tight-file", "tight-file", kind::toggle, false},
    {"--threads", "threads", kind::number, 0u},      // software backend: worker threads, 0 = one per processor
};
constexpr size_t spec_count = sizeof(specs) / sizeof(specs[0]);

// Number syntax of the reference (src/commands.cpp:95-111), quirks included: base-10 strtoul, so
// leading blanks, '+' and a wrapping '-' are accepted; ULONG_MAX and any trailing character are
// rejected; errno is tested without being cleared first, so an earlier ERANGE keeps rejecting.
bool read_number(const char *s, variant &out) {
    char *end = nullptr;
    const unsigned long v = std::strtoul(s, &end, 10);
    if(errno == ERANGE || v == ULONG_MAX || *end != '\0') return false;
    out = static_cast<u32>(v);
    return true;
}

bool read_value(const flag_spec &spec, const char *operand, variant &out) {
    switch(spec.type) {
        case kind::number: return read_number(operand, out);
        case kind::toggle: out = true; return true;
        case kind::text: out = std::string(operand); return true;
    }
    return false;
}

std::unique_ptr<ae::command_handler> *holder() {
    static std::unique_ptr<ae::command_handler> instance;
    return &instance;
}

}

namespace ae {

void command_handler::create(std::span<char *> arguments) {
    assert(!*holder());
    holder()->reset(new command_handler(arguments));
}

command_handler &command_handler::get() {
    assert(*holder());
    return **holder();
}

void command_handler::destroy() { holder()->reset(); }

command_handler::command_handler(std::span<char *> arguments) {
    std::bitset<spec_count> taken;

    size_t i = 0;
    while(i < arguments.size()) {
        size_t advance = 1;
        for(size_t j = 0; j < spec_count; j++) {
            const flag_spec &spec = specs[j];
            const size_t operands = spec.type == kind::toggle ? 0 : 1;
            if(taken[j] || std::strcmp(arguments[i], spec.flag) != 0) continue;
            if(i + operands >= arguments.size() && operands > 0) break;      // flag is last: no value to read
            variant parsed;
            // a toggle "parses" its own token; a valued flag parses the next one
            if(!read_value(spec, arguments[i + operands], parsed)) break;    // malformed: flag stays unset
            values_[spec.name] = parsed;
            taken[j] = true;
            advance = 1 + operands;
            break;
        }
        i += advance;
    }

    for(const flag_spec &spec : specs) {
        if(!values_.count(spec.name) && !std::holds_alternative<std::monostate>(spec.fallback)) {
            values_[spec.name] = spec.fallback;
        }
    }

    // src/commands.cpp:83-92
    for(const char *side : {"width", "height"}) {
        const u32 tile = ae::raytracer::tile_size;
        const u32 rounded = (std::get<u32>(values_[side]) + (tile - 1)) & ~(tile - 1);
        values_[side] = rounded < tile ? tile : rounded;
    }
}

bool command_handler::has(const std::string &name) const { return values_.count(name) != 0; }

command_handler::variant command_handler::value(const std::string &name) const {
    const auto it = values_.find(name);
    return it == values_.end() ? variant{} : it->second;
}

u32 command_handler::number(const std::string &name, u32 fallback) const {
    const variant v = value(name);
    return std::holds_alternative<u32>(v) ? std::get<u32>(v) : fallback;
}

bool command_handler::enabled(const std::string &name) const {
    const variant v = value(name);
    return std::holds_alternative<bool>(v) && std::get<bool>(v);
}

std::string command_handler::text(const std::string &name, const std::string &fallback) const {
    const variant v = value(name);
    return std::holds_alternative<std::string>(v) ? std::get<std::string>(v) : fallback;
}

}
