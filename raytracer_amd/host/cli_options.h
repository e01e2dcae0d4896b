// Command line of the host program, behind the reference's command_handler interface
// (src/commands.h:12-33): create()/get()/destroy() singleton, has(), value().
//
// The four reference flags behave exactly as in src/commands.cpp:39-122: a flag is taken the
// first time it appears with a value that parses; a malformed value leaves it unset so its
// default applies; unknown arguments are skipped; width and height are rounded up to a multiple
// of raytracer::tile_size, at least one tile.  The flags added for the HIP backend default to the
// reference's behaviour (1 sample at the pixel centre, depth 1, built-in scene).
#pragma once

#include "host_types.h"

#include <map>
#include <span>
#include <string>
#include <variant>

namespace ae {
    class command_handler {
    public:
        using variant = std::variant<std::monostate, bool, u32, std::string>;

        static void create(std::span<char *> arguments);
        static command_handler &get();
        static void destroy();

        bool has(const std::string &name) const;
        variant value(const std::string &name) const;

        u32 number(const std::string &name, u32 fallback) const;
        bool enabled(const std::string &name) const;
        std::string text(const std::string &name, const std::string &fallback) const;

    private:
        explicit command_handler(std::span<char *> arguments);
        std::map<std::string, variant> values_;
    };
}
