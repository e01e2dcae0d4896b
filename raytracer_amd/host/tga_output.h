// Memory-mapped TGA output file, compatible with the reference's ae::output
// (src/output.h:8-37, src/output.cpp:9-25, src/output_linux.cpp:12-63): an 18-byte header with
// image_type 2, 32 bits per pixel, descriptor 0, followed by width*height 0xAARRGGBB words in
// little-endian (B,G,R,A byte order), row 0 first.  get_buffer() points just past the header.
//
// The reference sizes the file as 18 * (w*h*4) bytes (src/output.cpp:11 multiplies where it
// means to add), leaving a sparse zero tail.  That size is reproduced by default so the file
// compares equal byte for byte; `tight` selects 18 + w*h*4 instead.
#pragma once

#include "host_types.h"

#include <string>

namespace ae {
    class output {
    public:
        output(const std::string &file_name, u32 width, u32 height, bool tight);
        ~output();

        output(const output &) = delete;
        output &operator=(const output &) = delete;

        void *get_buffer() const;
        size_t file_size() const { return size_; }
        bool ok() const { return mapping_ != nullptr; }

        static constexpr size_t header_size = 18;

    private:
        void *mapping_ = nullptr;
        size_t size_ = 0;
        int fd_ = -1;
    };
}
