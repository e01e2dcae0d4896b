// Integer and float aliases the host layer uses, spelled as in the code it plugs into
// (reference: src/common.h:6-19), so a class written against ae::raytracer compiles here as is.
// Only the ones the host layer needs.
#pragma once

#include <cstdint>

using u8 = std::uint8_t;     // TGA header fields (src/output.cpp:13-23)
using u16 = std::uint16_t;   // image sides in the TGA header
using u32 = std::uint32_t;   // packed 0xAARRGGBB pixels, resolutions
using u64 = std::uint64_t;   // path and ray counts
using f32 = float;           // colour channels, vec4f lanes
