// Fixed-width aliases used across the host layer (same spellings as the reference's
// src/common.h:6-19 so host code reads like the code it plugs into).
#pragma once

#include <cstddef>
#include <cstdint>

using i32 = std::int32_t;
using i64 = std::int64_t;
using u8 = std::uint8_t;
using u16 = std::uint16_t;
using u32 = std::uint32_t;
using u64 = std::uint64_t;
using f32 = float;
using f64 = double;
