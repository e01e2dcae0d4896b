// TEST INFRASTRUCTURE ONLY (oracle/_ref build).  Force-included (-include) when the reference's
// own translation units are compiled in place from /root/reference/src.
//
// libstdc++ 11 declares ::sqrtf but not std::sqrtf, which the reference spells
// (src/vec.h:149, src/shapes.cpp:23).  Newer standard libraries declare exactly this alias; it
// resolves to glibc's sqrtf (correctly rounded, `sqrtss`), so no arithmetic is replaced.
#pragma once
#ifdef __cplusplus
#include <cmath>
namespace std { using ::sqrtf; }
#endif
