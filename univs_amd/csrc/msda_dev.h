// Device-side typedefs shared by the MSDA kernels (msda_strips.hip).
#pragma once

namespace univs {

typedef float t3v2 __attribute__((ext_vector_type(2)));
typedef float t3v4 __attribute__((ext_vector_type(4)));
typedef unsigned t3u2 __attribute__((ext_vector_type(2)));
#define T3_LDS __attribute__((address_space(3)))

}  // namespace univs
