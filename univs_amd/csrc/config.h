// Process-wide settings of libunivs_hip.so (include/univs_hip.h: UnivsConfig, univs_configure).  Set through the C ABI, read
// by the operators at call time: no environment variable is consulted anywhere in the library.
#pragma once
#include "common.h"

namespace univs {

UnivsConfig config();   // a copy of the current settings (capi.hip)

}  // namespace univs
