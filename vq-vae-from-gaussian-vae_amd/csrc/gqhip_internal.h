// gqhip_internal.h -- what the translation units of libgqhip.so share on the host side.
#pragma once
#include <hip/hip_runtime.h>

#include "gqhip.h"

namespace gqhip {
extern thread_local int g_last_hip_error;   // last hipError_t seen by a failing call on this thread (gqhip_last_hip_error)
int check_launch();                          // hipGetLastError() -> GQHIP_OK / GQHIP_ERR_LAUNCH
}  // namespace gqhip
