// rtx_lanes.hip -- the per-ray closest-hit kernel (k_closest_lanes, rtx_closest.hip.h) as a translation unit of its own: its three
// pooled forms are two thirds of the library's compile time, and the Makefile builds this file beside rtx_device.hip.
// rtx_device.hip declares the two instantiations `extern template` and launches them.
#include <hip/hip_runtime.h>

#define RTX_LANES_TU 1

#include "rtx.h"
#include "rtx_math.hip.h"
#include "rtx_pack.h"
#include "rtx_kernels_common.hip.h"
#include "rtx_closest.hip.h"

namespace rtx {
template __global__ void k_closest_lanes<false>(SceneDev, LevelDev, const int*, int, int);
template __global__ void k_closest_lanes<true>(SceneDev, LevelDev, const int*, int, int);
}  // namespace rtx
