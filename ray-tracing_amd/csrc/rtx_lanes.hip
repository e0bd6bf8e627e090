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

#ifdef RTX_WAVE_CLOCK
// diagnosis build only: copies out (and clears) the wave clocks k_closest_lanes left; returns the number of records
extern "C" int rtx_debug_wave_clock(unsigned long long* out3, int capacity)
{
    unsigned int n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(rtx::g_wclk_n), sizeof n) != hipSuccess) return -1;
    if (n > rtx::kWaveClockCap) n = rtx::kWaveClockCap;
    if ((int)n > capacity) n = (unsigned)capacity;
    if (n && hipMemcpyFromSymbol(out3, HIP_SYMBOL(rtx::g_wclk), (size_t)n * 3 * sizeof(unsigned long long)) != hipSuccess) return -1;
    const unsigned int zero = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(rtx::g_wclk_n), &zero, sizeof zero);
    return (int)n;
}
#endif
