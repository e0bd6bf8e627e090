// Compile-time aid: instantiates only the shadow-pass kernels (register / LDS budget and ISA of one kernel in seconds
// instead of the whole translation unit's two minutes).  Not part of the library.
//   hipcc <flags of csrc/Makefile> -I ray-tracing_amd/csrc -S --cuda-device-only -Rpass-analysis=kernel-resource-usage tools/dbg/shadow_only.hip
#include <hip/hip_runtime.h>
#include "rtx.h"
#include "rtx_math.hip.h"
#include "rtx_pack.h"
#include "rtx_kernels_common.hip.h"
#include "rtx_closest.hip.h"
#include "rtx_shadow.hip.h"
namespace rtx {
template __global__ void k_shadow_pass<false, true>(SceneDev, ShadowQ, const int*, unsigned char*, int, int, int);
template __global__ void k_shadow_pass<false, false>(SceneDev, ShadowQ, const int*, unsigned char*, int, int, int);
#ifdef SHADOW_ONLY_ALL
template __global__ void k_shadow_affine<false, true, true>(SceneDev, ShadowQ, const int*, unsigned char*, int, int, int);
template __global__ void k_shadow_affine<false, true, false>(SceneDev, ShadowQ, const int*, unsigned char*, int, int, int);
template __global__ void k_shadow_affine<false, false, false>(SceneDev, ShadowQ, const int*, unsigned char*, int, int, int);
#endif
}
