// Compiler guard ("compiler trap"): every kernel of the library in one device-only translation unit
// without the host side, for `make -C ray-tracing_amd/csrc verify-mir` (LLVM's machine verifier over the AMDGPU
// back end: the miscompile of round 1 is a virtual register used before its definition, which the verifier reports)
// and for the IR check of tests/test_compiler_guard.py.  Not part of the library.
#include <hip/hip_runtime.h>
#include "rtx.h"
#include "rtx_math.hip.h"
#include "rtx_pack.h"
#include "rtx_kernels_common.hip.h"
#include "rtx_closest.hip.h"
#include "rtx_shadow.hip.h"
#include "rtx_cells.hip.h"
#include "rtx_shade.hip.h"
namespace rtx {
#define SHADOW_ARGS SceneDev, ShadowQ, const int*, unsigned char*, int, int, int
#if !defined(VERIFY_PART) || VERIFY_PART == 0
template __global__ void k_shadow_pass<false, true>(SHADOW_ARGS);
template __global__ void k_shadow_pass<false, false>(SHADOW_ARGS);
template __global__ void k_shadow_affine<false, true, true>(SHADOW_ARGS);
template __global__ void k_shadow_affine<false, true, false>(SHADOW_ARGS);
template __global__ void k_shadow_affine<false, false, false>(SHADOW_ARGS);
template __global__ void k_shadow_cells<false>(SceneDev, CellTab, ShadowQ, const int*, unsigned char*, int);
template __global__ void k_shadow_direct<false>(SceneDev, CellTab, LevelDev, const int*);
template __global__ void k_primary_cells<false>(SceneDev, CamTab, LevelDev, const int*);
// (k_cells_build and k_aff_build are not templates: this translation unit compiles them as they stand)
#endif
#if !defined(VERIFY_PART) || VERIFY_PART == 1
template __global__ void k_shadow_pass<true, true>(SHADOW_ARGS);
template __global__ void k_shadow_pass<true, false>(SHADOW_ARGS);
template __global__ void k_shadow_affine<true, true, true>(SHADOW_ARGS);
template __global__ void k_shadow_affine<true, true, false>(SHADOW_ARGS);
template __global__ void k_shadow_affine<true, false, false>(SHADOW_ARGS);
template __global__ void k_shadow_cells<true>(SceneDev, CellTab, ShadowQ, const int*, unsigned char*, int);
template __global__ void k_shadow_direct<true>(SceneDev, CellTab, LevelDev, const int*);
template __global__ void k_primary_cells<true>(SceneDev, CamTab, LevelDev, const int*);
#endif
#if !defined(VERIFY_PART) || VERIFY_PART == 2
template __global__ void k_closest_f<false, false>(SceneDev, LevelDev, const int*);
template __global__ void k_closest_f<true, false>(SceneDev, LevelDev, const int*);
template __global__ void k_closest_lanes<false>(SceneDev, LevelDev, const int*, int, int);
#endif
#if !defined(VERIFY_PART) || VERIFY_PART == 3
template __global__ void k_closest_f<false, true>(SceneDev, LevelDev, const int*);
template __global__ void k_closest_f<true, true>(SceneDev, LevelDev, const int*);
template __global__ void k_closest_lanes<true>(SceneDev, LevelDev, const int*, int, int);
template __global__ void k_closest<false>(SceneDev, LevelDev, const int*, unsigned long long*);
template __global__ void k_closest<true>(SceneDev, LevelDev, const int*, unsigned long long*);
template __global__ void k_shadow<false>(SceneDev, LevelDev, const int*, unsigned long long*);
template __global__ void k_shadow<true>(SceneDev, LevelDev, const int*, unsigned long long*);
template __global__ void k_setup<256>(SceneDev, LevelDev, LevelDev, const int*, int*, int, int*);
template __global__ void k_setup<1024>(SceneDev, LevelDev, LevelDev, const int*, int*, int, int*);
#endif
}
