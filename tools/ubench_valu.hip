// tools/ubench_valu.hip -- what does one gfx950 CU sustain for plain vs packed FP32 VALU ops?
// Decides whether the filter pass should use v_pk_* (two primitives per instruction).
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench_valu tools/ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 4096
#define REP8(x) x x x x x x x x

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = a0 * 0.5f, b1 = b0 + 1, b2 = b0 + 2, b3 = b0 + 3, b4 = b0 + 4, b5 = b0 + 5, b6 = b0 + 6, b7 = b0 + 7;
    float m = 0.999f, c = 0.001f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, b0}, p1 = {a1, b1}, p2 = {a2, b2}, p3 = {a3, b3}, p4 = {a4, b4}, p5 = {a5, b5}, p6 = {a6, b6}, p7 = {a7, b7};
    f2 mm = {m, m}, cc = {c, c};
    for (int i = 0; i < ITER; i++) {
        if (KIND == 0) {  // v_fma_f32, 8 independent chains
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if (KIND == 1) {  // v_pk_fma_f32
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(mm), "v"(cc));
        } else if (KIND == 2) {  // v_mul_f32 / v_add_f32 alternating
            asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n"
                         "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if (KIND == 3) {  // v_pk_mul_f32 / v_pk_add_f32
            asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %9\n v_pk_mul_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %9\n"
                         "v_pk_mul_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %9\n v_pk_mul_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %9\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(mm), "v"(cc));
        } else if (KIND == 4) {  // v_fma_f32 with an SGPR operand
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(seed), "v"(c));
        } else if (KIND == 5) {  // v_sqrt_f32 (transcendental rate)
            asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                         "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == 6) {  // v_cmp_lt_f32 to SGPR pair + v_fma
            unsigned long long mk;
            asm volatile("v_fma_f32 %0, %0, %9, %10\n v_cmp_lt_f32 %8, %0, %1\n v_fma_f32 %2, %2, %9, %10\n v_cmp_lt_f32 %8, %2, %3\n"
                         "v_fma_f32 %4, %4, %9, %10\n v_cmp_lt_f32 %8, %4, %5\n v_fma_f32 %6, %6, %9, %10\n v_cmp_lt_f32 %8, %6, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=s"(mk) : "v"(m), "v"(c));
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
    if (r == 12345.678f) out[0] = r;
}

template <int KIND>
void run(const char* name, int lanes_per_instr_ops, int blocks_per_cu)
{
    float* d;
    hipMalloc(&d, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, d, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, d, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double instr = 5.0 * grid * 4 /*waves*/ * (double)ITER * 8;
    double per_simd_cyc = ms * 1e-3 * 2.4e9 / (instr / (256.0 * 4));
    printf("%-28s blocks/CU %d: %.3f ms  wave-instr/s %.3e  cycles per wave-instr per SIMD (at 2.4GHz) %.2f  elem-ops/s %.3e\n", name,
           blocks_per_cu, ms, instr / (ms * 1e-3), per_simd_cyc, instr * 64 * lanes_per_instr_ops / (ms * 1e-3));
    hipFree(d);
}

int main()
{
    for (int b : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", 1, b);
        run<1>("v_pk_fma_f32", 2, b);
        run<2>("v_mul/v_add_f32", 1, b);
        run<3>("v_pk_mul/v_pk_add_f32", 2, b);
        run<4>("v_fma_f32 sgpr operand", 1, b);
        run<5>("v_sqrt_f32", 1, b);
        run<6>("v_fma+v_cmp(sgpr dst)", 1, b);
    }
    return 0;
}
