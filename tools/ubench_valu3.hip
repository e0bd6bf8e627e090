// tools/ubench_valu3.hip -- issue rates (wall clock, 8 waves/SIMD) of the remaining candidate
// instructions for the filter loop, and of a broadcast ds_read_b128 interleaved with VALU.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 8192
#define I_AND(i)    "v_and_b32_e32 %" #i ", %8, %" #i "\n"
#define I_OR(i)     "v_or_b32_e32 %" #i ", %8, %" #i "\n"
#define I_MIN(i)    "v_min_f32_e32 %" #i ", %8, %" #i "\n"
#define I_MAX3(i)   "v_max3_f32 %" #i ", %" #i ", %8, %9\n"
#define I_SUB(i)    "v_sub_f32_e32 %" #i ", %8, %" #i "\n"
#define I_FMANEG(i) "v_fma_f32 %" #i ", %" #i ", %8, -%9\n"
#define I_MOV(i)    "v_mov_b32_e32 %" #i ", %8\n"
#define I_ADD(i)    "v_add_f32_e32 %" #i ", %8, %" #i "\n"
#define I_ADDU(i)   "v_add_u32_e32 %" #i ", %8, %" #i "\n"
#define I_CNDMASK(i) "v_cndmask_b32_e32 %" #i ", %8, %" #i ", vcc\n"
#define I_MULLIT(i) "v_mul_f32_e32 %" #i ", 0x3f7fbe77, %" #i "\n"
#define I_MULINL(i) "v_mul_f32_e32 %" #i ", 0.5, %" #i "\n"
#define BODY8(INS) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c) : "vcc");

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, float seed)
{
    __shared__ float4 tile[256];
    tile[threadIdx.x] = make_float4(seed, seed + 1, seed + 2, seed + 3);
    __syncthreads();
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float m = 0.999f, c = 0.001f;
    float acc = 0;
    for (int i = 0; i < ITER; i++) {
        if (KIND == 0) BODY8(I_AND)
        if (KIND == 1) BODY8(I_OR)
        if (KIND == 2) BODY8(I_MIN)
        if (KIND == 3) BODY8(I_MAX3)
        if (KIND == 4) BODY8(I_SUB)
        if (KIND == 5) BODY8(I_FMANEG)
        if (KIND == 6) BODY8(I_MOV)
        if (KIND == 7) BODY8(I_ADD)
        if (KIND == 8) BODY8(I_ADDU)
        if (KIND == 9) BODY8(I_CNDMASK)
        if (KIND == 10) BODY8(I_MULLIT)
        if (KIND == 11) BODY8(I_MULINL)
        if (KIND == 12) {  // one broadcast ds_read_b128 per 8 fast VALU
            float4 b = tile[i & 255];
            BODY8(I_ADD)
            acc += b.x + b.y + b.z + b.w;
        }
        if (KIND == 13) {  // one broadcast ds_read_b128 per 16 fast VALU
            float4 b = tile[i & 255];
            BODY8(I_ADD)
            BODY8(I_ADD)
            acc += b.x + b.y + b.z + b.w;
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + acc;
    if (r == 12345.678f) out[0] = r;
}

template <int KIND>
void run(const char* name, double instr_per_iter)
{
    float* d;
    (void)hipMalloc(&d, 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    int grid = 256 * 8;
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, d, 1.0f);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, d, 1.0f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, d, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    double instr = (double)grid * 4 * ITER * instr_per_iter;
    printf("%-34s %.3f ms  %.3f G wave-instr/s/SIMD\n", name, ms, instr / (ms * 1e-3) / 1024 / 1e9);
    (void)hipFree(d);
}

int main()
{
    run<7>("v_add_f32_e32 (reference)", 8);
    run<0>("v_and_b32_e32", 8);
    run<1>("v_or_b32_e32", 8);
    run<2>("v_min_f32_e32", 8);
    run<3>("v_max3_f32", 8);
    run<4>("v_sub_f32_e32", 8);
    run<5>("v_fma_f32 neg modifier", 8);
    run<6>("v_mov_b32_e32", 8);
    run<8>("v_add_u32_e32", 8);
    run<9>("v_cndmask_b32_e32", 8);
    run<10>("v_mul_f32_e32 literal", 8);
    run<11>("v_mul_f32_e32 inline const", 8);
    run<12>("8 v_add + 1 ds_read_b128 bcast (+5 adds)", 13);
    run<13>("16 v_add + 1 ds_read_b128 bcast (+5 adds)", 21);
    return 0;
}
