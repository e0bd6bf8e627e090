// tools/ubench_valu2.hip -- cycle-accurate (s_memtime) issue rates of FP32 VALU forms on gfx950.
// Each wave times N instructions of one kind; with w waves resident per SIMD the SIMD's rate is
// w*N / elapsed.  One block of 256*w threads per CU => w waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define ITER 16384

#define BODY8(INS) \
    asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c), "s"(sg));
#define BODY8P(INS) \
    asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(mm), "v"(cc), "s"(sg2));

#define I_ADD32(i)  "v_add_f32_e32 %" #i ", %8, %" #i "\n"
#define I_MUL32(i)  "v_mul_f32_e32 %" #i ", %8, %" #i "\n"
#define I_ADD64(i)  "v_add_f32_e64 %" #i ", %8, %" #i "\n"
#define I_ADDS(i)   "v_add_f32_e32 %" #i ", %10, %" #i "\n"
#define I_FMAC32(i) "v_fmac_f32_e32 %" #i ", %8, %9\n"
#define I_FMA64(i)  "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define I_FMAS(i)   "v_fma_f32 %" #i ", %" #i ", %10, %9\n"
#define I_PKFMA(i)  "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define I_PKMUL(i)  "v_pk_mul_f32 %" #i ", %" #i ", %8\n"
#define I_PKADD(i)  "v_pk_add_f32 %" #i ", %" #i ", %9\n"
#define I_PKADDS(i) "v_pk_add_f32 %" #i ", %" #i ", %10\n"
#define I_MAX(i)    "v_max_f32_e32 %" #i ", %8, %" #i "\n"
#define I_SQRT(i)   "v_sqrt_f32_e32 %" #i ", %" #i "\n"
#define I_RCP(i)    "v_rcp_f32_e32 %" #i ", %" #i "\n"
#define I_CMP(i)    "v_cmp_lt_f32_e32 vcc, %8, %" #i "\n"
#define I_CMPS(i)   "v_cmp_lt_f32_e64 s[20:21], %8, %" #i "\n"
#define I_SUBREV(i) "v_subrev_f32_e32 %" #i ", %10, %" #i "\n"

template <int KIND>
__global__ void k(unsigned long long* out, float seed, float sg, int dummy)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float m = 0.999f, c = 0.001f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a1, a2}, p2 = {a2, a3}, p3 = {a3, a0}, p4 = {a4, a1}, p5 = {a5, a1}, p6 = {a6, a0}, p7 = {a7, a2};
    f2 mm = {m, m}, cc = {c, c};
    f2 sg2 = {sg, sg};
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITER; i++) {
        if (KIND == 0) BODY8(I_ADD32)
        if (KIND == 1) BODY8(I_MUL32)
        if (KIND == 2) BODY8(I_ADD64)
        if (KIND == 3) BODY8(I_ADDS)
        if (KIND == 4) BODY8(I_FMAC32)
        if (KIND == 5) BODY8(I_FMA64)
        if (KIND == 6) BODY8(I_FMAS)
        if (KIND == 7) BODY8P(I_PKFMA)
        if (KIND == 8) BODY8P(I_PKMUL)
        if (KIND == 9) BODY8P(I_PKADD)
        if (KIND == 10) BODY8(I_MAX)
        if (KIND == 11) BODY8(I_SQRT)
        if (KIND == 12) BODY8(I_RCP)
        if (KIND == 13) asm volatile(I_CMP(0) I_CMP(1) I_CMP(2) I_CMP(3) I_CMP(4) I_CMP(5) I_CMP(6) I_CMP(7)
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c), "s"(sg) : "vcc");
        if (KIND == 14) asm volatile(I_CMPS(0) I_CMPS(1) I_CMPS(2) I_CMPS(3) I_CMPS(4) I_CMPS(5) I_CMPS(6) I_CMPS(7)
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c), "s"(sg) : "s20", "s21");
        if (KIND == 15) BODY8(I_SUBREV)
        if (KIND == 16) BODY8P(I_PKADDS)
    }
    asm volatile("s_nop 0" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
    if ((threadIdx.x & 63) == 0) { int wv = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; out[2 * wv] = (t1 - t0) + (r == 12345.678f ? 1 : 0); out[2 * wv + 1] = r1 - r0; }
}

template <int KIND>
void run(const char* name)
{
    printf("%-26s", name);
    for (int w : {1, 2, 4, 8}) {
        int nblk = 256, nthr = 256 * w;
        if (nthr > 1024) { nblk = 256 * (nthr / 1024); nthr = 1024; }
        int nwaves = nblk * nthr / 64;
        unsigned long long* d;
        (void)hipMalloc(&d, nwaves * 16);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(nblk), dim3(nthr), 0, 0, d, 1.0f, 0.5f, 0);
        hipLaunchKernelGGL(k<KIND>, dim3(nblk), dim3(nthr), 0, 0, d, 1.0f, 0.5f, 0);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(nblk), dim3(nthr), 0, 0, d, 1.0f, 0.5f, 0);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> hh(2 * nwaves), h(nwaves), hr(nwaves);
        (void)hipMemcpy(hh.data(), d, nwaves * 16, hipMemcpyDeviceToHost);
        for (int i = 0; i < nwaves; i++) { h[i] = hh[2 * i]; hr[i] = hh[2 * i + 1]; }
        std::sort(h.begin(), h.end()); std::sort(hr.begin(), hr.end());
        double med = (double)h[nwaves / 2];
        double ghz = med / ((double)hr[nwaves / 2] * 10e-9) / 1e9;
        double wall_rate = (double)nwaves * ITER * 8 / (ms * 1e-3) / 1024.0;  // wave-instr per second per SIMD
        // cycles per instruction as seen by one wave, and SIMD throughput assuming w waves share it
        printf(" w=%d: %.2f tick/instr/wave, memtime %.2f GHz, wall %.3f Ginstr/s/SIMD |", w, med / (ITER * 8.0), ghz, wall_rate / 1e9);
        (void)hipFree(d);
    }
    printf("\n");
}

int main()
{
    run<0>("v_add_f32_e32");
    run<1>("v_mul_f32_e32");
    run<2>("v_add_f32_e64");
    run<3>("v_add_f32_e32 sgpr");
    run<15>("v_subrev_f32_e32 sgpr");
    run<4>("v_fmac_f32_e32");
    run<5>("v_fma_f32 (vop3)");
    run<6>("v_fma_f32 sgpr");
    run<7>("v_pk_fma_f32");
    run<8>("v_pk_mul_f32");
    run<9>("v_pk_add_f32");
    run<16>("v_pk_add_f32 sgpr");
    run<10>("v_max_f32_e32");
    run<11>("v_sqrt_f32");
    run<12>("v_rcp_f32");
    run<13>("v_cmp_lt_f32 vcc");
    run<14>("v_cmp_lt_f32 sgpr");
    return 0;
}
