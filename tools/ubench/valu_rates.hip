// VALU issue-rate microbenchmark for gfx950 (design input for k_demod_bits).
// Each kernel runs ITER iterations of 16 independent instances of one instruction.
// Reports wave-instructions per cycle per CU (at the measured kernel time and an assumed clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define ITER 4096
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

// scalar-register accumulators: a[16]
#define KERNEL1(NAME, ASMSTR, CONSTRAINT_EXTRA)                                                  \
    __global__ void NAME(float *out, float seed) {                                              \
        float a[16]; float b = seed, c = seed * 0.5f;                                            \
        for (int i = 0; i < 16; i++) a[i] = seed + i;                                            \
        for (int it = 0; it < ITER; it++) {                                                      \
            _Pragma("unroll") for (int i = 0; i < 16; i++)                                       \
                asm volatile(ASMSTR : "+v"(a[i]) : "v"(b), "v"(c) CONSTRAINT_EXTRA);             \
        }                                                                                        \
        float s = 0; for (int i = 0; i < 16; i++) s += a[i];                                     \
        if (s == 12345.678f) out[0] = s;                                                         \
    }

KERNEL1(k_fma, "v_fma_f32 %0, %1, %2, %0", )
KERNEL1(k_fmac, "v_fmac_f32 %0, %1, %2", )
KERNEL1(k_add, "v_add_f32 %0, %1, %0", )
KERNEL1(k_cvt_ub0, "v_cvt_f32_ubyte0 %0, %1", )
KERNEL1(k_cvt_ub3, "v_cvt_f32_ubyte3 %0, %1", )
KERNEL1(k_cvt_u32, "v_cvt_f32_u32 %0, %1", )
KERNEL1(k_max3, "v_max3_f32 %0, %0, |%1|, |%2|", )
KERNEL1(k_min3, "v_min3_f32 %0, %0, %1, %2", )
KERNEL1(k_alignbit, "v_alignbit_b32 %0, %1, %0, 1", )
KERNEL1(k_add_sdwa, "v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1", )
KERNEL1(k_fmamk, "v_fmamk_f32 %0, %1, 0x3e4ccccd, %0", )
KERNEL1(k_lshl_or, "v_lshl_or_b32 %0, %0, 1, %1", )
KERNEL1(k_and_or, "v_and_or_b32 %0, %1, %2, %0", )
KERNEL1(k_mul_lo, "v_mul_lo_u32 %0, %1, %0", )
KERNEL1(k_mad_u64, "v_mad_u32_u24 %0, %1, %2, %0", )

typedef float v2f __attribute__((ext_vector_type(2)));
#define KERNEL2(NAME, ASMSTR)                                                                    \
    __global__ void NAME(float *out, float seed) {                                              \
        v2f a[16]; v2f b = {seed, seed * 2}, c = {seed * 0.5f, seed * 0.25f};                    \
        for (int i = 0; i < 16; i++) { a[i].x = seed + i; a[i].y = seed - i; }                   \
        for (int it = 0; it < ITER; it++) {                                                      \
            _Pragma("unroll") for (int i = 0; i < 16; i++)                                       \
                asm volatile(ASMSTR : "+v"(a[i]) : "v"(b), "v"(c));                              \
        }                                                                                        \
        float s = 0; for (int i = 0; i < 16; i++) s += a[i].x + a[i].y;                          \
        if (s == 12345.678f) out[0] = s;                                                         \
    }
KERNEL2(k_pk_fma, "v_pk_fma_f32 %0, %1, %2, %0")
KERNEL2(k_pk_fma_mod, "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]")
KERNEL2(k_pk_add, "v_pk_add_f32 %0, %1, %0")
KERNEL2(k_pk_add_neg, "v_pk_add_f32 %0, %1, %0 neg_lo:[0,1] neg_hi:[0,1]")
KERNEL2(k_pk_mul, "v_pk_mul_f32 %0, %1, %0")

// packed fma with an SGPR-pair constant
__global__ void k_pk_fma_sgpr(float *out, float seed, float c0, float c1) {
    v2f a[16]; v2f b = {seed, seed * 2};
    for (int i = 0; i < 16; i++) { a[i].x = seed + i; a[i].y = seed - i; }
    v2f cc = {c0, c1};
    for (int it = 0; it < ITER; it++) {
        _Pragma("unroll") for (int i = 0; i < 16; i++)
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "s"(cc), "v"(b));
    }
    float s = 0; for (int i = 0; i < 16; i++) s += a[i].x + a[i].y;
    if (s == 12345.678f) out[0] = s;
}

__global__ void k_fma_f64(float *out, float seed) {
    double a[16]; double b = seed, c = seed * 0.5;
    for (int i = 0; i < 16; i++) a[i] = seed + i;
    for (int it = 0; it < ITER; it++) {
        _Pragma("unroll") for (int i = 0; i < 16; i++)
            asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
    }
    double s = 0; for (int i = 0; i < 16; i++) s += a[i];
    if (s == 12345.678) out[0] = (float)s;
}

template <class K, class... A>
void run(const char *name, K kern, int waves_per_simd, int ncu, double flops_per_inst, A... args) {
    float *out; CHK(hipMalloc(&out, 4));
    int blocks = ncu * waves_per_simd;  // 256 threads = 4 waves = 1 wave per SIMD per block
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, args...);
    CHK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, args...);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double winst = (double)blocks * 4 * ITER * 16;           // wave-instructions
    double per_cu_per_us = winst / ncu / (best * 1e3);       // wave-inst per CU per microsecond
    printf("%-14s w/SIMD=%d  %8.3f ms  %7.1f winst/CU/us  = %5.2f winst/clk/CU @2.4GHz  (%6.1f Tlane-op/s)\n", name,
           waves_per_simd, best, per_cu_per_us, per_cu_per_us / 2400.0, winst * 64 / (best * 1e-3) / 1e12);
    CHK(hipFree(out));
}

int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    int ncu = p.multiProcessorCount;
    printf("device %s CUs %d clock %d kHz\n", p.name, ncu, p.clockRate);
    for (int w : {1, 2, 4, 8}) {
        run("v_fma_f32", k_fma, w, ncu, 2, 1.0f);
        run("v_fmac_f32", k_fmac, w, ncu, 2, 1.0f);
        run("v_pk_fma_f32", k_pk_fma, w, ncu, 4, 1.0f);
        run("v_pk_fma_mod", k_pk_fma_mod, w, ncu, 4, 1.0f);
        run("v_pk_fma_sgpr", k_pk_fma_sgpr, w, ncu, 4, 1.0f, 0.5f, 0.25f);
        run("v_pk_add_f32", k_pk_add, w, ncu, 2, 1.0f);
        run("v_pk_add_neg", k_pk_add_neg, w, ncu, 2, 1.0f);
        run("v_pk_mul_f32", k_pk_mul, w, ncu, 2, 1.0f);
        run("v_add_f32", k_add, w, ncu, 1, 1.0f);
        run("v_fmamk_f32", k_fmamk, w, ncu, 2, 1.0f);
        run("cvt_f32_ubyte0", k_cvt_ub0, w, ncu, 1, 1.0f);
        run("cvt_f32_ubyte3", k_cvt_ub3, w, ncu, 1, 1.0f);
        run("cvt_f32_u32", k_cvt_u32, w, ncu, 1, 1.0f);
        run("v_max3_f32", k_max3, w, ncu, 1, 1.0f);
        run("v_min3_f32", k_min3, w, ncu, 1, 1.0f);
        run("v_alignbit", k_alignbit, w, ncu, 1, 1.0f);
        run("v_add_u32_sdwa", k_add_sdwa, w, ncu, 1, 1.0f);
        run("v_lshl_or", k_lshl_or, w, ncu, 1, 1.0f);
        run("v_and_or", k_and_or, w, ncu, 1, 1.0f);
        run("v_mul_lo_u32", k_mul_lo, w, ncu, 1, 1.0f);
        run("v_mad_u32_u24", k_mad_u64, w, ncu, 1, 1.0f);
        run("v_fma_f64", k_fma_f64, w, ncu, 2, 1.0f);
        printf("\n");
    }
    return 0;
}
