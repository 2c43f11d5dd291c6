// Round 2: more opcodes, and whether 2-cycle (fp32 fma/add/mul) and 4-cycle ops overlap.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define ITER 4096
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define KERNEL1(NAME, ASMSTR)                                                                    \
    __global__ void NAME(float *out, float seed) {                                              \
        float a[16]; float b = seed, c = seed * 0.5f;                                            \
        for (int i = 0; i < 16; i++) a[i] = seed + i;                                            \
        for (int it = 0; it < ITER; it++) {                                                      \
            _Pragma("unroll") for (int i = 0; i < 16; i++)                                       \
                asm volatile(ASMSTR : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");                      \
        }                                                                                        \
        float s = 0; for (int i = 0; i < 16; i++) s += a[i];                                     \
        if (s == 12345.678f) out[0] = s;                                                         \
    }
KERNEL1(k_max, "v_max_f32 %0, %1, %0")
KERNEL1(k_max_abs, "v_max_f32 %0, |%1|, %0")
KERNEL1(k_min_abs, "v_min_f32 %0, |%1|, %0")
KERNEL1(k_mul, "v_mul_f32 %0, %1, %0")
KERNEL1(k_mul_clamp, "v_mul_f32 %0, %1, %0 clamp")
KERNEL1(k_sub, "v_sub_f32 %0, %1, %0")
KERNEL1(k_add_abs, "v_add_f32 %0, |%1|, |%0|")
KERNEL1(k_and, "v_and_b32 %0, %1, %0")
KERNEL1(k_or, "v_or_b32 %0, %1, %0")
KERNEL1(k_xor, "v_xor_b32 %0, %1, %0")
KERNEL1(k_lshl, "v_lshlrev_b32 %0, 1, %0")
KERNEL1(k_lshr, "v_lshrrev_b32 %0, 1, %0")
KERNEL1(k_addu, "v_add_u32 %0, %1, %0")
KERNEL1(k_cndmask, "v_cndmask_b32 %0, %1, %0, vcc")
KERNEL1(k_mov, "v_mov_b32 %0, %1")
KERNEL1(k_bfe, "v_bfe_u32 %0, %1, 8, 8")
KERNEL1(k_perm, "v_perm_b32 %0, %1, %0, %2")
KERNEL1(k_cvt_i32, "v_cvt_f32_i32 %0, %1")
KERNEL1(k_cvt_sdwa, "v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1")
KERNEL1(k_cmp, "v_cmp_lt_f32 vcc, %1, %0")
KERNEL1(k_add_dpp, "v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL1(k_fma_lit, "v_fmaak_f32 %0, %1, %0, 0x3e4ccccd")
KERNEL1(k_mad_i24, "v_mad_i32_i24 %0, %1, %2, %0")
KERNEL1(k_add3, "v_add3_u32 %0, %1, %2, %0")
KERNEL1(k_dot2, "v_dot2_f32_f16 %0, %1, %2, %0")
KERNEL1(k_dot4, "v_dot4_i32_i8 %0, %1, %2, %0")
KERNEL1(k_pk_fma16, "v_pk_fma_f16 %0, %1, %2, %0")
KERNEL1(k_mad_mix, "v_fma_mix_f32 %0, %1, %2, %0")
KERNEL1(k_rcp, "v_rcp_f32 %0, %0")
KERNEL1(k_addc, "v_addc_co_u32 %0, vcc, %0, %0, vcc")

// mixes: 8 x A + 8 x B per iteration (16 instructions, as above)
#define KMIX(NAME, ASMA, ASMB)                                                                   \
    __global__ void NAME(float *out, float seed) {                                              \
        float a[16]; float b = seed, c = seed * 0.5f;                                            \
        for (int i = 0; i < 16; i++) a[i] = seed + i;                                            \
        for (int it = 0; it < ITER; it++) {                                                      \
            _Pragma("unroll") for (int i = 0; i < 8; i++) {                                      \
                asm volatile(ASMA : "+v"(a[2 * i]) : "v"(b), "v"(c) : "vcc");                    \
                asm volatile(ASMB : "+v"(a[2 * i + 1]) : "v"(b), "v"(c) : "vcc");                \
            }                                                                                    \
        }                                                                                        \
        float s = 0; for (int i = 0; i < 16; i++) s += a[i];                                     \
        if (s == 12345.678f) out[0] = s;                                                         \
    }
KMIX(k_mix_fma_cvt, "v_fma_f32 %0, %1, %2, %0", "v_cvt_f32_ubyte0 %0, %1")
KMIX(k_mix_fma_align, "v_fma_f32 %0, %1, %2, %0", "v_alignbit_b32 %0, %1, %0, 1")
KMIX(k_mix_fma_max3, "v_fma_f32 %0, %1, %2, %0", "v_max3_f32 %0, %0, |%1|, |%2|")
KMIX(k_mix_fma_pk, "v_fma_f32 %0, %1, %2, %0", "v_lshl_or_b32 %0, %0, 1, %1")
KMIX(k_mix_fma_f64, "v_fma_f32 %0, %1, %2, %0", "v_mul_lo_u32 %0, %1, %0")

template <class K>
void run(const char *name, K kern, int w, int ncu) {
    float *out; CHK(hipMalloc(&out, 4));
    int blocks = ncu * w;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    CHK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double winst = (double)blocks * 4 * ITER * 16;
    double per = winst / ncu / (best * 1e3);
    printf("%-16s w/SIMD=%d %8.3f ms  %5.2f winst/clk/CU @2.4GHz  -> %4.2f cyc/inst/SIMD @2.4, %4.2f @1.95\n", name, w, best,
           per / 2400.0, 4.0 / (per / 2400.0), 4.0 / (per / 1950.0));
    CHK(hipFree(out));
}

int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    int ncu = p.multiProcessorCount;
    for (int w : {2, 4}) {
#define R(n, k) run(n, k, w, ncu)
        R("v_max_f32", k_max); R("v_max_f32 |x|", k_max_abs); R("v_min_f32 |x|", k_min_abs); R("v_mul_f32", k_mul);
        R("v_mul_f32 clamp", k_mul_clamp); R("v_sub_f32", k_sub); R("v_add_f32 abs", k_add_abs);
        R("v_and_b32", k_and); R("v_or_b32", k_or); R("v_xor_b32", k_xor); R("v_lshlrev_b32", k_lshl);
        R("v_lshrrev_b32", k_lshr); R("v_add_u32", k_addu); R("v_cndmask_b32", k_cndmask); R("v_mov_b32", k_mov);
        R("v_bfe_u32", k_bfe); R("v_perm_b32", k_perm); R("v_cvt_f32_i32", k_cvt_i32); R("cvt_f32_i32_sdwa", k_cvt_sdwa);
        R("v_cmp_lt_f32", k_cmp); R("v_add_f32_dpp", k_add_dpp); R("v_fma_f32 lit", k_fma_lit);
        R("v_mad_i32_i24", k_mad_i24); R("v_add3_u32", k_add3); R("v_dot2_f32_f16", k_dot2); R("v_dot4_i32_i8", k_dot4);
        R("v_pk_fma_f16", k_pk_fma16); R("v_fma_mix_f32", k_mad_mix); R("v_rcp_f32", k_rcp); R("v_addc_co_u32", k_addc);
        R("mix fma+cvt", k_mix_fma_cvt); R("mix fma+alignbit", k_mix_fma_align); R("mix fma+max3", k_mix_fma_max3);
        R("mix fma+lshl_or", k_mix_fma_pk); R("mix fma+mul_lo", k_mix_fma_f64);
        printf("\n");
    }
    return 0;
}
