// Round 4: can the FIR pair sums be done as packed 16-bit integer adds on [0 Q 0 I] words whose
// halves are then read as f16 DENORMALS (value k * 2^-24, exact) by v_fma_mix_f32?
//   - issue rate of v_fma_mix_f32 (f16 operand), v_xad_u32, v_perm_b32, v_add_u32
//   - a per-sample instruction mix like the demod kernel's today vs the proposed one
//   - numerical check: fma_mix on denormal halves gives exactly fmaf(k * 2^-24, c, acc)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#define ITER 4096
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// MODE 0: fma_f32   1: fma_mix (f16 lo)   2: xad_u32   3: perm_b32   4: add_u32   5: fma_mix (f16 hi)
template <int MODE>
__global__ void k_rate(float *out, float seed) {
    float a[16]; float b = seed, c = seed * 0.5f;
    uint32_t u[16]; uint32_t ub = (uint32_t)seed, uc = 0x00FF00FFu;
    for (int i = 0; i < 16; i++) { a[i] = seed + i; u[i] = i; }
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            if (MODE == 1) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(b), "v"(c));
            if (MODE == 5) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(b), "v"(c));
            if (MODE == 2) asm volatile("v_xad_u32 %0, %1, %2, %0" : "+v"(u[i]) : "v"(ub), "v"(uc));
            if (MODE == 3) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(u[i]) : "v"(ub), "v"(uc));
            if (MODE == 4) asm volatile("v_add_u32 %0, %1, %0" : "+v"(u[i]) : "v"(ub));
        }
    }
    float s = 0; for (int i = 0; i < 16; i++) s += a[i] + (float)u[i];
    if (s == 12345.678f) out[0] = s;
}

// Per-sample mixes, two samples per body (so the half-rate min3 is a whole instruction).
// NEW = 0: today's kernel  (2 cvt, 8 add_f32, 10 fma, max3, mul+fma, 0.5 min3, alignbit)  = 24.5
// NEW = 1: proposed        (1 perm, 2 add_u32, 2 xad, 10 fma_mix, max3, mul+fma, 0.5 min3, alignbit) = 19.5
template <int NEW>
__global__ void k_sample_mix(float *out, float seed) {
    float w[8], acc[4], s[4]; uint32_t d[8], word = 0, raw = (uint32_t)seed * 0x01010101u;
    float fm = 0, nm = 1e30f, c = seed * 0.25f;
    for (int i = 0; i < 8; i++) { w[i] = seed + i; d[i] = i * 0x00010001u; }
    for (int i = 0; i < 4; i++) { acc[i] = seed; s[i] = seed; }
    uint32_t t[4];
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int smp = 0; smp < 2; smp++) {
            if (NEW) {
                asm volatile("v_perm_b32 %0, %1, %1, %2" : "=v"(d[smp]) : "v"(raw), "s"(0x0C010C00u));
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(t[0]) : "v"(d[0]), "v"(d[1]));
                asm volatile("v_xad_u32 %0, %1, %2, %3" : "=v"(t[1]) : "v"(d[2]), "s"(0x00FF00FFu), "v"(d[3]));
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(t[2]) : "v"(d[4]), "v"(d[5]));
                asm volatile("v_xad_u32 %0, %1, %2, %3" : "=v"(t[3]) : "v"(d[6]), "s"(0x00FF00FFu), "v"(d[7]));
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc[0]) : "v"(t[k & 3]), "v"(c));
                    asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc[1]) : "v"(t[k & 3]), "v"(c));
                }
            } else {
                asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(w[smp]) : "v"(raw));
                asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(w[smp + 2]) : "v"(raw));
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    asm volatile("v_add_f32 %0, %1, %2" : "=v"(s[k]) : "v"(w[k]), "v"(w[k + 4]));
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(t[k]) : "v"(w[k + 1]), "v"(w[(k + 5) & 7]));
                }
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(s[k & 3]), "v"(c));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[1]) : "v"(t[k & 3]), "v"(c));
                }
            }
            asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(fm) : "v"(acc[0]), "v"(acc[1]));
            float num;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(num) : "v"(acc[0]), "v"(acc[3]));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(num) : "v"(acc[1]), "v"(acc[2]));
            asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(word) : "v"(num));
            if (smp) asm volatile("v_min3_f32 %0, %0, |%1|, |%2|" : "+v"(nm) : "v"(num), "v"(acc[2]));
            acc[2] = acc[0]; acc[3] = acc[1];
        }
    }
    float r = fm + nm + (float)word;
    for (int i = 0; i < 4; i++) r += acc[i];
    if (r == 12345.678f) out[0] = r;
}

template <class K>
float run(const char *name, K kern, int w, int ncu, double n_instr) {
    float *out; CHK(hipMalloc(&out, 4));
    int blocks = ncu * w;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    CHK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double per_simd = (double)w * ITER * n_instr;
    printf("%-28s w/SIMD=%d %7.3f ms  %5.2f ns*2GHz-cycles/instr\n", name, w, best, best * 1e-3 * 2.0e9 / per_simd);
    CHK(hipFree(out));
    return best;
}

__global__ void k_check(const uint32_t *in, float *out, float c, float acc0) {
    const uint32_t d = in[threadIdx.x];
    float lo = acc0, hi = acc0;
    asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(d), "v"(c));
    asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(hi) : "v"(d), "v"(c));
    out[2 * threadIdx.x] = lo; out[2 * threadIdx.x + 1] = hi;
}

int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    int ncu = p.multiProcessorCount;
    // numerical check: halves 0..1023 as f16 denormals
    {
        uint32_t h[1024]; for (int i = 0; i < 1024; i++) h[i] = (uint32_t)i | ((uint32_t)(1023 - i) << 16);
        uint32_t *din; float *dout; CHK(hipMalloc(&din, sizeof h)); CHK(hipMalloc(&dout, 8192));
        CHK(hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice));
        const float c = 0.228626345955f * 16777216.0f, acc0 = -2.4391f;
        hipLaunchKernelGGL(k_check, dim3(1), dim3(1024), 0, 0, din, dout, c, acc0);
        float o[2048]; CHK(hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < 1024; i++) {
            const float el = fmaf((float)i * 5.9604644775390625e-8f, c, acc0), eh = fmaf((float)(1023 - i) * 5.9604644775390625e-8f, c, acc0);
            if (o[2 * i] != el || o[2 * i + 1] != eh) { if (bad < 5) printf("mismatch i=%d got %g %g want %g %g\n", i, o[2 * i], o[2 * i + 1], el, eh); bad++; }
        }
        printf("fma_mix on f16 denormal halves: %s (%d mismatches of 1024)\n", bad ? "WRONG" : "exact", bad);
    }
    for (int w : {4, 7}) {
        run("fma_f32", k_rate<0>, w, ncu, 16);
        run("fma_mix_f32 (f16 lo)", k_rate<1>, w, ncu, 16);
        run("fma_mix_f32 (f16 hi)", k_rate<5>, w, ncu, 16);
        run("xad_u32", k_rate<2>, w, ncu, 16);
        run("perm_b32", k_rate<3>, w, ncu, 16);
        run("add_u32", k_rate<4>, w, ncu, 16);
        float t0 = run("sample mix today (24.5/smp)", k_sample_mix<0>, w, ncu, 49);
        float t1 = run("sample mix proposed (19.5)", k_sample_mix<1>, w, ncu, 39);
        printf("  proposed / today time = %.3f (instruction ratio %.3f)\n\n", t1 / t0, 39.0 / 49.0);
    }
    return 0;
}
