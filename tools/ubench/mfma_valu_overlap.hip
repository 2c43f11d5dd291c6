// Does vector work of one wave issue while another wave of the same SIMD executes MFMAs?  (design input
// for k_demod_mfma.)  A 1024-thread workgroup = 16 waves = 4 per SIMD (wave i -> SIMD i % 4).  Roles by
// wave index / 4: MFMA-only, VALU-only, or every wave alternating bursts of 6 MFMAs and NV vector ops.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// NV vector ops on 8 independent chains (no dependency stalls); CLS 0: v_fma_f32, 1: v_max3_f32 (4-cycle class)
template <int NV, int CLS>
__device__ __forceinline__ float valu_block(float x, float y) {
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = x + j;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        if (CLS == 0) r[i & 7] = __builtin_fmaf(r[i & 7], y, 1.0f);
        else asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[i & 7]) : "v"(y), "v"(x));
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) s += r[j];
    return s;
}

// MODE 0: all waves alternate [6 MFMA][NV VALU] (dependent on the MFMA result, like the kernel)
// MODE 1: waves with (w / 4) < M do MFMA only, the others VALU only
// MODE 2: as 0 but the VALU work does not depend on the MFMA results
template <int MODE, int NV, int CLS, int ACC = 0>
__global__ __launch_bounds__(1024, 1) void k(float *out, int iters, int M) {
    const int w = threadIdx.x >> 6;
    h8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(threadIdx.x & 7); b[i] = (_Float16)((threadIdx.x >> 3) & 7); }
    f16v c0 = {0}, c1 = {0};
    float x = threadIdx.x * 1e-3f, y = 0.999f;
    for (int it = 0; it < iters; it++) {
        const bool do_m = MODE != 1 || (w >> 2) < M;
        const bool do_v = MODE != 1 || (w >> 2) >= M;
        if (do_m) {
            __builtin_amdgcn_sched_barrier(0);
            if (ACC) {  // accumulators in the ACC half of the register file
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b));
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c1) : "v"(b), "v"(a));
                }
            } else {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (do_v) {
            if (MODE == 0 && !ACC) x += c0[0] * 1e-30f;
            x = valu_block<NV, CLS>(x, y) * 1e-3f;
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = x + c0[3] + c1[5];
}

template <int MODE, int NV, int CLS = 0, int ACC = 0>
static void run(const char *name, int M, float *d) {
    const int iters = 2000, grid = 256;  // one workgroup per CU
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<MODE, NV, CLS, ACC>), dim3(grid), dim3(1024), 0, 0, d, iters, M);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE, NV, CLS, ACC>), dim3(grid), dim3(1024), 0, 0, d, iters, M);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    // per SIMD and iteration: cycles at 2 GHz
    printf("%-44s acc=%d cls=%d NV=%3d M=%d  %.3f ms  %.0f cycles@2GHz per iteration per SIMD\n", name, ACC, CLS, NV, M, ms, ms * 1e-3 * 2e9 / iters);
}

int main() {
    float *d; CHK(hipMalloc(&d, 256 * 1024 * 4));
    run<1, 85>("all 4 waves MFMA only (6 per iteration)", 4, d);
    run<1, 85>("all 4 waves VALU only", 0, d);
    run<1, 85>("2 waves MFMA only + 2 waves VALU only", 2, d);
    run<1, 85>("1 wave MFMA only + 3 waves VALU only", 1, d);
    run<1, 85>("3 waves MFMA only + 1 wave VALU only", 3, d);
    run<0, 85>("4 waves alternate burst / dependent VALU", 4, d);
    run<2, 85>("4 waves alternate burst / independent VALU", 4, d);
    run<0, 40>("4 waves alternate burst / dependent VALU", 4, d);
    run<0, 170>("4 waves alternate burst / dependent VALU", 4, d);
    run<1, 85, 1>("all 4 waves VALU only", 0, d);
    run<1, 85, 1>("2 waves MFMA only + 2 waves VALU only", 2, d);
    run<0, 85, 1>("4 waves alternate burst / dependent VALU", 4, d);
    run<2, 85, 1>("4 waves alternate burst / independent VALU", 4, d);
    run<1, 85, 0, 1>("all 4 waves MFMA only (6 per iteration)", 4, d);
    run<1, 85, 0, 1>("2 waves MFMA only + 2 waves VALU only", 2, d);
    run<1, 85, 0, 1>("1 wave MFMA only + 3 waves VALU only", 1, d);
    run<2, 85, 0, 1>("4 waves alternate burst / independent VALU", 4, d);
    run<2, 170, 0, 1>("4 waves alternate burst / independent VALU", 4, d);
    run<2, 170, 0, 0>("4 waves alternate burst / independent VALU", 4, d);
    return 0;
}
