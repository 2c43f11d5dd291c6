// Issue rate of v_smfmac_f32_32x32x32_f16 (2:4 sparse A, K = 32) against v_mfma_f32_32x32x16_f16 (K = 16):
// is a sparse step as cheap as a dense one?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16 __attribute__((ext_vector_type(16)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int SPARSE>
__global__ void k(float *out, int iters) {
    h8 a; h16 b;
    for (int i = 0; i < 8; i++) a[i] = (_Float16)(threadIdx.x & 7);
    for (int i = 0; i < 16; i++) b[i] = (_Float16)((threadIdx.x >> 3) & 7);
    h8 b8; for (int i = 0; i < 8; i++) b8[i] = b[i];
    f16v c0 = {0}, c1 = {0};
    const int idx = 0x4444 * (threadIdx.x & 1) + 0x9999;
    for (int it = 0; it < iters; it++) {
        if (SPARSE) {
            c0 = __builtin_amdgcn_smfmac_f32_32x32x32_f16(a, b, c0, idx, 0, 0);
            c1 = __builtin_amdgcn_smfmac_f32_32x32x32_f16(a, b, c1, idx, 0, 0);
        } else {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b8, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b8, c1, 0, 0, 0);
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = c0[3] + c1[5];
}

template <int SPARSE>
static void run(const char *name, float *d) {
    const int iters = 20000;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<SPARSE>, dim3(1024), dim3(64), 0, 0, d, iters);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<SPARSE>, dim3(1024), dim3(64), 0, 0, d, iters);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-40s %.3f ms for %d x 2 per wave, 1 wave per SIMD -> %.1f cycles@2GHz each\n", name, ms, iters, ms * 1e-3 * 2e9 / (2.0 * iters));
}

int main() {
    float *d; CHK(hipMalloc(&d, 1024 * 64 * 4));
    run<0>("v_mfma_f32_32x32x16_f16 (dense, K=16)", d);
    run<1>("v_smfmac_f32_32x32x32_f16 (sparse, K=32)", d);
    return 0;
}
