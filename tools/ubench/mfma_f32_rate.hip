// Issue rate of v_mfma_f32_32x32x2_f32 (and 16x16x4) with 2 or 4 independent accumulators per wave,
// 1..4 waves per SIMD: cycles per instruction and SIMD at an assumed 2.0 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define ITER 4096
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k32(float *out, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++) for (int e = 0; e < 16; e++) acc[i][e] = 0.0f;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < NACC; i++) for (int e = 0; e < 16; e++) s += acc[i][e];
    if (s == 12345.678f) out[0] = s;
}
template <int NACC>
__global__ void k16(float *out, float a, float b) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; i++) for (int e = 0; e < 4; e++) acc[i][e] = 0.0f;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < NACC; i++) for (int e = 0; e < 4; e++) s += acc[i][e];
    if (s == 12345.678f) out[0] = s;
}

template <class K>
void run(const char *name, K kern, int w, int ncu, int nacc, double flop_per_instr) {
    float *out; CHK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(kern, dim3(ncu * w), dim3(256), 0, 0, out, 1.0f, 0.5f);
    CHK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(ncu * w), dim3(256), 0, 0, out, 1.0f, 0.5f);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double per_simd = (double)w * ITER * nacc;  // instructions per SIMD
    double tf = per_simd * 4.0 * ncu * flop_per_instr / (best * 1e-3) / 1e12;
    printf("%-22s acc/wave=%d waves/SIMD=%d %8.3f ms  %6.1f cycles@2GHz per MFMA per SIMD  %6.1f TFLOP/s\n", name, nacc, w, best,
           best * 1e-3 * 2.0e9 / per_simd, tf);
    CHK(hipFree(out));
}

int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    int ncu = p.multiProcessorCount;
    for (int w : {1, 2, 4}) {
        run("mfma_f32_32x32x2f32", k32<2>, w, ncu, 2, 4096);
        run("mfma_f32_32x32x2f32", k32<4>, w, ncu, 4, 4096);
        run("mfma_f32_16x16x4f32", k16<4>, w, ncu, 4, 2048);
        run("mfma_f32_16x16x4f32", k16<8>, w, ncu, 8, 2048);
    }
    return 0;
}
