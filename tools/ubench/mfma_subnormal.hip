// Does v_mfma_f32_32x32x16_f16 take f16 SUBNORMAL B inputs at face value (no flush), exactly, and at the
// usual rate?  B = raw bytes k read as f16 bit patterns (k * 2^-24), A = integer taps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__global__ void k(float *out, int sub, int iters) {
    const int lane = threadIdx.x;
    u4 braw;
    for (int i = 0; i < 4; i++) {
        const uint32_t k0 = (lane * 7 + i * 31 + 5) & 255, k1 = (lane * 13 + i * 17 + 250) & 255;
        braw[i] = sub ? (k0 | (k1 << 16)) : ((0x6400u | k0) | ((0x6400u | k1) << 16));  // k * 2^-24  or  1024 + k
    }
    const h8 b = __builtin_bit_cast(h8, braw);
    h8 a;
    for (int j = 0; j < 8; j++) a[j] = (_Float16)(float)(((lane * 3 + j * 5) % 41) * 47 - 900);  // |a| <= 1027, integers
    f16v c = {0};
    for (int it = 0; it < iters; it++) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; r++) out[lane * 16 + r] = c[r];
}

int main() {
    float *d; CHK(hipMalloc(&d, 64 * 16 * 4));
    float h[2][1024];
    for (int sub = 0; sub < 2; sub++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, sub, 1);
        CHK(hipMemcpy(h[sub], d, sizeof h[sub], hipMemcpyDeviceToHost));
    }
    // host reference
    int bad[2] = {0, 0};
    for (int lane = 0; lane < 64; lane++)
        for (int r = 0; r < 16; r++) {
            const int col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            for (int sub = 0; sub < 2; sub++) {
                double acc = 0;
                for (int kk = 0; kk < 16; kk++) {
                    const int la = row + 32 * (kk >> 3), ja = kk & 7;          // A[row][k]: lane la, element ja
                    const int lb = col + 32 * (kk >> 3), jb = kk & 7;          // B[k][col]
                    const double av = ((la * 3 + ja * 5) % 41) * 47 - 900;
                    const int i = jb >> 1;
                    const unsigned kv = (jb & 1) ? ((lb * 13 + i * 17 + 250) & 255) : ((lb * 7 + i * 31 + 5) & 255);
                    acc += av * (sub ? kv / 16777216.0 : 1024.0 + kv);
                }
                if ((double)h[sub][lane * 16 + r] != acc) bad[sub]++;
            }
        }
    printf("normal inputs (1024 + k): %d of 1024 outputs differ from the exact sum\n", bad[0]);
    printf("subnormal inputs (k * 2^-24): %d of 1024 outputs differ from the exact sum (sample out[5] = %g)\n", bad[1], h[1][5]);
    // rate
    for (int sub = 0; sub < 2; sub++) {
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k, dim3(1024), dim3(64), 0, 0, d, sub, 20000);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(1024), dim3(64), 0, 0, d, sub, 20000);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s inputs: %.3f ms for 20000 dependent MFMAs per wave, 1 wave per SIMD -> %.1f cycles@2GHz each\n",
               sub ? "subnormal" : "normal", ms, ms * 1e-3 * 2e9 / 20000);
    }
    return 0;
}
