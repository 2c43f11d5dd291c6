// Round 3: does the slow (4-cycle) pipe overlap with the fast (2-cycle) pipe for mixes like the
// real kernel's (3:1, 7:1), with independent and with dependent chains?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define ITER 4096
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// NB = number of B ops per 16 instructions, placed evenly; DEP = 1: all A ops chain through 2 accumulators
template <int NB, int DEP>
__global__ void k_mix(float *out, float seed) {
    float a[16]; float b = seed, c = seed * 0.5f;
    for (int i = 0; i < 16; i++) a[i] = seed + i;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const bool isB = NB > 0 && (i % (16 / (NB > 0 ? NB : 1))) == 0;
            if (isB) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "+v"(a[i]) : "v"(b));
            else if (DEP) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i & 1]) : "v"(b), "v"(c));
            else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
        }
    }
    float s = 0; for (int i = 0; i < 16; i++) s += a[i];
    if (s == 12345.678f) out[0] = s;
}

template <class K>
void run(const char *name, K kern, int w, int ncu, int nA, int nB) {
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
    // SIMD-cycles per group of 16 instructions at an assumed 2.0 GHz
    double groups_per_simd = (double)w * ITER;
    double cyc = best * 1e-3 * 2.0e9 / groups_per_simd;
    printf("%-22s w/SIMD=%d %7.3f ms  %6.1f cyc/16-instr @2.0GHz   (serial %d, overlapped %d)\n", name, w, best, cyc,
           nA * 2 + nB * 4, (nA * 2 > nB * 4 ? nA * 2 : nB * 4));
    CHK(hipFree(out));
}

int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    int ncu = p.multiProcessorCount;
    for (int w : {4, 7}) {
        run("16A indep", k_mix<0, 0>, w, ncu, 16, 0);
        run("16A dep(2 chains)", k_mix<0, 1>, w, ncu, 16, 0);
        run("14A+2B indep", k_mix<2, 0>, w, ncu, 14, 2);
        run("12A+4B indep", k_mix<4, 0>, w, ncu, 12, 4);
        run("8A+8B indep", k_mix<8, 0>, w, ncu, 8, 8);
        run("14A+2B dep", k_mix<2, 1>, w, ncu, 14, 2);
        run("12A+4B dep", k_mix<4, 1>, w, ncu, 12, 4);
        run("8A+8B dep", k_mix<8, 1>, w, ncu, 8, 8);
        printf("\n");
    }
    return 0;
}
