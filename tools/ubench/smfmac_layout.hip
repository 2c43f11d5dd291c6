// Operand layouts of v_smfmac_f32_32x32x32_f16 (2:4-sparse A, K = 32) on gfx950, found by trying hypotheses against the
// instruction itself: random small integers, D compared with the dense product under each (A layout, B layout) pair.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16 __attribute__((ext_vector_type(16)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void k(const _Float16 *a, const _Float16 *b, const int *idx, float *d) {
    const int l = threadIdx.x;
    h8 av; h16 bv;
    for (int i = 0; i < 8; i++) av[i] = a[l * 8 + i];
    for (int i = 0; i < 16; i++) bv[i] = b[l * 16 + i];
    f16v c = {0};
    c = __builtin_amdgcn_smfmac_f32_32x32x32_f16(av, bv, c, idx[l], 0, 0);
    for (int i = 0; i < 16; i++) d[l * 16 + i] = c[i];
}

int main() {
    std::vector<_Float16> a(64 * 8), b(64 * 16);
    std::vector<int> idx(64);
    std::vector<float> d(64 * 16);
    srand(7);
    // positions of the two kept elements of each group of four: first < second
    const int pairs[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};
    for (int l = 0; l < 64; l++) {
        int w = 0;
        for (int g = 0; g < 4; g++) {
            const int *p = pairs[rand() % 6];
            w |= (p[0] | (p[1] << 2)) << (4 * g);
        }
        idx[l] = w;   // (upper 16 bits zero)
        for (int i = 0; i < 8; i++) a[l * 8 + i] = (_Float16)(float)(rand() % 7 - 3);
        for (int i = 0; i < 16; i++) b[l * 16 + i] = (_Float16)(float)(rand() % 9 - 4);
    }
    _Float16 *da, *db; int *di; float *dd;
    CHK(hipMalloc(&da, a.size() * 2)); CHK(hipMalloc(&db, b.size() * 2)); CHK(hipMalloc(&di, 64 * 4)); CHK(hipMalloc(&dd, d.size() * 4));
    CHK(hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice)); CHK(hipMemcpy(db, b.data(), b.size() * 2, hipMemcpyHostToDevice));
    CHK(hipMemcpy(di, idx.data(), 64 * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, di, dd);
    CHK(hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost));
    int found = 0;
    for (int ha = 0; ha < 2; ha++) for (int hb = 0; hb < 2; hb++) for (int hi = 0; hi < 2; hi++) {
        // dense A [32][32], B [32][32]
        float A[32][32] = {}, B[32][32] = {};
        for (int l = 0; l < 64; l++) {
            const int m = l % 32, h = l / 32;
            for (int i = 0; i < 8; i++) {
                const int c = ha == 0 ? 8 * h + i : (i < 4 ? 4 * h + i : 8 + 4 * h + (i - 4));   // compressed slot 0..15
                const int g = c / 2;                                                               // group of four K
                const int bits = hi == 0 ? (idx[l] >> (2 * i)) & 3 : (idx[l] >> (2 * (i ^ 1))) & 3;
                A[m][4 * g + bits] += (float)a[l * 8 + i];
            }
            for (int i = 0; i < 16; i++) {
                const int kk = hb == 0 ? 16 * h + i : (i < 8 ? 8 * h + i : 16 + 8 * h + (i - 8));
                B[kk][m] = (float)b[l * 16 + i];
            }
        }
        int bad = 0;
        for (int l = 0; l < 64; l++) for (int i = 0; i < 16; i++) {
            const int n = l % 32, m = 8 * (i / 4) + 4 * (l / 32) + (i % 4);
            float s = 0; for (int kk = 0; kk < 32; kk++) s += A[m][kk] * B[kk][n];
            if (s != d[l * 16 + i]) bad++;
        }
        printf("A layout %d (0: slot = 8h+i, 1: halves), B layout %d (0: k = 16h+i, 1: k = 8h+i | 16+8h+i-8), idx order %d: %s (%d of 1024 differ)\n",
               ha, hb, hi, bad ? "no" : "MATCH", bad);
        found += !bad;
    }
    if (!found) {
        printf("no hypothesis matched; lane 0: idx %04x a:", idx[0]);
        for (int i = 0; i < 8; i++) printf(" %g", (float)a[i]);
        printf("\n d[lane 0]:"); for (int i = 0; i < 16; i++) printf(" %g", d[i]);
        printf("\n");
    }
    return 0;
}
