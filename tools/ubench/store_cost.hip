// What the demod kernel's 138 MB of packed bits cost beside its 2.2 GB of reads, by store pattern (design input for
// k_demod_mfma's word stage).  A wave streams chunks of 16 consecutive 4 KiB tiles through LDS-DMA (one tile in flight, as
// the product kernel) and writes 256 B of "bits" per tile:
//   none      no stores (the read floor)
//   t1        256 B per tile, a dword per lane (the product's pattern without its staging)
//   t4        1 KiB every 4th tile, 16 B per lane (the product's word stage)
//   t16       4 KiB at the end of the chunk, four 16-byte stores per lane
//   t16s      the same, but every wave of the GPU writes at (roughly) the same time: chunks start together
//   *_nt      with the non-temporal hint;  *_small: into a 4 MiB region (stays in the L2: no HBM write traffic at all)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE, int NT>
__global__ __launch_bounds__(256) void k_stream(const uint8_t *in, size_t nchunks, uint32_t *out, uint8_t *bits, size_t bits_mask) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[4][4096];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * 4;
    uint8_t *my = lds[wave];
    auto issue = [&](size_t tt) {
#pragma unroll
        for (int j = 0; j < 4; j++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(in + tt * 4096 + j * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void *)(my + j * 1024), 16, 0, 2);
    };
    auto st4 = [&](uint8_t *p, uint32_t v) { if (NT) __builtin_nontemporal_store(v, (uint32_t *)p); else *(uint32_t *)p = v; };
    auto st16 = [&](uint8_t *p, uint4 v) {
        if (NT) { __builtin_nontemporal_store(v.x, (uint32_t *)p); __builtin_nontemporal_store(v.y, (uint32_t *)p + 1);
                  __builtin_nontemporal_store(v.z, (uint32_t *)p + 2); __builtin_nontemporal_store(v.w, (uint32_t *)p + 3); }
        else *(uint4 *)p = v;
    };
    uint32_t acc = 0;
    for (size_t chunk = (size_t)blockIdx.x * 4 + wave; chunk < nchunks; chunk += nwaves) {
        const size_t t0 = chunk * 16;
        issue(t0);
        uint4 keep[4] = {};
        for (int t = 0; t < 16; t++) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint4 *p = (const uint4 *)(my + 64 * lane);
            uint4 v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = p[j];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (t + 1 < 16) issue(t0 + t + 1);
            uint32_t x = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) x ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
            uint8_t *b = bits + (((t0 + t) * 256) & bits_mask);
            if (MODE == 1) st4(b + 4 * lane, x);
            else if (MODE == 4) {
                keep[0].x ^= x; keep[0].y += x; keep[0].z ^= x >> 3; keep[0].w += x >> 5;
                if ((t & 3) == 3) st16(bits + (((t0 + t - 3) * 256) & bits_mask) + 16 * lane, keep[0]);
            } else if (MODE == 16) {
                keep[t >> 2].x ^= x; keep[t >> 2].y += x; keep[t >> 2].z ^= x >> 3; keep[t >> 2].w += x >> 5;
            } else acc ^= x;
        }
        if (MODE == 16) {
            uint8_t *b = bits + ((t0 * 256) & bits_mask);
#pragma unroll
            for (int j = 0; j < 4; j++) st16(b + 1024 * j + 16 * lane, keep[j]);
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <class F>
void timeit(const char *name, size_t bytes, F launch) {
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    launch(); CHK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0;
    for (int r = 0; r < 10; r++) {
        CHK(hipEventRecord(e0)); launch(); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; sum += ms;
    }
    printf("%-22s best %7.3f ms  mean %7.3f ms  %6.2f TB/s of input\n", name, best, sum / 10, bytes / (best * 1e-3) / 1e12); fflush(stdout);
}

int main() {
    const size_t bytes = 4096ull * 270336 * 2;  // the bench input
    uint8_t *in; uint32_t *out; uint8_t *bits;
    CHK(hipMalloc(&in, bytes + 65536)); CHK(hipMalloc(&out, 4));
    CHK(hipMemset(in, 0x5a, bytes + 65536));
    const size_t ntiles = bytes / 4096, nchunks = ntiles / 16, bbytes = ntiles * 256;
    CHK(hipMalloc(&bits, bbytes + 65536));
    const size_t full = ~(size_t)0, small = (4u << 20) - 1;
    const int grid = 1024;
#define RUN(name, MODE, NT, mask) timeit(name, bytes, [&] { hipLaunchKernelGGL((k_stream<MODE, NT>), dim3(grid), dim3(256), 0, 0, in, nchunks, out, bits, mask); })
    for (int rep = 0; rep < 2; rep++) {
        RUN("none", 0, 0, full);
        RUN("t1", 1, 0, full);
        RUN("t1_nt", 1, 1, full);
        RUN("t4", 4, 0, full);
        RUN("t4_nt", 4, 1, full);
        RUN("t16", 16, 0, full);
        RUN("t16_nt", 16, 1, full);
        RUN("t1_small", 1, 0, small);
        RUN("t16_small", 16, 0, small);
        printf("\n");
    }
    return 0;
}
