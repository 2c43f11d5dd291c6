// Read-bandwidth ceilings on MI355X for a 2.2 GB buffer (design input for k_demod_bits).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// (a) plain coalesced 16-B loads, grid-stride, U loads in flight per lane
template <int U>
__global__ __launch_bounds__(256) void k_plain(const uint4 *in, size_t n16, uint32_t *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = in[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// (b) each wave streams 4 KiB tiles through LDS-DMA (like k_demod_bits), T tiles in flight per wave
template <int T>
__global__ __launch_bounds__(256) void k_glds(const uint8_t *in, size_t ntiles, uint32_t *out) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[4][T][4096];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * 4;
    uint32_t acc = 0;
    size_t tile = (size_t)blockIdx.x * 4 + wave;
    // prologue: T-1 tiles in flight
#pragma unroll
    for (int t = 0; t < T - 1; t++) {
        const size_t tt = tile + t * nwaves;
        if (tt < ntiles)
#pragma unroll
            for (int j = 0; j < 4; j++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(in + tt * 4096 + j * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void *)(&lds[wave][t][j * 1024]), 16, 0, 0);
    }
    int slot = 0;
    for (; tile < ntiles; tile += nwaves) {
        const size_t tt = tile + (T - 1) * nwaves;
        const int ns = (slot + T - 1) % T;
        if (tt < ntiles) {
#pragma unroll
            for (int j = 0; j < 4; j++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(in + tt * 4096 + j * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void *)(&lds[wave][ns][j * 1024]), 16, 0, 0);
            if (T == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (T == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (T == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const uint4 *p = (const uint4 *)(&lds[wave][slot][64 * lane]);
        uint4 a, b, c, d;
        asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n s_waitcnt lgkmcnt(0)"
                     : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"((uint32_t)(uintptr_t)p) : "memory");
        acc ^= a.x ^ b.y ^ c.z ^ d.w;
        slot = (slot + 1) % T;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

static size_t ntiles_for_bits(size_t bytes) { return bytes / 4096 * 64 * 4; }

// (c) the k_demod_bits loop order: wait -> LDS reads -> issue next tile into the SAME buffer -> store a word
template <int STORE, int HALO, int WORK>
__global__ __launch_bounds__(256) void k_like(const uint8_t *in, size_t ntiles, uint32_t *out, uint32_t *bits) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[4][32 + 4096];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * 4;
    uint8_t *my = lds[wave];
    size_t tile = (size_t)blockIdx.x * 4 + wave;
    auto issue = [&](size_t tt) {
#pragma unroll
        for (int j = 0; j < 4; j++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(in + tt * 4096 + j * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void *)(my + 32 + j * 1024), 16, 0, 0);
        if (HALO && lane < 2)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(in + tt * 4096 + (tt ? -32 : 0) + lane * 16),
                                             (__attribute__((address_space(3))) void *)(my), 16, 0, 0);
    };
    if (tile < ntiles) issue(tile);
    uint32_t acc = 0;
    while (tile < ntiles) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint4 *p = (const uint4 *)(my + 64 * lane);
        uint4 v[6];
#pragma unroll
        for (int j = 0; j < 6; j++) v[j] = p[j];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const size_t next = tile + nwaves;
        if (next < ntiles) issue(next);
        uint32_t x = 0;
#pragma unroll
        for (int j = 0; j < 6; j++) x ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
        if (WORK) {  // dependent fma chain standing in for the arithmetic
            float f = __uint_as_float(x & 0x3fffffff);
#pragma unroll 16
            for (int k = 0; k < WORK; k++) f = __builtin_fmaf(f, 1.0001f, 0.5f);
            x = __float_as_uint(f);
        }
        if (STORE) bits[tile * 64 + lane] = x; else acc ^= x;
        tile = next;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <class F>
void timeit(const char *name, size_t bytes, F launch) {
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    launch(); CHK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 8; r++) {
        CHK(hipEventRecord(e0)); launch(); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-28s %8.3f ms  %7.2f TB/s\n", name, best, bytes / (best * 1e-3) / 1e12);
}

int main() {
    const size_t bytes = 4096ull * 270336 * 2;  // the bench input
    uint8_t *in; uint32_t *out;
    CHK(hipMalloc(&in, bytes + 65536)); CHK(hipMalloc(&out, 4));
    CHK(hipMemset(in, 0x5a, bytes + 65536));
    const size_t n16 = bytes / 16, ntiles = bytes / 4096;
    uint32_t *bits; CHK(hipMalloc(&bits, ntiles_for_bits(bytes)));
    for (int wgpc : {4, 7}) {
        const int grid = 256 * wgpc;
        printf("-- k_like, %d workgroups per CU\n", wgpc);
        timeit("like: no store, no halo", bytes, [&] { hipLaunchKernelGGL((k_like<0,0,0>), dim3(grid), dim3(256), 0, 0, in, bytes / 4096, out, bits); });
        timeit("like: store", bytes, [&] { hipLaunchKernelGGL((k_like<1,0,0>), dim3(grid), dim3(256), 0, 0, in, bytes / 4096, out, bits); });
        timeit("like: store + halo", bytes, [&] { hipLaunchKernelGGL((k_like<1,1,0>), dim3(grid), dim3(256), 0, 0, in, bytes / 4096, out, bits); });
        timeit("like: store+halo+256 fma", bytes, [&] { hipLaunchKernelGGL((k_like<1,1,256>), dim3(grid), dim3(256), 0, 0, in, bytes / 4096, out, bits); });
        timeit("like: store+halo+1024 fma", bytes, [&] { hipLaunchKernelGGL((k_like<1,1,1024>), dim3(grid), dim3(256), 0, 0, in, bytes / 4096, out, bits); });
    }
    for (int wgpc : {4, 8, 16}) {
        const int grid = 256 * wgpc;
        printf("-- %d workgroups per CU\n", wgpc);
        timeit("plain dwordx4 U=1", bytes, [&] { hipLaunchKernelGGL(k_plain<1>, dim3(grid), dim3(256), 0, 0, (const uint4 *)in, n16, out); });
        timeit("plain dwordx4 U=4", bytes, [&] { hipLaunchKernelGGL(k_plain<4>, dim3(grid), dim3(256), 0, 0, (const uint4 *)in, n16, out); });
        timeit("plain dwordx4 U=8", bytes, [&] { hipLaunchKernelGGL(k_plain<8>, dim3(grid), dim3(256), 0, 0, (const uint4 *)in, n16, out); });
        if (wgpc <= 8) {
            timeit("glds 4KiB tiles T=1", bytes, [&] { hipLaunchKernelGGL(k_glds<1>, dim3(grid), dim3(256), 0, 0, in, ntiles, out); });
            if (wgpc <= 4) {
                timeit("glds 4KiB tiles T=2", bytes, [&] { hipLaunchKernelGGL(k_glds<2>, dim3(grid), dim3(256), 0, 0, in, ntiles, out); });
                timeit("glds 4KiB tiles T=3", bytes, [&] { hipLaunchKernelGGL(k_glds<3>, dim3(256 * 3), dim3(256), 0, 0, in, ntiles, out); });
            }
        }
    }
    return 0;
}
