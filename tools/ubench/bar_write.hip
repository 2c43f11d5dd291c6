// Can the host write a block straight into device memory (large BAR), and how fast?  For the streaming handle: the kernel's
// own reads of pinned host memory run at ~13 GB/s (128 KB = 10 us); a block pushed into HBM by the host would be read at HBM
// speed.  Runs each step in a child process: a fault in the host's store must not take the parent down.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/wait.h>
#include <unistd.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
__global__ void k_sum(const uint32_t *p, size_t n, uint32_t *out) {
    uint32_t s = 0;
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) s += p[i];
    atomicAdd(out, s);
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static int trial(int mode) {
    const size_t bytes = 128 * 1024, n = bytes / 4;
    void *dev = nullptr;
    if (mode == 0) CHK(hipMalloc(&dev, bytes));
    else if (mode == 1) CHK(hipExtMallocWithFlags(&dev, bytes, hipDeviceMallocFinegrained));
    else CHK(hipExtMallocWithFlags(&dev, bytes, hipDeviceMallocUncached));
    uint32_t *out; CHK(hipMalloc(&out, 4)); CHK(hipMemset(out, 0, 4)); CHK(hipDeviceSynchronize());
    uint32_t *src = (uint32_t *)aligned_alloc(64, bytes);
    uint32_t want = 0;
    for (size_t i = 0; i < n; i++) { src[i] = (uint32_t)(i * 2654435761u); want += src[i]; }
    printf("mode %d: device pointer %p - host memcpy into it ...\n", mode, dev); fflush(stdout);
    memcpy(dev, src, bytes);   // (faults here when the memory is not host-visible)
    double best = 1e9;
    for (int r = 0; r < 50; r++) {
        for (size_t i = 0; i < n; i++) src[i] += 1;
        const double t0 = now();
        memcpy(dev, src, bytes);
        __builtin_ia32_sfence();
        const double t1 = now();
        if (t1 - t0 < best) best = t1 - t0;
    }
    want = 0; for (size_t i = 0; i < n; i++) want += src[i];
    hipLaunchKernelGGL(k_sum, dim3(1), dim3(256), 0, 0, (const uint32_t *)dev, n, out);
    uint32_t got = 0; CHK(hipMemcpy(&got, out, 4, hipMemcpyDeviceToHost));
    printf("mode %d: host memcpy of 128 KB into device memory: best %.2f us = %.1f GB/s; kernel saw %s data\n", mode, best * 1e6, bytes / best / 1e9,
           got == want ? "the right" : "WRONG");
    fflush(stdout);
    return got == want ? 0 : 1;
}
int main() {
    for (int mode = 0; mode < 3; mode++) {
        fflush(stdout);
        pid_t pid = fork();   // (before any HIP call in this process)
        if (pid == 0) { _exit(trial(mode)); }
        int st = 0; waitpid(pid, &st, 0);
        if (WIFSIGNALED(st)) printf("mode %d: the host store faulted (signal %d): not host-visible\n", mode, WTERMSIG(st));
    }
    return 0;
}
