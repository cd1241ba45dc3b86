// Micro-benchmark: cost of a grid-wide barrier (global atomic counter) on MI355X, vs. a kernel boundary in a hipGraph.
// Build: hipcc -O3 --offload-arch=gfx950 scripts/ubench_barrier.hip -o /tmp/ubench_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* bar, unsigned target, unsigned* timeout_flag) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 20000000u) { *timeout_flag = 1; break; } // exit condition every wave reaches
        }
    }
    __syncthreads();
}

// payload: each WG reads `bytes_per_wg` of a buffer written in the previous phase by another WG (cross-XCD traffic)
__global__ void k_barriers(unsigned* bar, unsigned* timeout_flag, int n_iter, float* buf, int words_per_wg) {
    const unsigned G = gridDim.x;
    float acc = 0.f;
    for (int it = 0; it < n_iter; it++) {
        if (words_per_wg > 0) {
            const int src = (blockIdx.x + 37 * (it + 1)) % G;
            for (int i = threadIdx.x; i < words_per_wg; i += blockDim.x)
                acc += __builtin_nontemporal_load(buf + (size_t)src * words_per_wg + i);
            for (int i = threadIdx.x; i < words_per_wg; i += blockDim.x)
                buf[(size_t)blockIdx.x * words_per_wg + i] = acc + it;
            __threadfence();
        }
        grid_barrier(bar, (unsigned)(it + 1) * G, timeout_flag);
    }
    if (acc == 12345.678f) buf[0] = acc;
}
__global__ void k_empty(float* p) { if (p == nullptr) p[0] = 1; }

int main() {
    unsigned *bar, *flag; float* buf;
    CK(hipMalloc(&bar, 4)); CK(hipMalloc(&flag, 4)); CK(hipMalloc(&buf, 512 * 4096 * 4));
    CK(hipMemset(buf, 0, 512 * 4096 * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int threads : {256, 512}) for (int G : {128, 256, 512}) for (int words : {0, 1024}) {
        const int n_iter = 2000;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipMemsetAsync(bar, 0, 4, st)); CK(hipMemsetAsync(flag, 0, 4, st));
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(k_barriers, dim3(G), dim3(threads), 0, st, bar, flag, n_iter, buf, words);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned f; CK(hipMemcpy(&f, flag, 4, hipMemcpyDeviceToHost));
            if (rep == 1) printf("grid_barrier threads=%d G=%d words/wg=%d : %.3f us per barrier%s\n", threads, G, words, 1e3 * ms / n_iter, f ? "  (TIMEOUT!)" : "");
        }
    }
    // reference: empty dependent kernels in a graph
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 1000; i++) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, buf);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("graph node (empty kernel, 256 WG): %.3f us per node\n", ms);
    return 0;
}
