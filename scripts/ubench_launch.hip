// micro-benchmark: cost of a dependent kernel boundary (eager vs hipGraph) and the shader clock under this load
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_trivial(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void k_small(const float* a, float* b, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) b[i] = a[i] * 2.0f + 1.0f; }
__global__ void k_clock(unsigned long long* out, int iters) {
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float x = threadIdx.x;
    for (int i = 0; i < iters; i++) x = __builtin_fmaf(x, 1.0001f, 0.5f);
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = (unsigned long long)x; }
}
int main() {
    int* d; float *a, *b; unsigned long long* dc;
    CK(hipMalloc(&d, 4)); CK(hipMemset(d, 0, 4)); CK(hipMalloc(&a, 1 << 20)); CK(hipMalloc(&b, 1 << 20)); CK(hipMalloc(&dc, 64));
    hipStream_t st; CK(hipStreamCreate(&st));
    const int N = 1000;
    for (int variant = 0; variant < 3; variant++) {
        auto launch = [&]() {
            if (variant == 0) hipLaunchKernelGGL(k_trivial, dim3(1), dim3(64), 0, st, d);
            else if (variant == 1) hipLaunchKernelGGL(k_trivial, dim3(256), dim3(256), 0, st, d);
            else hipLaunchKernelGGL(k_small, dim3(1024), dim3(256), 0, st, a, b, 1 << 18);
        };
        for (int i = 0; i < 100; i++) launch();
        CK(hipStreamSynchronize(st));
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; i++) launch();
        CK(hipStreamSynchronize(st));
        double eager = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; i++) launch();
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 5; r++) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        double graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (5.0 * N);
        printf("variant %d: eager %.2f us/kernel, graph %.2f us/kernel\n", variant, eager, graph);
    }
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, st, dc, 200000);
        CK(hipStreamSynchronize(st));
        unsigned long long h[3]; CK(hipMemcpy(h, dc, 24, hipMemcpyDeviceToHost));
        printf("clock: %llu shader cycles / %llu ref ticks(100MHz) => %.0f MHz\n", h[0], h[1], (double)h[0] / (double)h[1] * 100.0);
    }
    return 0;
}
