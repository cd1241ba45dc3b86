// Is v_mfma_f32_32x32x1_2b_f32 (K = 1: one product per output per instruction) bit-identical to fmaf(a, b, c), including
// denormal inputs / results?  If so a chain of such MFMAs reproduces the spec's serial f32 fma chains exactly.
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off scripts/check_mfma_f32_k1.hip -o /tmp/chk && /tmp/chk
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef float f32x32 __attribute__((ext_vector_type(32)));
__global__ void k(const float* a, const float* b, float* d, int steps) { // a,b: [steps][64]; d: [32 regs][64 lanes]
    const int lane = threadIdx.x;
    f32x32 acc;
    for (int i = 0; i < 32; i++) acc[i] = 0.0f;
    for (int s = 0; s < steps; s++) acc = __builtin_amdgcn_mfma_f32_32x32x1f32(a[s * 64 + lane], b[s * 64 + lane], acc, 0, 0, 0);
    for (int i = 0; i < 32; i++) d[i * 64 + lane] = acc[i];
}
static uint32_t rng_state = 12345u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state; }
static float rand_float(int mode) {
    uint32_t u = rnd();
    float f = ((int32_t)u) * (1.0f / 2147483648.0f);            // [-1, 1)
    if (mode == 1) f = ldexpf(f, -70 - (int)(rnd() % 10));        // tiny: products and sums land in the denormal range
    if (mode == 2) { uint32_t m = rnd() & 0x807FFFFFu; memcpy(&f, &m, 4); } // denormal input
    if (mode == 3) f = ldexpf(f, (int)(rnd() % 40) - 20);         // wide dynamic range: cancellation
    return f;
}
int main() {
    const int steps = 8, trials = 200;
    float *da, *db, *dd;
    hipMalloc(&da, steps * 64 * 4); hipMalloc(&db, steps * 64 * 4); hipMalloc(&dd, 32 * 64 * 4);
    long bad = 0, total = 0, denorm_results = 0;
    for (int t = 0; t < trials; t++) {
        const int mode = t % 4;
        std::vector<float> a(steps * 64), b(steps * 64), d(32 * 64);
        for (auto& v : a) v = rand_float(mode == 2 ? (rnd() & 1 ? 2 : 0) : mode);
        for (auto& v : b) v = rand_float(mode == 1 ? 1 : mode == 3 ? 3 : 0);
        hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd, steps);
        hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost);
        for (int v = 0; v < 32; v++) for (int lane = 0; lane < 64; lane++) {
            const int blk = v / 16, vv = v % 16;
            const int i = 8 * (vv / 4) + 4 * (lane / 32) + vv % 4, j = lane % 32; // assumed output map: block = v/16
            float c = 0.0f;
            for (int s = 0; s < steps; s++) c = fmaf(a[s * 64 + blk * 32 + i], b[s * 64 + blk * 32 + j], c);
            uint32_t u1, u2; memcpy(&u1, &c, 4); float g = d[v * 64 + lane]; memcpy(&u2, &g, 4);
            total++;
            if (c != 0.0f && fabsf(c) < 1.17549435e-38f) denorm_results++;
            if (u1 != u2) { if (bad < 10) printf("mode %d v %d lane %d: host %.9g (%08x) mfma %.9g (%08x)\n", mode, v, lane, c, u1, g, u2); bad++; }
        }
    }
    printf("%ld / %ld mismatches (%ld denormal results checked)\n", bad, total, denorm_results);
    return bad != 0;
}
