// Do v_pk_mul_f32 / v_pk_fma_f32 treat f32 denormals like the scalar instructions on gfx950?  (the batched int8 GEMM runs its scale
// chain on packed pairs; bit-identity with the scalar GEMV path requires the same denormal behaviour)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* a, const float* b, const float* c, float* out_s, float* out_p, int n) {
    const int i = threadIdx.x;
    if (2 * i + 1 >= n) return;
    out_s[2 * i] = __builtin_fmaf(a[2 * i], b[2 * i], c[2 * i]);
    out_s[2 * i + 1] = __builtin_fmaf(a[2 * i + 1], b[2 * i + 1], c[2 * i + 1]);
    const f2 av = {a[2 * i], a[2 * i + 1]}, bv = {b[2 * i], b[2 * i + 1]}, cv = {c[2 * i], c[2 * i + 1]};
    const f2 m = av * bv;                                 // v_pk_mul_f32
    const f2 r = __builtin_elementwise_fma(av, bv, cv);   // v_pk_fma_f32
    out_p[2 * i] = r[0]; out_p[2 * i + 1] = r[1];
    out_s[n + 2 * i] = a[2 * i] * b[2 * i]; out_s[n + 2 * i + 1] = a[2 * i + 1] * b[2 * i + 1];
    out_p[n + 2 * i] = m[0]; out_p[n + 2 * i + 1] = m[1];
}
int main() {
    const int n = 8;
    // denormal inputs, denormal products, denormal sums
    float a[n] = {1e-40f, 3e-39f, 1e-20f, 1e-30f, 5e-39f, 1.0f, 2e-38f, 1e-45f};
    float b[n] = {1.0f, 0.5f, 1e-20f, 1e-10f, 2.0f, 1e-40f, 0.25f, 1.0f};
    float c[n] = {0.0f, 1e-39f, 0.0f, 1e-41f, -4e-39f, 1e-40f, 0.0f, 1e-45f};
    float *da, *db, *dc, *ds, *dp, hs[2 * n], hp[2 * n];
    hipMalloc(&da, sizeof(a)); hipMalloc(&db, sizeof(b)); hipMalloc(&dc, sizeof(c)); hipMalloc(&ds, sizeof(hs)); hipMalloc(&dp, sizeof(hp));
    hipMemcpy(da, a, sizeof(a), hipMemcpyHostToDevice); hipMemcpy(db, b, sizeof(b), hipMemcpyHostToDevice); hipMemcpy(dc, c, sizeof(c), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, ds, dp, n);
    hipMemcpy(hs, ds, sizeof(hs), hipMemcpyDeviceToHost); hipMemcpy(hp, dp, sizeof(hp), hipMemcpyDeviceToHost);
    int diff = 0;
    for (int i = 0; i < 2 * n; i++) {
        unsigned us, up; memcpy(&us, &hs[i], 4); memcpy(&up, &hp[i], 4);
        printf("%s[%d] scalar %08x (%g) packed %08x (%g)%s\n", i < n ? "fma" : "mul", i % n, us, hs[i], up, hp[i], us == up ? "" : "  <-- differs");
        diff += us != up;
    }
    printf("%d differences\n", diff);
    return 0;
}
