// Operand-layout probe for v_mfma_f32_32x32x16_f16 on gfx950 (used by the codec's split-f16 GEMM).
// Build + run on the GPU box: hipcc -O2 --offload-arch=gfx950 scripts/check_mfma_f16.hip -o /tmp/chk && /tmp/chk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
// hyp 0: lane l holds k = 8*(l/32) + i ; hyp 1: k = 4*(l/32) + (i&3) + 8*(i>>2)
__global__ void k(const float* A, const float* B, float* C, int hyp) { // A [32][16], B [16][32] row-major
    const int l = threadIdx.x, r = l & 31, g = l >> 5;
    h8 a, b;
    for (int i = 0; i < 8; i++) {
        const int kk = hyp == 0 ? 8 * g + i : 4 * g + (i & 3) + 8 * (i >> 2);
        a[i] = (_Float16)A[r * 16 + kk];
        b[i] = (_Float16)B[kk * 32 + r];
    }
    f16v c;
    for (int i = 0; i < 16; i++) c[i] = 0;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 16; i++) C[((i & 3) + 8 * (i >> 2) + 4 * g) * 32 + r] = c[i]; // row = (i&3)+8(i>>2)+4g, col = lane&31
}
int main() {
    std::vector<float> A(32 * 16), B(16 * 32), C(32 * 32), R(32 * 32, 0.f);
    srand(3);
    for (auto& v : A) v = (float)(rand() % 17 - 8);
    for (auto& v : B) v = (float)(rand() % 13 - 6);
    for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) for (int kk = 0; kk < 16; kk++) R[i * 32 + j] += A[i * 16 + kk] * B[kk * 32 + j];
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    for (int hyp = 0; hyp < 2; hyp++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, hyp);
        hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        double e = 0;
        for (size_t i = 0; i < C.size(); i++) e = e > fabs(C[i] - R[i]) ? e : fabs(C[i] - R[i]);
        printf("mfma_f32_32x32x16_f16 layout hypothesis %d: max |err| = %g\n", hyp, e);
    }
    return 0;
}
