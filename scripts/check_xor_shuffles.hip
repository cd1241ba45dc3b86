// Probe: exact xor-lane exchanges without LDS traffic on gfx950 (DPP quad_perm / row shifts, v_permlane16_swap, v_permlane32_swap)
// against __shfl_xor (ds_bpermute).  The arithmetic spec's butterflies need the SAME pairing, so every lane is compared.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2v __attribute__((ext_vector_type(2)));
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true)); }
__device__ __forceinline__ float x1(float v) { return dpp_mov<0xB1>(v); }
__device__ __forceinline__ float x2(float v) { return dpp_mov<0x4E>(v); }
__device__ __forceinline__ float x4(float v, int lane) { const float a = dpp_mov<0x104>(v), b = dpp_mov<0x114>(v); return (lane & 4) ? b : a; }
__device__ __forceinline__ float x8(float v, int lane) { const float a = dpp_mov<0x108>(v), b = dpp_mov<0x118>(v); return (lane & 8) ? b : a; }
__device__ __forceinline__ float x16(float v, int lane, int variant) {
    const u2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(((lane & 16) != 0) == (variant == 0) ? r[0] : r[1]);
}
__device__ __forceinline__ float x32(float v, int lane, int variant) {
    const u2v r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(((lane & 32) != 0) == (variant == 0) ? r[0] : r[1]);
}
__global__ void k(float* out) { // out[(test)*64 + lane]
    const int lane = threadIdx.x;
    const float v = (float)(lane * 3 + 1);
    out[0 * 64 + lane] = x1(v) - __shfl_xor(v, 1);
    out[1 * 64 + lane] = x2(v) - __shfl_xor(v, 2);
    out[2 * 64 + lane] = x4(v, lane) - __shfl_xor(v, 4);
    out[3 * 64 + lane] = x8(v, lane) - __shfl_xor(v, 8);
    out[4 * 64 + lane] = x16(v, lane, 0) - __shfl_xor(v, 16);
    out[5 * 64 + lane] = x16(v, lane, 1) - __shfl_xor(v, 16);
    out[6 * 64 + lane] = x32(v, lane, 0) - __shfl_xor(v, 32);
    out[7 * 64 + lane] = x32(v, lane, 1) - __shfl_xor(v, 32);
}
int main() {
    float* d; float h[8 * 64];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[8] = {"xor1 quad_perm", "xor2 quad_perm", "xor4 row_shl/shr 4", "xor8 row_shl/shr 8", "xor16 permlane16_swap v0", "xor16 permlane16_swap v1",
                            "xor32 permlane32_swap v0", "xor32 permlane32_swap v1"};
    for (int t = 0; t < 8; t++) { int bad = 0; for (int l = 0; l < 64; l++) bad += h[t * 64 + l] != 0.0f; printf("%-28s: %d mismatching lanes\n", names[t], bad); }
    return 0;
}
