// kdev.h -- device-side helpers shared by the .hip translation units
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/q3tts_spec.h"

namespace q3 {

__device__ __forceinline__ float h2f(uint32_t h) { return (float)__builtin_bit_cast(_Float16, (uint16_t)h); }
__device__ __forceinline__ uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
__device__ __forceinline__ float wave_sum_bfly(float v) { // spec butterfly: xor 32,16,8,4,2,1
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = v + __shfl_xor(v, s);
    return v;
}
__device__ __forceinline__ float wave_max_bfly(float v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = fmaxf(v, __shfl_xor(v, s));
    return v;
}
__device__ __forceinline__ int dot16(const uint4& a, const uint4& b) {
    int s = __builtin_amdgcn_sdot4((int)a.x, (int)b.x, 0, false);
    s = __builtin_amdgcn_sdot4((int)a.y, (int)b.y, s, false);
    s = __builtin_amdgcn_sdot4((int)a.z, (int)b.z, s, false);
    s = __builtin_amdgcn_sdot4((int)a.w, (int)b.w, s, false);
    return s;
}
__device__ __forceinline__ uint32_t half_of(const uint4& v, int b) { // b-th f16 of 8 packed halfs (b constant)
    uint32_t w = (b >> 1) == 0 ? v.x : (b >> 1) == 1 ? v.y : (b >> 1) == 2 ? v.z : v.w;
    return (b & 1) ? (w >> 16) : (w & 0xFFFFu);
}


typedef unsigned long long u64;
// argmax keys: high word = order-preserving image of the float, low word = ~index, so that an unsigned max
// picks the largest value and, among equal values, the SMALLEST index (first max, llama/mod.rs:690-701)
__host__ __device__ __forceinline__ u64 pack_key(float v, int idx) {
    uint32_t b = q3_f32_bits(v);
    if (b == 0x80000000u) b = 0u; // -0.0 compares equal to +0.0 in the reference's `>`
    uint32_t ord = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return ((u64)ord << 32) | (uint32_t)(~(uint32_t)idx);
}
__host__ __device__ __forceinline__ int key_code(u64 k) { return (int)(~(uint32_t)(k & 0xFFFFFFFFull)); }

} // namespace q3
