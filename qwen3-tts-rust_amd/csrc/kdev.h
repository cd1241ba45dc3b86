// kdev.h -- device-side helpers shared by the .hip translation units
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/q3tts_spec.h"

namespace q3 {

__device__ __forceinline__ float h2f(uint32_t h) { return (float)__builtin_bit_cast(_Float16, (uint16_t)h); }
// f32 -> f16, round-to-nearest-even of the *f32 value*.  The empty asm makes the operand opaque: without it the backend folds
// `f2h(fma(a, b, c))` into v_fma_mixlo_f16, which rounds the exact a*b+c once to f16 and differs from the spec's
// (and the oracle's) two-step rounding on f16 ties -- seen as a one-ulp K difference in ~1/8000 elements.
__device__ __forceinline__ uint16_t f2h(float f) { asm volatile("" : "+v"(f)); return __builtin_bit_cast(uint16_t, (_Float16)f); }
// Exact xor-lane exchanges without LDS traffic (HIP's __shfl_xor lowers to ds_bpermute_b32 + address arithmetic, ~100 cycles of latency
// per step of a dependent butterfly): DPP quad permutes for 1/2, DPP row shifts + select for 4/8, v_permlane16_swap / v_permlane32_swap
// (gfx950) for 16/32.  Same pairing as __shfl_xor for every lane (scripts/check_xor_shuffles.hip), so the spec's butterflies keep their bits.
typedef unsigned q3_u2v __attribute__((ext_vector_type(2)));
template <int CTRL> __device__ __forceinline__ int dpp_mov_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
template <int S> __device__ __forceinline__ int xor_lane(int v) {
    static_assert(S == 1 || S == 2 || S == 4 || S == 8 || S == 16 || S == 32, "xor_lane: power of two below 64");
    const int lane = threadIdx.x & 63;
    if (S == 1) return dpp_mov_i<0xB1>(v);                  // quad_perm [1,0,3,2]
    if (S == 2) return dpp_mov_i<0x4E>(v);                  // quad_perm [2,3,0,1]
    if (S == 4) { const int a = dpp_mov_i<0x104>(v), b = dpp_mov_i<0x114>(v); return (lane & 4) ? b : a; }   // row_shl:4 / row_shr:4
    if (S == 8) { const int a = dpp_mov_i<0x108>(v), b = dpp_mov_i<0x118>(v); return (lane & 8) ? b : a; }   // row_shl:8 / row_shr:8
    if (S == 16) { const q3_u2v r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false); return (int)((lane & 16) ? r[0] : r[1]); }
    const q3_u2v r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return (int)((lane & 32) ? r[0] : r[1]);
}
template <int S> __device__ __forceinline__ float xor_lane(float v) { return __int_as_float(xor_lane<S>(__float_as_int(v))); }
__device__ __forceinline__ float wave_sum_bfly(float v) { // spec butterfly: xor 32,16,8,4,2,1
    v = v + xor_lane<32>(v); v = v + xor_lane<16>(v); v = v + xor_lane<8>(v); v = v + xor_lane<4>(v); v = v + xor_lane<2>(v); v = v + xor_lane<1>(v);
    return v;
}
__device__ __forceinline__ float wave_max_bfly(float v) {
    v = fmaxf(v, xor_lane<32>(v)); v = fmaxf(v, xor_lane<16>(v)); v = fmaxf(v, xor_lane<8>(v)); v = fmaxf(v, xor_lane<4>(v));
    v = fmaxf(v, xor_lane<2>(v)); v = fmaxf(v, xor_lane<1>(v));
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
