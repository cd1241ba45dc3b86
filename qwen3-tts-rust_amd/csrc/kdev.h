// kdev.h -- device-side helpers shared by the .hip translation units
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/q3tts_spec.h"

namespace q3 {

// In-kernel phase stamps for scripts/ubench_chain.hip (built with -DQ3_STAMPS; the library build compiles them to nothing): thread 0 of every
// workgroup stores the 100 MHz s_memrealtime counter at up to 8 points of the kernel, so a phase can be placed on one time axis across workgroups.
#ifdef Q3_STAMPS
static __device__ unsigned long long* g_q3_stamps = nullptr; // one per translation unit, set through the unit's Q3_STAMP_SETTER function
#define Q3_STAMP_SETTER(name) void name(hipStream_t st, unsigned long long* p) { (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_q3_stamps), &p, sizeof(p), 0, hipMemcpyHostToDevice, st); (void)hipStreamSynchronize(st); }
// stamps live in SGPRs until the kernel's last instruction block (no store, no branch inside the phases); `dep` ties the read to a value the phase produced
#define Q3_STAMP_DECL unsigned long long q3_st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long q3_c0_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(q3_c0_) :: "memory")
#define Q3_STAMP(k) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(q3_st_[k]) :: "memory")
#define Q3_STAMP_AFTER(k, dep) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(q3_st_[k]), "+v"(dep) :: "memory")
#define Q3_STAMP_FLUSH() do { unsigned long long q3_c1_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(q3_c1_) :: "memory"); q3_st_[7] = q3_c1_ - q3_c0_; if (threadIdx.x == 0 && g_q3_stamps) for (int k_ = 0; k_ < 8; k_++) g_q3_stamps[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + k_] = q3_st_[k_]; } while (0)
#else
#define Q3_STAMP_SETTER(name)
#define Q3_STAMP_DECL
#define Q3_STAMP(k) do {} while (0)
#define Q3_STAMP_AFTER(k, dep) do {} while (0)
#define Q3_STAMP_FLUSH() do {} while (0)
#endif

// Workgroup barrier for LDS hand-offs that leaves the wave's GLOBAL loads and stores in flight.  __syncthreads() on gfx950 puts s_waitcnt vmcnt(0) in
// front of s_barrier (the target has no automatic wait), i.e. every barrier also drains the weight stream and waits for the epilogue's stores to land.
// Here only the LDS traffic is waited for, which is all an LDS producer -> consumer hand-off needs.
__device__ __forceinline__ void wg_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
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
