// wslice.h -- one lane's slice of a weight matrix for the GEMV family: the quants of (row, 256-segment) this lane multiplies, fetched from
// the Q8_0 tiles or from the packed K-quant planes (kernels.h), and the block chain of spec S3 that turns them into a partial sum.
// Shared by k_gemv_kq, k_gemv_q8_norm, and k_gateup_swiglu, so the fused decode path serves Q5_K_M files with the same
// launches as Q8_0 files.  The weight type is a TEMPLATE parameter: a kernel reads the type of its row group from the kernel arguments (a
// scalar) and enters the body compiled for it -- with the type as a run-time value inside one body the compiler merges the three load
// sequences with selects on loaded data, which parks the wave on the weight stream before the norm prologue (measured: 16 -> 20 us).
//   WT = 0        all-Q8_0 matrix, uniform 1-KiB tiles (the Q8_0 files; no row-group map)
//   WT = Q8_0     Q8_0 row group inside a K-quant matrix      acc = fma(f(isum), dw*dx, acc)
//   WT = Q5_K     acc = fma(d*f(sc*isum) - dmin*f(m*xsum), dx, acc)        xsum = sum of the activation block
//   WT = Q6_K     acc = fma(d*f(sc0*isum_lo + sc1*isum_hi), dx, acc)        the two 16-element halves are the sub-blocks; stored values are q + 32
#pragma once
#include <type_traits>
#include "kdev.h"
#include "kernels.h"

namespace q3 {

template <int V> using wt_tag = std::integral_constant<int, V>;

// 16 packed K-quant weights of (block, half) -> 16 unsigned bytes in k order: nibble words + one bit-plane word per extra bit
// (kernels.h: bit 8b+J of a plane word belongs to weight 4J+b of the block; J = 4*half + word index)
template <bool SIX>
__device__ __forceinline__ uint4 kq_unpack16(uint2 nib, uint32_t h4, uint32_t h5, int half) {
    const uint32_t m = 0x0F0F0F0Fu, one = 0x01010101u;
    uint4 v = make_uint4(nib.x & m, (nib.x >> 4) & m, nib.y & m, (nib.y >> 4) & m);
    const int J = 4 * half;
    v.x |= ((h4 >> (J + 0)) & one) << 4; v.y |= ((h4 >> (J + 1)) & one) << 4; v.z |= ((h4 >> (J + 2)) & one) << 4; v.w |= ((h4 >> (J + 3)) & one) << 4;
    if (SIX) { v.x |= ((h5 >> (J + 0)) & one) << 5; v.y |= ((h5 >> (J + 1)) & one) << 5; v.z |= ((h5 >> (J + 2)) & one) << 5; v.w |= ((h5 >> (J + 3)) & one) << 5; }
    return v;
}

// type of row group rg (wave-uniform) from the tensor map in the kernel arguments
__device__ __forceinline__ int wslice_type(const Q8Mat& w, int rg) {
    const int rgs = __builtin_amdgcn_readfirstlane(rg);
    return rgs >= w.p2_rg0 ? w.p2_type : rgs >= w.p1_rg0 ? w.p1_type : w.p0_type;
}
// run body(wt_tag<type>) for the type of row group rg.  TS = what the host knows about the launch: 0 = all-Q8_0 matrix (uniform tiles),
// Q8_0 / Q5_K / Q6_K = K-quant matrix whose rows all have that type (one body in the kernel: o-proj, down, gate/up), -1 = mixed (fused
// QKV of a Q5_K_M file: q, k Q5_K and v Q6_K) -- the type is read from the kernel arguments and all three bodies are in the kernel.
template <int TS, typename F>
__device__ __forceinline__ void wslice_dispatch(const Q8Mat& w, int rg, F&& body) {
    if (TS >= 0) { body(wt_tag<(TS >= 0 ? TS : 0)>{}); return; }
    const int wt = wslice_type(w, rg);
    if (wt == Q3_T_Q5_K) body(wt_tag<Q3_T_Q5_K>{});
    else if (wt == Q3_T_Q6_K) body(wt_tag<Q3_T_Q6_K>{});
    else body(wt_tag<Q3_T_Q8_0>{});
}
// host side: the TS of a matrix, and a switch that instantiates its statement once per value (the statement sees `constexpr int TS`)
inline int wslice_ts(const Q8Mat& w) { return !w.rg_type ? 0 : w.nparts == 1 ? w.p0_type : -1; }
#define Q3_TS_SWITCH(w, ...) do { switch (wslice_ts(w)) { \
    case 0: { constexpr int TS = 0; __VA_ARGS__; break; } case Q3_T_Q8_0: { constexpr int TS = Q3_T_Q8_0; __VA_ARGS__; break; } \
    case Q3_T_Q5_K: { constexpr int TS = Q3_T_Q5_K; __VA_ARGS__; break; } case Q3_T_Q6_K: { constexpr int TS = Q3_T_Q6_K; __VA_ARGS__; break; } \
    default: { constexpr int TS = -1; __VA_ARGS__; break; } } } while (0)

// LPR lanes share a row: lane = (r, half, bil) with r = lane % R the row inside the workgroup's R = 64 / LPR rows, half = which 16 of a
// block's 32 weights, bil = which of the BPL = LPR / 2 blocks of a load step.  A lane holds NLD = 8 / BPL loads of 16 weights.
template <int LPR, int WT>
struct WSlice {
    static constexpr int R = 64 / LPR, BPL = LPR / 2, NLD = 8 / BPL;
    static constexpr bool KQ = WT == Q3_T_Q5_K || WT == Q3_T_Q6_K;
    uint4 wv[NLD];
    uint4 dwv;          // Q8_0: the 8 block scales; K-quants: d, dmin in halfs 0, 1
    uint4 mv;           // K-quants: 16 metadata bytes of (row, segment)

    // issue the loads; the R <= 32 rows of a wave sit in one 32-row group, so the group's base address is a scalar
    __device__ __forceinline__ void load(const Q8Mat& w, int rg, int r32, int seg, int half, int bil) {
        const int nseg = w.K >> 8, nb = w.K >> 5;
        const size_t vidx = ((size_t)rg * nseg + seg) * 32 + r32;
        dwv = *reinterpret_cast<const uint4*>(w.sc + vidx * 8);
        if (WT == 0) {
            const uint8_t* base = w.qs + ((size_t)rg * nb + (size_t)seg * 8) * 1024 + half * 512 + r32 * 16;
#pragma unroll
            for (int i = 0; i < NLD; i++) wv[i] = *reinterpret_cast<const uint4*>(base + (size_t)(i * BPL + bil) * 1024);
            return;
        }
        const int rgs = __builtin_amdgcn_readfirstlane(rg);
        const bool in1 = rgs >= w.p1_rg0, in2 = rgs >= w.p2_rg0;
        const uint32_t poff = in2 ? w.p2_off : in1 ? w.p1_off : w.p0_off;
        const int prg0 = in2 ? w.p2_rg0 : in1 ? w.p1_rg0 : 0;
        constexpr int seg_bytes = WT == Q3_T_Q5_K ? 5120 : WT == Q3_T_Q6_K ? 6144 : 8192;
        const uint8_t* sbase = w.qs + (size_t)poff * 16 + ((size_t)(rgs - prg0) * nseg + seg) * seg_bytes;
        if (!KQ) {
            const uint8_t* base = sbase + half * 512 + r32 * 16;
#pragma unroll
            for (int i = 0; i < NLD; i++) wv[i] = *reinterpret_cast<const uint4*>(base + (size_t)(i * BPL + bil) * 1024);
            return;
        }
        mv = *reinterpret_cast<const uint4*>(w.meta + vidx * 16);
        // packed planes: the raw words park in wv (x, y = nibbles; z, w = bit planes) until finish() -- unpacking here would make the wave
        // wait for the weights before the norm prologue instead of behind it
        const uint8_t* hrow = sbase + 4096 + r32 * (WT == Q3_T_Q6_K ? 64 : 32);
#pragma unroll
        for (int i = 0; i < NLD; i++) {
            const int b = i * BPL + bil;
            const uint2 nib = *reinterpret_cast<const uint2*>(sbase + b * 512 + half * 256 + r32 * 8);
            if (WT == Q3_T_Q6_K) { const uint2 hh = *reinterpret_cast<const uint2*>(hrow + b * 8); wv[i] = make_uint4(nib.x, nib.y, hh.x, hh.y); }
            else wv[i] = make_uint4(nib.x, nib.y, *reinterpret_cast<const uint32_t*>(hrow + b * 4), 0u);
        }
    }

    // unpack the K-quant words parked by load() (unsigned values: Q5_K 0..31, Q6_K 0..63 = q + 32); call once, after the prologue
    __device__ __forceinline__ void finish(int half) {
        if (!KQ) return;
#pragma unroll
        for (int i = 0; i < NLD; i++) wv[i] = kq_unpack16<WT == Q3_T_Q6_K>(make_uint2(wv[i].x, wv[i].y), wv[i].z, wv[i].w, half);
    }

    __device__ __forceinline__ int mbyte(int k) const {
        const uint32_t ww = k < 4 ? mv.x : k < 8 ? mv.y : k < 12 ? mv.z : mv.w;
        return (int)((ww >> (8 * (k & 3))) & 0xFFu);
    }

    // acc + this segment's block chain for one token.  xseg: the token's 256 int8 activations of the segment (LDS or global, 16-B aligned),
    // dxv: their 8 f16 block scales.  Every lane of a row ends with the same value.
    __device__ __forceinline__ float chain(float acc, const int8_t* xseg, const uint4& dxv, int r, int half, int bil) const {
        if (!KQ) {
#pragma unroll
            for (int i = 0; i < NLD; i++) {
                const uint4 xv = *reinterpret_cast<const uint4*>(xseg + (i * BPL + bil) * 32 + half * 16);
                int isum = dot16(wv[i], xv);
                isum += xor_lane<R>(isum);
#pragma unroll
                for (int j = 0; j < BPL; j++) {
                    const int isj = (BPL == 1) ? isum : __shfl(isum, r + 2 * j * R);
                    const int b = i * BPL + j;
                    const float sc = h2f(half_of(dwv, b)) * h2f(half_of(dxv, b));
                    acc = q3_fmaf((float)isj, sc, acc);
                }
            }
            return acc;
        }
        const uint4 ones = make_uint4(0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u);
        const float d0 = h2f(half_of(dwv, 0)), d1 = h2f(half_of(dwv, 1));
        if (WT == Q3_T_Q5_K) {
#pragma unroll
            for (int i = 0; i < NLD; i++) {
                const uint4 xv = *reinterpret_cast<const uint4*>(xseg + (i * BPL + bil) * 32 + half * 16);
                int isum = dot16(wv[i], xv), xsum = dot16(ones, xv);
                isum += xor_lane<R>(isum); xsum += xor_lane<R>(xsum);
                // the block's term is finished in the lane pair that holds the block (all of its inputs are there); only the result travels
                const int bl = i * BPL + bil;
                const int i1 = __mul24(mbyte(bl), isum), i2 = __mul24(mbyte(8 + bl), xsum); // 6-bit scale x 17-bit dot: exact in v_mul_i32_i24
                const float a = d0 * (float)i1;
                const float a2 = d1 * (float)i2;
                const float diff = a - a2;
#pragma unroll
                for (int j = 0; j < BPL; j++) {
                    const float dj = (BPL == 1) ? diff : __shfl(diff, r + 2 * j * R);
                    acc = q3_fmaf(dj, h2f(half_of(dxv, i * BPL + j)), acc);
                }
            }
            return acc;
        }
#pragma unroll
        for (int i = 0; i < NLD; i++) { // Q6_K
            const uint4 xv = *reinterpret_cast<const uint4*>(xseg + (i * BPL + bil) * 32 + half * 16);
            const int bsel = i * BPL + bil; // block of THIS lane's data
            // sum (q - 32) x = sum (stored) x - 32 sum x, exact in int32; the lane's 16 weights are one sub-block
            int isum = __mul24(dot16(wv[i], xv) - 32 * dot16(ones, xv), (int)(int8_t)mbyte(2 * bsel + half));
            isum += xor_lane<R>(isum);
            const float a = d0 * (float)isum;
#pragma unroll
            for (int j = 0; j < BPL; j++) {
                const float aj = (BPL == 1) ? a : __shfl(a, r + 2 * j * R);
                acc = q3_fmaf(aj, h2f(half_of(dxv, i * BPL + j)), acc);
            }
        }
        return acc;
    }
};

} // namespace q3
