// wslice.h -- one lane's slice of a weight matrix for the GEMV family: the quants of (row, 256-segment) this lane multiplies, fetched from
// the Q8_0 tiles or from the packed K-quant planes (kernels.h), and the block chain of spec S3 that turns them into a partial sum.
// Shared by k_gemv_kq, k_gemv_q8_norm, k_gateup_swiglu and k_oproj_attn, so the fused decode path serves Q5_K_M files with the same
// launches as Q8_0 files.  KQ = false is the Q8_0-only form (no type test, no metadata: the code the Q8_0 kernels always had).
//   Q8_0: acc = fma(f(isum), dw*dx, acc)
//   Q5_K: acc = fma(d*f(sc*isum) - dmin*f(m*xsum), dx, acc)        xsum = sum of the activation block
//   Q6_K: acc = fma(d*f(sc0*isum_lo + sc1*isum_hi), dx, acc)        the two 16-element halves are the sub-blocks; stored values are q + 32
#pragma once
#include "kdev.h"
#include "kernels.h"

namespace q3 {

// 16 packed K-quant weights of (block, half) -> 16 unsigned bytes in k order: nibble words + one bit-plane word per extra bit
// (kernels.h: bit 8b+J of a plane word belongs to weight 4J+b of the block; J = 4*half + word index)
__device__ __forceinline__ uint4 kq_unpack16(uint2 nib, uint32_t h4, uint32_t h5, int half) {
    const uint32_t m = 0x0F0F0F0Fu, one = 0x01010101u;
    uint4 v = make_uint4(nib.x & m, (nib.x >> 4) & m, nib.y & m, (nib.y >> 4) & m);
    const int J = 4 * half;
    v.x |= ((h4 >> (J + 0)) & one) << 4; v.y |= ((h4 >> (J + 1)) & one) << 4; v.z |= ((h4 >> (J + 2)) & one) << 4; v.w |= ((h4 >> (J + 3)) & one) << 4;
    v.x |= ((h5 >> (J + 0)) & one) << 5; v.y |= ((h5 >> (J + 1)) & one) << 5; v.z |= ((h5 >> (J + 2)) & one) << 5; v.w |= ((h5 >> (J + 3)) & one) << 5;
    return v;
}

// LPR lanes share a row: lane = (r, half, bil) with r = lane % R the row inside the workgroup's R = 64 / LPR rows, half = which 16 of a
// block's 32 weights, bil = which of the BPL = LPR / 2 blocks of a load step.  A lane holds NLD = 8 / BPL loads of 16 weights.
template <int LPR, bool KQ>
struct WSlice {
    static constexpr int R = 64 / LPR, BPL = LPR / 2, NLD = 8 / BPL;
    uint4 wv[NLD];
    uint4 dwv;          // Q8_0: the 8 block scales; K-quants: d, dmin in halfs 0, 1
    uint4 mv;           // K-quants: 16 metadata bytes of (row, segment)
    int wt;

    __device__ __forceinline__ void load(const Q8Mat& w, int rg, int r32, int seg, int half, int bil) {
        const int nseg = w.K >> 8, nb = w.K >> 5;
        const size_t vidx = ((size_t)rg * nseg + seg) * 32 + r32;
        if (!KQ) {
            const uint8_t* base = w.qs + ((size_t)rg * nb + (size_t)seg * 8) * 1024 + half * 512 + r32 * 16;
#pragma unroll
            for (int i = 0; i < NLD; i++) wv[i] = *reinterpret_cast<const uint4*>(base + (size_t)(i * BPL + bil) * 1024);
            dwv = *reinterpret_cast<const uint4*>(w.sc + vidx * 8);
            return;
        }
        wt = __builtin_amdgcn_readfirstlane((int)w.rg_type[rg]); // the R <= 32 rows of a wave sit in one 32-row group: a scalar, so the type tests are scalar branches
        const uint8_t* rgbase = w.qs + (size_t)w.rg_off[rg] * 16;
        if (wt == Q3_T_Q8_0) {
            const uint8_t* base = rgbase + (size_t)seg * 8 * 1024 + half * 512 + r32 * 16;
#pragma unroll
            for (int i = 0; i < NLD; i++) wv[i] = *reinterpret_cast<const uint4*>(base + (size_t)(i * BPL + bil) * 1024);
        } else { // packed planes: the raw words park in wv (x, y = nibbles; z, w = bit planes) until finish() -- unpacking here would make
                 // the wave wait for the weights before the norm prologue instead of behind it
            const bool q6 = wt == Q3_T_Q6_K;
            const uint8_t* sbase = rgbase + (size_t)seg * (q6 ? 6144 : 5120);
            const uint8_t* hrow = sbase + 4096 + r32 * (q6 ? 64 : 32);
#pragma unroll
            for (int i = 0; i < NLD; i++) {
                const int b = i * BPL + bil;
                const uint2 nib = *reinterpret_cast<const uint2*>(sbase + b * 512 + half * 256 + r32 * 8);
                uint32_t h4, h5 = 0;
                if (q6) { const uint2 hh = *reinterpret_cast<const uint2*>(hrow + b * 8); h4 = hh.x; h5 = hh.y; }
                else h4 = *reinterpret_cast<const uint32_t*>(hrow + b * 4);
                wv[i] = make_uint4(nib.x, nib.y, h4, h5);
            }
        }
        dwv = *reinterpret_cast<const uint4*>(w.sc + vidx * 8);
        mv = make_uint4(0, 0, 0, 0);
        if (wt != Q3_T_Q8_0) mv = *reinterpret_cast<const uint4*>(w.meta + vidx * 16);
    }

    // unpack the K-quant words parked by load() (unsigned values: Q5_K 0..31, Q6_K 0..63 = q + 32); call once, after the prologue
    __device__ __forceinline__ void finish(int half) {
        if (!KQ) return;
        if (wt == Q3_T_Q8_0) return;
#pragma unroll
        for (int i = 0; i < NLD; i++) wv[i] = kq_unpack16(make_uint2(wv[i].x, wv[i].y), wv[i].z, wv[i].w, half);
    }

    __device__ __forceinline__ int mbyte(int k) const {
        const uint32_t ww = k < 4 ? mv.x : k < 8 ? mv.y : k < 12 ? mv.z : mv.w;
        return (int)((ww >> (8 * (k & 3))) & 0xFFu);
    }

    // acc + this segment's block chain for one token.  xseg: the token's 256 int8 activations of the segment (LDS or global, 16-B aligned),
    // dxv: their 8 f16 block scales.  Every lane of a row ends with the same value.
    __device__ __forceinline__ float chain(float acc, const int8_t* xseg, const uint4& dxv, int r, int half, int bil) const {
        if (!KQ || wt == Q3_T_Q8_0) {
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
        if (wt == Q3_T_Q5_K) {
#pragma unroll
            for (int i = 0; i < NLD; i++) {
                const uint4 xv = *reinterpret_cast<const uint4*>(xseg + (i * BPL + bil) * 32 + half * 16);
                int isum = dot16(wv[i], xv), xsum = dot16(ones, xv);
                isum += xor_lane<R>(isum); xsum += xor_lane<R>(xsum);
#pragma unroll
                for (int j = 0; j < BPL; j++) {
                    const int isj = (BPL == 1) ? isum : __shfl(isum, r + 2 * j * R);
                    const int xsj = (BPL == 1) ? xsum : __shfl(xsum, r + 2 * j * R);
                    const int b = i * BPL + j;
                    const int i1 = mbyte(b) * isj, i2 = mbyte(8 + b) * xsj;
                    const float a = d0 * (float)i1;
                    const float a2 = d1 * (float)i2;
                    const float diff = a - a2;
                    acc = q3_fmaf(diff, h2f(half_of(dxv, b)), acc);
                }
            }
            return acc;
        }
#pragma unroll
        for (int i = 0; i < NLD; i++) { // Q6_K
            const uint4 xv = *reinterpret_cast<const uint4*>(xseg + (i * BPL + bil) * 32 + half * 16);
            const int bsel = i * BPL + bil; // block of THIS lane's data
            // sum (q - 32) x = sum (stored) x - 32 sum x, exact in int32; the lane's 16 weights are one sub-block
            int isum = (dot16(wv[i], xv) - 32 * dot16(ones, xv)) * (int)(int8_t)mbyte(2 * bsel + half);
            isum += xor_lane<R>(isum);
#pragma unroll
            for (int j = 0; j < BPL; j++) {
                const int isj = (BPL == 1) ? isum : __shfl(isum, r + 2 * j * R);
                const int b = i * BPL + j;
                const float a = d0 * (float)isj;
                acc = q3_fmaf(a, h2f(half_of(dxv, b)), acc);
            }
        }
        return acc;
    }
};

} // namespace q3
