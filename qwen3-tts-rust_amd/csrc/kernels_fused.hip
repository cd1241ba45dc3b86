// kernels_fused.hip -- fused decode-step kernels (5 launches per transformer layer instead of 9).
//
// Batch-1 decode on MI355X is bound by the number of dependent launches (measured: 1.7 us per trivial graph
// node + 2-4 us of load->reduce->store latency per kernel, profiles/r01_*), not by HBM, so every
// elementwise / normalisation / quantisation step is folded into the kernel that produces or consumes it:
//   A  k_gemv_q8_norm     [residual + RMSNorm + int8 quant] prologue -> Q8_0 GEMV  (QKV; head with atomic argmax)
//   B  k_attention_fused  per-head q/k RMSNorm + M-RoPE + KV append + paged GQA attention + int8 quant
//   C  k_gemv_q8          o-proj (kernels.hip)
//   D  k_gateup_swiglu    [residual + RMSNorm + quant] -> gate & up GEMV -> SwiGLU -> int8 quant
//   E  k_gemv_q8          down-proj (kernels.hip)
// Arithmetic is include/q3tts_spec.h's, bit-identical to the unfused kernels and to oracle/.
#include "kernels.h"
#include "kdev.h"
#include "wslice.h"
#include <cstdlib>
#include "q3_common.h"

namespace q3 {
Q3_STAMP_SETTER(set_stamp_buffer_fused)

// ---------------------------------------------------------------------------------------------------
// one wave: h = h_in (+ parts, in order); RMSNorm; quantise to LDS.  (same arithmetic as k_rmsnorm_quant)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void norm_quant_token(const NormPro& a, int d, int tok, int ntok, int lane, int8_t* xq_dst,
                                                 uint16_t* xd_dst, bool write_global) {
    const int nch = d >> 8;
    const float* hin = a.h_in;
    if (a.idx_keys) hin += (size_t)key_code(a.idx_keys[(size_t)tok * a.idx_stride]) * a.h_stride;
    else hin += (size_t)tok * a.h_stride;
    float4 x[8];
    float p = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        if (c < nch) {
            float4 v = *reinterpret_cast<const float4*>(hin + 256 * c + 4 * lane);
            if (a.nparts > 0) {
                const float* pp = a.parts + (size_t)tok * a.parts_stride + 256 * c + 4 * lane;
                float4 y = *reinterpret_cast<const float4*>(pp);
                for (int s = 1; s < a.nparts; s++) {
                    const float4 z = *reinterpret_cast<const float4*>(pp + (size_t)s * a.parts_slab);
                    y.x = y.x + z.x; y.y = y.y + z.y; y.z = y.z + z.z; y.w = y.w + z.w;
                }
                v.x = v.x + y.x; v.y = v.y + y.y; v.z = v.z + y.z; v.w = v.w + y.w;
            }
            if (write_global && a.h_out) *reinterpret_cast<float4*>(a.h_out + (size_t)tok * d + 256 * c + 4 * lane) = v;
            x[c] = v;
            p = q3_fmaf(v.x, v.x, p); p = q3_fmaf(v.y, v.y, p); p = q3_fmaf(v.z, v.z, p); p = q3_fmaf(v.w, v.w, p);
        }
    }
    const float ss = wave_sum_bfly(p);
    const float mean = ss / (float)d;
    const float scale = 1.0f / q3_sqrtf(mean + a.eps);
#pragma unroll
    for (int c = 0; c < 8; c++) {
        if (c < nch) {
            const float4 g = *reinterpret_cast<const float4*>(a.g + 256 * c + 4 * lane);
            float4 y;
            y.x = (x[c].x * scale) * g.x; y.y = (x[c].y * scale) * g.y;
            y.z = (x[c].z * scale) * g.z; y.w = (x[c].w * scale) * g.w;
            if (write_global && a.xn_out) *reinterpret_cast<float4*>(a.xn_out + (size_t)tok * d + 256 * c + 4 * lane) = y;
            float amax = fmaxf(fmaxf(q3_fabsf(y.x), q3_fabsf(y.y)), fmaxf(q3_fabsf(y.z), q3_fabsf(y.w)));
            amax = fmaxf(amax, xor_lane<1>(amax)); amax = fmaxf(amax, xor_lane<2>(amax)); amax = fmaxf(amax, xor_lane<4>(amax));
            const float dd = amax / 127.0f;
            const float id = (dd != 0.0f) ? (1.0f / dd) : 0.0f;
            const int q0 = (int)q3_rintf(y.x * id), q1 = (int)q3_rintf(y.y * id), q2 = (int)q3_rintf(y.z * id), q3v = (int)q3_rintf(y.w * id);
            const uint32_t pk = (uint32_t)(q0 & 0xFF) | ((uint32_t)(q1 & 0xFF) << 8) | ((uint32_t)(q2 & 0xFF) << 16) | ((uint32_t)(q3v & 0xFF) << 24);
            *reinterpret_cast<uint32_t*>(xq_dst + 256 * c + 4 * lane) = pk;
            if ((lane & 7) == 0) xd_dst[8 * c + (lane >> 3)] = f2h(dd);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Workgroup-cooperative form of the same prologue (identical arithmetic): wave w fetches 256-chunk w (all global
// loads of the token in ONE round trip instead of a serial chain in a single wave), wave 0 runs the spec's 64-lane
// sum-of-squares chain over the chunks from LDS, then every wave scales + quantises its own chunk.
// Requires blockDim.x == 64 * (d/256).  Contains a workgroup barrier: all threads of the block must call it; vbuf is read by EVERY wave after that
// barrier, so a caller that reuses vbuf (a loop over tokens) puts a barrier between two calls.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void norm_quant_wg(const NormPro& a, int d, int tok, bool tok_valid, int lane, int wave, int8_t* xq_dst,
                                              uint16_t* xd_dst, float* vbuf, float* scal, bool write_global) {
    const int c = wave; // chunk owned by this wave
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f), g = v;
    if (tok_valid) {
        const float* hin = a.h_in;
        if (a.idx_keys) hin += (size_t)key_code(a.idx_keys[(size_t)tok * a.idx_stride]) * a.h_stride;
        else hin += (size_t)tok * a.h_stride;
        v = *reinterpret_cast<const float4*>(hin + 256 * c + 4 * lane);
        g = *reinterpret_cast<const float4*>(a.g + 256 * c + 4 * lane);
        if (a.nparts > 0) {
            const float* pp = a.parts + (size_t)tok * a.parts_stride + 256 * c + 4 * lane;
            float4 z[4];
            const int np = a.nparts < 4 ? a.nparts : 4;
#pragma unroll
            for (int s = 0; s < 4; s++) if (s < np) z[s] = *reinterpret_cast<const float4*>(pp + (size_t)s * a.parts_slab);
            float4 y = z[0];
#pragma unroll
            for (int s = 1; s < 4; s++) if (s < np) { y.x = y.x + z[s].x; y.y = y.y + z[s].y; y.z = y.z + z[s].z; y.w = y.w + z[s].w; }
            for (int s = 4; s < a.nparts; s++) {
                const float4 zz = *reinterpret_cast<const float4*>(pp + (size_t)s * a.parts_slab);
                y.x = y.x + zz.x; y.y = y.y + zz.y; y.z = y.z + zz.z; y.w = y.w + zz.w;
            }
            v.x = v.x + y.x; v.y = v.y + y.y; v.z = v.z + y.z; v.w = v.w + y.w;
        }
        if (write_global && a.h_out) *reinterpret_cast<float4*>(a.h_out + (size_t)tok * d + 256 * c + 4 * lane) = v;
    }
    *reinterpret_cast<float4*>(vbuf + 256 * c + 4 * lane) = v;
    wg_barrier_lds();
    // every wave runs the spec's 64-lane sum-of-squares chain itself (same order, same result in each wave): cheaper than wave 0 computing it
    // and a second barrier + LDS round trip to hand the scale over
    float scale;
    {
        const int nch = d >> 8; // <= 8 (d <= 2048): all chunk reads first, then the chain in chunk order (a rolled loop pays an LDS round trip per chunk)
        float4 u[8];
#pragma unroll
        for (int cc = 0; cc < 8; cc++) u[cc] = *reinterpret_cast<const float4*>(vbuf + 256 * (cc < nch ? cc : 0) + 4 * lane);
        float p = 0.0f;
#pragma unroll
        for (int cc = 0; cc < 8; cc++)
            if (cc < nch) { p = q3_fmaf(u[cc].x, u[cc].x, p); p = q3_fmaf(u[cc].y, u[cc].y, p); p = q3_fmaf(u[cc].z, u[cc].z, p); p = q3_fmaf(u[cc].w, u[cc].w, p); }
        const float ss = wave_sum_bfly(p);
        const float mean = ss / (float)d;
        scale = 1.0f / q3_sqrtf(mean + a.eps);
    }
    (void)scal;
    float4 y;
    y.x = (v.x * scale) * g.x; y.y = (v.y * scale) * g.y; y.z = (v.z * scale) * g.z; y.w = (v.w * scale) * g.w;
    if (tok_valid && write_global && a.xn_out) *reinterpret_cast<float4*>(a.xn_out + (size_t)tok * d + 256 * c + 4 * lane) = y;
    float amax = fmaxf(fmaxf(q3_fabsf(y.x), q3_fabsf(y.y)), fmaxf(q3_fabsf(y.z), q3_fabsf(y.w)));
    amax = fmaxf(amax, xor_lane<1>(amax)); amax = fmaxf(amax, xor_lane<2>(amax)); amax = fmaxf(amax, xor_lane<4>(amax));
    const float dd = amax / 127.0f;
    const float id = (dd != 0.0f) ? (1.0f / dd) : 0.0f;
    const int q0 = (int)q3_rintf(y.x * id), q1 = (int)q3_rintf(y.y * id), q2 = (int)q3_rintf(y.z * id), q3v = (int)q3_rintf(y.w * id);
    const uint32_t pk = (uint32_t)(q0 & 0xFF) | ((uint32_t)(q1 & 0xFF) << 8) | ((uint32_t)(q2 & 0xFF) << 16) | ((uint32_t)(q3v & 0xFF) << 24);
    *reinterpret_cast<uint32_t*>(xq_dst + 256 * c + 4 * lane) = pk;
    if ((lane & 7) == 0) xd_dst[8 * c + (lane >> 3)] = f2h(dd);
}

// standalone workgroup-per-token norm + quant (batched-step path): same arithmetic as k_rmsnorm_quant, one global
// round trip instead of a serial chain in a single wave
__global__ void __launch_bounds__(512) k_rmsnorm_quant_wg(NormPro a, int d, int8_t* __restrict__ xq, uint16_t* __restrict__ xd) {
    __shared__ __attribute__((aligned(16))) float vbuf_s[2048];
    __shared__ float scal_s[1];
    const int tok = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    Q3_STAMP_DECL;
    Q3_STAMP(0);
    // the quantised row and its block scales go straight to global memory (a dword per lane = 256 contiguous bytes per wave): staging them in LDS for
    // 16-byte stores cost a barrier and an LDS round trip that this latency-bound kernel does not get back
    norm_quant_wg(a, d, tok, true, lane, wave, xq + (size_t)tok * d, xd + (size_t)tok * (d / 32), vbuf_s, scal_s, true);
    Q3_STAMP(1);
    Q3_STAMP(6);
    Q3_STAMP_FLUSH();
}
void launch_rmsnorm_quant_wg(hipStream_t st, const NormPro& a, int d, int8_t* xq, uint16_t* xd, int ntok) {
    hipLaunchKernelGGL(k_rmsnorm_quant_wg, dim3(ntok), dim3(64 * (d / 256)), 0, st, a, d, xq, xd);
}

// ===================================================================================================
// A: norm prologue + GEMV (K = d <= 2048, one super-segment).  EPI 0: store f32; EPI 1: atomic argmax.
// ===================================================================================================
template <int LPR, int MT, int EPI, int TS>
__global__ void __launch_bounds__(512) k_gemv_q8_norm(Q8Mat w, int row0, int nrows, NormPro a, float* __restrict__ out,
                                                      int out_stride, int ntok, ArgmaxEpi am) {
    constexpr int R = 64 / LPR;
    __shared__ float red[8][R * MT];
    __shared__ __attribute__((aligned(16))) int8_t xq_s[MT][2048];
    __shared__ __attribute__((aligned(16))) uint16_t xd_s[MT][64];
    __shared__ __attribute__((aligned(16))) float vbuf_s[2048];
    __shared__ float scal_s[1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane % R, q = lane / R, half = q & 1, bil = q >> 1;
    const int nseg = w.K >> 8;
    const int seg = wave; // single super-segment
    const int tok0 = blockIdx.z * MT;
    int row = row0 + blockIdx.x * R + r;
    if (row > w.Npad - 1) row = w.Npad - 1;
    // one body per weight type (wslice.h); every wave of the workgroup has the same row group, hence the same type and the same barriers
    wslice_dispatch<TS>(w, row >> 5, [&](auto tag) {
        // weight stream first: independent of the activations, flies while the prologue runs
        WSlice<LPR, decltype(tag)::value> ws;
        ws.load(w, row >> 5, row & 31, seg, half, bil);
        for (int m = 0; m < MT; m++) { // (nwaves == K/256 by construction of the launch)
            const int tok = tok0 + m;
            norm_quant_wg(a, w.K, tok, tok < ntok, lane, wave, xq_s[m], xd_s[m], vbuf_s, scal_s, blockIdx.x == 0);
            if (m + 1 < MT) wg_barrier_lds(); // every wave reads all of vbuf_s for its sum of squares: the next token may not overwrite it earlier
        }
        wg_barrier_lds();
        ws.finish(half);
        float acc[MT];
#pragma unroll
        for (int m = 0; m < MT; m++) {
            const int mm = (tok0 + m < ntok) ? m : 0; // clamp like the unfused kernel (result unused)
            const uint4 dxv = *reinterpret_cast<const uint4*>(&xd_s[mm][seg * 8]);
            acc[m] = ws.chain(0.0f, &xq_s[mm][seg * 256], dxv, r, half, bil);
        }
        if (q == 0) {
#pragma unroll
            for (int m = 0; m < MT; m++) red[wave][m * R + r] = acc[m];
        }
    });
    wg_barrier_lds();
    for (int t = threadIdx.x; t < R * MT; t += blockDim.x) { // whole R-lane groups stay together (blockDim % 64 == 0)
        const int m = t / R, rr = t % R;
        float v[8]; // all eight reads first, then the in-order adds (a rolled loop pays one LDS round trip per add); rows s >= nseg are read and ignored
#pragma unroll
        for (int s = 0; s < 8; s++) v[s] = red[s][t];
        float S = v[0];
#pragma unroll
        for (int s = 1; s < 8; s++) S = (s < nseg) ? S + v[s] : S;
        const int orow = blockIdx.x * R + rr, tok = tok0 + m;
        const bool ok = orow < nrows && tok < ntok;
        if (EPI == 0) {
            if (ok) out[(size_t)tok * out_stride + orow] = S;
        } else {
            // first-max over this workgroup's R rows (R <= 32 lanes share a token), then one atomic per token
            const int gi = orow + am.idx_add;
            const int mk = (ok && am.mask_per_tok) ? am.mask_per_tok[tok] : -1;
            u64 key = (ok && gi != mk && S > -INFINITY) ? pack_key(S, gi) : 0ull;
#pragma unroll
            for (int sft = R / 2; sft >= 1; sft >>= 1) {
                const u64 o = __shfl_xor(key, sft);
                key = o > key ? o : key;
            }
            if (rr == 0 && key != 0ull && tok < ntok) atomicMax(am.keys + (size_t)tok * am.key_stride, key);
        }
    }
}

template <int LPR, int MT, int EPI>
static void gemv_norm_launch(hipStream_t st, const Q8Mat& w, int row0, int nrows, const NormPro& a, float* out, int out_stride,
                             int ntok, const ArgmaxEpi& am) {
    constexpr int R = 64 / LPR;
    const int nseg = w.K >> 8;
    dim3 grid((nrows + R - 1) / R, 1, (ntok + MT - 1) / MT);
    Q3_TS_SWITCH(w, hipLaunchKernelGGL((k_gemv_q8_norm<LPR, MT, EPI, TS>), grid, dim3(64 * nseg), 0, st, w, row0, nrows, a, out, out_stride, ntok, am));
}
template <int LPR, int EPI>
static void gemv_norm_mt(hipStream_t st, const Q8Mat& w, int row0, int nrows, const NormPro& a, float* out, int out_stride, int ntok,
                         const ArgmaxEpi& am) {
    if (ntok == 1) gemv_norm_launch<LPR, 1, EPI>(st, w, row0, nrows, a, out, out_stride, ntok, am);
    else if (ntok == 2) gemv_norm_launch<LPR, 2, EPI>(st, w, row0, nrows, a, out, out_stride, ntok, am);
    else if (ntok <= 4) gemv_norm_launch<LPR, 4, EPI>(st, w, row0, nrows, a, out, out_stride, ntok, am);
    else gemv_norm_launch<LPR, 8, EPI>(st, w, row0, nrows, a, out, out_stride, ntok, am);
}
void launch_gemv_q8_norm(hipStream_t st, const Q8Mat& w, int row0, int nrows, const NormPro& a, float* out, int out_stride, int ntok,
                         const ArgmaxEpi* am) {
    ArgmaxEpi none{};
    if (am) { gemv_norm_mt<2, 1>(st, w, row0, nrows, a, out, out_stride, ntok, *am); return; }
    if (nrows / 32 >= 512) gemv_norm_mt<2, 0>(st, w, row0, nrows, a, out, out_stride, ntok, none);
    else if (nrows / 16 >= 256) gemv_norm_mt<4, 0>(st, w, row0, nrows, a, out, out_stride, ntok, none);
    else gemv_norm_mt<8, 0>(st, w, row0, nrows, a, out, out_stride, ntok, none);
}

// ===================================================================================================
// D: norm prologue + gate & up GEMV + SwiGLU + int8 quant.  Workgroup = 32 gate rows + the 32 matching up
// rows (one output quant block); wave = one segment of both (16 weight loads in flight per lane).
// ===================================================================================================
// NSEG (waves = 256-element segments of K; 8 = the talker's K = 2048) is a template parameter only so that profiles list the talker's
// and the predictor's launches as different kernels: bench.py's roofline line is about k_gateup_swiglu<1, 8>.
template <int MT, int NSEG, int TS>
__global__ void __launch_bounds__(64 * NSEG) k_gateup_swiglu(Q8Mat w, int ff, NormPro a, int8_t* __restrict__ aq,
                                                             uint16_t* __restrict__ ad, int ntok) {
    __shared__ float red[8][2][32 * MT];
    __shared__ __attribute__((aligned(16))) int8_t xq_s[MT][2048];
    __shared__ __attribute__((aligned(16))) uint16_t xd_s[MT][64];
    __shared__ __attribute__((aligned(16))) float vbuf_s[2048];
    __shared__ float scal_s[1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, half = lane >> 5;
    const int nseg = w.K >> 8;
    const int seg = wave;
    const int tok0 = blockIdx.z * MT;
    const int rgG = blockIdx.x, rgU = (ff >> 5) + blockIdx.x;
    // one body per weight type (gate and up rows of a file share their type: launch_gateup_swiglu checks)
    wslice_dispatch<TS>(w, rgG, [&](auto tag) {
        constexpr int WT = decltype(tag)::value;
        WSlice<2, WT> wsg, wsu;
        if (WT == 0) { // the two weight streams interleaved, as the Q8_0 kernel always issued them
            const int nb = w.K >> 5;
            const uint8_t* baseG = w.qs + ((size_t)rgG * nb + (size_t)seg * 8) * 1024 + half * 512 + r * 16;
            const uint8_t* baseU = w.qs + ((size_t)rgU * nb + (size_t)seg * 8) * 1024 + half * 512 + r * 16;
#pragma unroll
            for (int i = 0; i < 8; i++) { wsg.wv[i] = *reinterpret_cast<const uint4*>(baseG + (size_t)i * 1024); wsu.wv[i] = *reinterpret_cast<const uint4*>(baseU + (size_t)i * 1024); }
            wsg.dwv = *reinterpret_cast<const uint4*>(w.sc + (((size_t)rgG * nseg + seg) * 32 + r) * 8);
            wsu.dwv = *reinterpret_cast<const uint4*>(w.sc + (((size_t)rgU * nseg + seg) * 32 + r) * 8);
        } else {
            wsg.load(w, rgG, r, seg, half, 0);
            wsu.load(w, rgU, r, seg, half, 0);
        }
        for (int m = 0; m < MT; m++) {
            const int tok = tok0 + m;
            norm_quant_wg(a, w.K, tok, tok < ntok, lane, wave, xq_s[m], xd_s[m], vbuf_s, scal_s, blockIdx.x == 0);
            if (m + 1 < MT) wg_barrier_lds(); // every wave reads all of vbuf_s for its sum of squares: the next token may not overwrite it earlier
        }
        wg_barrier_lds();
        wsg.finish(half); wsu.finish(half);
#pragma unroll
        for (int m = 0; m < MT; m++) {
            const int mm = (tok0 + m < ntok) ? m : 0;
            const uint4 dxv = *reinterpret_cast<const uint4*>(&xd_s[mm][seg * 8]);
            float ag = 0.0f, au = 0.0f;
            if (WT == 0) {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const uint4 xv = *reinterpret_cast<const uint4*>(&xq_s[mm][seg * 256 + i * 32 + half * 16]);
                    int ig = dot16(wsg.wv[i], xv), iu = dot16(wsu.wv[i], xv);
                    ig += xor_lane<32>(ig); iu += xor_lane<32>(iu);
                    const float dx = h2f(half_of(dxv, i));
                    ag = q3_fmaf((float)ig, h2f(half_of(wsg.dwv, i)) * dx, ag);
                    au = q3_fmaf((float)iu, h2f(half_of(wsu.dwv, i)) * dx, au);
                }
            } else {
                ag = wsg.chain(0.0f, &xq_s[mm][seg * 256], dxv, r, half, 0);
                au = wsu.chain(0.0f, &xq_s[mm][seg * 256], dxv, r, half, 0);
            }
            if (half == 0) { red[wave][0][m * 32 + r] = ag; red[wave][1][m * 32 + r] = au; }
        }
    });
    wg_barrier_lds();
    for (int t = threadIdx.x; t < 32 * MT; t += blockDim.x) {
        const int m = t >> 5, rr = t & 31, tok = tok0 + m;
        float vg[8], vu[8]; // (all reads first: see k_gemv_q8_norm)
#pragma unroll
        for (int s = 0; s < 8; s++) { vg[s] = red[s][0][t]; vu[s] = red[s][1][t]; }
        float G = vg[0], U = vu[0];
#pragma unroll
        for (int s = 1; s < 8; s++) { G = (s < nseg) ? G + vg[s] : G; U = (s < nseg) ? U + vu[s] : U; }
        const float y = q3_swiglu(G, U);
        float amax = q3_fabsf(y);
        amax = fmaxf(amax, xor_lane<16>(amax)); amax = fmaxf(amax, xor_lane<8>(amax)); amax = fmaxf(amax, xor_lane<4>(amax));
        amax = fmaxf(amax, xor_lane<2>(amax)); amax = fmaxf(amax, xor_lane<1>(amax));
        const float dd = amax / 127.0f;
        const float id = (dd != 0.0f) ? (1.0f / dd) : 0.0f;
        if (tok < ntok) {
            aq[(size_t)tok * ff + blockIdx.x * 32 + rr] = (int8_t)(int)q3_rintf(y * id);
            if (rr == 0) ad[(size_t)tok * (ff >> 5) + blockIdx.x] = f2h(dd);
        }
    }
}
void launch_gateup_swiglu(hipStream_t st, const Q8Mat& w, int ff, const NormPro& a, int8_t* aq, uint16_t* ad, int ntok) {
    const int nseg = w.K >> 8;
    const int mt = ntok == 1 ? 1 : ntok == 2 ? 2 : 4;
    dim3 grid(ff / 32, 1, (ntok + mt - 1) / mt);
#define Q3_GU(MTV, NS) Q3_TS_SWITCH(w, hipLaunchKernelGGL((k_gateup_swiglu<MTV, NS, TS>), grid, dim3(64 * NS), 0, st, w, ff, a, aq, ad, ntok))
#define Q3_GU_NS(NS) do { if (mt == 1) Q3_GU(1, NS); else if (mt == 2) Q3_GU(2, NS); else Q3_GU(4, NS); } while (0)
    switch (nseg) {
        case 1: Q3_GU_NS(1); break; case 2: Q3_GU_NS(2); break; case 3: Q3_GU_NS(3); break; case 4: Q3_GU_NS(4); break;
        case 5: Q3_GU_NS(5); break; case 6: Q3_GU_NS(6); break; case 7: Q3_GU_NS(7); break; case 8: Q3_GU_NS(8); break;
        default: throw Error("k_gateup_swiglu: K must be at most 2048");
    }
#undef Q3_GU_NS
#undef Q3_GU
}

// ===================================================================================================
// B: fused per-head norm + RoPE + KV append + attention for DECODE steps (every token of the launch belongs
// to a different sequence, so no token needs another token's K/V from the same launch).  The current
// position's K/V are taken from LDS (not re-read from global) so no intra-launch global visibility is needed.
// ===================================================================================================
__global__ void __launch_bounds__(256) k_attention_fused(const float* __restrict__ qkv, int qkv_stride, int n_head, int n_kv,
                                                         const float* __restrict__ q_norm_w, const float* __restrict__ k_norm_w,
                                                         float eps, const float* __restrict__ rope_cos,
                                                         const float* __restrict__ rope_sin, int n_ctx,
                                                         const int32_t* __restrict__ mrope_sec, TokMeta tm, KvCache kv, int layer,
                                                         int8_t* __restrict__ aq, uint16_t* __restrict__ ad) {
    __shared__ __attribute__((aligned(16))) float q_s[128];
    __shared__ __attribute__((aligned(16))) uint16_t kcur_s[128];
    __shared__ __attribute__((aligned(16))) uint16_t vcur_s[128];
    __shared__ float p_s[256];
    __shared__ float red_s[4][128];
    __shared__ float wmax_s[4];
    const int h = blockIdx.x, tok = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int seq = tm.seq_of(tok), slot = tm.slot_of(tok), n = slot + 1;
    const int grp = n_head / n_kv, kvh = h / grp;
    const size_t head_off = (size_t)layer * kv.layer_stride() + (size_t)kvh * 8192;
    const int jj = lane >> 4, dc = lane & 15;
    // ---- latency plan: every global load of the first 256-chunk is issued before the prologue's arithmetic ----
    // (K/V of cached positions depend only on the page table, not on this token's q/k/v)
    uint4 kreg[16], vreg[16];
    {
        const int jbase = wave * 64;
        if (jbase < n) {
            const int page = kv.page_of(seq, jbase >> 6);
            const uint16_t* Kb = kv.k + (size_t)page * kv.page_stride() + head_off;
#pragma unroll
            for (int d8 = 0; d8 < 16; d8++) kreg[d8] = *reinterpret_cast<const uint4*>(Kb + (d8 * 64 + lane) * 8);
        }
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int jg = 16 * u + 4 * wave + jj;
            vreg[u] = make_uint4(0, 0, 0, 0);
            if (jg < n && jg != slot) {
                const int page = kv.page_of(seq, jg >> 6);
                vreg[u] = *reinterpret_cast<const uint4*>(kv.v + (size_t)page * kv.page_stride() + head_off + (jg & 63) * 128 + dc * 8);
            }
        }
    }
    // ---- prologue: wave 0 = q head, wave 1 = k head, wave 2 = v head ----
    if (wave < 3) {
        const float* vec = qkv + (size_t)tok * qkv_stride + (wave == 0 ? (size_t)h * 128 : wave == 1 ? (size_t)(n_head + kvh) * 128 : (size_t)(n_head + n_kv + kvh) * 128);
        const float x1 = vec[lane], x2 = vec[lane + 64];
        const int page = kv.page_of(seq, slot >> 6), ps = slot & 63;
        if (wave < 2) {
            const float* wn = wave == 0 ? q_norm_w : k_norm_w;
            const float w1 = wn[lane], w2 = wn[lane + 64];
            int32_t sec[4] = { mrope_sec[0], mrope_sec[1], mrope_sec[2], mrope_sec[3] };
            int pp = tm.pos_of(tok, q3_mrope_stream(lane, sec));
            if (pp < 0) pp = 0;
            if (pp > n_ctx - 1) pp = n_ctx - 1;
            const float cs = rope_cos[(size_t)pp * 64 + lane], sn = rope_sin[(size_t)pp * 64 + lane];
            float p = x1 * x1;
            p = q3_fmaf(x2, x2, p);
            const float ss = wave_sum_bfly(p);
            const float mean = ss / 128.0f;
            const float scale = 1.0f / q3_sqrtf(mean + eps);
            const float y1 = (x1 * scale) * w1, y2 = (x2 * scale) * w2;
            float o1, o2;
            q3_rope_pair(y1, y2, cs, sn, &o1, &o2);
            if (wave == 0) { q_s[lane] = o1; q_s[lane + 64] = o2; }
            else {
                const uint16_t k1 = f2h(o1), k2 = f2h(o2);
                kcur_s[lane] = k1; kcur_s[lane + 64] = k2;
                if (h % grp == 0) { // one writer per kv head
                    uint16_t* Kb = kv.k + (size_t)page * kv.page_stride() + head_off;
                    Kb[((lane >> 3) * 64 + ps) * 8 + (lane & 7)] = k1;
                    Kb[(((lane + 64) >> 3) * 64 + ps) * 8 + (lane & 7)] = k2;
                }
            }
        } else {
            const uint16_t v1 = f2h(x1), v2 = f2h(x2);
            vcur_s[lane] = v1; vcur_s[lane + 64] = v2;
            if (h % grp == 0) {
                uint16_t* Vb = kv.v + (size_t)page * kv.page_stride() + head_off;
                Vb[ps * 128 + lane] = v1; Vb[ps * 128 + lane + 64] = v2;
            }
        }
    }
    wg_barrier_lds();
    const float scale = 0.08838834764831845f;
    float M = 0.0f, L = 0.0f, O[8];
#pragma unroll
    for (int i = 0; i < 8; i++) O[i] = 0.0f;
    for (int c0 = 0; c0 < n; c0 += 256) {
        const int jbase = c0 + wave * 64;
        const int jme = jbase + lane;
        const bool valid = jme < n;
        if (c0 > 0) { // later chunks: load here (the first chunk was prefetched above)
            if (jbase < n) {
                const int page = kv.page_of(seq, jbase >> 6);
                const uint16_t* Kb = kv.k + (size_t)page * kv.page_stride() + head_off;
#pragma unroll
                for (int d8 = 0; d8 < 16; d8++) kreg[d8] = *reinterpret_cast<const uint4*>(Kb + (d8 * 64 + lane) * 8);
            }
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int jg = c0 + 16 * u + 4 * wave + jj;
                vreg[u] = make_uint4(0, 0, 0, 0);
                if (jg < n && jg != slot) {
                    const int page = kv.page_of(seq, jg >> 6);
                    vreg[u] = *reinterpret_cast<const uint4*>(kv.v + (size_t)page * kv.page_stride() + head_off + (jg & 63) * 128 + dc * 8);
                }
            }
        }
        float s = -INFINITY;
        if (jbase < n) {
            const bool cur = (jme == slot);
            float acc = 0.0f;
#pragma unroll
            for (int d8 = 0; d8 < 16; d8++) {
                uint4 kk = kreg[d8];
                if (cur) kk = *reinterpret_cast<const uint4*>(&kcur_s[8 * d8]);
                const float4 qa = *reinterpret_cast<const float4*>(&q_s[8 * d8]);
                const float4 qb = *reinterpret_cast<const float4*>(&q_s[8 * d8 + 4]);
                acc = q3_fmaf(qa.x, h2f(kk.x & 0xFFFFu), acc); acc = q3_fmaf(qa.y, h2f(kk.x >> 16), acc);
                acc = q3_fmaf(qa.z, h2f(kk.y & 0xFFFFu), acc); acc = q3_fmaf(qa.w, h2f(kk.y >> 16), acc);
                acc = q3_fmaf(qb.x, h2f(kk.z & 0xFFFFu), acc); acc = q3_fmaf(qb.y, h2f(kk.z >> 16), acc);
                acc = q3_fmaf(qb.z, h2f(kk.w & 0xFFFFu), acc); acc = q3_fmaf(qb.w, h2f(kk.w >> 16), acc);
            }
            if (valid) s = acc * scale;
        }
        const float wm = wave_max_bfly(s);
        if (lane == 0) wmax_s[wave] = wm;
        wg_barrier_lds();
        const float mc = fmaxf(fmaxf(wmax_s[0], wmax_s[1]), fmaxf(wmax_s[2], wmax_s[3]));
        const float p = valid ? q3_expf(s - mc) : 0.0f;
        p_s[wave * 64 + lane] = p;
        wg_barrier_lds();
        float S[8];
#pragma unroll
        for (int i = 0; i < 8; i++) S[i] = 0.0f;
        const int cn = (n - c0) < 256 ? (n - c0) : 256;
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (16 * u < cn) {
                const int jl = 16 * u + 4 * wave + jj;
                const int jg = c0 + jl;
                const float pj = p_s[jl];
                uint4 vv = vreg[u];
                if (jg == slot) vv = *reinterpret_cast<const uint4*>(&vcur_s[dc * 8]);
                S[0] = q3_fmaf(pj, h2f(vv.x & 0xFFFFu), S[0]); S[1] = q3_fmaf(pj, h2f(vv.x >> 16), S[1]);
                S[2] = q3_fmaf(pj, h2f(vv.y & 0xFFFFu), S[2]); S[3] = q3_fmaf(pj, h2f(vv.y >> 16), S[3]);
                S[4] = q3_fmaf(pj, h2f(vv.z & 0xFFFFu), S[4]); S[5] = q3_fmaf(pj, h2f(vv.z >> 16), S[5]);
                S[6] = q3_fmaf(pj, h2f(vv.w & 0xFFFFu), S[6]); S[7] = q3_fmaf(pj, h2f(vv.w >> 16), S[7]);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float a2 = S[i] + xor_lane<16>(S[i]);
            const float T = a2 + xor_lane<32>(a2);
            if (jj == 0) red_s[wave][dc * 8 + i] = T;
        }
        wg_barrier_lds();
        if (wave == 0) {
            float a2 = p_s[lane];
            a2 = a2 + p_s[lane + 64]; a2 = a2 + p_s[lane + 128]; a2 = a2 + p_s[lane + 192];
            const float lc = wave_sum_bfly(a2);
            float oc[8];
#pragma unroll
            for (int i = 0; i < 8; i++) oc[i] = (red_s[0][dc * 8 + i] + red_s[1][dc * 8 + i]) + (red_s[2][dc * 8 + i] + red_s[3][dc * 8 + i]);
            if (c0 == 0) {
                M = mc; L = lc;
#pragma unroll
                for (int i = 0; i < 8; i++) O[i] = oc[i];
            } else {
                const float mn = fmaxf(M, mc);
                const float ea = q3_expf(M - mn), eb = q3_expf(mc - mn);
                const float t2 = lc * eb;
                L = q3_fmaf(L, ea, t2);
#pragma unroll
                for (int i = 0; i < 8; i++) { const float u2 = oc[i] * eb; O[i] = q3_fmaf(O[i], ea, u2); }
                M = mn;
            }
        }
        wg_barrier_lds();
    }
    if (wave == 0) {
        float y[8];
        float amax = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; i++) { y[i] = O[i] / L; amax = fmaxf(amax, q3_fabsf(y[i])); }
        amax = fmaxf(amax, xor_lane<1>(amax));
        amax = fmaxf(amax, xor_lane<2>(amax));
        const float dd = amax / 127.0f;
        const float id = (dd != 0.0f) ? (1.0f / dd) : 0.0f;
        if (jj == 0) {
            const size_t o = ((size_t)tok * n_head + h) * 128 + dc * 8;
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                lo |= (uint32_t)((int)q3_rintf(y[i] * id) & 0xFF) << (8 * i);
                hi |= (uint32_t)((int)q3_rintf(y[i + 4] * id) & 0xFF) << (8 * i);
            }
            *reinterpret_cast<uint2*>(aq + o) = make_uint2(lo, hi);
            if ((dc & 3) == 0) ad[o >> 5] = f2h(dd);
        }
    }
}
void launch_attention_fused(hipStream_t st, const float* qkv, int qkv_stride, int n_head, int n_kv, const float* q_norm_w,
                            const float* k_norm_w, float eps, const float* rope_cos, const float* rope_sin, int n_ctx,
                            const int32_t* mrope_sec, const TokMeta& tm, const KvCache& kv, int layer, int8_t* aq, uint16_t* ad,
                            int ntok) {
    hipLaunchKernelGGL(k_attention_fused, dim3(n_head, ntok), dim3(256), 0, st, qkv, qkv_stride, n_head, n_kv, q_norm_w, k_norm_w,
                       eps, rope_cos, rope_sin, n_ctx, mrope_sec, tm, kv, layer, aq, ad);
}

// ===================================================================================================
// Single-wave attention for sequences of at most 64 cached positions (the code predictor: <= 17): no workgroup barrier, K rows and the
// cached V rows live in registers (both fetched before the prologue's arithmetic).  HPW = q heads per wave: 2 = one wave per (token,
// kv head) serves both q heads of the group (fewest waves: wide steps); 1 = one wave per q head, the kv head's K/V row computed by both
// waves of the pair and written by the even one (half the serial work per wave: narrow steps).  Same arithmetic as k_attention_fused (S7).
// ===================================================================================================
template <int HPW>
__global__ void __launch_bounds__(64) k_attention_short(const float* __restrict__ qkv, int qkv_stride, int n_head, int n_kv,
                                                        const float* __restrict__ q_norm_w, const float* __restrict__ k_norm_w, float eps,
                                                        const float* __restrict__ rope_cos, const float* __restrict__ rope_sin, int n_ctx,
                                                        const int32_t* __restrict__ mrope_sec, TokMeta tm, KvCache kv, int layer,
                                                        int8_t* __restrict__ aq, uint16_t* __restrict__ ad, float* __restrict__ att) {
    __shared__ __attribute__((aligned(16))) float q_s[HPW][128];
    __shared__ __attribute__((aligned(16))) uint16_t kcur_s[128];
    __shared__ __attribute__((aligned(16))) uint16_t vcur_s[128];
    __shared__ float p_s[HPW][64];
    const int kvh = HPW == 2 ? blockIdx.x : blockIdx.x >> 1, h0 = HPW == 2 ? 2 * blockIdx.x : blockIdx.x; // first q head of this wave
    Q3_STAMP_DECL;
    Q3_STAMP(0);
    const bool writer = HPW == 2 || (blockIdx.x & 1) == 0;
    const int tok = blockIdx.y, lane = threadIdx.x;
    const float scale = 0.08838834764831845f;
    const int jj = lane >> 4, dc = lane & 15;
    const int seq = tm.seq_of(tok), slot = tm.slot_of(tok), n = slot + 1;
    const size_t head_off = (size_t)layer * kv.layer_stride() + (size_t)kvh * 8192;
    const int page = kv.page_of(seq, 0);
    const uint16_t* Kb = kv.k + (size_t)page * kv.page_stride() + head_off;
    const uint16_t* Vb = kv.v + (size_t)page * kv.page_stride() + head_off;
    uint4 kreg[16], vreg[4][4];
#pragma unroll
    for (int d8 = 0; d8 < 16; d8++) kreg[d8] = (lane < slot) ? *reinterpret_cast<const uint4*>(Kb + (d8 * 64 + lane) * 8) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int ww = 0; ww < 4; ww++)
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int jl = 16 * u + 4 * ww + jj;
            vreg[ww][u] = (jl < slot) ? *reinterpret_cast<const uint4*>(Vb + jl * 128 + dc * 8) : make_uint4(0, 0, 0, 0);
        }
    int32_t sec[4] = { mrope_sec[0], mrope_sec[1], mrope_sec[2], mrope_sec[3] };
    int pp = tm.pos_of(tok, q3_mrope_stream(lane, sec));
    if (pp < 0) pp = 0;
    if (pp > n_ctx - 1) pp = n_ctx - 1;
    const float cs = rope_cos[(size_t)pp * 64 + lane], sn = rope_sin[(size_t)pp * 64 + lane];
    const float* tv = qkv + (size_t)tok * qkv_stride;
#pragma unroll
    for (int which = 0; which < HPW + 1; which++) { // q heads, then k
        const float* vec = tv + (which < HPW ? (size_t)(h0 + which) * 128 : (size_t)(n_head + kvh) * 128);
        const float* wn = which < HPW ? q_norm_w : k_norm_w;
        const float x1 = vec[lane], x2 = vec[lane + 64];
        float p = x1 * x1;
        p = q3_fmaf(x2, x2, p);
        const float ss = wave_sum_bfly(p);
        const float mean = ss / 128.0f;
        const float sc2 = 1.0f / q3_sqrtf(mean + eps);
        const float y1 = (x1 * sc2) * wn[lane], y2 = (x2 * sc2) * wn[lane + 64];
        float o1, o2;
        q3_rope_pair(y1, y2, cs, sn, &o1, &o2);
        if (which < HPW) { q_s[which][lane] = o1; q_s[which][lane + 64] = o2; }
        else {
            const uint16_t k1 = f2h(o1), k2 = f2h(o2);
            kcur_s[lane] = k1; kcur_s[lane + 64] = k2;
            if (writer) {
                uint16_t* Kw = kv.k + (size_t)page * kv.page_stride() + head_off;
                Kw[((lane >> 3) * 64 + slot) * 8 + (lane & 7)] = k1;
                Kw[(((lane + 64) >> 3) * 64 + slot) * 8 + (lane & 7)] = k2;
            }
        }
    }
    {
        const float* vec = tv + (size_t)(n_head + n_kv + kvh) * 128;
        const uint16_t v1 = f2h(vec[lane]), v2 = f2h(vec[lane + 64]);
        vcur_s[lane] = v1; vcur_s[lane + 64] = v2;
        if (writer) {
            uint16_t* Vw = kv.v + (size_t)page * kv.page_stride() + head_off;
            Vw[slot * 128 + lane] = v1; Vw[slot * 128 + lane + 64] = v2;
        }
    }
    wg_barrier_lds();
    Q3_STAMP(1);
    const bool valid = lane < n, cur = lane == slot;
    float a[HPW];
#pragma unroll
    for (int hh = 0; hh < HPW; hh++) a[hh] = 0.0f;
#pragma unroll
    for (int d8 = 0; d8 < 16; d8++) {
        uint4 kk = kreg[d8];
        if (cur) kk = *reinterpret_cast<const uint4*>(&kcur_s[8 * d8]);
        const float kf[8] = { h2f(kk.x & 0xFFFFu), h2f(kk.x >> 16), h2f(kk.y & 0xFFFFu), h2f(kk.y >> 16),
                              h2f(kk.z & 0xFFFFu), h2f(kk.z >> 16), h2f(kk.w & 0xFFFFu), h2f(kk.w >> 16) };
#pragma unroll
        for (int e = 0; e < 8; e++)
#pragma unroll
            for (int hh = 0; hh < HPW; hh++) a[hh] = q3_fmaf(q_s[hh][8 * d8 + e], kf[e], a[hh]);
    }
    Q3_STAMP_AFTER(2, a[0]);
    float L[HPW];
#pragma unroll
    for (int hh = 0; hh < HPW; hh++) {
        const float sc = valid ? a[hh] * scale : -INFINITY;
        const float m = wave_max_bfly(sc);
        const float pv = valid ? q3_expf(sc - m) : 0.0f;
        p_s[hh][lane] = pv;
        L[hh] = wave_sum_bfly(pv);
    }
    wg_barrier_lds();
    Q3_STAMP(3);
    const int dq = n_head * 128;
#pragma unroll
    for (int hh = 0; hh < HPW; hh++) {
        float s01[8], s23[8], y[8];
#pragma unroll
        for (int ww = 0; ww < 4; ww++) {
            float S[8];
#pragma unroll
            for (int i = 0; i < 8; i++) S[i] = 0.0f;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (16 * u < n) {
                    const int jl = 16 * u + 4 * ww + jj;
                    const float pj = p_s[hh][jl];
                    uint4 vv = vreg[ww][u]; // zero beyond the sequence
                    if (jl == slot) vv = *reinterpret_cast<const uint4*>(&vcur_s[dc * 8]);
                    S[0] = q3_fmaf(pj, h2f(vv.x & 0xFFFFu), S[0]); S[1] = q3_fmaf(pj, h2f(vv.x >> 16), S[1]);
                    S[2] = q3_fmaf(pj, h2f(vv.y & 0xFFFFu), S[2]); S[3] = q3_fmaf(pj, h2f(vv.y >> 16), S[3]);
                    S[4] = q3_fmaf(pj, h2f(vv.z & 0xFFFFu), S[4]); S[5] = q3_fmaf(pj, h2f(vv.z >> 16), S[5]);
                    S[6] = q3_fmaf(pj, h2f(vv.w & 0xFFFFu), S[6]); S[7] = q3_fmaf(pj, h2f(vv.w >> 16), S[7]);
                }
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const float a2 = S[i] + xor_lane<16>(S[i]);
                const float T = a2 + xor_lane<32>(a2);
                if (ww == 0) s01[i] = T; else if (ww == 1) s01[i] = s01[i] + T; else if (ww == 2) s23[i] = T; else s23[i] = s23[i] + T;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++) y[i] = (s01[i] + s23[i]) / L[hh];
        const int hq = h0 + hh;
        if (att && jj == 0) { // float-weight models consume the f32 rows
            float* o = att + (size_t)tok * dq + (size_t)hq * 128 + dc * 8;
            *reinterpret_cast<float4*>(o) = make_float4(y[0], y[1], y[2], y[3]);
            *reinterpret_cast<float4*>(o + 4) = make_float4(y[4], y[5], y[6], y[7]);
        }
        float amax = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; i++) amax = fmaxf(amax, q3_fabsf(y[i]));
        amax = fmaxf(amax, xor_lane<1>(amax));
        amax = fmaxf(amax, xor_lane<2>(amax));
        const float dd = amax / 127.0f;
        const float id = (dd != 0.0f) ? (1.0f / dd) : 0.0f;
        if (jj == 0) {
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                lo |= (uint32_t)((int)q3_rintf(y[i] * id) & 0xFF) << (8 * i);
                hi |= (uint32_t)((int)q3_rintf(y[i + 4] * id) & 0xFF) << (8 * i);
            }
            *reinterpret_cast<uint2*>(aq + (size_t)tok * dq + (size_t)hq * 128 + dc * 8) = make_uint2(lo, hi);
            if ((dc & 3) == 0) ad[(size_t)tok * (dq >> 5) + hq * 4 + (dc >> 2)] = f2h(dd);
        }
    }
    Q3_STAMP(6);
    Q3_STAMP_FLUSH();
}
void launch_attention_short(hipStream_t st, const float* qkv, int qkv_stride, int n_head, int n_kv, const float* q_norm_w,
                            const float* k_norm_w, float eps, const float* rope_cos, const float* rope_sin, int n_ctx,
                            const int32_t* mrope_sec, const TokMeta& tm, const KvCache& kv, int layer, int8_t* aq, uint16_t* ad, int ntok, float* att) {
    // one q head per wave while that still leaves the chip under-filled (measured crossover: see DESIGN section 4)
    static const int hpw1_max = [] { const char* e = std::getenv("Q3_SHORT_ATTN_HPW1_MAX"); return e ? atoi(e) : 128; }();
    if (ntok <= hpw1_max)
        hipLaunchKernelGGL(k_attention_short<1>, dim3(n_head, ntok), dim3(64), 0, st, qkv, qkv_stride, n_head, n_kv, q_norm_w, k_norm_w, eps, rope_cos,
                           rope_sin, n_ctx, mrope_sec, tm, kv, layer, aq, ad, att);
    else
        hipLaunchKernelGGL(k_attention_short<2>, dim3(n_kv, ntok), dim3(64), 0, st, qkv, qkv_stride, n_head, n_kv, q_norm_w, k_norm_w, eps, rope_cos,
                           rope_sin, n_ctx, mrope_sec, tm, kv, layer, aq, ad, att);
}

// ===================================================================================================
// projection (assets_manager.rs:383-399 order preserved exactly)
// ===================================================================================================
// Wblk is [n_out/16][n_in][16]: a workgroup's 16 output columns are one contiguous slab.
// Thread = one (token, output) chain (the reference's order: products added one by one in input order, no fma).  The chain is 2048
// dependent adds; what made the first version slow (104 us at 64 tokens) was not the chain but 128 serialised batches of global weight loads behind it.
// Now the workgroup's weight slab [n_in][16] and its 16 activation rows go through LDS in chunks of 512 inputs -- fetched with 16-byte loads one chunk
// ahead (registers), stored transposed ([output][input], [token][input]; row stride 516 floats keeps the 16-byte reads of 16 rows on distinct banks) -- and
// the chain reads 4 inputs per ds_read_b128.
__global__ void __launch_bounds__(256) k_project_mt(const float* __restrict__ x, int x_stride, const float* __restrict__ Wblk,
                                                    const float* __restrict__ b, int n_in, int n_out, float* __restrict__ out,
                                                    int out_stride, int ntok) {
    constexpr int KC = 512, LDP = KC + 4;
    __shared__ __attribute__((aligned(16))) float xs[16][LDP];
    __shared__ __attribute__((aligned(16))) float ws[16][LDP];
    const int ob = blockIdx.x, t0 = blockIdx.y * 16, tid = threadIdx.x;
    const int o16 = tid & 15, tl = tid >> 4;
    const float* wslab = Wblk + (size_t)ob * n_in * 16;
    float4 rx[8], rw[8];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int e = tid + 256 * j;               // x: 16 tokens x 128 float4
            const int t = e >> 7, i4 = e & 127;
            rx[j] = (t0 + t < ntok) ? *reinterpret_cast<const float4*>(x + (size_t)(t0 + t) * x_stride + k0 + 4 * i4) : make_float4(0.f, 0.f, 0.f, 0.f);
            rw[j] = *reinterpret_cast<const float4*>(wslab + (size_t)k0 * 16 + 4 * (size_t)e); // W: 512 inputs x 4 float4 (outputs 4q .. 4q + 3 of input e / 4)
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int e = tid + 256 * j;
            *reinterpret_cast<float4*>(&xs[e >> 7][4 * (e & 127)]) = rx[j];
            const int i = e >> 2, q = (e & 3) * 4;
            ws[q + 0][i] = rw[j].x; ws[q + 1][i] = rw[j].y; ws[q + 2][i] = rw[j].z; ws[q + 3][i] = rw[j].w;
        }
    };
    const int o = ob * 16 + o16;
    float sum = b[o];
    fetch(0);
    for (int k0 = 0; k0 < n_in; k0 += KC) {
        stash();
        __syncthreads();
        if (k0 + KC < n_in) fetch(k0 + KC);
#pragma unroll 8
        for (int i = 0; i < KC; i += 4) {
            const float4 xv = *reinterpret_cast<const float4*>(&xs[tl][i]), wv = *reinterpret_cast<const float4*>(&ws[o16][i]);
            float t = xv.x * wv.x; sum = sum + t;
            t = xv.y * wv.y; sum = sum + t;
            t = xv.z * wv.z; sum = sum + t;
            t = xv.w * wv.w; sum = sum + t;
        }
        __syncthreads();
    }
    if (t0 + tl < ntok) out[(size_t)(t0 + tl) * out_stride + o] = sum;
}
// (one kernel for every token count: at one token it replaced a 27.6 us single-workgroup-per-output-block form, C2 frame 2.816 -> 2.797 ms)
void init_fused_kernel_attributes() {} // nothing needs a per-device opt-in any more (kept: the engine calls it before its first capture)
void launch_project_blk(hipStream_t st, const float* x, int x_stride, const float* Wblk, const float* b, int n_in, int n_out,
                        float* out, int out_stride, int ntok) {
    if (n_in % 512 != 0 || x_stride % 4 != 0 || (reinterpret_cast<uintptr_t>(x) & 15) != 0) throw Error("k_project_mt: n_in must be a multiple of 512 and the rows 16-byte aligned");
    hipLaunchKernelGGL(k_project_mt, dim3(n_out / 16, (ntok + 15) / 16), dim3(256), 0, st, x, x_stride, Wblk, b, n_in, n_out, out, out_stride, ntok);
}

// ===================================================================================================
// code consumers working on argmax keys
// ===================================================================================================
__global__ void __launch_bounds__(256) k_feedback_keys(const float* const* __restrict__ tables, const int64_t* __restrict__ table_rows,
                                                       const u64* __restrict__ keys, int key_stride,
                                                       const float* __restrict__ tts_pad, float* __restrict__ out) {
    const int tok = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    float acc = 0.0f;
#pragma unroll
    for (int q = 0; q < 16; q++) {
        int c = key_code(keys[(size_t)tok * key_stride + q]);
        if (c < 0) c = 0;
        const float v = ((int64_t)c < table_rows[q]) ? tables[q][(size_t)c * 2048 + i] : 0.0f;
        acc = acc + v;
    }
    acc = acc + tts_pad[i];
    out[(size_t)tok * 2048 + i] = acc;
}
void launch_feedback_keys(hipStream_t st, const float* const* tables, const int64_t* table_rows, const u64* keys, int key_stride,
                          const float* tts_pad, float* out, int ntok) {
    hipLaunchKernelGGL(k_feedback_keys, dim3(8, ntok), dim3(256), 0, st, tables, table_rows, keys, key_stride, tts_pad, out);
}

// project(m_hidden) and the pre-projected E_0[code_0] row side by side: predictor pass-A input rows [0,B) and [B,2B)
__global__ void __launch_bounds__(256) k_gather_rows_keys(const float* __restrict__ table, int64_t rows, const u64* __restrict__ keys,
                                                          int key_stride, int row_len, float* __restrict__ dst) {
    const int tok = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= row_len) return;
    int c = key_code(keys[(size_t)tok * key_stride]);
    if (c < 0) c = 0;
    dst[(size_t)tok * row_len + i] = ((int64_t)c < rows) ? table[(size_t)c * row_len + i] : 0.0f;
}
void launch_gather_rows_keys(hipStream_t st, const float* table, int64_t rows, const u64* keys, int key_stride, int row_len,
                             float* dst, int ntok) {
    hipLaunchKernelGGL(k_gather_rows_keys, dim3((row_len + 255) / 256, ntok), dim3(256), 0, st, table, rows, keys, key_stride, row_len, dst);
}

__global__ void k_advance_keys(AdvanceKeysArgs a) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= a.B) return;
    u64* k = a.keys + (size_t)b * 16;
    const bool live = !a.finished[b] && a.n_frames[b] < a.max_frames[b];
    if (live) {
        const int c0 = key_code(k[0]);
        if (c0 == Q3_CODEC_EOS || c0 == Q3_TEXT_EOS) a.finished[b] = 1; // engine.rs:558-561
        else {
            int32_t* dst = a.hist + (size_t)b * a.hist_stride + (size_t)a.n_frames[b] * 16;
            for (int q = 0; q < 16; q++) dst[q] = key_code(k[q]);
            a.n_frames[b] = a.n_frames[b] + 1;
            a.t_slot[b] = a.t_slot[b] + 1;
            a.t_pos[b * 4 + 0] += 1; a.t_pos[b * 4 + 1] += 1; a.t_pos[b * 4 + 2] += 1;
        }
    }
    // hand the next frame its code_0 (argmax of the talker logits just produced) and re-arm every key
    k[0] = a.next_key0[b];
    a.next_key0[b] = pack_key(-INFINITY, 0);
    for (int q = 1; q < 16; q++) k[q] = pack_key(-INFINITY, 0);
}
void launch_advance_keys(hipStream_t st, const AdvanceKeysArgs& a) {
    hipLaunchKernelGGL(k_advance_keys, dim3((a.B + 63) / 64), dim3(64), 0, st, a);
}

} // namespace q3
