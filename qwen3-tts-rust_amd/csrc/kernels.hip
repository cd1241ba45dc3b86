// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the Qwen3-TTS autoregressive decode path.
//
// Replaces what the reference reaches through llama_decode (/root/reference/src/models/llama/mod.rs:442-451)
// and its own CPU glue (assets_manager.rs:383-399 project, engine.rs:622-631 feedback, llama/mod.rs:690-701
// argmax).  All arithmetic follows include/q3tts_spec.h so results are bit-identical to oracle/.
//
// These are HBM/L2-bound integer-dot + small-reduction kernels: the design rules that matter are
// 16-B-per-lane coalesced weight streams (1 KiB per wave instruction), all of a wave's weight loads issued
// before first use, wave-shuffle reductions, and >=256 workgroups per launch.  No MFMA here on purpose.
#include "kernels.h"
#include <cstdlib>
#include "q3_common.h"
#include "../../include/q3tts_spec.h"
#include "kdev.h"
#include "wslice.h"

namespace q3 {

Q3_STAMP_SETTER(set_stamp_buffer)

// =====================================================================================================
// Q8_0 GEMV / skinny GEMM (spec S3).  One wave = R rows x one 256-element segment; LPR = 64/R lanes share
// a row, each lane streaming 16 B (half a 32-block) per load.  A workgroup = up to 8 waves = one
// 2048-element super-segment; segment chains are combined in order through LDS.
// =====================================================================================================
template <int LPR, int MT>
__global__ void __launch_bounds__(512) k_gemv_q8(Q8Mat w, int row0, int nrows, const int8_t* __restrict__ xq,
                                                 const uint16_t* __restrict__ xd, float* __restrict__ out,
                                                 int out_stride, int ntok) {
    constexpr int R = 64 / LPR, BPL = LPR / 2, NLD = 8 / BPL;
    __shared__ float red[8][R * MT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane % R, q = lane / R, half = q & 1, bil = q >> 1;
    const int nseg = w.K >> 8, nb = w.K >> 5;
    const int sseg = blockIdx.y, seg = sseg * 8 + wave;
    const int tok0 = blockIdx.z * MT;
    const bool active = seg < nseg; // wave-uniform
    float acc[MT];
#pragma unroll
    for (int m = 0; m < MT; m++) acc[m] = 0.0f;
    if (active) {
        int row = row0 + blockIdx.x * R + r;
        if (row > w.Npad - 1) row = w.Npad - 1;
        const int rg = row >> 5, r32 = row & 31;
        const uint8_t* base = w.qs + ((size_t)rg * nb + (size_t)seg * 8) * 1024 + half * 512 + r32 * 16;
        uint4 wv[NLD];
#pragma unroll
        for (int i = 0; i < NLD; i++) wv[i] = *reinterpret_cast<const uint4*>(base + (size_t)(i * BPL + bil) * 1024);
        const uint4 dwv = *reinterpret_cast<const uint4*>(w.sc + (((size_t)rg * nseg + seg) * 32 + r32) * 8);
#pragma unroll
        for (int m = 0; m < MT; m++) {
            int tok = tok0 + m;
            if (tok > ntok - 1) tok = ntok - 1;
            const int8_t* xp = xq + (size_t)tok * w.K + seg * 256;
            const uint4 dxv = *reinterpret_cast<const uint4*>(xd + (size_t)tok * nb + seg * 8);
#pragma unroll
            for (int i = 0; i < NLD; i++) {
                const uint4 xv = *reinterpret_cast<const uint4*>(xp + (i * BPL + bil) * 32 + half * 16);
                int isum = dot16(wv[i], xv);
                isum += xor_lane<R>(isum); // the other half of the same 32-block: exact integer add
#pragma unroll
                for (int j = 0; j < BPL; j++) {
                    const int isj = (BPL == 1) ? isum : __shfl(isum, r + 2 * j * R);
                    const int b = i * BPL + j;
                    const float sc = h2f(half_of(dwv, b)) * h2f(half_of(dxv, b));
                    acc[m] = q3_fmaf((float)isj, sc, acc[m]);
                }
            }
        }
        if (q == 0) {
#pragma unroll
            for (int m = 0; m < MT; m++) red[wave][m * R + r] = acc[m];
        }
    }
    wg_barrier_lds();
    for (int t = threadIdx.x; t < R * MT; t += blockDim.x) { // blockDim may be smaller than R*MT (few segments, many tokens)
        const int m = t / R, rr = t % R;
        int nsg = nseg - sseg * 8;
        if (nsg > 8) nsg = 8;
        float v[8]; // all eight reads first, then the in-order adds (a rolled loop pays one LDS round trip per add); rows s >= nsg are read and ignored
#pragma unroll
        for (int s = 0; s < 8; s++) v[s] = red[s][t];
        float S = v[0];
#pragma unroll
        for (int s = 1; s < 8; s++) S = (s < nsg) ? S + v[s] : S;
        const int orow = blockIdx.x * R + rr, tok = tok0 + m;
        if (orow < nrows && tok < ntok) out[((size_t)sseg * ntok + tok) * out_stride + orow] = S;
    }
}

__global__ void k_gemm_q8_tok(Q8Mat w, int row0, int nrows, const int8_t* __restrict__ xq, const uint16_t* __restrict__ xd,
                              float* __restrict__ out, int out_stride, int ntok);
// Register budget: 4 waves per SIMD (128 VGPRs), stated explicitly.  Measured on MI355X at 64 slots: with 2 waves per SIMD (256 VGPRs) the
// kernel has room for a further exact trick -- start the int8 accumulator at the bit pattern of 1.5 * 2^23 so that float(dot) becomes one
// v_pk_add_f32 per pair instead of two v_cvt_f32_i32 -- but the lost occupancy costs more than the 8 VALU instructions per block it saves
// (AR step 5.64 vs 5.54 ms, prefill 52.7 vs 45.8 ms); squeezed into 128 VGPRs the constant tile spills (7.75 ms).
#define Q3_GEMM_Q8_BUDGET __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4)))
template <bool GU>
__global__ void Q3_GEMM_Q8_BUDGET k_gemm_q8_mfma(Q8Mat w, int row0, int nrows, const int8_t* __restrict__ xq, const uint16_t* __restrict__ xd,
                               float* __restrict__ out, int out_stride, int ntok, int ff, int8_t* __restrict__ aq, uint16_t* __restrict__ ad);
template <bool GU>
__global__ void Q3_GEMM_Q8_BUDGET k_gemm_q8_tile1(Q8Mat w, int row0, int nrows, const int8_t* __restrict__ xq, const uint16_t* __restrict__ xd,
                               float* __restrict__ out, int out_stride, int ntok, int ff, int8_t* __restrict__ aq, uint16_t* __restrict__ ad);
template <int TYPE> // register budget cut for 2 waves per SIMD = 256 VGPRs (no spill, 1 workgroup per CU; the 128-VGPR build spilled 200 B per lane in its K loop)
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_gemm_float_mfma(const void* __restrict__ wt, int K, int tile0, int nrows, const float* __restrict__ x, int x_stride, float* __restrict__ out, int out_stride, int ntok);
void init_fused_kernel_attributes();
// dynamic-LDS opt-ins (per device): done once at engine construction so that no attribute call happens inside a stream capture
void init_kernel_attributes() {
    static bool done[64] = {};
    int dev = 0;
    Q3_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev > 63 || done[dev]) return;
    Q3_HIP(hipFuncSetAttribute((const void*)k_gemm_q8_tok, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    constexpr int lds = Q3_SSEG_SEGS * 64 * 33 * (int)sizeof(float); // k_gemm_float_mfma: [8 segments][64 rows][FM_PAD]
    Q3_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_float_mfma<Q3_T_F32>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    Q3_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_float_mfma<Q3_T_F16>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    Q3_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_float_mfma<Q3_T_BF16>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    init_fused_kernel_attributes();
    done[dev] = true;
}
static bool kq_mfma() { static const bool on = [] { const char* e = std::getenv("Q3_KQ_MFMA"); return e ? e[0] == '1' : true; }(); return on; } // 0: K-quant batches through the z-tiled GEMV
template <bool GU, int TS>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_gemm_kq_mfma(Q8Mat w, int row0, int nrows, const int8_t* __restrict__ xq, const uint16_t* __restrict__ xd, float* __restrict__ out, int out_stride,
               int ntok, int ff, int8_t* __restrict__ aq, uint16_t* __restrict__ ad);
// token tiles per launch dimension z: as few as keep >= 256 workgroups in flight (z = 1 streams the weights exactly once)
static int mfma_ztiles(int rowgroups, int nsseg, int ntok) {
    const int ntiles = (ntok + 31) / 32;
    static const int target = [] { const char* e = std::getenv("Q3_MFMA_WGS"); return e ? atoi(e) : 1024; }(); // measured on MI355X at 64 tokens: 128 -> 6.74, 256 -> 6.16, 512..2048 -> 6.14 ms/step (prefill 50.6 -> 48.1 ms at 1024)
    int z = (target + rowgroups * nsseg - 1) / (rowgroups * nsseg);
    return z < 1 ? 1 : (z > ntiles ? ntiles : z);
}

// =====================================================================================================
// Mixed-type form (Q5_K_M files: Q5_K + Q6_K + Q8_0 rows in one fused matrix).  Same wave mapping and the same
// int8 tile layout as k_gemv_q8; the per-block arithmetic follows the weight type of the row group (spec S3):
//   Q8_0: acc = fma(f(isum), dw*dx, acc)
//   Q5_K: acc = fma(d*f(sc*isum1) - dmin*f(m*isum2), dx, acc)      isum2 = sum of the activation block
//   Q6_K: acc = fma(d*f(sc0*isum_lo + sc1*isum_hi), dx, acc)        (the two 16-element halves are the sub-blocks)
// =====================================================================================================
template <int LPR, int MT, int TS>
__global__ void __launch_bounds__(512) k_gemv_kq(Q8Mat w, int row0, int nrows, const int8_t* __restrict__ xq,
                                                 const uint16_t* __restrict__ xd, float* __restrict__ out, int out_stride, int ntok) {
    constexpr int R = 64 / LPR;
    __shared__ float red[8][R * MT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane % R, q = lane / R, half = q & 1, bil = q >> 1;
    const int nseg = w.K >> 8, nb = w.K >> 5;
    const int sseg = blockIdx.y, seg = sseg * 8 + wave;
    const int tok0 = blockIdx.z * MT;
    const bool active = seg < nseg;
    float acc[MT];
#pragma unroll
    for (int m = 0; m < MT; m++) acc[m] = 0.0f;
    if (active) {
        int row = row0 + blockIdx.x * R + r;
        if (row > w.Npad - 1) row = w.Npad - 1;
        wslice_dispatch<TS>(w, row >> 5, [&](auto tag) {
            WSlice<LPR, decltype(tag)::value> ws;
            ws.load(w, row >> 5, row & 31, seg, half, bil);
            ws.finish(half);
#pragma unroll
            for (int m = 0; m < MT; m++) {
                int tok = tok0 + m;
                if (tok > ntok - 1) tok = ntok - 1;
                const uint4 dxv = *reinterpret_cast<const uint4*>(xd + (size_t)tok * nb + seg * 8);
                acc[m] = ws.chain(acc[m], xq + (size_t)tok * w.K + seg * 256, dxv, r, half, bil);
            }
        });
        if (q == 0) {
#pragma unroll
            for (int m = 0; m < MT; m++) red[wave][m * R + r] = acc[m];
        }
    }
    wg_barrier_lds();
    for (int t = threadIdx.x; t < R * MT; t += blockDim.x) {
        const int m = t / R, rr = t % R;
        int nsg = nseg - sseg * 8;
        if (nsg > 8) nsg = 8;
        float v[8]; // all eight reads first, then the in-order adds (a rolled loop pays one LDS round trip per add); rows s >= nsg are read and ignored
#pragma unroll
        for (int s = 0; s < 8; s++) v[s] = red[s][t];
        float S = v[0];
#pragma unroll
        for (int s = 1; s < 8; s++) S = (s < nsg) ? S + v[s] : S;
        const int orow = blockIdx.x * R + rr, tok = tok0 + m;
        if (orow < nrows && tok < ntok) out[((size_t)sseg * ntok + tok) * out_stride + orow] = S;
    }
}
template <int LPR>
static void gemv_kq_mt(hipStream_t st, const Q8Mat& w, int row0, int nrows, const int8_t* xq, const uint16_t* xd, float* out,
                       int out_stride, int ntok) {
    constexpr int R = 64 / LPR;
    const int nseg = w.K >> 8, nsseg = (nseg + 7) / 8, nw = nseg < 8 ? nseg : 8;
    const int mt = ntok == 1 ? 1 : ntok == 2 ? 2 : ntok <= 4 ? 4 : 8;
    dim3 grid((nrows + R - 1) / R, nsseg, (ntok + mt - 1) / mt);
    if (mt == 1) Q3_TS_SWITCH(w, hipLaunchKernelGGL((k_gemv_kq<LPR, 1, TS>), grid, dim3(64 * nw), 0, st, w, row0, nrows, xq, xd, out, out_stride, ntok));
    else if (mt == 2) Q3_TS_SWITCH(w, hipLaunchKernelGGL((k_gemv_kq<LPR, 2, TS>), grid, dim3(64 * nw), 0, st, w, row0, nrows, xq, xd, out, out_stride, ntok));
    else if (mt == 4) Q3_TS_SWITCH(w, hipLaunchKernelGGL((k_gemv_kq<LPR, 4, TS>), grid, dim3(64 * nw), 0, st, w, row0, nrows, xq, xd, out, out_stride, ntok));
    else Q3_TS_SWITCH(w, hipLaunchKernelGGL((k_gemv_kq<LPR, 8, TS>), grid, dim3(64 * nw), 0, st, w, row0, nrows, xq, xd, out, out_stride, ntok));
}

template <int LPR, int MT>
static void gemv_launch(hipStream_t st, const Q8Mat& w, int row0, int nrows, const int8_t* xq, const uint16_t* xd,
                        float* out, int out_stride, int ntok) {
    constexpr int R = 64 / LPR;
    const int nseg = w.K >> 8, nsseg = (nseg + 7) / 8;
    dim3 grid((nrows + R - 1) / R, nsseg, (ntok + MT - 1) / MT);
    const int nw = nseg < 8 ? nseg : 8;
    hipLaunchKernelGGL((k_gemv_q8<LPR, MT>), grid, dim3(64 * nw), 0, st, w, row0, nrows, xq, xd, out, out_stride, ntok);
}
template <int LPR>
static void gemv_launch_mt(hipStream_t st, const Q8Mat& w, int row0, int nrows, const int8_t* xq, const uint16_t* xd,
                           float* out, int out_stride, int ntok) {
    if (ntok == 1) gemv_launch<LPR, 1>(st, w, row0, nrows, xq, xd, out, out_stride, ntok);
    else if (ntok == 2) gemv_launch<LPR, 2>(st, w, row0, nrows, xq, xd, out, out_stride, ntok);
    else if (ntok <= 4) gemv_launch<LPR, 4>(st, w, row0, nrows, xq, xd, out, out_stride, ntok);
    else gemv_launch<LPR, 8>(st, w, row0, nrows, xq, xd, out, out_stride, ntok);
}
void launch_gemv_q8(hipStream_t st, const Q8Mat& w, int row0, int nrows, const int8_t* xq, const uint16_t* xd,
                    float* out, int out_stride, int ntok, int lpr_hint) {
    const int nsseg = ((w.K >> 8) + 7) / 8;
    if (w.rg_type && ntok >= 16 && !lpr_hint && kq_mfma()) { // K-quant rows on the matrix cores (k_gemm_kq_mfma)
        const int nseg = w.K >> 8, nw = nseg < 8 ? nseg : 8;
        const int rgs = (nrows + 31) / 32;
        Q3_TS_SWITCH(w, hipLaunchKernelGGL((k_gemm_kq_mfma<false, TS>), dim3(rgs, nsseg, mfma_ztiles(rgs, nsseg, ntok)), dim3(64 * nw), 0, st, w, row0, nrows, xq, xd, out, out_stride, ntok, 0, (int8_t*)nullptr, (uint16_t*)nullptr));
        return;
    }
    if (w.rg_type) { // mixed K-quant matrix: one kernel handles every type; tokens beyond 8 go to z tiles
        int lpr = lpr_hint;
        if (!lpr) lpr = ((long)(nrows / 32) * nsseg >= 512) ? 2 : ((long)(nrows / 16) * nsseg >= 256) ? 4 : 8;
        // Q6_K rows with 8 lanes per row: measured 14.6-62 us per launch for the predictor's 1024 x 3072 down-projection (2.1 MB; the Q5_K
        // layers of the same shape take 4.7 us) with the same instruction and fetch counts but 4 x the SQ busy cycles
        // (gpurun_out/pmc_kq.txt, round 2); 4 lanes per row does not show it, so Q6_K matrices stop at 4.  Q3_KQ_Q6_LPR8=1 restores 8 for A/B.
        static const bool q6_lpr8 = [] { const char* e = std::getenv("Q3_KQ_Q6_LPR8"); return e && e[0] == '1'; }();
        if (lpr == 8 && !lpr_hint && !q6_lpr8 && wslice_ts(w) != Q3_T_Q5_K && wslice_ts(w) != Q3_T_Q8_0) lpr = 4;
        if (lpr == 2) gemv_kq_mt<2>(st, w, row0, nrows, xq, xd, out, out_stride, ntok);
        else if (lpr == 4) gemv_kq_mt<4>(st, w, row0, nrows, xq, xd, out, out_stride, ntok);
        else gemv_kq_mt<8>(st, w, row0, nrows, xq, xd, out, out_stride, ntok);
        return;
    }
    if (ntok >= 16 && !lpr_hint) { // matrix-core path: exact int8 block dots for 32 tokens x 32 rows per MFMA
        const int nseg = w.K >> 8, nw = nseg < 8 ? nseg : 8;
        const int rgs = (nrows + 31) / 32;
        const int z = mfma_ztiles(rgs, nsseg, ntok);
        // one token tile per workgroup and K <= 1024 (4 waves): the latency-tuned form.  Measured in-graph at 64 tokens (profiles/r03_gemm_stamps.txt): predictor
        // q,k,v 4.67 -> 4.46 us, head 4.50 -> 4.35; with 8 waves per workgroup (K >= 2048) its extra matrix instruction per block costs more than the
        // combine saves (o-projection 5.3 -> 5.9 us), so those shapes stay on the resident-tile kernel
        if (z == (ntok + 31) / 32 && nw <= 4 && !(nrows & 3) && !(out_stride & 3) && !((uintptr_t)out & 15))
            hipLaunchKernelGGL((k_gemm_q8_tile1<false>), dim3(rgs, nsseg, z), dim3(64 * nw), 0, st, w, row0, nrows, xq, xd, out, out_stride, ntok, 0, (int8_t*)nullptr, (uint16_t*)nullptr);
        else
            hipLaunchKernelGGL((k_gemm_q8_mfma<false>), dim3(rgs, nsseg, z), dim3(64 * nw), 0, st, w, row0, nrows, xq, xd, out, out_stride, ntok, 0, (int8_t*)nullptr, (uint16_t*)nullptr);
        return;
    }
    if (ntok > 8 && !lpr_hint) {
        const int nseg = w.K >> 8, nw = nseg < 8 ? nseg : 8;
        const size_t lds = 16 * 2048 + 16 * 128 + 8 * 16 * 64 * 4; // 66 KiB: activations of 16 tokens + segment sums
        init_kernel_attributes(); // (normally done at engine construction; a no-op then)
        hipLaunchKernelGGL(k_gemm_q8_tok, dim3((nrows + 63) / 64, nsseg, (ntok + 15) / 16), dim3(64 * nw), lds, st, w, row0, nrows, xq, xd, out,
                           out_stride, ntok);
        return;
    }
    int lpr = lpr_hint;
    if (!lpr) { // enough workgroups to cover 256 CUs twice, else narrower row groups
        if ((long)(nrows / 32) * nsseg >= 512) lpr = 2;
        else if ((long)(nrows / 16) * nsseg >= 256) lpr = 4;
        else lpr = 8;
    }
    if (lpr == 2) gemv_launch_mt<2>(st, w, row0, nrows, xq, xd, out, out_stride, ntok);
    else if (lpr == 4) gemv_launch_mt<4>(st, w, row0, nrows, xq, xd, out, out_stride, ntok);
    else gemv_launch_mt<8>(st, w, row0, nrows, xq, xd, out, out_stride, ntok);
}

// gate/up GEMM + SwiGLU + int8 quantisation for batched steps (ntok >= 16, pure Q8_0, K = 1024 or 2048)
bool launch_gateup_mfma(hipStream_t st, const Q8Mat& wgu, int ff, const int8_t* xq, const uint16_t* xd, int8_t* aq, uint16_t* ad, int ntok) {
    if (ntok < 16 || (wgu.K != 1024 && wgu.K != 2048) || (ff & 31)) return false;
    const int rgs = ff / 32, ntiles = (ntok + 31) / 32;
    if (wgu.rg_type) { // K-quant gate / up rows (one type for both, see Transformer): the matrix-core form with the SwiGLU + quant epilogue
        if (!kq_mfma() || wgu.nparts != 1) return false;
        int z = mfma_ztiles(rgs, 1, ntok);
        if (z < (ntiles + 3) / 4) z = (ntiles + 3) / 4; // a workgroup parks at most 4 tiles of gate sums
        Q3_TS_SWITCH(wgu, hipLaunchKernelGGL((k_gemm_kq_mfma<true, TS>), dim3(rgs, 1, z), dim3(wgu.K / 4), 0, st, wgu, 0, ff, xq, xd, (float*)nullptr, 0, ntok, ff, aq, ad));
        return true;
    }
    int z = mfma_ztiles(rgs, 1, ntok);
    if (z < (ntiles + 3) / 4) z = (ntiles + 3) / 4; // a workgroup parks at most 4 tiles of gate sums
    if (z == ntiles && !((uintptr_t)aq & 3)) hipLaunchKernelGGL((k_gemm_q8_tile1<true>), dim3(rgs, 1, z), dim3(wgu.K / 4), 0, st, wgu, 0, ff, xq, xd, (float*)nullptr, 0, ntok, ff, aq, ad);
    else hipLaunchKernelGGL((k_gemm_q8_mfma<true>), dim3(rgs, 1, z), dim3(wgu.K / 4), 0, st, wgu, 0, ff, xq, xd, (float*)nullptr, 0, ntok, ff, aq, ad);
    return true;
}

// =====================================================================================================
// Batched-step form (ntok > 8): weight-stationary.  A wave keeps its 32 rows x 256-element segment of weights in
// registers (8 x 16 B per lane) and sweeps up to 32 tokens, so weights are read once per 32 tokens instead of once
// per 8; per (row, token) the arithmetic is the same block chain, and segments are combined in order through LDS.
// =====================================================================================================
__global__ void __launch_bounds__(512) k_gemm_q8_tok(Q8Mat w, int row0, int nrows, const int8_t* __restrict__ xq,
                                                     const uint16_t* __restrict__ xd, float* __restrict__ out, int out_stride,
                                                     int ntok) {
    // lane = row (64 rows per wave: two 32-row groups), wave = segment: no cross-lane traffic in the token sweep
    constexpr int TT = 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int8_t* xq_s = reinterpret_cast<int8_t*>(smem);                                   // [TT][2048]
    uint16_t* xd_s = reinterpret_cast<uint16_t*>(smem + TT * 2048);                   // [TT][64]
    float* red = reinterpret_cast<float*>(smem + TT * 2048 + TT * 128);               // [8][TT][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nseg = w.K >> 8, nb = w.K >> 5;
    const int sseg = blockIdx.y, seg = sseg * 8 + wave;
    const int tok0 = blockIdx.z * TT;
    int nsg = nseg - sseg * 8;
    if (nsg > 8) nsg = 8;
    const int kspan = nsg * 256;
    const int nt = (ntok - tok0) < TT ? (ntok - tok0) : TT;
    const bool active = seg < nseg;
    uint4 wlo[8], whi[8];
    float dwf[8];
    if (active) {
        int row = row0 + blockIdx.x * 64 + lane;
        if (row > w.Npad - 1) row = w.Npad - 1;
        const int rg = row >> 5, r32 = row & 31;
        const uint8_t* base = w.qs + ((size_t)rg * nb + (size_t)seg * 8) * 1024 + r32 * 16;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            wlo[i] = *reinterpret_cast<const uint4*>(base + (size_t)i * 1024);
            whi[i] = *reinterpret_cast<const uint4*>(base + (size_t)i * 1024 + 512);
        }
        const uint4 dwv = *reinterpret_cast<const uint4*>(w.sc + (((size_t)rg * nseg + seg) * 32 + r32) * 8);
#pragma unroll
        for (int i = 0; i < 8; i++) dwf[i] = h2f(half_of(dwv, i));
    }
    for (int e = threadIdx.x; e < nt * (kspan / 16); e += blockDim.x) {
        const int m = e / (kspan / 16), c = e % (kspan / 16);
        *reinterpret_cast<uint4*>(xq_s + m * 2048 + c * 16) =
            *reinterpret_cast<const uint4*>(xq + (size_t)(tok0 + m) * w.K + sseg * 2048 + c * 16);
    }
    for (int e = threadIdx.x; e < nt * nsg; e += blockDim.x) {
        const int m = e / nsg, c = e % nsg;
        *reinterpret_cast<uint4*>(xd_s + m * 64 + c * 8) = *reinterpret_cast<const uint4*>(xd + (size_t)(tok0 + m) * nb + (sseg * 8 + c) * 8);
    }
    __syncthreads();
    if (active) {
        for (int m = 0; m < nt; m++) {
            const int8_t* xp = xq_s + m * 2048 + wave * 256;
            const uint4 dxv = *reinterpret_cast<const uint4*>(xd_s + m * 64 + wave * 8);
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint4 xa = *reinterpret_cast<const uint4*>(xp + i * 32);       // wave-uniform address: LDS broadcast
                const uint4 xb = *reinterpret_cast<const uint4*>(xp + i * 32 + 16);
                const int isum = dot16(wlo[i], xa) + dot16(whi[i], xb);
                const float sc = dwf[i] * h2f(half_of(dxv, i));
                acc = q3_fmaf((float)isum, sc, acc);
            }
            red[(wave * TT + m) * 64 + lane] = acc;
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < nt * 64; t += blockDim.x) {
        const int m = t >> 6, rr = t & 63;
        float S = red[(0 * TT + m) * 64 + rr];
        for (int s = 1; s < nsg; s++) S = S + red[(s * TT + m) * 64 + rr];
        const int orow = blockIdx.x * 64 + rr, tok = tok0 + m;
        if (orow < nrows) out[((size_t)sseg * ntok + tok) * out_stride + orow] = S;
    }
}

// =====================================================================================================
// Batched-step / prefill form on the matrix cores (ntok >= 16): v_mfma_i32_32x32x32_i8 computes, for one 32-block of
// K, the EXACT integer dots of 32 tokens x 32 rows; the per-block f32 scale + fma chain of spec S3 then runs on the
// 16 accumulator values each lane owns.  B operand = a weight tile exactly as stored (lane = half*32 + row), A operand
// = 16 activation bytes of token (lane & 31), half (lane >> 5).  Wave = segment, workgroup = super-segment.
// =====================================================================================================
#ifndef Q3_GEMM_AB
#define Q3_GEMM_AB 4
#endif
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef float f32x32q __attribute__((ext_vector_type(32)));
// The workgroup keeps its weight tile (32 rows x one super-segment) in registers and loops over 32-token tiles
// (tile = blockIdx.z, += gridDim.z), so weights are streamed once per launch when gridDim.z = 1.
// GU = gate/up form for K <= 2048: pass 0 runs the 32 gate rows over the workgroup's tiles and parks the sums in LDS, pass 1
// runs the 32 matching up rows and finishes with SwiGLU + int8 quantisation of the 32-row block (spec S8, S2); the f32
// gate/up matrix never reaches memory.  A GU workgroup handles at most 4 token tiles (launcher picks gridDim.z accordingly).
// The per-block scales d_x[token] * d_w[row] come from the f32 matrix pipe: both factors are f16 values, so the product is exact in
// f32, and it is an outer product -- one v_mfma_f32_32x32x1_2b_f32 (K = 1: one exact product per output, C = 0) delivers the scale tiles of
// two blocks in the C layout of the int8 MFMA (no LDS staging of scales, no v_pk_mul on the VALU-bound chain).
template <bool GU>
__global__ void Q3_GEMM_Q8_BUDGET k_gemm_q8_mfma(Q8Mat w, int row0, int nrows, const int8_t* __restrict__ xq,
                                                      const uint16_t* __restrict__ xd, float* __restrict__ out, int out_stride,
                                                      int ntok, int ff, int8_t* __restrict__ aq, uint16_t* __restrict__ ad) {
    __shared__ __attribute__((aligned(16))) float red[8][32][33];
    __shared__ float gate_s[GU ? 4 : 1][GU ? 1024 : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, half = lane >> 5;
    const int nseg = w.K >> 8, nb = w.K >> 5;
    const int sseg = blockIdx.y, seg = sseg * 8 + wave;
    int nsg = nseg - sseg * 8;
    if (nsg > 8) nsg = 8;
    const bool active = seg < nseg;
    const int ntiles = (ntok + 31) >> 5;
    constexpr int NM = GU ? 2 : 1;
    Q3_STAMP_DECL;
    Q3_STAMP(0);
#pragma unroll 1
    for (int q = 0; q < NM; q++) {
        i32x4v wv[8];
        uint4 dwv = make_uint4(0, 0, 0, 0);
        if (active) {
            int row = row0 + blockIdx.x * 32 + r + q * ff;
            if (row > w.Npad - 1) row = w.Npad - 1;
            const int rg = row >> 5, r32 = row & 31;
            const uint8_t* base = w.qs + ((size_t)rg * nb + (size_t)seg * 8) * 1024 + half * 512 + r32 * 16;
#pragma unroll
            for (int i = 0; i < 8; i++) wv[i] = *reinterpret_cast<const i32x4v*>(base + (size_t)i * 1024);
            dwv = *reinterpret_cast<const uint4*>(w.sc + (((size_t)rg * nseg + seg) * 32 + r32) * 8);
        }
        int lt = 0;
#pragma unroll 1
        for (int tt = blockIdx.z; tt < ntiles; tt += gridDim.z, lt++) {
            const int tok0 = tt * 32;
            if (active) {
                int atok = tok0 + r; // A operand: this lane feeds token (lane & 31)
                if (atok > ntok - 1) atok = ntok - 1;
                const int8_t* xp = xq + (size_t)atok * w.K + seg * 256 + half * 16;
                const uint4 dxa = *reinterpret_cast<const uint4*>(xd + (size_t)atok * nb + seg * 8); // this token's 8 block scales
                // C layout: column = lane & 31 (weight row), C row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5) (token)
                // the scale chain runs on register pairs so it can issue as v_pk_mul_f32 / v_pk_fma_f32 (IEEE per component, same bits)
                f32x2v acc2[8];
#pragma unroll
                for (int g = 0; g < 8; g++) acc2[g] = f32x2v{0.0f, 0.0f};
                // activation blocks fetched at a time: 2 in the gate/up form keeps the kernel inside 128 VGPRs without spilling.  (A per-block 16-register scale
                // tile -- v_mfma_f32_32x32x2_f32 with the second k fed zeros -- frees registers for 4 or 8 blocks in flight; built and measured in round 2: C3 612 / 548
                // vs 616 audio-s/s, the fetch latency is already hidden by the SIMD's other waves, so it was removed again.)
                constexpr int AB = GU ? 2 : Q3_GEMM_AB;
#pragma unroll
                for (int ih = 0; ih < 8 / AB; ih++) {
                    i32x4v av[AB];
#pragma unroll
                    for (int i = 0; i < AB; i++) av[i] = *reinterpret_cast<const i32x4v*>(xp + (AB * ih + i) * 32);
                    f32x32q D;
#pragma unroll
                    for (int i4 = 0; i4 < AB; i4++) {
                        const int i = AB * ih + i4;
                        if ((i4 & 1) == 0) { // scale tiles of blocks i (lanes 0..31 feed it) and i + 1 (lanes 32..63)
#pragma unroll
                            for (int g = 0; g < 32; g++) D[g] = 0.0f;
                            const uint32_t ex = half ? half_of(dxa, i + 1) : half_of(dxa, i), ew = half ? half_of(dwv, i + 1) : half_of(dwv, i);
                            D = __builtin_amdgcn_mfma_f32_32x32x1f32(h2f(ex), h2f(ew), D, 0, 0, 0);
                        }
                        i32x16 c;
#pragma unroll
                        for (int g = 0; g < 16; g++) c[g] = 0;
                        c = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[i4], wv[i], c, 0, 0, 0);
#pragma unroll
                        for (int g4 = 0; g4 < 4; g4++) {
                            const int o = 16 * (i4 & 1) + 4 * g4;
                            const f32x2v sc_a = f32x2v{D[o], D[o + 1]}, sc_b = f32x2v{D[o + 2], D[o + 3]};
                            const f32x2v ca = f32x2v{(float)c[4 * g4], (float)c[4 * g4 + 1]}, cb = f32x2v{(float)c[4 * g4 + 2], (float)c[4 * g4 + 3]};
                            acc2[2 * g4] = __builtin_elementwise_fma(ca, sc_a, acc2[2 * g4]);
                            acc2[2 * g4 + 1] = __builtin_elementwise_fma(cb, sc_b, acc2[2 * g4 + 1]);
                        }
                    }
                }
                if (lt == 0) { if (q == 0) Q3_STAMP_AFTER(1, acc2[7][1]); else Q3_STAMP_AFTER(4, acc2[7][1]); }
#pragma unroll
                for (int g = 0; g < 16; g++) red[wave][(g & 3) + 8 * (g >> 2) + 4 * half][r] = acc2[g >> 1][g & 1];
            }
            wg_barrier_lds();
            if (lt == 0) { if (q == 0) Q3_STAMP(2); else Q3_STAMP(5); }
            // segment sums added in spec order.  All of a round's LDS reads are issued before the first add (the rolled form -- one dependent ds_read per
            // add, trip count nsg -- cost ~250 ns per output: in-kernel stamps, profiles/r03_gemm_stamps.txt); rows s2 >= nsg are read and ignored
#pragma unroll 1
            for (int t0 = threadIdx.x; t0 < 32 * 32; t0 += 2 * blockDim.x) { // two outputs per round; whole 32-lane groups share a token (blockDim % 64 == 0)
                float v[2][8];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int t = (t0 + u * (int)blockDim.x) & 1023, m = t >> 5, rr = t & 31;
#pragma unroll
                    for (int s2 = 0; s2 < 8; s2++) v[u][s2] = red[s2][m][rr];
                }
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int t = t0 + u * (int)blockDim.x, m = t >> 5, rr = t & 31, tok = tok0 + m;
                    if (t >= 32 * 32) break; // (uniform over each 32-lane group)
                    float S = v[u][0];
#pragma unroll
                    for (int s2 = 1; s2 < 8; s2++) S = (s2 < nsg) ? S + v[u][s2] : S;
                    if (!GU) {
                        const int orow = blockIdx.x * 32 + rr;
                        if (orow < nrows && tok < ntok) out[((size_t)sseg * ntok + tok) * out_stride + orow] = S;
                    } else if (q == 0) gate_s[lt][t] = S; // read back by the same thread in pass 1
                    else {
                        const float y = q3_swiglu(gate_s[lt][t], S);
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
            }
            // red is rewritten by the next tile / the next pass; behind the last one the workgroup simply ends
            if (q + 1 < NM || tt + (int)gridDim.z < ntiles) wg_barrier_lds();
            if (q == 0 && lt == 0) Q3_STAMP(3);
        }
    }
    Q3_STAMP(6);
    Q3_STAMP_FLUSH();
}


// -----------------------------------------------------------------------------------------------------
// One-tile form of the batched int8 GEMM: every workgroup owns exactly ONE 32-token tile (gridDim.z = token tiles; decode steps of up to ~128
// sequences, where a launch is latency-bound, not throughput-bound).  Same arithmetic as k_gemm_q8_mfma per (row, token): same block dots, same
// fma chain, same in-order segment sums -> same bits.  What the in-kernel stamps of that kernel showed (profiles/r03_gemm_stamps.txt) and this one removes:
//   * activations arrived in 2 (plain) or 4 (gate/up) separately exposed L2 round trips inside the chain: here all 8 blocks of the token tile are
//     fetched first (they are L2-resident and land before the weights), and the gate/up form fetches them ONCE for both passes.  The registers come
//     from the scale tile: v_mfma_f32_32x32x2_f32 with the second k fed zeros lays ONE block's d_x d_w products out in the C layout (16 registers
//     instead of the 32 of the two-block form; one more matrix instruction per block, on a pipe that is ~10 % busy at these sizes).
//   * the up rows' weight stream started behind the gate combine and its barriers: here it is issued right behind the gate rows' matrix work and
//     flies under them -- the barriers are LDS-only (wg_barrier_lds), they do not drain the stream.
//   * the combine read one ds_read_b32 per add, 1 024 times per tile: here 256 threads read 8 float4 each (rows padded to 36 floats), add the four
//     chains side by side and store 16 B (plain) / one packed dword of int8 (gate/up); blockDim / gridDim come from the arguments, not from a
//     dispatch-packet load behind the barrier; no barrier behind the last combine.
// Launch: grid (row groups, super-segments, token tiles), 64 * min(K / 256, 8) threads; plain form needs nrows % 4 == 0, out_stride % 4 == 0.
// -----------------------------------------------------------------------------------------------------
typedef float f32x16q __attribute__((ext_vector_type(16)));
template <bool GU>
__global__ void Q3_GEMM_Q8_BUDGET k_gemm_q8_tile1(Q8Mat w, int row0, int nrows, const int8_t* __restrict__ xq, const uint16_t* __restrict__ xd,
                                                  float* __restrict__ out, int out_stride, int ntok, int ff, int8_t* __restrict__ aq, uint16_t* __restrict__ ad) {
    __shared__ __attribute__((aligned(16))) float red[8][32][36];
    __shared__ __attribute__((aligned(16))) float gate_s[GU ? 1024 : 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, half = lane >> 5;
    const int nseg = w.K >> 8, nb = w.K >> 5;
    const int sseg = blockIdx.y, seg = sseg * 8 + wave;
    int nsg = nseg - sseg * 8;
    if (nsg > 8) nsg = 8;
    const bool active = seg < nseg;
    const int tok0 = blockIdx.z * 32;
    Q3_STAMP_DECL;
    Q3_STAMP(0);
    i32x4v av[8], wv[8];
    uint4 dxa = make_uint4(0, 0, 0, 0), dwv = dxa;
    // A operand: this lane feeds token (lane & 31), 16 bytes of every block of the wave's segment; + the token's 8 block scales.  The gate/up form fetches
    // them again for the up pass (L2 hits, in flight under the gate combine like the up rows' weights) instead of holding 32 registers across the
    // combine: held, one weight block spills, and a kernel with scratch pays ~2 us per launch.  (The pointer is laundered through an empty asm: the
    // compiler would otherwise prove the second fetch redundant and keep the registers.)
    // NLATE: the gate/up form fetches its last 4 blocks from inside the chain (block 4 + i behind the matrix work of block i: L2 hits with four blocks
    // of work to land under).  All 8 up front does not fit its 128 registers -- 48 of 16-aligned matrix tuples + 72 of operands + lane constants: one
    // operand block spills, and a kernel with ANY scratch was measured ~2 us slower per launch (1, 2: still spills; 4: 126 registers, none).
    constexpr int NLATE = GU ? 4 : 0;
    const int8_t* xp = xq;
    auto load_a = [&]() {
        if (!active) return;
        int atok = tok0 + r;
        if (atok > ntok - 1) atok = ntok - 1;
        xp = xq + (size_t)atok * w.K + seg * 256 + half * 16;
        asm volatile("" : "+v"(xp) :: "memory"); // (pins the fetch: see load_w)
#pragma unroll
        for (int i = 0; i < 8 - NLATE; i++) av[i] = *reinterpret_cast<const i32x4v*>(xp + i * 32);
        dxa = *reinterpret_cast<const uint4*>(xd + (size_t)atok * nb + seg * 8);
    };
    load_a();
    // the wave's 32 rows x 256-element segment: gate rows / the plain rows (q = 0) or the matching up rows (q = 1); blocks [i0, i1) (+ the scales with block 0)
    auto load_w = [&](int q, int i0, int i1) {
        if (!active) return;
        int row = row0 + blockIdx.x * 32 + r + q * ff;
        if (row > w.Npad - 1) row = w.Npad - 1;
        const int rg = row >> 5, r32 = row & 31;
        const uint8_t* base = w.qs + ((size_t)rg * nb + (size_t)seg * 8) * 1024 + half * 512 + r32 * 16;
        // (the weights are read-only, no-alias arguments: their loads may be scheduled anywhere, barriers included, unless the address depends on something
        // that cannot move -- an empty volatile asm with a memory clobber pins this fetch behind the code in front of it)
        asm volatile("" : "+v"(base) :: "memory");
#pragma unroll
        for (int i = 0; i < 8; i++) if (i >= i0 && i < i1) wv[i] = *reinterpret_cast<const i32x4v*>(base + (size_t)i * 1024);
        if (i0 == 0) dwv = *reinterpret_cast<const uint4*>(w.sc + (((size_t)rg * nseg + seg) * 32 + r32) * 8);
    };
    auto chain = [&]() { // 8 exact int8 block dots + the spec's per-block fma chain -> this wave's segment sums in red[wave]
        if (!active) return;
        // C layout: column = lane & 31 (weight row), C row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5) (token)
        f32x2v acc2[8];
#pragma unroll
        for (int g = 0; g < 8; g++) acc2[g] = f32x2v{0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < 8; i++) {
            f32x16q D;
#pragma unroll
            for (int g = 0; g < 16; g++) D[g] = 0.0f;
            // k = 0 (lanes 0..31): d_x[token] x d_w[row], both f16 values -> exact in f32; k = 1 (lanes 32..63): 0 x 0
            const float ex = half ? 0.0f : h2f(half_of(dxa, i)), ew = half ? 0.0f : h2f(half_of(dwv, i));
            D = __builtin_amdgcn_mfma_f32_32x32x2f32(ex, ew, D, 0, 0, 0);
            i32x16 c;
#pragma unroll
            for (int g = 0; g < 16; g++) c[g] = 0;
            c = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[i], wv[i], c, 0, 0, 0);
            if (i < NLATE) {
                const int8_t* xl = xp;
                asm volatile("" : "+v"(xl), "+v"(c) :: "memory"); // behind block i's matrix instruction, not hoisted to the top
                av[8 - NLATE + i] = *reinterpret_cast<const i32x4v*>(xl + (8 - NLATE + i) * 32);
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                const f32x2v sc_a = f32x2v{D[4 * g4], D[4 * g4 + 1]}, sc_b = f32x2v{D[4 * g4 + 2], D[4 * g4 + 3]};
                const f32x2v ca = f32x2v{(float)c[4 * g4], (float)c[4 * g4 + 1]}, cb = f32x2v{(float)c[4 * g4 + 2], (float)c[4 * g4 + 3]};
                acc2[2 * g4] = __builtin_elementwise_fma(ca, sc_a, acc2[2 * g4]);
                acc2[2 * g4 + 1] = __builtin_elementwise_fma(cb, sc_b, acc2[2 * g4 + 1]);
            }
        }
#pragma unroll
        for (int g = 0; g < 16; g++) red[wave][(g & 3) + 8 * (g >> 2) + 4 * half][r] = acc2[g >> 1][g & 1];
    };
    // slot t < 256 owns token m = t >> 3 and the 4 rows c4 .. c4 + 3 of the 32-row block: segment sums added in spec order, four chains side by side.
    // Workgroups of 256 / 512 threads (K >= 1024) give every slot its own thread; narrower ones (K = 256 .. 768) walk the slots.
    auto seg_sums = [&](int m, int c4) -> float4 { // (rows s2 >= nsg are read and ignored; two rounds of 4 reads: all 8 at once costs the gate/up form registers it does not have)
        float4 v[4];
#pragma unroll
        for (int s2 = 0; s2 < 4; s2++) v[s2] = *reinterpret_cast<const float4*>(&red[s2][m][c4]);
        float4 S = v[0];
#pragma unroll
        for (int s2 = 1; s2 < 4; s2++)
            if (s2 < nsg) { S.x = S.x + v[s2].x; S.y = S.y + v[s2].y; S.z = S.z + v[s2].z; S.w = S.w + v[s2].w; }
        if (nsg > 4) { // wave-uniform
            asm volatile("" ::: "memory");
#pragma unroll
            for (int s2 = 0; s2 < 4; s2++) v[s2] = *reinterpret_cast<const float4*>(&red[4 + s2][m][c4]);
#pragma unroll
            for (int s2 = 0; s2 < 4; s2++)
                if (4 + s2 < nsg) { S.x = S.x + v[s2].x; S.y = S.y + v[s2].y; S.z = S.z + v[s2].z; S.w = S.w + v[s2].w; }
        }
        return S;
    };
    const int nthr = 64 * (nseg < 8 ? nseg : 8); // = blockDim.x (from the arguments: no dispatch-packet load behind the barrier)
    load_w(0, 0, 8);
    chain();
    Q3_STAMP(1);
    // the up rows' stream (and the tile's activations again: L2 hits) starts under the gate combine
    if (GU) { load_a(); load_w(1, 0, 8); }
    wg_barrier_lds();
    Q3_STAMP(2);
    if (!GU) {
        for (int t = threadIdx.x; t < 256; t += nthr) {
            const int m = t >> 3, c4 = (t & 7) * 4, tok = tok0 + m;
            const float4 S = seg_sums(m, c4);
            const int orow = blockIdx.x * 32 + c4;
            if (orow < nrows && tok < ntok) *reinterpret_cast<float4*>(out + ((size_t)sseg * ntok + tok) * out_stride + orow) = S;
        }
        Q3_STAMP(6);
        Q3_STAMP_FLUSH();
        return;
    }
    for (int t = threadIdx.x; t < 256; t += nthr) *reinterpret_cast<float4*>(&gate_s[4 * t]) = seg_sums(t >> 3, (t & 7) * 4); // read back by the same thread
    wg_barrier_lds(); // red is rewritten by the up pass
    Q3_STAMP(3);
    chain();
    Q3_STAMP(4);
    wg_barrier_lds();
    Q3_STAMP(5);
    for (int t = threadIdx.x; t < 256; t += nthr) {
        const int m = t >> 3, c4 = (t & 7) * 4, tok = tok0 + m;
        const float4 U = seg_sums(m, c4), G = *reinterpret_cast<const float4*>(&gate_s[4 * t]);
        const float y0 = q3_swiglu(G.x, U.x), y1 = q3_swiglu(G.y, U.y), y2 = q3_swiglu(G.z, U.z), y3 = q3_swiglu(G.w, U.w);
        float amax = fmaxf(fmaxf(q3_fabsf(y0), q3_fabsf(y1)), fmaxf(q3_fabsf(y2), q3_fabsf(y3)));
        amax = fmaxf(amax, xor_lane<4>(amax)); amax = fmaxf(amax, xor_lane<2>(amax)); amax = fmaxf(amax, xor_lane<1>(amax)); // the 8 lanes of the token's 32-row block
        const float dd = amax / 127.0f;
        const float id = (dd != 0.0f) ? (1.0f / dd) : 0.0f;
        if (tok < ntok) {
            const uint32_t pk = ((uint32_t)(int)q3_rintf(y0 * id) & 0xFFu) | (((uint32_t)(int)q3_rintf(y1 * id) & 0xFFu) << 8) |
                                (((uint32_t)(int)q3_rintf(y2 * id) & 0xFFu) << 16) | (((uint32_t)(int)q3_rintf(y3 * id) & 0xFFu) << 24);
            *reinterpret_cast<uint32_t*>(aq + (size_t)tok * ff + blockIdx.x * 32 + c4) = pk;
            if (c4 == 0) ad[(size_t)tok * (ff >> 5) + blockIdx.x] = f2h(dd);
        }
    }
    Q3_STAMP(6);
    Q3_STAMP_FLUSH();
}

// -----------------------------------------------------------------------------------------------------
// K-quant form of the matrix-core GEMM (Q5_K_M files, >= 16 tokens): the packed Q5_K / Q6_K planes are unpacked into the same int8 B operand
// the Q8_0 kernel loads ready-made (WSlice<2, WT>: lane = half * 32 + row, 16 weights of the block), and v_mfma_i32_32x32x32_i8 again delivers the
// exact integer block dots of 32 tokens x 32 rows.  What differs is the chain that follows (spec S3, K-quant rows):
//   Q5_K  acc = fma(d * f(sc_b * idot) - dmin * f(m_b * xsum), dx, acc)     xsum[token] = sum of the activation block -- a second MFMA against an
//                                                                          all-ones B tile puts it in every column of the C layout
//   Q6_K  acc = fma(d * f(s_lo * idot_lo + s_hi * idot_hi), dx, acc)         the two 16-element halves are separate sub-blocks: two MFMAs with the
//         other half's lanes of B zeroed; stored values are q + 32, so idot_x = (stored dot) - 32 * xsum_x (two more MFMAs against half-ones tiles)
// d, dmin, sc_b, m_b belong to the lane's weight row (C column); dx[token, block] comes, as in the Q8_0 kernel, from the f32 matrix pipe: an outer
// product with 1.0 lays the activation scales of two blocks out in the C layout.  Built for 2 waves per SIMD (the extra accumulators do not fit 128 VGPRs).
// Before this kernel the batched path of a Q5_K_M file swept the weights once per 8 tokens (k_gemv_kq z tiles): C3 247 audio-s/s against 620 on Q8_0.
// -----------------------------------------------------------------------------------------------------
template <bool GU, int TS>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_gemm_kq_mfma(Q8Mat w, int row0, int nrows, const int8_t* __restrict__ xq, const uint16_t* __restrict__ xd, float* __restrict__ out, int out_stride,
               int ntok, int ff, int8_t* __restrict__ aq, uint16_t* __restrict__ ad) {
    __shared__ float red[8][32][33];
    __shared__ float gate_s[GU ? 4 : 1][GU ? 1024 : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, half = lane >> 5;
    const int nseg = w.K >> 8, nb = w.K >> 5;
    const int sseg = blockIdx.y, seg = sseg * 8 + wave;
    int nsg = nseg - sseg * 8;
    if (nsg > 8) nsg = 8;
    const bool active = seg < nseg;
    const int ntiles = (ntok + 31) >> 5;
    constexpr int NM = GU ? 2 : 1;
    const int row_first = row0 + blockIdx.x * 32; // (a workgroup's rows sit in one 32-row group: one weight type)
    wslice_dispatch<TS>(w, (row_first > w.Npad - 1 ? w.Npad - 1 : row_first) >> 5, [&](auto tag) {
        constexpr int WT = decltype(tag)::value;
        constexpr bool Q5 = WT == Q3_T_Q5_K, Q6 = WT == Q3_T_Q6_K;
#pragma unroll 1
        for (int q = 0; q < NM; q++) {
            WSlice<2, WT> ws;
            ws.dwv = make_uint4(0, 0, 0, 0); ws.mv = make_uint4(0, 0, 0, 0);
            int row = row_first + r + q * ff;
            if (row > w.Npad - 1) row = w.Npad - 1;
            if (active) { ws.load(w, row >> 5, row & 31, seg, half, 0); ws.finish(half); }
            const float d0 = h2f(half_of(ws.dwv, 0)), d1 = h2f(half_of(ws.dwv, 1));
            int lt = 0;
#pragma unroll 1
            for (int tt = blockIdx.z; tt < ntiles; tt += gridDim.z, lt++) {
                const int tok0 = tt * 32;
                if (active) {
                    int atok = tok0 + r; // A operand: this lane feeds token (lane & 31)
                    if (atok > ntok - 1) atok = ntok - 1;
                    const int8_t* xp = xq + (size_t)atok * w.K + seg * 256 + half * 16;
                    const uint4 dxa = *reinterpret_cast<const uint4*>(xd + (size_t)atok * nb + seg * 8); // this token's 8 block scales
                    float acc[16];
#pragma unroll
                    for (int g = 0; g < 16; g++) acc[g] = 0.0f;
                    const i32x4v ones = i32x4v{0x01010101, 0x01010101, 0x01010101, 0x01010101}, zero4 = i32x4v{0, 0, 0, 0};
#pragma unroll
                    for (int ip = 0; ip < 4; ip++) { // block pairs: one f32 MFMA lays out the activation scales of blocks 2 ip (lanes 0..31) and 2 ip + 1
                        f32x32q D;
#pragma unroll
                        for (int g = 0; g < 32; g++) D[g] = 0.0f;
                        const uint32_t ex = half ? half_of(dxa, 2 * ip + 1) : half_of(dxa, 2 * ip);
                        D = __builtin_amdgcn_mfma_f32_32x32x1f32(h2f(ex), 1.0f, D, 0, 0, 0);
#pragma unroll
                        for (int i2 = 0; i2 < 2; i2++) {
                            const int i = 2 * ip + i2;
                            const i32x4v av = *reinterpret_cast<const i32x4v*>(xp + i * 32);
                            const i32x4v wv = i32x4v{(int)ws.wv[i].x, (int)ws.wv[i].y, (int)ws.wv[i].z, (int)ws.wv[i].w};
                            i32x16 z16;
#pragma unroll
                            for (int g = 0; g < 16; g++) z16[g] = 0;
                            if (Q5) {
                                const i32x16 c = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, wv, z16, 0, 0, 0);
                                const i32x16 cx = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, ones, z16, 0, 0, 0);
                                const int scb = ws.mbyte(i), mb = ws.mbyte(8 + i);
#pragma unroll
                                for (int g = 0; g < 16; g++) {
                                    const int i1 = __mul24(scb, c[g]), i2v = __mul24(mb, cx[g]); // |idot| <= 32 * 31 * 127 and the 6-bit scales fit v_mul_i32_i24 (full rate; v_mul_lo_u32 is quarter rate)
                                    const float a = d0 * (float)i1;
                                    const float a2 = d1 * (float)i2v;
                                    const float diff = a - a2;
                                    acc[g] = q3_fmaf(diff, D[16 * i2 + g], acc[g]);
                                }
                            } else if (Q6) {
                                const i32x4v wlo = half ? zero4 : wv, whi = half ? wv : zero4, olo = half ? zero4 : ones, ohi = half ? ones : zero4;
                                const i32x16 clo = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, wlo, z16, 0, 0, 0);
                                const i32x16 chi = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, whi, z16, 0, 0, 0);
                                const i32x16 xlo = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, olo, z16, 0, 0, 0);
                                const i32x16 xhi = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, ohi, z16, 0, 0, 0);
                                const int s0 = (int)(int8_t)ws.mbyte(2 * i), s1 = (int)(int8_t)ws.mbyte(2 * i + 1);
#pragma unroll
                                for (int g = 0; g < 16; g++) {
                                    const int isum = __mul24(clo[g] - 32 * xlo[g], s0) + __mul24(chi[g] - 32 * xhi[g], s1); // 17-bit x 8-bit factors: exact in the 24-bit multiplier
                                    const float a = d0 * (float)isum;
                                    acc[g] = q3_fmaf(a, D[16 * i2 + g], acc[g]);
                                }
                            } else { // Q8_0 row group of a K-quant matrix: f(idot) * (dw * dx)
                                const i32x16 c = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, wv, z16, 0, 0, 0);
                                const float dwf = h2f(half_of(ws.dwv, i));
#pragma unroll
                                for (int g = 0; g < 16; g++) acc[g] = q3_fmaf((float)c[g], dwf * D[16 * i2 + g], acc[g]);
                            }
                        }
                    }
#pragma unroll
                    for (int g = 0; g < 16; g++) red[wave][(g & 3) + 8 * (g >> 2) + 4 * half][r] = acc[g];
                }
                wg_barrier_lds(); // only LDS is shared here: __syncthreads() would also drain the loads and stores in flight (s_waitcnt vmcnt(0))
                for (int t = threadIdx.x; t < 32 * 32; t += blockDim.x) { // whole 32-lane groups share a token (blockDim % 64 == 0)
                    const int m = t >> 5, rr = t & 31, tok = tok0 + m;
                    float v8[8]; // all reads first, then the adds in segment order (a rolled loop pays one LDS round trip per add; rows s2 >= nsg are read and ignored)
#pragma unroll
                    for (int s2 = 0; s2 < 8; s2++) v8[s2] = red[s2][m][rr];
                    float S = v8[0];
#pragma unroll
                    for (int s2 = 1; s2 < 8; s2++) S = (s2 < nsg) ? S + v8[s2] : S;
                    if (!GU) {
                        const int orow = blockIdx.x * 32 + rr;
                        if (orow < nrows && tok < ntok) out[((size_t)sseg * ntok + tok) * out_stride + orow] = S;
                    } else if (q == 0) gate_s[lt][t] = S; // read back by the same thread in pass 1
                    else {
                        const float y = q3_swiglu(gate_s[lt][t], S);
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
                wg_barrier_lds(); // red is rewritten by the next tile
            }
        }
    });
}


// =====================================================================================================
// residual + RMSNorm + int8 activation quantisation (spec S4, S2, S9).  One wave per token: lane l owns
// elements 256c+4l..+3 of every 256-chunk; fma chain in index order; xor butterfly.
// =====================================================================================================
__global__ void __launch_bounds__(64) k_rmsnorm_quant(NormArgs a) {
    const int tok = blockIdx.x, lane = threadIdx.x;
    const int nch = a.d >> 8;
    const float* hin = a.h_in;
    if (a.idx_keys) hin += (size_t)key_code(a.idx_keys[(size_t)tok * a.idx_stride]) * a.h_stride;
    else if (a.idx) hin += (size_t)a.idx[(size_t)tok * a.idx_stride] * a.h_stride;
    else hin += (size_t)tok * a.h_stride;
    float4 x[8];
    float p = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        if (c < nch) {
            float4 v = *reinterpret_cast<const float4*>(hin + 256 * c + 4 * lane);
            if (a.nparts > 0) { // y = S0; y += S1; ...; h = h + y
                const float* pp = a.parts + (size_t)tok * a.parts_stride + 256 * c + 4 * lane;
                float4 y = *reinterpret_cast<const float4*>(pp);
                for (int s = 1; s < a.nparts; s++) {
                    const float4 z = *reinterpret_cast<const float4*>(pp + (size_t)s * gridDim.x * a.parts_stride);
                    y.x = y.x + z.x; y.y = y.y + z.y; y.z = y.z + z.z; y.w = y.w + z.w;
                }
                v.x = v.x + y.x; v.y = v.y + y.y; v.z = v.z + y.z; v.w = v.w + y.w;
            }
            if (a.h_out) *reinterpret_cast<float4*>(a.h_out + (size_t)tok * a.d + 256 * c + 4 * lane) = v;
            x[c] = v;
            p = q3_fmaf(v.x, v.x, p); p = q3_fmaf(v.y, v.y, p); p = q3_fmaf(v.z, v.z, p); p = q3_fmaf(v.w, v.w, p);
        }
    }
    const float ss = wave_sum_bfly(p);
    const float mean = ss / (float)a.d;
    const float scale = 1.0f / q3_sqrtf(mean + a.eps);
#pragma unroll
    for (int c = 0; c < 8; c++) {
        if (c < nch) {
            const float4 g = *reinterpret_cast<const float4*>(a.g + 256 * c + 4 * lane);
            float4 y;
            y.x = (x[c].x * scale) * g.x; y.y = (x[c].y * scale) * g.y;
            y.z = (x[c].z * scale) * g.z; y.w = (x[c].w * scale) * g.w;
            if (a.xn_out) *reinterpret_cast<float4*>(a.xn_out + (size_t)tok * a.d + 256 * c + 4 * lane) = y;
            float amax = fmaxf(fmaxf(q3_fabsf(y.x), q3_fabsf(y.y)), fmaxf(q3_fabsf(y.z), q3_fabsf(y.w)));
            amax = fmaxf(amax, xor_lane<1>(amax)); amax = fmaxf(amax, xor_lane<2>(amax)); amax = fmaxf(amax, xor_lane<4>(amax));
            const float dd = amax / 127.0f;
            const float id = (dd != 0.0f) ? (1.0f / dd) : 0.0f;
            const int q0 = (int)q3_rintf(y.x * id), q1 = (int)q3_rintf(y.y * id), q2 = (int)q3_rintf(y.z * id), q3v = (int)q3_rintf(y.w * id);
            const uint32_t pk = (uint32_t)(q0 & 0xFF) | ((uint32_t)(q1 & 0xFF) << 8) | ((uint32_t)(q2 & 0xFF) << 16) | ((uint32_t)(q3v & 0xFF) << 24);
            *reinterpret_cast<uint32_t*>(a.xq + (size_t)tok * a.d + 256 * c + 4 * lane) = pk;
            if ((lane & 7) == 0) a.xd[(size_t)tok * (a.d >> 5) + 8 * c + (lane >> 3)] = f2h(dd);
        }
    }
}
void launch_rmsnorm_quant(hipStream_t st, const NormArgs& a, int ntok) {
    hipLaunchKernelGGL(k_rmsnorm_quant, dim3(ntok), dim3(64), 0, st, a);
}

// =====================================================================================================
// per-head RMSNorm + NeoX (M-)RoPE + KV append (spec S4', S5, S6).  One wave per (token, head vector);
// lane l owns the rotation pair (x[l], x[l+64]).
// =====================================================================================================
__global__ void __launch_bounds__(64) k_qk_rope_append(const float* __restrict__ qkv, int qkv_stride, int n_head, int n_kv,
                                                       const float* __restrict__ q_norm_w, const float* __restrict__ k_norm_w,
                                                       float eps, const float* __restrict__ rope_cos,
                                                       const float* __restrict__ rope_sin, int n_ctx,
                                                       const int32_t* __restrict__ mrope_sec, TokMeta tm, KvCache kv,
                                                       int layer, float* __restrict__ qrot) {
    const int hv = blockIdx.x, tok = blockIdx.y, lane = threadIdx.x;
    const float* vec = qkv + (size_t)tok * qkv_stride + (size_t)hv * 128;
    float x1 = vec[lane], x2 = vec[lane + 64];
    const int seq = tm.seq_of(tok), slot = tm.slot_of(tok);
    const int page = kv.page_of(seq, slot >> 6);
    const int ps = slot & 63;
    if (hv < n_head + n_kv) {
        const float* w = hv < n_head ? q_norm_w : k_norm_w;
        float p = x1 * x1;
        p = q3_fmaf(x2, x2, p);
        const float ss = wave_sum_bfly(p);
        const float mean = ss / 128.0f;
        const float scale = 1.0f / q3_sqrtf(mean + eps);
        const float y1 = (x1 * scale) * w[lane], y2 = (x2 * scale) * w[lane + 64];
        int32_t sec[4] = { mrope_sec[0], mrope_sec[1], mrope_sec[2], mrope_sec[3] };
        int pp = tm.pos_of(tok, q3_mrope_stream(lane, sec));
        if (pp < 0) pp = 0;
        if (pp > n_ctx - 1) pp = n_ctx - 1;
        float o1, o2;
        q3_rope_pair(y1, y2, rope_cos[(size_t)pp * 64 + lane], rope_sin[(size_t)pp * 64 + lane], &o1, &o2);
        if (hv < n_head) {
            float* dst = qrot + ((size_t)tok * n_head + hv) * 128;
            dst[lane] = o1; dst[lane + 64] = o2;
        } else {
            const int kh = hv - n_head;
            uint16_t* Kb = kv.k + (size_t)page * kv.page_stride() + (size_t)layer * kv.layer_stride() + (size_t)kh * 8192;
            // element d -> [d/8][pos][d%8]
            Kb[((lane >> 3) * 64 + ps) * 8 + (lane & 7)] = f2h(o1);
            Kb[(((lane + 64) >> 3) * 64 + ps) * 8 + (lane & 7)] = f2h(o2);
        }
    } else {
        const int vh = hv - n_head - n_kv;
        uint16_t* Vb = kv.v + (size_t)page * kv.page_stride() + (size_t)layer * kv.layer_stride() + (size_t)vh * 8192;
        Vb[ps * 128 + lane] = f2h(x1);
        Vb[ps * 128 + lane + 64] = f2h(x2);
    }
}
void launch_qk_rope_append(hipStream_t st, const float* qkv, int qkv_stride, const float*, int n_head, int n_kv,
                           const float* q_norm_w, const float* k_norm_w, float eps, const float* rope_cos,
                           const float* rope_sin, int n_ctx, const int32_t* mrope_sec, const TokMeta& tm,
                           const KvCache& kv, int layer, float* qrot, int ntok) {
    hipLaunchKernelGGL(k_qk_rope_append, dim3(n_head + 2 * n_kv, ntok), dim3(64), 0, st, qkv, qkv_stride, n_head, n_kv,
                       q_norm_w, k_norm_w, eps, rope_cos, rope_sin, n_ctx, mrope_sec, tm, kv, layer, qrot);
}

// =====================================================================================================
// causal GQA decode attention over the paged f16 cache (spec S7).  Workgroup = 4 waves per (head, token):
// wave w scores positions c0+64w+lane of each 256-chunk (K stored [d/8][pos][8] so a wave load is 1 KiB
// contiguous), then accumulates PV chains r = 4w + lane/16 over positions j = r (mod 16).
// =====================================================================================================
__global__ void __launch_bounds__(256) k_attention(const float* __restrict__ qrot, int n_head, int n_kv, TokMeta tm,
                                                   KvCache kv, int layer, float* __restrict__ att,
                                                   int8_t* __restrict__ aq, uint16_t* __restrict__ ad) {
    __shared__ float q_s[128];
    __shared__ float p_s[256];
    __shared__ float red_s[4][128];
    __shared__ float wmax_s[4];
    const int h = blockIdx.x, tok = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int seq = tm.seq_of(tok), n = tm.slot_of(tok) + 1;
    const int kvh = h / (n_head / n_kv);
    const size_t head_off = (size_t)layer * kv.layer_stride() + (size_t)kvh * 8192;
    if (tid < 128) q_s[tid] = qrot[((size_t)tok * n_head + h) * 128 + tid];
    __syncthreads();
    const float scale = 0.08838834764831845f;
    const int jj = lane >> 4, dc = lane & 15;
    float M = 0.0f, L = 0.0f, O[8];
#pragma unroll
    for (int i = 0; i < 8; i++) O[i] = 0.0f;
    for (int c0 = 0; c0 < n; c0 += 256) {
        const int jbase = c0 + wave * 64;
        const bool valid = (jbase + lane) < n;
        float s = -INFINITY;
        if (jbase < n) {
            const int page = kv.page_of(seq, jbase >> 6);
            const uint16_t* Kb = kv.k + (size_t)page * kv.page_stride() + head_off;
            float acc = 0.0f;
#pragma unroll
            for (int d8 = 0; d8 < 16; d8++) {
                const uint4 kk = *reinterpret_cast<const uint4*>(Kb + (d8 * 64 + lane) * 8);
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
        __syncthreads();
        const float mc = fmaxf(fmaxf(wmax_s[0], wmax_s[1]), fmaxf(wmax_s[2], wmax_s[3]));
        const float p = valid ? q3_expf(s - mc) : 0.0f;
        p_s[wave * 64 + lane] = p;
        __syncthreads();
        // PV: 16 chains; this wave owns chains 4*wave+jj
        float S[8];
#pragma unroll
        for (int i = 0; i < 8; i++) S[i] = 0.0f;
        const int cn = (n - c0) < 256 ? (n - c0) : 256;
#pragma unroll 4
        for (int u = 0; u < 16; u++) {
            if (16 * u < cn) {
                const int jl = 16 * u + 4 * wave + jj;
                const int jg = c0 + jl;
                const float pj = p_s[jl];
                uint4 vv = make_uint4(0, 0, 0, 0);
                if (jg < n) {
                    const int page = kv.page_of(seq, jg >> 6);
                    vv = *reinterpret_cast<const uint4*>(kv.v + (size_t)page * kv.page_stride() + head_off + (jg & 63) * 128 + dc * 8);
                }
                S[0] = q3_fmaf(pj, h2f(vv.x & 0xFFFFu), S[0]); S[1] = q3_fmaf(pj, h2f(vv.x >> 16), S[1]);
                S[2] = q3_fmaf(pj, h2f(vv.y & 0xFFFFu), S[2]); S[3] = q3_fmaf(pj, h2f(vv.y >> 16), S[3]);
                S[4] = q3_fmaf(pj, h2f(vv.z & 0xFFFFu), S[4]); S[5] = q3_fmaf(pj, h2f(vv.z >> 16), S[5]);
                S[6] = q3_fmaf(pj, h2f(vv.w & 0xFFFFu), S[6]); S[7] = q3_fmaf(pj, h2f(vv.w >> 16), S[7]);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float a = S[i] + xor_lane<16>(S[i]);   // (S0+S1) | (S2+S3)
            const float T = a + xor_lane<32>(a);          // (S0+S1)+(S2+S3)
            if (jj == 0) red_s[wave][dc * 8 + i] = T;
        }
        __syncthreads();
        if (wave == 0) {
            float a = p_s[lane];                            // spec: a = 0 + p0 (== p0), then + p1, p2, p3
            a = a + p_s[lane + 64]; a = a + p_s[lane + 128]; a = a + p_s[lane + 192];
            const float lc = wave_sum_bfly(a);
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
                const float t = lc * eb;
                L = q3_fmaf(L, ea, t);
#pragma unroll
                for (int i = 0; i < 8; i++) { const float u2 = oc[i] * eb; O[i] = q3_fmaf(O[i], ea, u2); }
                M = mn;
            }
        }
        __syncthreads();
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
            if (att) {
#pragma unroll
                for (int i = 0; i < 8; i++) att[o + i] = y[i];
            }
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
void launch_attention(hipStream_t st, const float* qrot, int n_head, int n_kv, const TokMeta& tm, const KvCache& kv,
                      int layer, float* att, int8_t* aq, uint16_t* ad, int ntok) {
    hipLaunchKernelGGL(k_attention, dim3(n_head, ntok), dim3(256), 0, st, qrot, n_head, n_kv, tm, kv, layer, att, aq, ad);
}

// =====================================================================================================
// SwiGLU + quantise (spec S8, S2): thread handles 4 consecutive elements; 8 threads = one 32-block.
// =====================================================================================================
__global__ void __launch_bounds__(256) k_swiglu_quant(const float* __restrict__ gu, int ff, int8_t* __restrict__ aq,
                                                      uint16_t* __restrict__ ad) {
    const int tok = blockIdx.y;
    const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= ff) return; // ff % 32 == 0 and 8-lane groups are aligned, so whole groups exit together
    const float4 g = *reinterpret_cast<const float4*>(gu + (size_t)tok * 2 * ff + e);
    const float4 u = *reinterpret_cast<const float4*>(gu + (size_t)tok * 2 * ff + ff + e);
    float4 y;
    y.x = q3_swiglu(g.x, u.x); y.y = q3_swiglu(g.y, u.y); y.z = q3_swiglu(g.z, u.z); y.w = q3_swiglu(g.w, u.w);
    float amax = fmaxf(fmaxf(q3_fabsf(y.x), q3_fabsf(y.y)), fmaxf(q3_fabsf(y.z), q3_fabsf(y.w)));
    amax = fmaxf(amax, xor_lane<1>(amax)); amax = fmaxf(amax, xor_lane<2>(amax)); amax = fmaxf(amax, xor_lane<4>(amax));
    const float dd = amax / 127.0f;
    const float id = (dd != 0.0f) ? (1.0f / dd) : 0.0f;
    const uint32_t pk = (uint32_t)((int)q3_rintf(y.x * id) & 0xFF) | ((uint32_t)((int)q3_rintf(y.y * id) & 0xFF) << 8) |
                        ((uint32_t)((int)q3_rintf(y.z * id) & 0xFF) << 16) | ((uint32_t)((int)q3_rintf(y.w * id) & 0xFF) << 24);
    *reinterpret_cast<uint32_t*>(aq + (size_t)tok * ff + e) = pk;
    if ((threadIdx.x & 7) == 0) ad[(size_t)tok * (ff >> 5) + (e >> 5)] = f2h(dd);
}
void launch_swiglu_quant(hipStream_t st, const float* gu, int ff, int8_t* aq, uint16_t* ad, int ntok) {
    hipLaunchKernelGGL(k_swiglu_quant, dim3((ff / 4 + 255) / 256, ntok), dim3(256), 0, st, gu, ff, aq, ad);
}

// =====================================================================================================
// first-max argmax (llama/mod.rs:690-701)
// =====================================================================================================
__global__ void __launch_bounds__(256) k_argmax(const float* __restrict__ logits, int stride, int start, int end,
                                                const int32_t* __restrict__ mask_per_tok, int32_t* __restrict__ out,
                                                int out_stride, int add) {
    __shared__ float bv[256];
    __shared__ int bi[256];
    const int tok = blockIdx.x, t = threadIdx.x;
    const int mask_idx = mask_per_tok ? mask_per_tok[tok] : -1;
    const float* lg = logits + (size_t)tok * stride;
    float mv = -INFINITY;
    int mi = start;
    for (int i = start + t; i < end; i += 256) {
        const float v = (i == mask_idx) ? -INFINITY : lg[i];
        if (v > mv) { mv = v; mi = i; } // ascending i within a thread: first max kept
    }
    bv[t] = mv; bi[t] = mi;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (t < s) {
            const float v2 = bv[t + s]; const int i2 = bi[t + s];
            if (v2 > bv[t] || (v2 == bv[t] && i2 < bi[t])) { bv[t] = v2; bi[t] = i2; }
        }
        __syncthreads();
    }
    if (t == 0) out[(size_t)tok * out_stride] = bi[0] + add;
}
void launch_argmax(hipStream_t st, const float* logits, int stride, int start, int end, const int32_t* mask_per_tok,
                   int32_t* out, int out_stride, int add, int ntok) {
    hipLaunchKernelGGL(k_argmax, dim3(ntok), dim3(256), 0, st, logits, stride, start, end, mask_per_tok, out, out_stride, add);
}

// =====================================================================================================
// 2048 -> 1024 projection with the reference's exact order (assets_manager.rs:383-399): one thread per
// output, bias first, ascending i, separate mul and add.  Wt is the transposed weight [n_in][n_out] so
// that a wave reads 256 contiguous bytes per step.
// =====================================================================================================
__global__ void __launch_bounds__(256) k_project(const float* __restrict__ x, int x_stride, const float* __restrict__ Wt,
                                                 const float* __restrict__ b, int n_in, int n_out,
                                                 float* __restrict__ out, int out_stride) {
    extern __shared__ float xs[];
    const int tok = blockIdx.y, o = blockIdx.x * 256 + threadIdx.x;
    for (int i = threadIdx.x; i < n_in; i += 256) xs[i] = x[(size_t)tok * x_stride + i];
    __syncthreads();
    if (o >= n_out) return;
    float sum = b[o];
#pragma unroll 8
    for (int i = 0; i < n_in; i++) { const float t = xs[i] * Wt[(size_t)i * n_out + o]; sum = sum + t; }
    out[(size_t)tok * out_stride + o] = sum;
}
void launch_project(hipStream_t st, const float* x, int x_stride, const float* Wt, const float* b, int n_in, int n_out,
                    float* out, int out_stride, int ntok) {
    hipLaunchKernelGGL(k_project, dim3((n_out + 255) / 256, ntok), dim3(256), n_in * sizeof(float), st, x, x_stride, Wt, b,
                       n_in, n_out, out, out_stride);
}
template <int CT>
__global__ void __launch_bounds__(256) k_project_table(const float* __restrict__ table, int64_t rows,
                                                       const float* __restrict__ Wt, const float* __restrict__ b,
                                                       int n_in, int n_out, float* __restrict__ out) {
    extern __shared__ float xs[]; // [CT][n_in]
    const int64_t r0 = (int64_t)blockIdx.y * CT;
    const int o = blockIdx.x * 256 + threadIdx.x;
    for (int i = threadIdx.x; i < CT * n_in; i += 256) {
        const int64_t r = r0 + i / n_in;
        xs[i] = r < rows ? table[(size_t)r * n_in + (i % n_in)] : 0.0f;
    }
    __syncthreads();
    if (o >= n_out) return;
    float sum[CT];
#pragma unroll
    for (int c = 0; c < CT; c++) sum[c] = b[o];
    for (int i = 0; i < n_in; i++) {
        const float wv = Wt[(size_t)i * n_out + o];
#pragma unroll
        for (int c = 0; c < CT; c++) { const float t = xs[c * n_in + i] * wv; sum[c] = sum[c] + t; }
    }
#pragma unroll
    for (int c = 0; c < CT; c++) if (r0 + c < rows) out[(size_t)(r0 + c) * n_out + o] = sum[c];
}
void launch_project_table(hipStream_t st, const float* table, int64_t rows, const float* Wt, const float* b, int n_in,
                          int n_out, float* out) {
    constexpr int CT = 4;
    hipLaunchKernelGGL((k_project_table<CT>), dim3((n_out + 255) / 256, (unsigned)((rows + CT - 1) / CT)), dim3(256),
                       CT * n_in * sizeof(float), st, table, rows, Wt, b, n_in, n_out, out);
}

// argmax over stored logits -> argmax key (batched-step path; same first-max semantics as the fused epilogue)
__global__ void __launch_bounds__(256) k_argmax_keys(const float* __restrict__ logits, int stride, int n, const int32_t* __restrict__ mask_per_tok,
                                                     unsigned long long* __restrict__ keys, int key_stride) {
    __shared__ unsigned long long best[256];
    const int tok = blockIdx.x, t = threadIdx.x;
    const int mk = mask_per_tok ? mask_per_tok[tok] : -1;
    unsigned long long k = pack_key(-INFINITY, 0);
    for (int i = t; i < n; i += 256) {
        const float v = logits[(size_t)tok * stride + i];
        if (i != mk && v > -INFINITY) { const unsigned long long c = pack_key(v, i); k = c > k ? c : k; }
    }
    best[t] = k;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) { if (t < s && best[t + s] > best[t]) best[t] = best[t + s]; __syncthreads(); }
    if (t == 0) keys[(size_t)tok * key_stride] = best[0];
}
void launch_argmax_keys(hipStream_t st, const float* logits, int stride, int n, const int32_t* mask_per_tok, unsigned long long* keys,
                        int key_stride, int ntok) {
    hipLaunchKernelGGL(k_argmax_keys, dim3(ntok), dim3(256), 0, st, logits, stride, n, mask_per_tok, keys, key_stride);
}


// =====================================================================================================
// float-weight GEMV (bf16 / f16 / f32 rows, row-major as stored in the GGUF): one wave per output row; lane l owns the
// 8-element sub-chunks l, l+64, ... of the row (16 B of bf16 per load, 1 KiB per wave instruction); the 4 lanes of a
// 32-block combine with two xor shuffles, the 8 block terms of a 256-segment are chained in order, segments and
// super-segments in order (spec S3, float form).
// =====================================================================================================
template <int TYPE> // 0 f32, 1 f16, 30 bf16
__device__ __forceinline__ void load8(const void* row, int e, float* v) {
    if (TYPE == Q3_T_F32) {
        const float4 a = *reinterpret_cast<const float4*>((const float*)row + e), b = *reinterpret_cast<const float4*>((const float*)row + e + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
        const uint4 u = *reinterpret_cast<const uint4*>((const uint16_t*)row + e);
        const uint32_t wds[4] = { u.x, u.y, u.z, u.w };
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (TYPE == Q3_T_BF16) { v[2 * i] = q3_bits_f32(wds[i] << 16); v[2 * i + 1] = q3_bits_f32(wds[i] & 0xFFFF0000u); }
            else { v[2 * i] = h2f(wds[i] & 0xFFFFu); v[2 * i + 1] = h2f(wds[i] >> 16); }
        }
    }
}
__device__ __forceinline__ float quad_xor1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false)); } // quad_perm [1,0,3,2]
__device__ __forceinline__ float quad_xor2(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false)); } // quad_perm [2,3,0,1]
template <typename T> __device__ __forceinline__ float lane_bcast(float v, T lane_const) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane_const)); }
// MT tokens x RW rows per wave: a weight row chunk is loaded once and applied to MT activation rows, and an activation chunk is
// loaded once and applied to RW weight rows (batched steps and prefill: far fewer loads per fma); per (row, token) the arithmetic
// is unchanged.
template <int TYPE, int MT, int RW>
__global__ void __launch_bounds__(256) k_gemv_float(const void* __restrict__ w, int K, int row0, int nrows, const float* __restrict__ x,
                                                    int x_stride, float* __restrict__ out, int out_stride, int ntok) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = (blockIdx.x * 4 + wave) * RW, tok0 = blockIdx.y * MT;
    if (r0 >= nrows) return;
    const size_t esz = TYPE == Q3_T_F32 ? 4 : 2;
    const char* rows[RW];
#pragma unroll
    for (int q = 0; q < RW; q++) { const int rr = r0 + q < nrows ? r0 + q : nrows - 1; rows[q] = (const char*)w + (size_t)(row0 + rr) * K * esz; }
    const int nseg = K >> 8;
    float y[MT][RW], S[MT][RW];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int q = 0; q < RW; q++) { y[m][q] = 0.0f; S[m][q] = 0.0f; }
    for (int e0 = 0; e0 < K; e0 += 512) { // one wave instruction covers 512 elements = 2 segments
        const int e = e0 + 8 * lane;
        float wv[RW][8];
        if (e < K) {
#pragma unroll
            for (int q = 0; q < RW; q++) load8<TYPE>(rows[q], e, wv[q]);
        }
#pragma unroll
        for (int m = 0; m < MT; m++) {
            int tok = tok0 + m;
            if (tok > ntok - 1) tok = ntok - 1;
            const float* xv = x + (size_t)tok * x_stride;
            float4 xa = make_float4(0.f, 0.f, 0.f, 0.f), xb = xa;
            if (e < K) { xa = *reinterpret_cast<const float4*>(xv + e); xb = *reinterpret_cast<const float4*>(xv + e + 4); }
#pragma unroll
            for (int q = 0; q < RW; q++) {
                float bt = 0.0f;
                if (e < K) {
                    float c = 0.0f;
                    c = q3_fmaf(wv[q][0], xa.x, c); c = q3_fmaf(wv[q][1], xa.y, c); c = q3_fmaf(wv[q][2], xa.z, c); c = q3_fmaf(wv[q][3], xa.w, c);
                    c = q3_fmaf(wv[q][4], xb.x, c); c = q3_fmaf(wv[q][5], xb.y, c); c = q3_fmaf(wv[q][6], xb.z, c); c = q3_fmaf(wv[q][7], xb.w, c);
                    const float a = c + quad_xor1(c); // (c0+c1) | (c2+c3)     (DPP quad permutes: no LDS traffic)
                    bt = a + quad_xor2(a);            // (c0+c1)+(c2+c3)
                }
#pragma unroll
                for (int sg = 0; sg < 2; sg++) { // the two segments of this load: lanes [32*sg, 32*sg+32)
                    const int s = (e0 >> 8) + sg;
                    if (s < nseg) {
                        float acc = 0.0f;
#pragma unroll
                        for (int j = 0; j < 8; j++) acc = acc + lane_bcast(bt, 32 * sg + 4 * j); // v_readlane: block sums in order
                        S[m][q] = (s % Q3_SSEG_SEGS == 0) ? acc : S[m][q] + acc;
                        if (s % Q3_SSEG_SEGS == Q3_SSEG_SEGS - 1 || s == nseg - 1) y[m][q] = (s < Q3_SSEG_SEGS) ? S[m][q] : y[m][q] + S[m][q];
                    }
                }
            }
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int q = 0; q < RW; q++) if (tok0 + m < ntok && r0 + q < nrows) out[(size_t)(tok0 + m) * out_stride + r0 + q] = y[m][q];
    }
}
template <int TYPE>
static void gemv_float_mt(hipStream_t st, const FMat& w, int row0, int nrows, const float* x, int x_stride, float* out, int out_stride, int ntok) {
    const int mt = ntok == 1 ? 1 : ntok == 2 ? 2 : ntok <= 4 ? 4 : 8;
    // rows per wave (RW) > 1 shares activation loads between rows; measured at 32 sequences, bf16: RW 1 -> 68.8, 2 -> 68.3, 4 -> 60.2 audio-s/s,
    // so every token count uses one row per wave (the in-order block sums, not the loads, bound this kernel)
    dim3 grid((nrows + 3) / 4, (ntok + mt - 1) / mt);
    if (mt == 1) hipLaunchKernelGGL((k_gemv_float<TYPE, 1, 1>), grid, dim3(256), 0, st, w.w, w.K, row0, nrows, x, x_stride, out, out_stride, ntok);
    else if (mt == 2) hipLaunchKernelGGL((k_gemv_float<TYPE, 2, 1>), grid, dim3(256), 0, st, w.w, w.K, row0, nrows, x, x_stride, out, out_stride, ntok);
    else if (mt == 4) hipLaunchKernelGGL((k_gemv_float<TYPE, 4, 1>), grid, dim3(256), 0, st, w.w, w.K, row0, nrows, x, x_stride, out, out_stride, ntok);
    else hipLaunchKernelGGL((k_gemv_float<TYPE, 8, 1>), grid, dim3(256), 0, st, w.w, w.K, row0, nrows, x, x_stride, out, out_stride, ntok);
}

// -----------------------------------------------------------------------------------------------------
// Many-token form of the float matmul (batched steps, prefill) on the matrix cores.
// v_mfma_f32_32x32x1_2b_f32 has K = 1: every output receives exactly one product per instruction, d = a*b + c with one rounding,
// denormals kept -- bit-identical to fmaf (scripts/check_mfma_f32_k1.hip: 0 mismatches in 409 600 chains incl. 90 000 denormal
// results).  A run of 8 such instructions on one accumulator tile IS the spec's 8-element fma chain for 64 rows x 32 tokens at
// once, with both operands coming from plain vector registers (lane = weight row for A, lane = token for B).
//   * weights: tiled copy FMat::wt [N/64][K/8][64 rows][8 elements] -> one contiguous 1/2 KB load per 8-element chunk;
//   * a wave owns (64 rows, 32 tokens, one 256-element segment): 4 chains x 8 MFMAs per 32-element block, block sum
//     (c0+c1)+(c2+c3) and segment sum on whole accumulator tiles (v_pk_add_f32);
//   * the 8 waves of a workgroup take the 8 segments of one super-segment; segment sums meet in LDS and are added in spec order.
// -----------------------------------------------------------------------------------------------------
// XCD-aware tile order for the matrix-core float kernels: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so the
// token tiles that share one weight tile are given ids that are congruent mod 8 and consecutive on that XCD -- the weight tile is then
// fetched into ONE L2 once and hit by the following token tiles, instead of being fetched by up to 8 L2s (or evicted in between).
//   id -> xcd = id % 8, slot = id / 8;  row tile = (slot / n_tok_tiles) * 8 + xcd;  token tile = slot % n_tok_tiles
__device__ __forceinline__ bool xcd_tile(int n_row_tiles, int n_tok_tiles, int* row_tile, int* tok_tile) {
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    *row_tile = (slot / n_tok_tiles) * 8 + xcd;
    *tok_tile = slot % n_tok_tiles;
    return *row_tile < n_row_tiles;
}
static inline unsigned xcd_grid(int n_row_tiles, int n_tok_tiles) { return (unsigned)(((n_row_tiles + 7) / 8) * 8 * n_tok_tiles); }
typedef float f32x32 __attribute__((ext_vector_type(32)));
template <int TYPE> struct RawChunk { uint4 a; };
template <> struct RawChunk<Q3_T_F32> { uint4 a, b; };
template <int TYPE>
__device__ __forceinline__ RawChunk<TYPE> load_raw(const char* p) {
    RawChunk<TYPE> r;
    r.a = *reinterpret_cast<const uint4*>(p);
    if constexpr (TYPE == Q3_T_F32) r.b = *reinterpret_cast<const uint4*>(p + 16);
    return r;
}
template <int TYPE>
__device__ __forceinline__ void unpack_raw(const RawChunk<TYPE>& r, float* v) {
    if constexpr (TYPE == Q3_T_F32) {
        v[0] = __uint_as_float(r.a.x); v[1] = __uint_as_float(r.a.y); v[2] = __uint_as_float(r.a.z); v[3] = __uint_as_float(r.a.w);
        v[4] = __uint_as_float(r.b.x); v[5] = __uint_as_float(r.b.y); v[6] = __uint_as_float(r.b.z); v[7] = __uint_as_float(r.b.w);
    } else {
        const uint32_t wds[4] = { r.a.x, r.a.y, r.a.z, r.a.w };
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if constexpr (TYPE == Q3_T_BF16) { v[2 * i] = q3_bits_f32(wds[i] << 16); v[2 * i + 1] = q3_bits_f32(wds[i] & 0xFFFF0000u); }
            else { v[2 * i] = h2f(wds[i] & 0xFFFFu); v[2 * i + 1] = h2f(wds[i] >> 16); }
        }
    }
}
constexpr int FM_TOK = 32, FM_PAD = 33; // tokens per workgroup tile; LDS row pitch (conflict-free transposed read)
template <int TYPE>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) k_gemm_float_mfma(const void* __restrict__ wt, int K, int tile0, int nrows, const float* __restrict__ x,
                                                         int x_stride, float* __restrict__ out, int out_stride, int ntok) {
    extern __shared__ float segsum[]; // [8 segments][64 rows][FM_PAD]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int tile, tt;
    if (!xcd_tile((nrows + 63) >> 6, (ntok + FM_TOK - 1) / FM_TOK, &tile, &tt)) return; // (whole workgroup: no barrier is skipped)
    const int tok0 = tt * FM_TOK;
    const int nseg = K >> 8;
    constexpr size_t CH = (TYPE == Q3_T_F32 ? 32 : 16) * 64; // bytes of one 8-element chunk of a 64-row tile
    const char* wbase = (const char*)wt + (size_t)(tile0 + tile) * (size_t)(K >> 3) * CH + (size_t)lane * (CH / 64);
    int tok = tok0 + (lane & 31);
    if (tok > ntok - 1) tok = ntok - 1;
    const float* xrow = x + (size_t)tok * x_stride;
    const int orow = threadIdx.x & 63, otok = threadIdx.x >> 6; // outputs this thread combines: (orow, otok + 8 i), i < 4
    float y[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
    for (int ss = 0; ss * Q3_SSEG_SEGS < nseg; ss++) {
        const int s = ss * Q3_SSEG_SEGS + wave;
        if (s < nseg) {
            f32x32 acc;
#pragma unroll
            for (int v = 0; v < 32; v++) acc[v] = 0.0f;
            const char* wp = wbase + (size_t)s * 32 * CH;
            const float* xp = xrow + (s << 8);
            // operands are fetched one block ahead.  Two blocks ahead for the weights (DEEP) measured slower on MI355X: 200 vs 157 us
            // for the 12288 x 2048 x 256-token prefill GEMM -- the extra live registers cost more than the latency they hide.
            RawChunk<TYPE> raw[4];
            float4 xb[8];
#pragma unroll
            for (int j = 0; j < 4; j++) raw[j] = load_raw<TYPE>(wp + (size_t)j * CH);
#pragma unroll
            for (int j = 0; j < 8; j++) xb[j] = *reinterpret_cast<const float4*>(xp + 4 * j);
#pragma unroll 1
            for (int b = 0; b < 8; b++) {
                const int b1 = b < 7 ? b + 1 : 7; // (the tail re-reads the last block; unused)
                // Rolling prefetch: as soon as chunk j of this block has been multiplied, its registers take chunk j of the NEXT block (weights and
                // activations), three quarters of a block ahead of their use.  The earlier form held a whole second set (next weights + next
                // activations: 48 more VGPRs) next to four 32-register accumulators and spilled 200 B per lane inside this loop.
                f32x32 p01, p23;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float wv[8];
                    unpack_raw<TYPE>(raw[j], wv);
                    const float xv[8] = { xb[2 * j].x, xb[2 * j].y, xb[2 * j].z, xb[2 * j].w, xb[2 * j + 1].x, xb[2 * j + 1].y, xb[2 * j + 1].z, xb[2 * j + 1].w };
                    xb[2 * j] = *reinterpret_cast<const float4*>(xp + b1 * 32 + 8 * j);
                    xb[2 * j + 1] = *reinterpret_cast<const float4*>(xp + b1 * 32 + 8 * j + 4);
                    raw[j] = load_raw<TYPE>(wp + (size_t)(b1 * 4 + j) * CH);
                    f32x32 c;
#pragma unroll
                    for (int v = 0; v < 32; v++) c[v] = 0.0f;
#pragma unroll
                    for (int i = 0; i < 8; i++) c = __builtin_amdgcn_mfma_f32_32x32x1f32(wv[i], xv[i], c, 0, 0, 0);
                    if (j == 0) p01 = c; else if (j == 1) p01 = p01 + c; else if (j == 2) p23 = c; else p23 = p23 + c;
                }
                acc = acc + (p01 + p23);
            }
            float* dst = segsum + (size_t)wave * 64 * FM_PAD;
#pragma unroll
            for (int v = 0; v < 32; v++) {
                const int row = 32 * (v >> 4) + 8 * ((v & 15) >> 2) + 4 * (lane >> 5) + (v & 3);
                dst[row * FM_PAD + (lane & 31)] = acc[v];
            }
        }
        wg_barrier_lds(); // LDS only: a __syncthreads() also drains the weight / activation loads in flight (s_waitcnt vmcnt(0))
        const int nsl = nseg - ss * Q3_SSEG_SEGS < Q3_SSEG_SEGS ? nseg - ss * Q3_SSEG_SEGS : Q3_SSEG_SEGS;
        static_assert(Q3_SSEG_SEGS == 8, "segment combine below is unrolled for 8 segments");
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float a8[8]; // all reads first, then the adds in segment order (a rolled loop pays one LDS round trip per add)
#pragma unroll
            for (int sg = 0; sg < 8; sg++) a8[sg] = segsum[((size_t)(sg < nsl ? sg : 0) * 64 + orow) * FM_PAD + otok + 8 * i];
            float S = a8[0];
#pragma unroll
            for (int sg = 1; sg < 8; sg++) S = sg < nsl ? S + a8[sg] : S;
            y[i] = ss == 0 ? S : y[i] + S;
        }
        wg_barrier_lds();
    }
    const int r = tile * 64 + orow;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int t = tok0 + otok + 8 * i;
        if (t < ntok && r < nrows) out[(size_t)t * out_stride + r] = y[i];
    }
}
// Few-token form (batched decode steps: 12..64 tokens): v_mfma_f32_16x16x1_4b_f32 runs the FOUR chains of a 32-element block as its
// four blocks -- lane (u, i) feeds weight w[row i][8u + step], lane (u, j) feeds x[token j][8u + step] -- so after 8 instructions
// registers v, v+4, v+8, v+12 of a lane hold c0..c3 of the same output and the block sum (c0+c1)+(c2+c3) is four in-lane adds.
// A wave owns (16 rows, 16 tokens, one segment): 64 MFMAs of 8 passes, ~50 registers -> 16x finer tasks than the 32x32 kernel,
// which is what a 32-token step on 256 CUs needs (the 32x32 form leaves most SIMDs idle there).
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int FS_PAD = 17;
template <int TYPE>
__global__ void __launch_bounds__(512) k_gemm_float_mfma16(const void* __restrict__ wt, int K, int tile0, int nrows, const float* __restrict__ x,
                                                           int x_stride, float* __restrict__ out, int out_stride, int ntok) {
    __shared__ float segsum[Q3_SSEG_SEGS][16][FS_PAD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int u = lane >> 4, li = lane & 15;
    int rt, tt; // 16-row tile (relative to tile0 * 4), 16-token tile
    if (!xcd_tile((nrows + 15) >> 4, (ntok + 15) >> 4, &rt, &tt)) return;
    const int tok0 = tt * 16;
    const int nseg = K >> 8;
    constexpr size_t ESZ = TYPE == Q3_T_F32 ? 4 : 2;
    constexpr size_t CH = 8 * ESZ * 64; // bytes of one 8-element chunk of a 64-row tile
    const char* wbase = (const char*)wt + (size_t)(tile0 + (rt >> 2)) * (size_t)(K >> 3) * CH + (size_t)u * CH + (size_t)((rt & 3) * 16 + li) * 8 * ESZ;
    int tok = tok0 + li;
    if (tok > ntok - 1) tok = ntok - 1;
    const float* xrow = x + (size_t)tok * x_stride + 8 * u;
    const int orow = threadIdx.x & 15, otok = (threadIdx.x >> 4) & 15; // threads < 256 combine one output each
    float y = 0.0f;
    for (int ss = 0; ss * Q3_SSEG_SEGS < nseg; ss++) {
        const int s = ss * Q3_SSEG_SEGS + wave;
        if (s < nseg) {
            float acc[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
            const char* wp = wbase + (size_t)s * 32 * CH;
            const float* xp = xrow + (s << 8);
            RawChunk<TYPE> raw[3];
            float4 xa[3][2];
#pragma unroll
            for (int d = 0; d < 2; d++) {
                raw[d] = load_raw<TYPE>(wp + (size_t)(4 * d) * CH);
                xa[d][0] = *reinterpret_cast<const float4*>(xp + 32 * d); xa[d][1] = *reinterpret_cast<const float4*>(xp + 32 * d + 4);
            }
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const int b2 = b < 6 ? b + 2 : 7; // (the tail re-reads the last block; unused)
                raw[(b + 2) % 3] = load_raw<TYPE>(wp + (size_t)(4 * b2) * CH);
                xa[(b + 2) % 3][0] = *reinterpret_cast<const float4*>(xp + 32 * b2); xa[(b + 2) % 3][1] = *reinterpret_cast<const float4*>(xp + 32 * b2 + 4);
                float wv[8];
                unpack_raw<TYPE>(raw[b % 3], wv);
                const float4 x0 = xa[b % 3][0], x1 = xa[b % 3][1];
                const float xv[8] = { x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w };
                f32x16 c;
#pragma unroll
                for (int v = 0; v < 16; v++) c[v] = 0.0f;
#pragma unroll
                for (int i = 0; i < 8; i++) c = __builtin_amdgcn_mfma_f32_16x16x1f32(wv[i], xv[i], c, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; r++) acc[r] = acc[r] + ((c[r] + c[r + 4]) + (c[r + 8] + c[r + 12]));
            }
#pragma unroll
            for (int r = 0; r < 4; r++) segsum[wave][4 * u + r][li] = acc[r];
        }
        wg_barrier_lds();
        if (threadIdx.x < 256) {
            const int nsl = nseg - ss * Q3_SSEG_SEGS < Q3_SSEG_SEGS ? nseg - ss * Q3_SSEG_SEGS : Q3_SSEG_SEGS;
            float a8[8];
#pragma unroll
            for (int sg = 0; sg < 8; sg++) a8[sg] = segsum[sg < nsl ? sg : 0][orow][otok];
            float S = a8[0];
#pragma unroll
            for (int sg = 1; sg < 8; sg++) S = sg < nsl ? S + a8[sg] : S;
            y = ss == 0 ? S : y + S;
        }
        wg_barrier_lds();
    }
    const int r = rt * 16 + orow, t = tok0 + otok;
    if (threadIdx.x < 256 && t < ntok && r < nrows) out[(size_t)t * out_stride + r] = y;
}
// Gate/up pair of the float-weight MLP in one launch: a workgroup computes the 16 gate rows AND the 16 matching up rows for its 16
// tokens (the x operand is loaded once for both), combines the segment sums in spec order and writes silu(g)*u (q3_swiglu) -- the
// [ntok][2 ff] gate/up round trip and the separate SwiGLU launch of the unfused form disappear.  Same arithmetic per output.
template <int TYPE>
__global__ void __launch_bounds__(512) k_gateup_float_mfma16(const void* __restrict__ wt, int K, int ff, const float* __restrict__ x, int x_stride,
                                                             float* __restrict__ act, int ntok) {
    __shared__ float segsum[2][Q3_SSEG_SEGS][16][FS_PAD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int u = lane >> 4, li = lane & 15;
    int rt, tt;
    if (!xcd_tile(ff >> 4, (ntok + 15) >> 4, &rt, &tt)) return;
    const int tok0 = tt * 16;
    const int nseg = K >> 8;
    constexpr size_t ESZ = TYPE == Q3_T_F32 ? 4 : 2;
    constexpr size_t CH = 8 * ESZ * 64;
    const int rtu = rt + (ff >> 4); // the up rows' 16-row tile
    const char* wg = (const char*)wt + (size_t)(rt >> 2) * (size_t)(K >> 3) * CH + (size_t)u * CH + (size_t)((rt & 3) * 16 + li) * 8 * ESZ;
    const char* wu = (const char*)wt + (size_t)(rtu >> 2) * (size_t)(K >> 3) * CH + (size_t)u * CH + (size_t)((rtu & 3) * 16 + li) * 8 * ESZ;
    int tok = tok0 + li;
    if (tok > ntok - 1) tok = ntok - 1;
    const float* xrow = x + (size_t)tok * x_stride + 8 * u;
    const int orow = threadIdx.x & 15, otok = (threadIdx.x >> 4) & 15;
    float yg = 0.0f, yu = 0.0f;
    for (int ss = 0; ss * Q3_SSEG_SEGS < nseg; ss++) {
        const int s = ss * Q3_SSEG_SEGS + wave;
        if (s < nseg) {
            float ag[4] = { 0.0f, 0.0f, 0.0f, 0.0f }, au[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
            const char* pg = wg + (size_t)s * 32 * CH;
            const char* pu = wu + (size_t)s * 32 * CH;
            const float* xp = xrow + (s << 8);
            RawChunk<TYPE> rg[3], ru[3];
            float4 xa[3][2];
#pragma unroll
            for (int d = 0; d < 2; d++) {
                rg[d] = load_raw<TYPE>(pg + (size_t)(4 * d) * CH); ru[d] = load_raw<TYPE>(pu + (size_t)(4 * d) * CH);
                xa[d][0] = *reinterpret_cast<const float4*>(xp + 32 * d); xa[d][1] = *reinterpret_cast<const float4*>(xp + 32 * d + 4);
            }
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const int b2 = b < 6 ? b + 2 : 7;
                rg[(b + 2) % 3] = load_raw<TYPE>(pg + (size_t)(4 * b2) * CH); ru[(b + 2) % 3] = load_raw<TYPE>(pu + (size_t)(4 * b2) * CH);
                xa[(b + 2) % 3][0] = *reinterpret_cast<const float4*>(xp + 32 * b2); xa[(b + 2) % 3][1] = *reinterpret_cast<const float4*>(xp + 32 * b2 + 4);
                float wgv[8], wuv[8];
                unpack_raw<TYPE>(rg[b % 3], wgv); unpack_raw<TYPE>(ru[b % 3], wuv);
                const float4 x0 = xa[b % 3][0], x1 = xa[b % 3][1];
                const float xv[8] = { x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w };
                f32x16 cg, cu;
#pragma unroll
                for (int v = 0; v < 16; v++) { cg[v] = 0.0f; cu[v] = 0.0f; }
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    cg = __builtin_amdgcn_mfma_f32_16x16x1f32(wgv[i], xv[i], cg, 0, 0, 0);
                    cu = __builtin_amdgcn_mfma_f32_16x16x1f32(wuv[i], xv[i], cu, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    ag[r] = ag[r] + ((cg[r] + cg[r + 4]) + (cg[r + 8] + cg[r + 12]));
                    au[r] = au[r] + ((cu[r] + cu[r + 4]) + (cu[r + 8] + cu[r + 12]));
                }
            }
#pragma unroll
            for (int r = 0; r < 4; r++) { segsum[0][wave][4 * u + r][li] = ag[r]; segsum[1][wave][4 * u + r][li] = au[r]; }
        }
        wg_barrier_lds();
        if (threadIdx.x < 256) {
            const int nsl = nseg - ss * Q3_SSEG_SEGS < Q3_SSEG_SEGS ? nseg - ss * Q3_SSEG_SEGS : Q3_SSEG_SEGS;
            float g8[8], u8[8];
#pragma unroll
            for (int sg = 0; sg < 8; sg++) { g8[sg] = segsum[0][sg < nsl ? sg : 0][orow][otok]; u8[sg] = segsum[1][sg < nsl ? sg : 0][orow][otok]; }
            float Sg = g8[0], Su = u8[0];
#pragma unroll
            for (int sg = 1; sg < 8; sg++) { Sg = sg < nsl ? Sg + g8[sg] : Sg; Su = sg < nsl ? Su + u8[sg] : Su; }
            yg = ss == 0 ? Sg : yg + Sg; yu = ss == 0 ? Su : yu + Su;
        }
        wg_barrier_lds();
    }
    const int r = rt * 16 + orow, t = tok0 + otok;
    if (threadIdx.x < 256 && t < ntok && r < ff) act[(size_t)t * ff + r] = q3_swiglu(yg, yu);
}
// row-major -> tiled copy (model load): one thread per 8-element chunk
template <int ESZ>
__global__ void k_tile_rows(const char* __restrict__ w, char* __restrict__ wt, int N, int K) {
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; // ((tile * K/8) + kc) * 64 + lane
    const size_t kc8 = (size_t)(K >> 3);
    const size_t ntile = (size_t)(N + 63) / 64;
    if (id >= ntile * kc8 * 64) return;
    const int lane = (int)(id & 63); const size_t kc = (id >> 6) % kc8, tile = (id >> 6) / kc8;
    const size_t row = tile * 64 + lane;
    uint4 a = make_uint4(0, 0, 0, 0), b = a;
    if (row < (size_t)N) {
        const char* src = w + (row * K + kc * 8) * ESZ;
        a = *reinterpret_cast<const uint4*>(src);
        if (ESZ == 4) b = *reinterpret_cast<const uint4*>(src + 16);
    }
    char* dst = wt + id * 8 * ESZ;
    *reinterpret_cast<uint4*>(dst) = a;
    if (ESZ == 4) *reinterpret_cast<uint4*>(dst + 16) = b;
}
void launch_tile_float(hipStream_t st, const void* w, void* wt, int type, int N, int K) {
    const size_t n = (size_t)((N + 63) / 64) * (K >> 3) * 64;
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (type == Q3_T_F32) hipLaunchKernelGGL((k_tile_rows<4>), dim3(grid), dim3(256), 0, st, (const char*)w, (char*)wt, N, K);
    else hipLaunchKernelGGL((k_tile_rows<2>), dim3(grid), dim3(256), 0, st, (const char*)w, (char*)wt, N, K);
}
template <int TYPE>
static bool gemm_float_mfma(hipStream_t st, const FMat& w, int row0, int nrows, const float* x, int x_stride, float* out, int out_stride, int ntok) {
    // measured on MI355X (bf16, AR step ms): 2 sequences 5.58 vs 6.15 (GEMV), 8 sequences 6.02 vs 11.87 -- the matrix-core form wins
    // from two tokens up even though a 16-token tile is then mostly padding (the step is latency-, not throughput-bound)
    static const int min_tok = [] { const char* e = std::getenv("Q3_FLOAT_MFMA_MIN"); return e ? atoi(e) : 2; }();
    if (!w.wt || min_tok <= 0 || ntok < min_tok || row0 % 64 != 0 || (w.K & 255) != 0 || (x_stride & 3) != 0 || ((uintptr_t)x & 15) != 0) return false;
    constexpr size_t lds = (size_t)Q3_SSEG_SEGS * 64 * FM_PAD * sizeof(float);
    init_kernel_attributes(); // (normally done at engine construction; a no-op then)
    static const int wide_tok = [] { const char* e = std::getenv("Q3_FLOAT_MFMA_WIDE"); return e ? atoi(e) : 96; }();
    if (ntok < wide_tok) { // few tokens: 16 x 16 tiles, 16x more workgroups
        dim3 grid16(xcd_grid((nrows + 15) / 16, (ntok + 15) / 16));
        hipLaunchKernelGGL((k_gemm_float_mfma16<TYPE>), grid16, dim3(512), 0, st, w.wt, w.K, row0 / 64, nrows, x, x_stride, out, out_stride, ntok);
        return true;
    }
    dim3 grid(xcd_grid((nrows + 63) / 64, (ntok + FM_TOK - 1) / FM_TOK));
    hipLaunchKernelGGL((k_gemm_float_mfma<TYPE>), grid, dim3(512), lds, st, w.wt, w.K, row0 / 64, nrows, x, x_stride, out, out_stride, ntok);
    return true;
}
// fused gate/up + SwiGLU for batched steps; false = not applicable (caller runs the matmul and k_swiglu_f32)
bool launch_gateup_float(hipStream_t st, const FMat& wgu, int ff, const float* x, int x_stride, float* act, int ntok) {
    static const int max_tok = [] { const char* e = std::getenv("Q3_FLOAT_GU_FUSED_MAX"); return e ? atoi(e) : 95; }();
    if (!wgu.wt || ntok < 2 || ntok > max_tok || wgu.N != 2 * ff || (ff & 15) != 0 || (wgu.K & 255) != 0 || (x_stride & 3) != 0 || ((uintptr_t)x & 15) != 0) return false;
    dim3 grid(xcd_grid(ff / 16, (ntok + 15) / 16));
    if (wgu.type == Q3_T_F32) hipLaunchKernelGGL((k_gateup_float_mfma16<Q3_T_F32>), grid, dim3(512), 0, st, wgu.wt, wgu.K, ff, x, x_stride, act, ntok);
    else if (wgu.type == Q3_T_F16) hipLaunchKernelGGL((k_gateup_float_mfma16<Q3_T_F16>), grid, dim3(512), 0, st, wgu.wt, wgu.K, ff, x, x_stride, act, ntok);
    else hipLaunchKernelGGL((k_gateup_float_mfma16<Q3_T_BF16>), grid, dim3(512), 0, st, wgu.wt, wgu.K, ff, x, x_stride, act, ntok);
    return true;
}
void launch_gemv_float(hipStream_t st, const FMat& w, int row0, int nrows, const float* x, int x_stride, float* out, int out_stride, int ntok) {
    if (w.type == Q3_T_F32 ? gemm_float_mfma<Q3_T_F32>(st, w, row0, nrows, x, x_stride, out, out_stride, ntok)
        : w.type == Q3_T_F16 ? gemm_float_mfma<Q3_T_F16>(st, w, row0, nrows, x, x_stride, out, out_stride, ntok)
                             : gemm_float_mfma<Q3_T_BF16>(st, w, row0, nrows, x, x_stride, out, out_stride, ntok)) return;
    if (w.type == Q3_T_F32) gemv_float_mt<Q3_T_F32>(st, w, row0, nrows, x, x_stride, out, out_stride, ntok);
    else if (w.type == Q3_T_F16) gemv_float_mt<Q3_T_F16>(st, w, row0, nrows, x, x_stride, out, out_stride, ntok);
    else gemv_float_mt<Q3_T_BF16>(st, w, row0, nrows, x, x_stride, out, out_stride, ntok);
}
__global__ void k_swiglu_f32(const float* __restrict__ gu, int ff, float* __restrict__ out) {
    const int tok = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
    if (e < ff) out[(size_t)tok * ff + e] = q3_swiglu(gu[(size_t)tok * 2 * ff + e], gu[(size_t)tok * 2 * ff + ff + e]);
}
void launch_swiglu_f32(hipStream_t st, const float* gu, int ff, float* out, int ntok) {
    hipLaunchKernelGGL(k_swiglu_f32, dim3((ff + 255) / 256, ntok), dim3(256), 0, st, gu, ff, out);
}

__global__ void k_copy_f32(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
void launch_copy_f32(hipStream_t st, const float* src, float* dst, size_t n) {
    hipLaunchKernelGGL(k_copy_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, n);
}

} // namespace q3
