// codec.hip -- streaming codec decoder on MI355X: codes -> 24 kHz PCM, "Q3TTS-codec-synth" (see codec.h).
//
// Every convolution / linear is one implicit-GEMM kernel on exact-f32 MFMA (v_mfma_f32_32x32x2_f32; bf16 would
// not hold the 1e-4 RMS PCM bar).  Activations are time-major [T][C]; a causal conv with k taps and dilation d
// reads its A operand straight from an "extended" buffer whose first (k-1)*d rows are the stream's history, so
// streaming state lives on the device and chunked decoding equals a full-sequence pass exactly.
// Transposed convs (k = 2*stride) are the same GEMM with two taps and N = stride*cout columns (r, co).
#include "codec.h"
#include "gguf.h"
#include <cmath>
#include <cstdlib>
#include <map>

namespace q3 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { EPI_NONE = 0, EPI_GELU = 1, EPI_RES_SCALE = 2, EPI_RES = 3, EPI_SNAKE = 4 };

struct GemmArgs {
    const float* A; int lda;        // A rows: time-major activations (extended buffer), row stride lda floats
    int cin, dil;                   // K index kk -> tap j = kk / cin, channel ci = kk % cin ; A row = m + j*dil
    const float* W;                 // [N][K] row-major (rearranged at load)
    const float* bias;              // [N] or null
    float* out; int ldo;            // out[m][n]
    int M, N, K;
    int epi; const float* res; int ldr; const float* scale; // EPI_RES_SCALE: out = res + scale[n]*(acc+bias); EPI_RES: res + acc+bias
    const float *snake_ea, *snake_ib;  // EPI_SNAKE: v + ib[n]*sin^2(v*ea[n]) applied to acc+bias
    float* ws;                          // split-K partial slabs [z][M][N]
    // Batched decoding of G streams: row m belongs to segment m / segT.  A rows live in a batched extended buffer whose
    // segments carry a_skip extra (history) rows each; out / res rows get o_skip / r_skip extra FLOATS per segment.
    int a_segT, a_skip; int o_segT; long long o_skip; int r_segT; long long r_skip;
    // Pre-split operands: a row of C values stored as [C x f16 hi][C x f16 lo] (the same 4 C bytes as C floats; hi = f16(x), lo = f16(x - hi)).
    // The SnakeBeta producers (k_snake, the EPI_SNAKE epilogue) write this form once, so the split-f16 GEMMs copy it into LDS instead of
    // redoing two conversions + a subtraction per element for every tap and every column tile (SQ counters: the VALU was 57 % busy in the
    // N = 96 stage, the matrix pipe 20 %; profiles/r02_codec_sq_counters.md).  Kernels without an f16 path read hi + lo back as f32.
    int a_split, o_split;
    int xcd_swizzle;   // k_conv_gemm_h3: XCD-aware tile order (Q3_CODEC_XCD=0 turns it off)
};
#define SEG_NONE 0x3fffffff

// segment of a row; the unsegmented case (SEG_NONE, a wave-uniform test) skips the ~35-instruction integer division: it sat in the epilogue
// of every output element and cost more VALU time than the SnakeBeta sine
__device__ __forceinline__ int seg_of(int row, int segT) { return segT == SEG_NONE ? 0 : row / segT; }
__device__ __forceinline__ size_t out_off(const GemmArgs& g, int row, int col) { return (size_t)row * g.ldo + (size_t)seg_of(row, g.o_segT) * g.o_skip + col; }
typedef _Float16 h4v_ __attribute__((ext_vector_type(4)));
// 4 consecutive channels ci..ci+3 of A row `row` as f32 (split rows: hi + lo)
__device__ __forceinline__ float4 load_a4(const GemmArgs& g, size_t row, int ci) {
    if (!g.a_split) return *reinterpret_cast<const float4*>(g.A + row * g.lda + ci);
    const _Float16* base = reinterpret_cast<const _Float16*>(g.A + row * g.lda);
    const h4v_ hi = *reinterpret_cast<const h4v_*>(base + ci), lo = *reinterpret_cast<const h4v_*>(base + g.cin + ci);
    return make_float4((float)hi[0] + (float)lo[0], (float)hi[1] + (float)lo[1], (float)hi[2] + (float)lo[2], (float)hi[3] + (float)lo[3]);
}
// the same 4 channels as raw hi / lo halfs packed into a float4's bits (x,y = hi; z,w = lo): the f16 GEMMs stage these without arithmetic
__device__ __forceinline__ float4 load_a4_split_raw(const GemmArgs& g, size_t row, int ci) {
    const _Float16* base = reinterpret_cast<const _Float16*>(g.A + row * g.lda);
    const float2 hi = *reinterpret_cast<const float2*>(base + ci), lo = *reinterpret_cast<const float2*>(base + g.cin + ci);
    return make_float4(hi.x, hi.y, lo.x, lo.y);
}
// result element -> memory: plain f32, or the hi / lo half planes of a split row (row width = g.N values)
__device__ __forceinline__ void gemm_store(const GemmArgs& g, int row, int col, float v) {
    const size_t o = out_off(g, row, 0);
    if (!g.o_split) { g.out[o + col] = v; return; }
    _Float16* base = reinterpret_cast<_Float16*>(g.out + o);
    const _Float16 hi = (_Float16)v;
    base[col] = hi; base[g.N + col] = (_Float16)(v - (float)hi);
}

// sin(x) for the SnakeBeta activation.  libm's sinf compiles to ~200 VALU instructions with a Payne-Hanek branch -- in a GEMM epilogue that is
// more work per output than the whole K loop of the narrow stages (16-48 outputs per lane).  This form: two-term Cody-Waite reduction by
// pi/2 (exact for |x| up to ~1e5), degree-7 / degree-6 minimax polynomials on [-pi/4, pi/4]; ~20 instructions, |error| < 1e-7 up to |x| = 1e5
// (checked against double sin on 8.8 M points) (the PCM bar is 1e-4 RMS; the oracle computes sin in double).
__device__ __forceinline__ float snake_sin(float x) {
    const float n = rintf(x * 0.63661977236758134f);              // x * 2/pi
    float r = fmaf(n, -1.57079637050628662f, x);                   // pi/2 = hi + lo: hi = float(pi/2) ...
    r = fmaf(n, 4.37113900018624283e-8f, r);                       // ... lo = pi/2 - hi = -4.37e-8
    const int q = (int)n;
    const float r2 = r * r;
    // sin(r) = r + r^3 (s1 + r^2 (s2 + r^2 s3)),  cos(r) = 1 + r^2 (c1 + r^2 (c2 + r^2 c3))   on |r| <= pi/4
    const float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f), r2 * r, r);
    const float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f), r2 * r2, fmaf(-0.5f, r2, 1.0f));
    const float v = (q & 1) ? cp : sp;
    return (q & 2) ? -v : v;
}
// epilogue shared by all GEMM forms: v = acc (+bias) -> activation / residual -> out
__device__ __forceinline__ float gemm_epilogue(const GemmArgs& g, float v, int row, int col) {
    if (g.bias) v += g.bias[col];
    if (g.epi == EPI_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
    else if (g.epi == EPI_RES_SCALE) v = g.res[(size_t)row * g.ldr + (size_t)seg_of(row, g.r_segT) * g.r_skip + col] + g.scale[col] * v;
    else if (g.epi == EPI_RES) v = g.res[(size_t)row * g.ldr + (size_t)seg_of(row, g.r_segT) * g.r_skip + col] + v;
    else if (g.epi == EPI_SNAKE) { const float sn = snake_sin(v * g.snake_ea[col]); v = v + g.snake_ib[col] * (sn * sn); }
    return v;
}

// the same epilogue for 4 consecutive columns of one row, with 16-byte parameter / residual loads and one 16-byte store (callers check alignment on the host)
typedef _Float16 h4e_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gemm_epilogue_store4(const GemmArgs& g, float4 t, int row, int col) {
    float v[4] = {t.x, t.y, t.z, t.w};
    if (g.bias) { const float4 b = *reinterpret_cast<const float4*>(g.bias + col); v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
    if (g.epi == EPI_GELU) {
#pragma unroll
        for (int c = 0; c < 4; c++) v[c] = 0.5f * v[c] * (1.0f + erff(v[c] * 0.70710678118654752f));
    } else if (g.epi == EPI_RES_SCALE || g.epi == EPI_RES) {
        const float4 rs = *reinterpret_cast<const float4*>(g.res + (size_t)row * g.ldr + (size_t)seg_of(row, g.r_segT) * g.r_skip + col);
        if (g.epi == EPI_RES_SCALE) {
            const float4 sc = *reinterpret_cast<const float4*>(g.scale + col);
            v[0] = rs.x + sc.x * v[0]; v[1] = rs.y + sc.y * v[1]; v[2] = rs.z + sc.z * v[2]; v[3] = rs.w + sc.w * v[3];
        } else { v[0] = rs.x + v[0]; v[1] = rs.y + v[1]; v[2] = rs.z + v[2]; v[3] = rs.w + v[3]; }
    } else if (g.epi == EPI_SNAKE) {
        const float4 ea = *reinterpret_cast<const float4*>(g.snake_ea + col), ib = *reinterpret_cast<const float4*>(g.snake_ib + col);
        const float eav[4] = {ea.x, ea.y, ea.z, ea.w}, ibv[4] = {ib.x, ib.y, ib.z, ib.w};
#pragma unroll
        for (int c = 0; c < 4; c++) { const float sn = snake_sin(v[c] * eav[c]); v[c] = v[c] + ibv[c] * (sn * sn); }
    }
    const size_t o = out_off(g, row, 0);
    if (!g.o_split) *reinterpret_cast<float4*>(g.out + o + col) = make_float4(v[0], v[1], v[2], v[3]);
    else {
        _Float16* base = reinterpret_cast<_Float16*>(g.out + o);
        h4e_ hi, lo;
#pragma unroll
        for (int c = 0; c < 4; c++) { hi[c] = (_Float16)v[c]; lo[c] = (_Float16)(v[c] - (float)hi[c]); }
        *reinterpret_cast<h4e_*>(base + col) = hi;
        *reinterpret_cast<h4e_*>(base + g.N + col) = lo;
    }
}
// host side: every pointer and stride the 4-wide epilogue touches is 16-byte aligned / a multiple of 4 floats
static bool epilogue4_ok(const GemmArgs& g) {
    auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return g.N % 4 == 0 && g.ldo % 4 == 0 && g.o_skip % 4 == 0 && a16(g.out) && (!g.bias || a16(g.bias)) &&
           (!(g.epi == EPI_RES || g.epi == EPI_RES_SCALE) || (g.ldr % 4 == 0 && g.r_skip % 4 == 0 && a16(g.res))) &&
           (g.epi != EPI_RES_SCALE || a16(g.scale)) && (g.epi != EPI_SNAKE || (a16(g.snake_ea) && a16(g.snake_ib)));
}

// Workgroup = 4 waves; WM = waves along M.  Tile (32*WM) x (32*(4/WM)), BK = 16, LDS tiles stored k-major so the
// MFMA fragment reads (lane -> 32 consecutive m or n) are conflict-free.  The next K tile is fetched into registers
// while the current one is multiplied.  gridDim.z > 1 = split-K: each z writes a partial slab (no epilogue) that
// k_splitk_reduce (or the consumer) sums in slab order.
template <int WM>
__global__ void __launch_bounds__(256) k_conv_gemm(GemmArgs g) {
    constexpr int WN = 4 / WM, BM = 32 * WM, BN = 32 * WN, BK = 16;
    constexpr int NA = BM * 4 / 256 > 0 ? BM * 4 / 256 : 1, NB = BN * 4 / 256 > 0 ? BN * 4 / 256 : 1;
    __shared__ float As[BK][BM];
    __shared__ float Bs[BK][BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int ksplit = gridDim.z, kper = g.K / ksplit, kbeg = blockIdx.z * kper, kend = kbeg + kper;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;
    float4 ra[NA], rb[NB];
    int arow[NA]; // A row of this thread's i-th fetch (tap 0), including the history rows of the segments before it
#pragma unroll
    for (int i = 0; i < NA; i++) { const int mm = m0 + (tid + i * 256) % BM; arow[i] = mm + seg_of(mm, g.a_segT) * g.a_skip; }
    auto fetch = [&](int k0) {
        const int j = k0 / g.cin, ci0 = k0 % g.cin; // BK divides cin, so a K tile never straddles taps
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int e = tid + i * 256, r = e % BM, qd = e / BM;
            ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < BM * 4 && m0 + r < g.M) ra[i] = load_a4(g, (size_t)(arow[i] + j * g.dil), ci0 + 4 * qd);
        }
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const int e = tid + i * 256, r = e % BN, qd = e / BN;
            rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < BN * 4 && n0 + r < g.N) rb[i] = *reinterpret_cast<const float4*>(g.W + (size_t)(n0 + r) * g.K + k0 + 4 * qd);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int e = tid + i * 256, r = e % BM, qd = e / BM;
            if (e < BM * 4) { As[4 * qd + 0][r] = ra[i].x; As[4 * qd + 1][r] = ra[i].y; As[4 * qd + 2][r] = ra[i].z; As[4 * qd + 3][r] = ra[i].w; }
        }
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const int e = tid + i * 256, r = e % BN, qd = e / BN;
            if (e < BN * 4) { Bs[4 * qd + 0][r] = rb[i].x; Bs[4 * qd + 1][r] = rb[i].y; Bs[4 * qd + 2][r] = rb[i].z; Bs[4 * qd + 3][r] = rb[i].w; }
        }
    };
    fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        stash();
        __syncthreads();
        if (k0 + BK < kend) fetch(k0 + BK);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float a = As[kk + (lane >> 5)][wm * 32 + (lane & 31)];
            const float b = Bs[kk + (lane >> 5)][wn * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int col = n0 + wn * 32 + (lane & 31);
    if (col < g.N) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < g.M) {
                if (ksplit > 1) g.ws[((size_t)blockIdx.z * g.M + row) * g.N + col] = acc[r];
                else gemm_store(g, row, col, gemm_epilogue(g, acc[r], row, col));
            }
        }
    }
}
// Split-f16 form of the same GEMM for large M: every f32 operand x is carried as hi = f16(x), lo = f16(x - hi) (22 significant bits),
// and a*b is accumulated in f32 as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_32x32x16_f16 -- three passes at the 16x f16 MFMA
// rate instead of one at the f32 rate.  Weights are split once at load (Wh/Wl, [N][K] f16); activations are split while the tile
// is staged into LDS.  Relative error per product ~2^-22 (f32: 2^-24); PCM stays within 1e-5 RMS of the double-precision oracle.
// Same tiling, implicit-GEMM addressing, split-K and epilogues as k_conv_gemm<2>; BK = 32 (divides every cin of the decoder).
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
// EPL = 1: finished tile (or split-K partial) through LDS, 16-byte epilogue accesses (see k_conv_gemm_h3)
// SPLIT = pre-split operands (g.a_split); as in k_conv_gemm_h3 every K-loop load is unconditional (clamped rows) and the loop's barriers wait for LDS only
template <int MR, int EPL, bool SPLIT> // MR = 32-row tiles per wave along M: workgroup tile (64*MR) x 64; MR = 2 halves the weight re-reads and the LDS traffic per MFMA
__global__ void __launch_bounds__(256) k_conv_gemm_h(GemmArgs g, const _Float16* __restrict__ Wh, const _Float16* __restrict__ Wl) {
    constexpr int BM = 64 * MR, BN = 64, BK = 32, LD = 40; // LD: padded row stride (f16) of the LDS tiles
    constexpr int NA = BM * 8 / 256;                        // float4 fetches of A per thread per K tile
    constexpr int CLD = BN + 4;
    struct Stage { _Float16 Ah[BM][LD], Al[BM][LD], Bh[BN][LD], Bl[BN][LD]; };
    union Shared { Stage s; float Ct[EPL ? BM : 1][CLD]; };   // the finished tile reuses the operand tiles' space
    __shared__ __attribute__((aligned(16))) Shared sh;
    auto& Ah = sh.s.Ah; auto& Al = sh.s.Al; auto& Bh = sh.s.Bh; auto& Bl = sh.s.Bl;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int ksplit = gridDim.z, kper = g.K / ksplit, kbeg = blockIdx.z * kper, kend = kbeg + kper;
    f32x16 acc[MR];
#pragma unroll
    for (int t = 0; t < MR; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[t][i] = 0.0f;
    float4 ra[NA];
    uint4 rh, rl;
    int arow[NA];
#pragma unroll
    for (int i = 0; i < NA; i++) { int mm = m0 + (tid + i * 256) / 8; mm = mm < g.M ? mm : g.M - 1; arow[i] = mm + seg_of(mm, g.a_segT) * g.a_skip; }
    const int wrow = (n0 + tid / 4 < g.N) ? n0 + tid / 4 : g.N - 1, wk = 8 * (tid & 3); // columns beyond N re-read row N - 1 (their outputs are never stored)
    auto fetch = [&](int k0) {
        const int j = k0 / g.cin, ci0 = k0 % g.cin; // BK divides cin, so a K tile never straddles taps
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int kq = (tid + i * 256) % 8;
            const size_t row = (size_t)(arow[i] + j * g.dil);
            if constexpr (SPLIT) ra[i] = load_a4_split_raw(g, row, ci0 + 4 * kq);
            else ra[i] = *reinterpret_cast<const float4*>(g.A + row * g.lda + ci0 + 4 * kq);
        }
        rh = *reinterpret_cast<const uint4*>(Wh + (size_t)wrow * g.K + k0 + wk);
        rl = *reinterpret_cast<const uint4*>(Wl + (size_t)wrow * g.K + k0 + wk);
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int e = tid + i * 256, r = e / 8, kq = e % 8;
            if constexpr (SPLIT) { // operands arrive as hi / lo halfs: a copy
                *reinterpret_cast<float2*>(&Ah[r][4 * kq]) = make_float2(ra[i].x, ra[i].y);
                *reinterpret_cast<float2*>(&Al[r][4 * kq]) = make_float2(ra[i].z, ra[i].w);
            } else {
                const float x[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
                h4v hi, lo;
#pragma unroll
                for (int c = 0; c < 4; c++) { hi[c] = (_Float16)x[c]; lo[c] = (_Float16)(x[c] - (float)hi[c]); }
                *reinterpret_cast<h4v*>(&Ah[r][4 * kq]) = hi;
                *reinterpret_cast<h4v*>(&Al[r][4 * kq]) = lo;
            }
        }
        *reinterpret_cast<uint4*>(&Bh[tid / 4][wk]) = rh;
        *reinterpret_cast<uint4*>(&Bl[tid / 4][wk]) = rl;
    };
    auto multiply = [&]() {
#pragma unroll
        for (int kk = 0; kk < BK; kk += 16) {
            const int ko = kk + 8 * (lane >> 5); // operand lane l: row l & 31, 8 consecutive k of half l >> 5 (A and B use the same split)
            const h8v bh = *reinterpret_cast<const h8v*>(&Bh[wn * 32 + (lane & 31)][ko]);
            const h8v bl = *reinterpret_cast<const h8v*>(&Bl[wn * 32 + (lane & 31)][ko]);
#pragma unroll
            for (int t = 0; t < MR; t++) {
                const int ar = (wm * MR + t) * 32 + (lane & 31);
                const h8v ah = *reinterpret_cast<const h8v*>(&Ah[ar][ko]);
                const h8v al = *reinterpret_cast<const h8v*>(&Al[ar][ko]);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
            }
        }
    };
    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    fetch(kbeg);
    int k0 = kbeg;
    for (; k0 + BK < kend; k0 += BK) { // every refill unconditional (see k_conv_gemm_h3)
        stash();
        lds_barrier();
        fetch(k0 + BK);
        multiply();
        lds_barrier();
    }
    stash();
    lds_barrier();
    multiply();
    __syncthreads(); // the epilogue reuses the operand tiles
    if constexpr (EPL != 0) {
        auto& Ct = sh.Ct; // (the K loop ended on a barrier: the operand tiles are dead)
#pragma unroll
        for (int t = 0; t < MR; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) Ct[(wm * MR + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][wn * 32 + (lane & 31)] = acc[t][r];
        __syncthreads();
        for (int e = tid; e < BM * (BN / 4); e += 256) {
            const int rl = e / (BN / 4), cl = (e % (BN / 4)) * 4, row = m0 + rl, col = n0 + cl;
            if (row >= g.M || col >= g.N) continue; // N % 4 == 0 (host check): a 4-group is inside or outside as a whole
            const float4 t4 = *reinterpret_cast<const float4*>(&Ct[rl][cl]);
            if (ksplit > 1) *reinterpret_cast<float4*>(g.ws + ((size_t)blockIdx.z * g.M + row) * g.N + col) = t4;
            else gemm_epilogue_store4(g, t4, row, col);
        }
    } else {
        const int col = n0 + wn * 32 + (lane & 31);
        if (col < g.N) {
#pragma unroll
            for (int t = 0; t < MR; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = m0 + (wm * MR + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (row < g.M) {
                        if (ksplit > 1) g.ws[((size_t)blockIdx.z * g.M + row) * g.N + col] = acc[t][r];
                        else gemm_store(g, row, col, gemm_epilogue(g, acc[t][r], row, col));
                    }
                }
        }
    }
}
// 128 x (32 NT) form: the 4 waves stack along M (32 rows each) and every wave owns NT column tiles, so a k-step of 16 reads 2 + 2 NT LDS
// fragments for 3 NT matrix instructions (NT = 3: 8 for 9, against 4 for 3 in k_conv_gemm_h<1>), a 96-column stage has no padded columns
// (BN = 64 wastes a quarter of the matrix work at N = 96) and the activation tile is fetched once per 96 output columns instead of once per 64.
// One K tile in flight and ~110 VGPRs / 36 KB of LDS keep 4 workgroups per CU -- the middle ground between k_conv_gemm_h<1> (8 per CU, the
// L2 -> LDS traffic 2.3 x larger) and a 256 x 96 register-blocked tile (2 per CU, spills; removed).  Reads pre-split operands as copies.
// EPL = 1: the finished tile goes through LDS once so that every lane handles 4 consecutive columns of a row -- bias / residual / SnakeBeta parameter
// loads and the output stores become 16-byte accesses (a row of the tile = one 384-byte burst) instead of 48 four-byte accesses per lane.
// PF = K tiles whose loads are in flight in registers; SPLIT = operands arrive pre-split (g.a_split) -- a template parameter, and every load of the K loop is
// unconditional (rows beyond M re-read row M - 1, spare loader slots re-read a valid weight row; their results are never stored): with exec-mask
// branches around the loads the compiler's wait-count pass falls back to s_waitcnt vmcnt(0) at every join, which serialises any prefetch.
template <int MR, int NT, int BK, int EPL, int PF, bool SPLIT>
__global__ void __launch_bounds__(256) k_conv_gemm_h3(GemmArgs g, const _Float16* __restrict__ Wh, const _Float16* __restrict__ Wl) {
    constexpr int BM = 128 * MR, BN = 32 * NT, LD = BK + 8; // LD: padded row stride (f16); 80 / 144 B keep the b128 fragment reads conflict-free
    constexpr int KQ = BK / 4, KC = BK / 8;                  // float4 pieces of an A row, uint4 pieces of a B row per K tile
    constexpr int NA = BM * KQ / 256;
    constexpr int NB = (BN * KC + 255) / 256;
    constexpr int CLD = BN + 4;                              // row stride (floats) of the staged output tile
    struct Stage { _Float16 Ah[BM][LD], Al[BM][LD], Bh[BN][LD], Bl[BN][LD]; };
    union Shared { Stage s; float Ct[EPL ? BM : 1][CLD]; };   // the finished tile reuses the operand tiles' space
    __shared__ __attribute__((aligned(16))) Shared sh;
    auto& Ah = sh.s.Ah; auto& Al = sh.s.Al; auto& Bh = sh.s.Bh; auto& Bl = sh.s.Bl;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware tile order: workgroups are dealt to the 8 XCDs round-robin in launch order, so the column tiles of one row tile (and neighbouring row tiles, which
    // share the taps' halo rows) would land on 8 different private L2s and each fetch the activations from HBM again (measured: 532 MB per launch for a 63 MB
    // activation tensor at N = 192).  The remap gives every XCD one contiguous range of tiles (bijective for any workgroup count).
    int tile_m = blockIdx.y, tile_n = blockIdx.x;
    if (g.xcd_swizzle) {
        const int nwg = gridDim.x * gridDim.y, orig = blockIdx.y * gridDim.x + blockIdx.x;
        const int xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
        const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
        tile_n = wgid % gridDim.x; tile_m = wgid / gridDim.x;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int ksplit = gridDim.z, kper = g.K / ksplit, kbeg = blockIdx.z * kper, kend = kbeg + kper;
    f32x16 acc[MR][NT];
#pragma unroll
    for (int t = 0; t < MR; t++)
#pragma unroll
        for (int u = 0; u < NT; u++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[t][u][i] = 0.0f;
    int arow[NA];
#pragma unroll
    for (int i = 0; i < NA; i++) { int mm = m0 + (tid + i * 256) / KQ; mm = mm < g.M ? mm : g.M - 1; arow[i] = mm + seg_of(mm, g.a_segT) * g.a_skip; }
    struct Regs { float4 ra[NA]; uint4 rh[NB], rl[NB]; };
    auto fetch = [&](Regs& s, int k0) {
        const int j = k0 / g.cin, ci0 = k0 % g.cin;
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int kq = (tid + i * 256) % KQ;
            const size_t row = (size_t)(arow[i] + j * g.dil);
            if constexpr (SPLIT) s.ra[i] = load_a4_split_raw(g, row, ci0 + 4 * kq);
            else s.ra[i] = *reinterpret_cast<const float4*>(g.A + row * g.lda + ci0 + 4 * kq);
        }
#pragma unroll
        for (int i = 0; i < NB; i++) {
            int e = tid + i * 256;
            e = e < BN * KC ? e : e - 256;              // spare slots of the last round re-read a valid piece (never written to LDS)
            const int r = e / KC, wk = 8 * (e % KC);
            s.rh[i] = *reinterpret_cast<const uint4*>(Wh + (size_t)(n0 + r) * g.K + k0 + wk);
            s.rl[i] = *reinterpret_cast<const uint4*>(Wl + (size_t)(n0 + r) * g.K + k0 + wk);
        }
    };
    auto stash = [&](const Regs& s) {
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int e = tid + i * 256, r = e / KQ, kq = e % KQ;
            if constexpr (SPLIT) {
                *reinterpret_cast<float2*>(&Ah[r][4 * kq]) = make_float2(s.ra[i].x, s.ra[i].y);
                *reinterpret_cast<float2*>(&Al[r][4 * kq]) = make_float2(s.ra[i].z, s.ra[i].w);
            } else {
                const float x[4] = {s.ra[i].x, s.ra[i].y, s.ra[i].z, s.ra[i].w};
                h4v hi, lo;
#pragma unroll
                for (int c = 0; c < 4; c++) { hi[c] = (_Float16)x[c]; lo[c] = (_Float16)(x[c] - (float)hi[c]); }
                *reinterpret_cast<h4v*>(&Ah[r][4 * kq]) = hi;
                *reinterpret_cast<h4v*>(&Al[r][4 * kq]) = lo;
            }
        }
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const int e = tid + i * 256, r = e / KC, wk = 8 * (e % KC);
            if (e < BN * KC) { *reinterpret_cast<uint4*>(&Bh[r][wk]) = s.rh[i]; *reinterpret_cast<uint4*>(&Bl[r][wk]) = s.rl[i]; }
        }
    };
    auto multiply = [&]() {
#pragma unroll
        for (int kk = 0; kk < BK; kk += 16) {
            const int ko = kk + 8 * (lane >> 5);
            if constexpr (MR == 1) {
                const int ar = wave * 32 + (lane & 31);
                const h8v ah = *reinterpret_cast<const h8v*>(&Ah[ar][ko]);
                const h8v al = *reinterpret_cast<const h8v*>(&Al[ar][ko]);
#pragma unroll
                for (int u = 0; u < NT; u++) {
                    const h8v bh = *reinterpret_cast<const h8v*>(&Bh[u * 32 + (lane & 31)][ko]);
                    const h8v bl = *reinterpret_cast<const h8v*>(&Bl[u * 32 + (lane & 31)][ko]);
                    acc[0][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[0][u], 0, 0, 0);
                    acc[0][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[0][u], 0, 0, 0);
                    acc[0][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[0][u], 0, 0, 0);
                }
            } else {
                h8v bh[NT], bl[NT];
#pragma unroll
                for (int u = 0; u < NT; u++) {
                    bh[u] = *reinterpret_cast<const h8v*>(&Bh[u * 32 + (lane & 31)][ko]);
                    bl[u] = *reinterpret_cast<const h8v*>(&Bl[u * 32 + (lane & 31)][ko]);
                }
#pragma unroll
                for (int t = 0; t < MR; t++) {
                    const int ar = (wave * MR + t) * 32 + (lane & 31);
                    const h8v ah = *reinterpret_cast<const h8v*>(&Ah[ar][ko]);
                    const h8v al = *reinterpret_cast<const h8v*>(&Al[ar][ko]);
#pragma unroll
                    for (int u = 0; u < NT; u++) {
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[u], acc[t][u], 0, 0, 0);
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[u], acc[t][u], 0, 0, 0);
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[u], acc[t][u], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0); // keep one row tile's operands live at a time (the scheduler otherwise hoists every LDS read and spills)
                }
            }
        }
    };
    // workgroup barriers of the K loop wait for LDS traffic only: __syncthreads() would also drain the prefetch (s_waitcnt vmcnt(0))
    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    Regs st[PF];
#pragma unroll
    for (int j = 0; j < PF; j++)
        if (kbeg + j * BK < kend) fetch(st[j], kbeg + j * BK);
    int k0 = kbeg;
    // steady state: every refill is unconditional, so the wait before a stash is a counted vmcnt that leaves the other stages' loads in flight
    // (a conditional refill makes the wait-count pass assume the worst path and drain everything)
    for (; k0 + 2 * PF * BK <= kend; k0 += PF * BK) {
#pragma unroll
        for (int j = 0; j < PF; j++) {
            stash(st[j]);
            lds_barrier();
            fetch(st[j], k0 + (j + PF) * BK);
            multiply();
            lds_barrier();
        }
    }
    for (; k0 < kend; k0 += PF * BK) { // the last tiles
#pragma unroll
        for (int j = 0; j < PF; j++) {
            if (k0 + j * BK < kend) { // (uniform)
                stash(st[j]);
                lds_barrier();
                if (k0 + (j + PF) * BK < kend) fetch(st[j], k0 + (j + PF) * BK);
                multiply();
                lds_barrier();
            }
        }
    }
    __syncthreads(); // the epilogue reuses the operand tiles
    // rows outermost: one row's addresses live at a time (columns outermost made the compiler keep every row's 64-bit addresses in
    // registers: 256 VGPRs, one wave per SIMD).  One call per row tile: a `for t` loop is "too large to unroll" and turns acc[t] into a
    // scratch array.
    auto epilogue_rows = [&](const f32x16 (&a)[NT], int row0) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < g.M) {
#pragma unroll
                for (int u = 0; u < NT; u++) {
                    const int col = n0 + u * 32 + (lane & 31);
                    if (col < g.N) {
                        if (ksplit > 1) g.ws[((size_t)blockIdx.z * g.M + row) * g.N + col] = a[u][r];
                        else gemm_store(g, row, col, gemm_epilogue(g, a[u][r], row, col));
                    }
                }
            }
        }
    };
    if constexpr (EPL != 0 && MR == 1) {
        // (the K loop ended on a barrier: the operand tiles are dead)
        auto& Ct = sh.Ct;
#pragma unroll
        for (int u = 0; u < NT; u++)
#pragma unroll
            for (int r = 0; r < 16; r++) Ct[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][u * 32 + (lane & 31)] = acc[0][u][r];
        __syncthreads();
        constexpr int C4 = BN / 4;
        for (int e = tid; e < BM * C4; e += 256) {
            const int rl = e / C4, col = n0 + (e % C4) * 4, row = m0 + rl;
            if (row >= g.M) continue;
            const float4 t4 = *reinterpret_cast<const float4*>(&Ct[rl][(e % C4) * 4]);
            if (ksplit > 1) *reinterpret_cast<float4*>(g.ws + ((size_t)blockIdx.z * g.M + row) * g.N + col) = t4;
            else gemm_epilogue_store4(g, t4, row, col);
        }
    } else {
        epilogue_rows(acc[0], m0 + wave * MR * 32);
        if constexpr (MR > 1) epilogue_rows(acc[1], m0 + (wave * MR + 1) * 32);
    }
    static_assert(MR <= 2, "one epilogue call per row tile");
}
// 4 columns per thread (slab rows are N floats, N % 4 == 0): 16-byte slab reads, 16-byte epilogue accesses
__global__ void __launch_bounds__(256) k_splitk_reduce4(GemmArgs g, int ksplit) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4, mn = (size_t)g.M * g.N;
    if (i >= mn) return;
    const int row = (int)(i / g.N), col = (int)(i % g.N);
    float4 v = *reinterpret_cast<const float4*>(g.ws + i);
    for (int s = 1; s < ksplit; s++) {
        const float4 w = *reinterpret_cast<const float4*>(g.ws + (size_t)s * mn + i);
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    gemm_epilogue_store4(g, v, row, col);
}
__global__ void __launch_bounds__(256) k_splitk_reduce(GemmArgs g, int ksplit) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)g.M * g.N) return;
    const int row = (int)(i / g.N), col = (int)(i % g.N);
    float v = g.ws[i];
    for (int s = 1; s < ksplit; s++) v += g.ws[(size_t)s * g.M * g.N + i];
    gemm_store(g, row, col, gemm_epilogue(g, v, row, col));
}

// Skinny GEMM for M <= 16 (transformer, ConvNeXt MLPs, conv_in, first transposed conv): weights are streamed exactly
// once; workgroup = 8 waves = 8 output columns sharing an LDS copy of the A rows; each wave's lanes split K.
template <int MT>
__global__ void __launch_bounds__(512) k_skinny_gemm(GemmArgs g) {
    constexpr int KT = 512;
    __shared__ __attribute__((aligned(16))) float As[MT][KT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.x * 8 + wave;
    const int nn = n < g.N ? n : g.N - 1;
    float acc[MT];
#pragma unroll
    for (int m = 0; m < MT; m++) acc[m] = 0.0f;
    for (int k0 = 0; k0 < g.K; k0 += KT) {
        const int kt = (g.K - k0) < KT ? (g.K - k0) : KT;
        // weights first (independent of the LDS staging)
        float4 w4[2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int kk = 4 * (lane + 64 * i);
            w4[i] = (kk < kt) ? *reinterpret_cast<const float4*>(g.W + (size_t)nn * g.K + k0 + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads(); // previous tile fully consumed
        for (int e = tid; e < MT * (KT / 4); e += 512) {
            const int m = e / (KT / 4), kk = 4 * (e % (KT / 4));
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < g.M && kk < kt) {
                const int kg = k0 + kk, j = kg / g.cin, ci = kg % g.cin;
                v = load_a4(g, (size_t)(m + seg_of(m, g.a_segT) * g.a_skip + j * g.dil), ci);
            }
            *reinterpret_cast<float4*>(&As[m][kk]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int kk = 4 * (lane + 64 * i);
#pragma unroll
            for (int m = 0; m < MT; m++) {
                const float4 a = *reinterpret_cast<const float4*>(&As[m][kk]);
                acc[m] += w4[i].x * a.x + w4[i].y * a.y + w4[i].z * a.z + w4[i].w * a.w;
            }
        }
    }
    float mine = 0.0f;
#pragma unroll
    for (int m = 0; m < MT; m++) {
        float v = acc[m];
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        if (lane == m) mine = v;
    }
    if (lane < MT && lane < g.M && n < g.N) gemm_store(g, lane, n, gemm_epilogue(g, mine, lane, n));
}
static const bool g_codec_f32 = [] { const char* e = std::getenv("Q3_CODEC_F32"); return e && e[0] == '1'; }(); // A/B switch
static void gemm(hipStream_t st, GemmArgs g, float* ws, size_t ws_floats, const _Float16* wh = nullptr, const _Float16* wl = nullptr) {
    if (g.M <= 16) {
        dim3 grid((g.N + 7) / 8);
        if (g.M <= 4) hipLaunchKernelGGL((k_skinny_gemm<4>), grid, dim3(512), 0, st, g);
        else if (g.M <= 8) hipLaunchKernelGGL((k_skinny_gemm<8>), grid, dim3(512), 0, st, g);
        else hipLaunchKernelGGL((k_skinny_gemm<16>), grid, dim3(512), 0, st, g);
        return;
    }
    const bool small = g.M <= 32;
    const int tiles = small ? ((g.N + 127) / 128) * ((g.M + 31) / 32) : ((g.N + 63) / 64) * ((g.M + 63) / 64);
    int ksplit = 1;
    while (tiles * ksplit < 256 && ksplit < 16 && (g.K / (ksplit * 2)) % 16 == 0 && g.K / (ksplit * 2) >= 128 &&
           (size_t)(ksplit * 2) * g.M * g.N <= ws_floats) ksplit *= 2;
    g.ws = ws;
    static const int xcd_on = [] { const char* e = std::getenv("Q3_CODEC_XCD"); return e ? atoi(e) : 1; }();
    g.xcd_swizzle = xcd_on;
    static const int h3_min_wgs = [] { const char* e = std::getenv("Q3_CODEC_H3_MIN_WGS"); return e ? atoi(e) : 128; }(); // workgroups from which the 128 x 96 tile serves (0 = never)
    static const int h3_max_n = [] { const char* e = std::getenv("Q3_CODEC_H3_MAXN"); return e ? atoi(e) : 1 << 30; }();
    static const int epl_on = [] { const char* e = std::getenv("Q3_CODEC_H3_EPL"); return e ? atoi(e) : 1; }();
    // split-K for the 128 x 96 tile: deep-K GEMMs with few tiles (N = 768: K = 5376 = 168 K tiles on 128-256 workgroups, one wave per SIMD) are latency-bound;
    // K is cut until Q3_CODEC_H3_WGS workgroups exist, partial slabs summed by k_splitk_reduce4.  Default 0 = never: alone the codec gains 5-8 % at 512-768
    // (3.68 -> 3.40 ms per 16-stream pass), but next to the frame loop the extra workgroups cost more than they save (C3 664 -> 650-657 audio-s/s)
    static const int h3_wgs = [] { const char* e = std::getenv("Q3_CODEC_H3_WGS"); return e ? atoi(e) : 0; }();
    static const int h3_min_tiles = [] { const char* e = std::getenv("Q3_CODEC_H3_MIN_TILES"); return e ? atoi(e) : 96; }();
    if (!small && wh && !g_codec_f32 && g.cin % 32 == 0 && h3_min_wgs > 0 && g.N % 96 == 0 && g.N <= h3_max_n) {
        const int tiles3 = (g.N / 96) * ((g.M + 127) / 128);
        const bool vec_ok = epl_on && epilogue4_ok(g) && (reinterpret_cast<uintptr_t>(ws) & 15) == 0;
        int ks = 1;
        if (vec_ok && h3_wgs > 0 && tiles3 >= h3_min_tiles && tiles3 < h3_wgs) {
            static const int cand[] = {2, 3, 4, 6, 7, 8, 12, 14, 16};
            for (int c : cand) {
                if (g.K % c != 0 || (g.K / c) % 32 != 0 || g.K / c < 256 || (size_t)c * g.M * g.N > ws_floats) continue;
                ks = c;
                if (tiles3 * c >= h3_wgs) break;
            }
        }
        if (tiles3 * ks >= h3_min_wgs) {
            g.ws = ws;
            const dim3 grid3(g.N / 96, (g.M + 127) / 128, vec_ok ? ks : 1);
#define Q3_H3(EPLV, PFV) do { if (g.a_split) hipLaunchKernelGGL((k_conv_gemm_h3<1, 3, 32, EPLV, PFV, true>), grid3, dim3(256), 0, st, g, wh, wl); \
                              else hipLaunchKernelGGL((k_conv_gemm_h3<1, 3, 32, EPLV, PFV, false>), grid3, dim3(256), 0, st, g, wh, wl); } while (0)
            if (!vec_ok) Q3_H3(0, 1); else Q3_H3(1, 1); // PF = 2 / 3 measured equal at 64 slots and slower at 256 (profiles/r02_codec_ab_sweeps.txt)
#undef Q3_H3
            if (ks > 1) {
                const size_t n = (size_t)g.M * g.N;
                hipLaunchKernelGGL(k_splitk_reduce4, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, g, ks);
            }
            return;
        }
    }
    if (!small && wh && !g_codec_f32 && g.cin % 32 == 0) { // large M: split-f16 matrix cores (K tile 32)
        static const int big_m = [] { const char* e = std::getenv("Q3_CODEC_BM128"); return e ? atoi(e) : 1024; }(); // rows from which the 128-row tile is used
        // measured inside the 64-stream pipeline: N = 768 (K = 5376) 96 -> 64 us with the tall tile, N <= 384 no gain
        const bool tall = g.M >= big_m && g.N >= 512 && ((g.N + 63) / 64) * ((g.M + 127) / 128) >= 160;
        const int t2 = tall ? ((g.N + 63) / 64) * ((g.M + 127) / 128) : tiles;
        static const int wg_target = [] { const char* e = std::getenv("Q3_CODEC_WGS"); return e ? atoi(e) : 256; }(); // split-K until this many workgroups
        int ks = 1;
        while (t2 * ks < wg_target && ks < 16 && (g.K / (ks * 2)) % 32 == 0 && g.K / (ks * 2) >= 128 && (size_t)(ks * 2) * g.M * g.N <= ws_floats) ks *= 2;
        static const int epl4 = [] { const char* e = std::getenv("Q3_CODEC_EPL4"); return e ? atoi(e) : 1; }();
        const bool vec_ok = epl4 && epilogue4_ok(g) && (reinterpret_cast<uintptr_t>(g.ws) & 15) == 0;
#define Q3_H(MRV, EPLV, GRID) do { if (g.a_split) hipLaunchKernelGGL((k_conv_gemm_h<MRV, EPLV, true>), GRID, dim3(256), 0, st, g, wh, wl); \
                                   else hipLaunchKernelGGL((k_conv_gemm_h<MRV, EPLV, false>), GRID, dim3(256), 0, st, g, wh, wl); } while (0)
        const dim3 grid_t((g.N + 63) / 64, (g.M + 127) / 128, ks), grid_s((g.N + 63) / 64, (g.M + 63) / 64, ks);
        if (vec_ok) { if (tall) Q3_H(2, 1, grid_t); else Q3_H(1, 1, grid_s); }
        else { if (tall) Q3_H(2, 0, grid_t); else Q3_H(1, 0, grid_s); }
#undef Q3_H
        if (ks > 1) {
            const size_t n = (size_t)g.M * g.N;
            if (vec_ok) hipLaunchKernelGGL(k_splitk_reduce4, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, g, ks);
            else hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g, ks);
        }
        return;
    }
    if (small) hipLaunchKernelGGL((k_conv_gemm<1>), dim3((g.N + 127) / 128, (g.M + 31) / 32, ksplit), dim3(256), 0, st, g);
    else hipLaunchKernelGGL((k_conv_gemm<2>), dim3((g.N + 63) / 64, (g.M + 63) / 64, ksplit), dim3(256), 0, st, g);
    if (ksplit > 1) {
        const size_t n = (size_t)g.M * g.N;
        hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g, ksplit);
    }
}

// ---------------- small elementwise / reduction kernels ----------------
// Row maps: a batched extended buffer holds G segments of (H + T) rows; logical row r (= g*T + t) lives at physical row
// r + (r / segT) * skip + off  (segT = T, skip = H, off = 0 for the "history first" view, off = H for the "current rows" view).
struct RowMap { int segT, skip, off; };
__device__ __forceinline__ size_t map_row(const RowMap& m, int r) { return (size_t)r + (size_t)(m.segT == SEG_NONE ? 0 : r / m.segT) * m.skip + m.off; }
static RowMap plain_map() { return RowMap{SEG_NONE, 0, 0}; }

__global__ void k_rvq_sum(const int64_t* __restrict__ codes, const float* const* __restrict__ cb, int n_q, int cb_size, int cb_dim,
                          float* __restrict__ out, int ldo, RowMap om) {
    const int t = blockIdx.y, d = blockIdx.x * 256 + threadIdx.x;
    if (d >= cb_dim) return;
    float acc = 0.0f;
    for (int q = 0; q < n_q; q++) {
        long long c = codes[(size_t)t * n_q + q];
        c = c < 0 ? 0 : (c >= cb_size ? cb_size - 1 : c);
        acc += cb[q][(size_t)c * cb_dim + d];
    }
    out[map_row(om, t) * ldo + d] = acc;
}
__global__ void __launch_bounds__(256) k_rmsnorm_rows(const float* __restrict__ x, int ldx, const float* __restrict__ g, int C, float eps,
                                                      float* __restrict__ y, int ldy) {
    __shared__ float red[4];
    const int t = blockIdx.x;
    float s = 0.0f;
    for (int c = threadIdx.x; c < C; c += 256) { const float v = x[(size_t)t * ldx + c]; s += v * v; }
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    const float sc = 1.0f / sqrtf(tot / (float)C + eps);
    for (int c = threadIdx.x; c < C; c += 256) y[(size_t)t * ldy + c] = x[(size_t)t * ldx + c] * sc * g[c];
}
__global__ void __launch_bounds__(256) k_layernorm_rows(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                        const float* __restrict__ b, int C, float eps, float* __restrict__ y, int ldy) {
    __shared__ float red[8];
    const int t = blockIdx.x;
    float s = 0.0f;
    for (int c = threadIdx.x; c < C; c += 256) s += x[(size_t)t * ldx + c];
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float mu = (red[0] + red[1] + red[2] + red[3]) / (float)C;
    float v = 0.0f;
    for (int c = threadIdx.x; c < C; c += 256) { const float dlt = x[(size_t)t * ldx + c] - mu; v += dlt * dlt; }
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = v;
    __syncthreads();
    const float var = (red[4] + red[5] + red[6] + red[7]) / (float)C;
    const float inv = 1.0f / sqrtf(var + eps);
    for (int c = threadIdx.x; c < C; c += 256) y[(size_t)t * ldy + c] = (x[(size_t)t * ldx + c] - mu) * inv * w[c] + b[c];
}
// per-call stream metadata on device: meta[0..G) = stream index, meta[G..2G) = valid K/V history rows, meta[2G..3G) = frames seen
// NeoX RoPE on q and k inside a packed [G*T0][3*H] qkv buffer (head_dim hd), absolute position seen[g] + t
// RoPE of q and k plus the K / V window append in one launch: thread e < H/2 rotates pair e of q (in place) and of k (in place and into the K window);
// every thread also copies two V elements into the V window
__global__ void k_codec_rope_append(float* __restrict__ qkv, int ld, int H, int hd, const float* __restrict__ cs, const float* __restrict__ sn,
                                    const int64_t* __restrict__ meta, int G, int T0, int max_pos, float* __restrict__ kext, float* __restrict__ vext, RowMap om) {
    const int r = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x, half = hd / 2;
    if (e >= H / 2) return;
    const int head = e / half, i = e % half;
    long long p = meta[2 * G + r / T0] + (r % T0);
    if (p >= max_pos) p = max_pos - 1;
    const float c = cs[(size_t)p * half + i], s = sn[(size_t)p * half + i];
    const size_t o = map_row(om, r) * H;
    float* row = qkv + (size_t)r * ld;
    {
        float* v = row + head * hd;
        const float a = v[i], b = v[i + half];
        v[i] = a * c - b * s; v[i + half] = b * c + a * s;
    }
    {
        float* v = row + H + head * hd;
        const float a = v[i], b = v[i + half];
        const float x = a * c - b * s, y = b * c + a * s;
        v[i] = x; v[i + half] = y;
        kext[o + head * hd + i] = x; kext[o + head * hd + i + half] = y;
    }
    vext[o + e] = row[2 * H + e];
    vext[o + e + H / 2] = row[2 * H + e + H / 2];
}
// sliding-window attention: one wave per (head, row); the segment of stream g holds its W-1 history rows RIGHT-aligned
// (only the last kvlen[g] are valid) followed by the T0 new rows; keys of row t: max(t, W-1-kvlen) .. W-1+t
__global__ void __launch_bounds__(64) k_codec_attn(const float* __restrict__ qkv, int ld, int H, int hd, const float* __restrict__ kext,
                                                   const float* __restrict__ vext, const int64_t* __restrict__ meta, int G, int T0, int W,
                                                   float* __restrict__ out, int ldo) {
    __shared__ float p_s[128];
    __shared__ float q_s[128];
    const int head = blockIdx.x, r = blockIdx.y, lane = threadIdx.x;
    const int g = r / T0, t = r % T0;
    const int kvl = (int)meta[G + g];
    const size_t seg = (size_t)g * (W - 1 + T0);
    const int j1 = W - 1 + t, jv = W - 1 - kvl, j0 = t > jv ? t : jv, nk = j1 - j0 + 1;
    for (int d = lane; d < hd; d += 64) q_s[d] = qkv[(size_t)r * ld + head * hd + d];
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)hd);
    float mx = -INFINITY;
    for (int jj = lane; jj < nk; jj += 64) {
        const float* kr = kext + (seg + j0 + jj) * H + head * hd;
        float a = 0.0f;
        if ((hd & 3) == 0) { // 16-byte loads of the lane's K row (rows start on hd-float boundaries); same order of adds
            for (int d = 0; d < hd; d += 4) {
                const float4 kk = *reinterpret_cast<const float4*>(kr + d);
                a += q_s[d] * kk.x; a += q_s[d + 1] * kk.y; a += q_s[d + 2] * kk.z; a += q_s[d + 3] * kk.w;
            }
        } else for (int d = 0; d < hd; d++) a += q_s[d] * kr[d];
        a *= scale;
        p_s[jj] = a;
        mx = fmaxf(mx, a);
    }
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float den = 0.0f;
    for (int jj = lane; jj < nk; jj += 64) { const float e = expf(p_s[jj] - mx); p_s[jj] = e; den += e; }
    for (int o = 32; o >= 1; o >>= 1) den += __shfl_xor(den, o);
    __syncthreads();
    for (int d = lane; d < hd; d += 64) {
        float a = 0.0f;
        for (int jj = 0; jj < nk; jj++) a += p_s[jj] * vext[(seg + j0 + jj) * H + head * hd + d];
        out[(size_t)r * ldo + head * hd + d] = a / den;
    }
}
__global__ void k_swiglu_rows(const float* __restrict__ gu, int ff, float* __restrict__ out) { // gu [T][2ff] -> out [T][ff]
    const int t = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= ff) return;
    const float g = gu[(size_t)t * 2 * ff + c], u = gu[(size_t)t * 2 * ff + ff + c];
    out[(size_t)t * ff + c] = (g / (1.0f + expf(-g))) * u;
}
// depthwise causal conv k=7 over a batched extended buffer (6 history rows per segment), w [C][7]; out plain rows
__global__ void k_dwconv7(const float* __restrict__ in_ext, int C, const float* __restrict__ w, const float* __restrict__ b,
                          float* __restrict__ out, RowMap im) {
    const int t = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const size_t r0 = map_row(im, t);
    float a = b[c];
#pragma unroll
    for (int j = 0; j < 7; j++) a += w[c * 7 + j] * in_ext[(r0 + j) * C + c];
    out[(size_t)t * C + c] = a;
}
// y = x + inv_eb[c] * sin^2(x*ea[c]); src plain rows [T][C] -> dst rows through the map
__global__ void k_snake(const float* __restrict__ src, float* __restrict__ dst, int C, const float* __restrict__ ea,
                        const float* __restrict__ inv_eb, size_t n4, RowMap dm, int split) { // 4 channels per thread (C % 4 == 0)
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int c4 = C >> 2, c = (int)(i % c4) * 4, r = (int)(i / c4);
    const float4 v = reinterpret_cast<const float4*>(src)[i];
    const float4 a = *reinterpret_cast<const float4*>(ea + c), b = *reinterpret_cast<const float4*>(inv_eb + c);
    const float s0 = snake_sin(v.x * a.x), s1 = snake_sin(v.y * a.y), s2 = snake_sin(v.z * a.z), s3 = snake_sin(v.w * a.w);
    const float4 y = make_float4(v.x + b.x * (s0 * s0), v.y + b.y * (s1 * s1), v.z + b.z * (s2 * s2), v.w + b.w * (s3 * s3));
    float* drow = dst + map_row(dm, r) * C;
    if (!split) { *reinterpret_cast<float4*>(drow + c) = y; return; }
    _Float16* row = reinterpret_cast<_Float16*>(drow); // split row: [C hi][C lo]
    h4v_ hi, lo;
    hi[0] = (_Float16)y.x; hi[1] = (_Float16)y.y; hi[2] = (_Float16)y.z; hi[3] = (_Float16)y.w;
    lo[0] = (_Float16)(y.x - (float)hi[0]); lo[1] = (_Float16)(y.y - (float)hi[1]); lo[2] = (_Float16)(y.z - (float)hi[2]); lo[3] = (_Float16)(y.w - (float)hi[3]);
    *reinterpret_cast<h4v_*>(row + c) = hi; *reinterpret_cast<h4v_*>(row + C + c) = lo;
}
__global__ void k_copy_rows(const float* __restrict__ src, float* __restrict__ dst, size_t n, int C, RowMap dm) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C), r = (int)(i / C);
    dst[map_row(dm, r) * C + c] = src[i];
}
// all convolutions' histories in one launch (blockIdx.z = conv): a pass loads every history up front and saves every one at the end
struct HistDesc { float* work; float* hist; int H, C, T, pad; };
struct HistTable { HistDesc d[48]; int n; };
__global__ void k_hist_all(HistTable tab, const int64_t* __restrict__ meta, int save) { // 4 floats per thread (every C is a multiple of 4)
    const HistDesc e = tab.d[blockIdx.z];
    const int g = blockIdx.y;
    const size_t n = (size_t)e.H * e.C, i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    float* w = e.work + (size_t)g * (e.H + e.T) * e.C;
    float* h = e.hist + (size_t)meta[g] * n;
    if (save) *reinterpret_cast<float4*>(h + i) = *reinterpret_cast<const float4*>(w + (size_t)e.T * e.C + i); // the last H rows of (history + new rows) become the next call's history
    else *reinterpret_cast<float4*>(w + i) = *reinterpret_cast<const float4*>(h + i);
}
// a stream's histories back to zero (new utterance): one launch for every convolution instead of one memset node per history (37 per request)
__global__ void k_hist_zero(HistTable tab, int stream) {
    const HistDesc e = tab.d[blockIdx.z];
    const size_t n = (size_t)e.H * e.C, i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) e.hist[(size_t)stream * n + i] = 0.0f;
}
// final conv (cout = 1): one wave per output sample, lanes split the 7*C products (coalesced rows), butterfly sum, clamp
__global__ void __launch_bounds__(256) k_conv_out_wave(const float* __restrict__ in_ext, int C, const float* __restrict__ w, float bias,
                                                       float* __restrict__ pcm, int n, RowMap im) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= n) return;
    const float* row = in_ext + map_row(im, t) * C; // 7 consecutive rows of C floats = one contiguous span of 7*C
    float a = 0.0f;
    if ((C & 3) == 0) { // 16-byte loads (rows start on C-float boundaries; C % 4 == 0 for every decoder width)
        for (int i = lane; i < 7 * C / 4; i += 64) {
            const float4 x = *reinterpret_cast<const float4*>(row + 4 * i), ww = *reinterpret_cast<const float4*>(w + 4 * i);
            a += ww.x * x.x + ww.y * x.y + ww.z * x.z + ww.w * x.w;
        }
    } else for (int i = lane; i < 7 * C; i += 64) a += w[i] * row[i];
    for (int o = 32; o >= 1; o >>= 1) a += __shfl_xor(a, o);
    a += bias;
    if (lane == 0) pcm[t] = a < -1.0f ? -1.0f : (a > 1.0f ? 1.0f : a);
}
// Window form of the output convolution: a workgroup owns TB = 128 consecutive output samples of ONE segment (host guarantees segT % 128 == 0), copies their
// 128 + 6 input rows into LDS once (16-byte loads: the one-wave-per-sample form fetched every row 7 times, 236 MB for a 94 MB input) and each wave
// finishes 32 samples from there: lane l holds 3 float4 of the 7*C weights and reads the matching pieces of the sample's 7-row span.
template <int C>
__global__ void __launch_bounds__(256) k_conv_out_win(const float* __restrict__ in_ext, const float* __restrict__ w, float bias, float* __restrict__ pcm, int n, RowMap im) {
    constexpr int TB = 128, ROWS = TB + 6, SPAN4 = 7 * C / 4, NW = (SPAN4 + 63) / 64;
    __shared__ __attribute__((aligned(16))) float xs[ROWS * C];
    const int t0 = blockIdx.x * TB, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* src = in_ext + map_row(im, t0) * C; // the TB samples of a block lie in one segment: their ROWS input rows are contiguous
    for (int e = tid; e < ROWS * C / 4; e += 256) *reinterpret_cast<float4*>(xs + 4 * e) = *reinterpret_cast<const float4*>(src + 4 * e);
    float4 wv[NW];
#pragma unroll
    for (int k = 0; k < NW; k++) { const int i = lane + 64 * k; wv[k] = i < SPAN4 ? *reinterpret_cast<const float4*>(w + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f); }
    __syncthreads();
    for (int j = 0; j < TB / 4; j++) {
        const int tl = wave * (TB / 4) + j;
        float a = 0.0f;
#pragma unroll
        for (int k = 0; k < NW; k++) {
            const int i = lane + 64 * k;
            if (i < SPAN4) {
                const float4 x = *reinterpret_cast<const float4*>(xs + tl * C + 4 * i);
                a += wv[k].x * x.x + wv[k].y * x.y + wv[k].z * x.z + wv[k].w * x.w;
            }
        }
        for (int o = 32; o >= 1; o >>= 1) a += __shfl_xor(a, o);
        a += bias;
        if (lane == 0 && t0 + tl < n) pcm[t0 + tl] = a < -1.0f ? -1.0f : (a > 1.0f ? 1.0f : a);
    }
}
// ---------------- host side ----------------
struct ConvW { DevBuf<float> w, b; DevBuf<_Float16> wh, wl; int N = 0, K = 0, cin = 0, taps = 1, dil = 1; };
struct Snake { DevBuf<float> ea, inv_eb; int C = 0; };
// A causal conv's streaming state.  `hist` is the persistent store [n_streams][H][C]; the rows a call works on live in a
// per-lane batched extended buffer [G][(H + T)][C] (Scratch::work[id]): history first, then the call's T new rows.
struct Ext { DevBuf<float> hist; int H = 0, C = 0, Tmax = 0, id = -1; };

struct CodecDecoder::Impl {
    int n_q, cb_size, cb_dim, hidden, n_layers, n_heads, head_dim, ffn, window, n_up, dec_dim, n_dec;
    int up_ratios[4], dec_rates[8];
    float rope_base, eps;
    int n_streams, max_frames, max_pos = 8192;
    std::vector<DevBuf<float>> codebooks; DevBuf<const float*> d_cb_ptrs;
    ConvW pre_conv;
    struct TfL { DevBuf<float> attn_norm, ffn_norm, ls_attn, ls_ffn; ConvW wqkv, wo, wgu, wdown; };
    std::vector<TfL> tf; DevBuf<float> tf_norm, rope_c, rope_s;
    struct Up { ConvW ct, pw1, pw2; DevBuf<float> dw_w, dw_b, ln_w, ln_b, gamma; };
    std::vector<Up> up;
    ConvW conv_in;
    struct RU { Snake s1, s2; ConvW c1, c2; };
    struct Blk { Snake snake; ConvW ct; RU ru[3]; int cin, cout, rate; };
    std::vector<Blk> blk;
    Snake snake_out; DevBuf<float> conv_out_w; float conv_out_b = 0;
    // per-stream state
    Ext z_ext, convin_ext, out_ext; std::vector<Ext> dw_ext, ct_ext, k_ext, v_ext; std::vector<std::vector<Ext>> ru_ext;
    std::vector<Ext*> all_ext;
    bool presplit = false; // Q3_CODEC_PRESPLIT=1: SnakeBeta outputs feeding GEMMs are stored as hi / lo f16 planes (measured: no gain)
    std::vector<int> kv_len; std::vector<long long> n_seen;
    int gmax = 1; // streams decoded together in one pass
    struct Scratch { // one set per concurrency lane: independent decodes run on different HIP streams at the same time
        DevBuf<float> h, xn, qkv, att, gu, act, t1, t2, pcm, splitk_ws; DevBuf<int64_t> d_in; // d_in = codes then meta
        std::vector<DevBuf<float>> work;              // batched extended buffers, indexed by Ext::id
        int64_t* h_in = nullptr; int ring_idx = 0;    // pinned staging ring for the (tiny) code + metadata uploads of async calls
    };
    std::vector<Scratch> lanes; Scratch* S = nullptr; int ring = 256;
    size_t in_slot() const { return (size_t)gmax * max_frames * n_q + 3 * (size_t)gmax; }
    double flops_frame = 0;

    static std::vector<float> tensor(const Gguf& g, const std::string& name) {
        const GgufTensor& t = g.need(name);
        Q3_CHECK(t.type == Q3_T_F32, "codec tensor not F32: " + name);
        const float* p = reinterpret_cast<const float*>(t.data);
        return std::vector<float>(p, p + t.nbytes / 4);
    }
    static void up_f(DevBuf<float>& d, const std::vector<float>& v) { d.alloc(v.size()); d.upload(v.data(), v.size()); }
    // hi/lo f16 split of a rearranged weight matrix for k_conv_gemm_h (only matrices that can see >= 64 rows per call need it)
    static void split_w(ConvW& c, const std::vector<float>& r) {
        std::vector<_Float16> h(r.size()), l(r.size());
        for (size_t i = 0; i < r.size(); i++) { h[i] = (_Float16)r[i]; l[i] = (_Float16)(r[i] - (float)h[i]); }
        c.wh.alloc(h.size()); c.wh.upload(h.data(), h.size());
        c.wl.alloc(l.size()); c.wl.upload(l.data(), l.size());
    }
    // causal conv weight [cout][cin][k] -> Wr[co][j*cin + ci]
    void make_conv(ConvW& c, const Gguf& g, const std::string& wn, const std::string& bn, int cout, int cin, int k, int dil) {
        auto w = tensor(g, wn);
        Q3_CHECK((int64_t)w.size() == (int64_t)cout * cin * k && cin % 16 == 0, "conv shape " + wn);
        std::vector<float> r((size_t)cout * cin * k);
        for (int co = 0; co < cout; co++) for (int ci = 0; ci < cin; ci++) for (int j = 0; j < k; j++)
            r[((size_t)co * k + j) * cin + ci] = w[((size_t)co * cin + ci) * k + j];
        up_f(c.w, r); split_w(c, r);
        if (!bn.empty()) up_f(c.b, tensor(g, bn));
        c.N = cout; c.K = cin * k; c.cin = cin; c.taps = k; c.dil = dil;
    }
    void make_linear(ConvW& c, const std::vector<float>& w, int N, int K) { up_f(c.w, w); split_w(c, w); c.N = N; c.K = K; c.cin = K; c.taps = 1; c.dil = 1; }
    // transposed conv weight [cin][cout][k], stride s (k == 2s or k == s) -> Wr[(r,co)][j*cin+ci], tap 0 = previous frame
    void make_convt(ConvW& c, const Gguf& g, const std::string& wn, const std::string& bn, int cin, int cout, int k, int s) {
        auto w = tensor(g, wn);
        auto b = tensor(g, bn);
        Q3_CHECK((int64_t)w.size() == (int64_t)cin * cout * k && (k == 2 * s || k == s) && cin % 16 == 0, "convT shape " + wn);
        const int taps = k / s;
        std::vector<float> r((size_t)s * cout * taps * cin), br((size_t)s * cout);
        for (int rr = 0; rr < s; rr++) for (int co = 0; co < cout; co++) {
            br[(size_t)rr * cout + co] = b[co];
            for (int ci = 0; ci < cin; ci++) {
                if (taps == 2) { // A row = [x[t-1] ; x[t]]
                    r[((size_t)(rr * cout + co) * 2 + 0) * cin + ci] = w[((size_t)ci * cout + co) * k + rr + s];
                    r[((size_t)(rr * cout + co) * 2 + 1) * cin + ci] = w[((size_t)ci * cout + co) * k + rr];
                } else r[(size_t)(rr * cout + co) * cin + ci] = w[((size_t)ci * cout + co) * k + rr];
            }
        }
        up_f(c.w, r); split_w(c, r); up_f(c.b, br);
        c.N = s * cout; c.K = taps * cin; c.cin = cin; c.taps = taps; c.dil = 1;
    }
    void make_snake(Snake& s, const Gguf& g, const std::string& an, const std::string& bn, int C) {
        auto a = tensor(g, an), b = tensor(g, bn);
        std::vector<float> ea(C), ib(C);
        for (int i = 0; i < C; i++) { ea[i] = expf(a[i]); ib[i] = 1.0f / (expf(b[i]) + 1e-9f); }
        up_f(s.ea, ea); up_f(s.inv_eb, ib); s.C = C;
    }
    void make_ext(Ext& e, int H, int C, int Tmax) {
        e.H = H; e.C = C; e.Tmax = Tmax; e.id = (int)all_ext.size();
        e.hist.alloc((size_t)n_streams * H * C); e.hist.zero();
        all_ext.push_back(&e);
    }
    // where a GEMM operand lives: a plain [M][ld] buffer, or the batched extended buffer of `e` for a call with T rows per stream
    struct Loc { float* p; int segT; long long skip_rows; };
    Loc plain(float* p) { return Loc{p, SEG_NONE, 0}; }
    float* work(const Ext& e) { return S->work[e.id].p; }
    Loc ext_base(const Ext& e, int T) { return Loc{work(e), T, e.H}; }                       // rows t + j*dil, history first
    Loc ext_cur(const Ext& e, int T) { return Loc{work(e) + (size_t)e.H * e.C, T, e.H}; }   // the call's new rows
    RowMap cur_map(const Ext& e, int T) { return RowMap{T, e.H, e.H}; }
    RowMap base_map(const Ext& e, int T) { return RowMap{T, e.H, 0}; }
    // out_rows_per_m: a transposed conv writes f consecutive C-wide rows per GEMM row (ldo = f*C); skip stays H*C floats per segment
    void run_conv(hipStream_t st, const ConvW& c, Loc A, int lda, int M, Loc out, int ldo, int out_C, int epi = EPI_NONE,
                  Loc res = Loc{nullptr, SEG_NONE, 0}, int ldr = 0, const float* scale = nullptr, const Snake* sn = nullptr, bool a_split = false,
                  bool o_split = false) {
        GemmArgs g{};
        g.a_split = a_split ? 1 : 0; g.o_split = o_split ? 1 : 0;
        g.A = A.p; g.lda = lda; g.cin = c.cin; g.dil = c.dil; g.W = c.w.p; g.bias = c.b.n ? c.b.p : nullptr; g.out = out.p; g.ldo = ldo;
        g.M = M; g.N = c.N; g.K = c.K; g.epi = epi; g.res = res.p; g.ldr = ldr; g.scale = scale;
        g.a_segT = A.segT; g.a_skip = (int)A.skip_rows;
        g.o_segT = out.segT; g.o_skip = out.skip_rows * out_C;
        g.r_segT = res.segT; g.r_skip = res.skip_rows * ldr;
        if (sn) { g.snake_ea = sn->ea.p; g.snake_ib = sn->inv_eb.p; }
        gemm(st, g, S->splitk_ws.p, S->splitk_ws.n, c.wh.n ? c.wh.p : nullptr, c.wl.n ? c.wl.p : nullptr);
    }
    // histories are moved by two launches per pass (k_hist_all): decode_group_async registers every conv it is about to run with
    // the row count of this call, loads all histories before the first kernel and saves them after the last
    HistTable table;
    void plan_hist(const Ext& e, int T) {
        if (e.H == 0) return;
        Q3_CHECK(table.n < 48, "too many streaming convolutions");
        Q3_CHECK(e.C % 4 == 0, "streaming state rows must be a multiple of 4 channels (k_hist_all moves float4s)");
        table.d[table.n++] = HistDesc{work(e), e.hist.p, e.H, e.C, T, 0};
    }
    void move_hist(hipStream_t st, int G, const int64_t* meta, bool save) {
        size_t mx = 0;
        for (int i = 0; i < table.n; i++) mx = std::max(mx, (size_t)table.d[i].H * table.d[i].C);
        hipLaunchKernelGGL(k_hist_all, dim3((unsigned)((mx / 4 + 255) / 256), G, table.n), dim3(256), 0, st, table, meta, save ? 1 : 0);
    }

    void snake(hipStream_t st, const Snake& s, const float* src, float* dst, int rows, RowMap dm, bool split = false) {
        Q3_CHECK(s.C % 4 == 0, "SnakeBeta channel count must be a multiple of 4");
        const size_t n = (size_t)rows * (s.C / 4);
        hipLaunchKernelGGL(k_snake, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, s.C, s.ea.p, s.inv_eb.p, n, dm, split ? 1 : 0);
    }
    void copy_rows(hipStream_t st, const float* src, float* dst, int rows, int C, RowMap dm) {
        const size_t n = (size_t)rows * C;
        hipLaunchKernelGGL(k_copy_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, n, C, dm);
    }
};

CodecDecoder::CodecDecoder(const std::string& path, int n_streams, int max_frames, int n_lanes, int max_group) : impl_(new Impl()) {
    Impl& m = *impl_;
    if (const char* e = std::getenv("Q3_CODEC_PRESPLIT")) m.presplit = !(e[0] == '0');
    if (g_codec_f32) m.presplit = false; // the f32-only A/B mode keeps full f32 operands
    Gguf g(path);
    m.n_streams = n_streams; m.max_frames = max_frames; m.gmax = max_group > 0 ? max_group : 1;
    m.n_q = (int)g.kv_int("codec.n_codebooks", 16); m.cb_size = (int)g.kv_int("codec.codebook_size", 2048);
    m.cb_dim = (int)g.kv_int("codec.codebook_dim", 512); m.hidden = (int)g.kv_int("codec.hidden", 1024);
    m.n_layers = (int)g.kv_int("codec.n_layers", 8); m.n_heads = (int)g.kv_int("codec.n_heads", 16);
    m.head_dim = (int)g.kv_int("codec.head_dim", 64); m.ffn = (int)g.kv_int("codec.ffn", 3072);
    m.window = (int)g.kv_int("codec.window", 72); m.dec_dim = (int)g.kv_int("codec.dec_dim", 1536);
    m.rope_base = (float)g.kv_float("codec.rope_base", 10000.0); m.eps = (float)g.kv_float("codec.eps", 1e-5);
    m.n_up = (int)g.kv_int("codec.n_up", 2); m.n_dec = (int)g.kv_int("codec.n_dec", 4);
    static const int def_rates[8] = {8, 5, 4, 3, 2, 2, 2, 2};
    for (int i = 0; i < m.n_up; i++) m.up_ratios[i] = (int)g.kv_int("codec.up_ratio." + std::to_string(i), 2);
    for (int i = 0; i < m.n_dec; i++) m.dec_rates[i] = (int)g.kv_int("codec.dec_rate." + std::to_string(i), def_rates[i]);
    const int H = m.hidden, A = m.n_heads * m.head_dim;
    Q3_CHECK(A == H && m.head_dim <= 128 && m.window <= 128 && m.cb_dim % 16 == 0 && H % 16 == 0 && m.ffn % 16 == 0, "unsupported codec dims");
    std::vector<const float*> cbp;
    m.codebooks.resize(m.n_q);
    for (int q = 0; q < m.n_q; q++) { Impl::up_f(m.codebooks[q], Impl::tensor(g, "codec.codebook." + std::to_string(q))); cbp.push_back(m.codebooks[q].p); }
    m.d_cb_ptrs.alloc(m.n_q); m.d_cb_ptrs.upload(cbp.data(), m.n_q);
    m.make_conv(m.pre_conv, g, "codec.pre_conv.weight", "codec.pre_conv.bias", H, m.cb_dim, 3, 1);
    m.tf.resize(m.n_layers);
    double fl = 0;
    fl += 2.0 * H * m.cb_dim * 3;
    for (int l = 0; l < m.n_layers; l++) {
        auto& L = m.tf[l];
        const std::string p = "codec.tf." + std::to_string(l) + ".";
        Impl::up_f(L.attn_norm, Impl::tensor(g, p + "attn_norm")); Impl::up_f(L.ffn_norm, Impl::tensor(g, p + "ffn_norm"));
        Impl::up_f(L.ls_attn, Impl::tensor(g, p + "ls_attn")); Impl::up_f(L.ls_ffn, Impl::tensor(g, p + "ls_ffn"));
        auto wq = Impl::tensor(g, p + "wq"), wk = Impl::tensor(g, p + "wk"), wv = Impl::tensor(g, p + "wv");
        std::vector<float> qkv; qkv.insert(qkv.end(), wq.begin(), wq.end()); qkv.insert(qkv.end(), wk.begin(), wk.end()); qkv.insert(qkv.end(), wv.begin(), wv.end());
        m.make_linear(L.wqkv, qkv, 3 * H, H);
        m.make_linear(L.wo, Impl::tensor(g, p + "wo"), H, A);
        auto wg = Impl::tensor(g, p + "w_gate"), wu = Impl::tensor(g, p + "w_up");
        std::vector<float> gu; gu.insert(gu.end(), wg.begin(), wg.end()); gu.insert(gu.end(), wu.begin(), wu.end());
        m.make_linear(L.wgu, gu, 2 * m.ffn, H);
        m.make_linear(L.wdown, Impl::tensor(g, p + "w_down"), H, m.ffn);
        fl += 2.0 * (4.0 * H * H + 3.0 * H * m.ffn);
    }
    Impl::up_f(m.tf_norm, Impl::tensor(g, "codec.tf.norm"));
    {
        const int half = m.head_dim / 2;
        std::vector<float> c((size_t)m.max_pos * half), s((size_t)m.max_pos * half);
        for (int p = 0; p < m.max_pos; p++) for (int i = 0; i < half; i++) {
            const double ang = (double)p * std::pow((double)m.rope_base, -(double)i / half);
            c[(size_t)p * half + i] = (float)std::cos(ang); s[(size_t)p * half + i] = (float)std::sin(ang);
        }
        Impl::up_f(m.rope_c, c); Impl::up_f(m.rope_s, s);
    }
    int rate = 1;
    m.up.resize(m.n_up);
    for (int i = 0; i < m.n_up; i++) {
        auto& U = m.up[i];
        const std::string p = "codec.up." + std::to_string(i) + ".";
        const int f = m.up_ratios[i];
        m.make_convt(U.ct, g, p + "convt.weight", p + "convt.bias", H, H, f, f);
        Impl::up_f(U.dw_w, Impl::tensor(g, p + "dw.weight")); Impl::up_f(U.dw_b, Impl::tensor(g, p + "dw.bias"));
        Impl::up_f(U.ln_w, Impl::tensor(g, p + "ln.weight")); Impl::up_f(U.ln_b, Impl::tensor(g, p + "ln.bias"));
        m.make_linear(U.pw1, Impl::tensor(g, p + "pw1.weight"), 4 * H, H); Impl::up_f(U.pw1.b, Impl::tensor(g, p + "pw1.bias"));
        m.make_linear(U.pw2, Impl::tensor(g, p + "pw2.weight"), H, 4 * H); Impl::up_f(U.pw2.b, Impl::tensor(g, p + "pw2.bias"));
        Impl::up_f(U.gamma, Impl::tensor(g, p + "gamma"));
        fl += rate * 2.0 * H * H * f;
        rate *= f;
        fl += rate * 2.0 * (7.0 * H + 8.0 * H * H);
    }
    m.make_conv(m.conv_in, g, "codec.dec.conv_in.weight", "codec.dec.conv_in.bias", m.dec_dim, H, 7, 1);
    fl += rate * 2.0 * 7.0 * H * m.dec_dim;
    int ch = m.dec_dim;
    m.blk.resize(m.n_dec);
    for (int b = 0; b < m.n_dec; b++) {
        auto& B = m.blk[b];
        const std::string p = "codec.dec." + std::to_string(b) + ".";
        const int r = m.dec_rates[b], co = ch / 2;
        B.cin = ch; B.cout = co; B.rate = r;
        m.make_snake(B.snake, g, p + "snake.alpha", p + "snake.beta", ch);
        m.make_convt(B.ct, g, p + "convt.weight", p + "convt.bias", ch, co, 2 * r, r);
        fl += rate * 2.0 * ch * co * 2.0 * r;
        rate *= r;
        static const int dil[3] = {1, 3, 9};
        for (int u = 0; u < 3; u++) {
            const std::string q = p + "ru." + std::to_string(u) + ".";
            m.make_snake(B.ru[u].s1, g, q + "snake1.alpha", q + "snake1.beta", co);
            m.make_conv(B.ru[u].c1, g, q + "conv1.weight", q + "conv1.bias", co, co, 7, dil[u]);
            m.make_snake(B.ru[u].s2, g, q + "snake2.alpha", q + "snake2.beta", co);
            m.make_conv(B.ru[u].c2, g, q + "conv2.weight", q + "conv2.bias", co, co, 1, 1);
            fl += rate * 2.0 * co * co * 8.0;
        }
        ch = co;
    }
    m.make_snake(m.snake_out, g, "codec.dec.snake_out.alpha", "codec.dec.snake_out.beta", ch);
    {   // conv_out weight [1][ch][7] -> [j][c]
        auto w = Impl::tensor(g, "codec.dec.conv_out.weight");
        std::vector<float> r((size_t)7 * ch);
        for (int c = 0; c < ch; c++) for (int j = 0; j < 7; j++) r[(size_t)j * ch + c] = w[(size_t)c * 7 + j];
        Impl::up_f(m.conv_out_w, r);
        m.conv_out_b = Impl::tensor(g, "codec.dec.conv_out.bias")[0];
        fl += rate * 2.0 * 7.0 * ch;
    }
    m.flops_frame = fl;
    // ---- per-stream state + scratch ----
    const int T0 = max_frames, GM = m.gmax;
    m.make_ext(m.z_ext, 2, m.cb_dim, T0);
    m.k_ext.resize(m.n_layers); m.v_ext.resize(m.n_layers);
    for (int l = 0; l < m.n_layers; l++) { m.make_ext(m.k_ext[l], m.window - 1, H, T0); m.make_ext(m.v_ext[l], m.window - 1, H, T0); }
    int T = T0;
    m.dw_ext.resize(m.n_up);
    for (int i = 0; i < m.n_up; i++) { T *= m.up_ratios[i]; m.make_ext(m.dw_ext[i], 6, H, T); }
    const int Tlat = T;
    m.make_ext(m.convin_ext, 6, H, Tlat);
    m.ct_ext.resize(m.n_dec); m.ru_ext.resize(m.n_dec);
    size_t max_act = (size_t)Tlat * 4 * H;
    ch = m.dec_dim;
    static const int dil[3] = {1, 3, 9};
    for (int b = 0; b < m.n_dec; b++) {
        m.make_ext(m.ct_ext[b], 1, ch, T);
        T *= m.dec_rates[b]; ch /= 2;
        m.ru_ext[b].resize(3);
        for (int u = 0; u < 3; u++) m.make_ext(m.ru_ext[b][u], 6 * dil[u], ch, T);
        max_act = std::max(max_act, (size_t)T * ch);
    }
    m.make_ext(m.out_ext, 6, ch, T);
    max_act = std::max(max_act, (size_t)Tlat * m.dec_dim);
    max_act *= GM;
    m.kv_len.assign(n_streams, 0); m.n_seen.assign(n_streams, 0);
    m.lanes.resize(n_lanes > 0 ? n_lanes : 1);
    for (auto& L : m.lanes) {
        const size_t R0 = (size_t)GM * T0;
        L.h.alloc(R0 * H); L.xn.alloc(R0 * H); L.qkv.alloc(R0 * 3 * H); L.att.alloc(R0 * H);
        L.gu.alloc(R0 * 2 * m.ffn); L.act.alloc(R0 * m.ffn);
        L.t1.alloc(max_act); L.t2.alloc(max_act); L.pcm.alloc((size_t)GM * T);
        L.work.resize(m.all_ext.size());
        for (Ext* e : m.all_ext) L.work[e->id].alloc((size_t)GM * (e->H + e->Tmax) * e->C);
        L.d_in.alloc(m.in_slot());
        L.splitk_ws.alloc(std::max<size_t>(max_act * 2, (size_t)1 << 20));
        Q3_HIP(hipHostMalloc((void**)&L.h_in, (size_t)m.ring * m.in_slot() * sizeof(int64_t)));
    }
    m.S = &m.lanes[0];
    Q3_HIP(hipDeviceSynchronize());
}
CodecDecoder::~CodecDecoder() { if (impl_) for (auto& L : impl_->lanes) if (L.h_in) (void)hipHostFree(L.h_in); }
int CodecDecoder::max_group() const { return impl_->gmax; }
int CodecDecoder::n_lanes() const { return (int)impl_->lanes.size(); }
int CodecDecoder::samples_per_frame() const {
    int s = 1;
    for (int i = 0; i < impl_->n_up; i++) s *= impl_->up_ratios[i];
    for (int i = 0; i < impl_->n_dec; i++) s *= impl_->dec_rates[i];
    return s;
}
double CodecDecoder::flops_per_frame() const { return impl_->flops_frame; }

void CodecDecoder::reset(int s) {
    Impl& m = *impl_;
    Q3_CHECK(s >= 0 && s < m.n_streams, "stream out of range");
    std::lock_guard<std::mutex> cap(capture_mutex());
    for (Ext* e : m.all_ext) if (e->H) Q3_HIP(hipMemset(e->hist.p + (size_t)s * e->H * e->C, 0, (size_t)e->H * e->C * 4));
    m.kv_len[s] = 0; m.n_seen[s] = 0;
}

// ---- DecoderState export / import (onnx.rs:461-496: the reference carries the streaming state as named tensors between decode calls;
// here it lives on the device, and these calls move one stream's state out / in as a flat f32 blob with a published layout) ----
std::vector<CodecStateEntry> CodecDecoder::state_layout() const {
    const Impl& m = *impl_;
    std::vector<CodecStateEntry> out;
    size_t off = 0;
    auto add = [&](const std::string& name, const Ext& e) {
        if (!e.H) return;
        out.push_back(CodecStateEntry{name, (int64_t)off, e.H, e.C});
        off += (size_t)e.H * e.C;
    };
    add("pre_conv_history", m.z_ext);                                   // onnx.rs:476 (rows = time, columns = channels)
    for (size_t l = 0; l < m.k_ext.size(); l++) { add("past_key_" + std::to_string(l), m.k_ext[l]); add("past_value_" + std::to_string(l), m.v_ext[l]); } // :484-492
    for (size_t i = 0; i < m.dw_ext.size(); i++) add("conv_history.upsample." + std::to_string(i), m.dw_ext[i]);            // :480 conv_history, split
    add("conv_history.conv_in", m.convin_ext);
    for (size_t b = 0; b < m.ct_ext.size(); b++) {
        add("conv_history.block." + std::to_string(b) + ".transposed", m.ct_ext[b]);
        for (size_t u = 0; u < m.ru_ext[b].size(); u++) add("conv_history.block." + std::to_string(b) + ".unit." + std::to_string(u), m.ru_ext[b][u]);
    }
    add("conv_history.conv_out", m.out_ext);
    out.push_back(CodecStateEntry{"counters(kv_len,frames_seen)", (int64_t)off, 1, 2});
    return out;
}
size_t CodecDecoder::state_floats() const { const auto l = state_layout(); return (size_t)l.back().offset + 2; }
void CodecDecoder::state_export(int s, float* out, size_t n_floats) const {
    const Impl& m = *impl_;
    Q3_CHECK(s >= 0 && s < m.n_streams && out, "stream out of range");
    Q3_CHECK(n_floats == state_floats(), "state_export: buffer holds " + std::to_string(n_floats) + " floats, the state has " + std::to_string(state_floats()));
    std::lock_guard<std::mutex> cap(capture_mutex()); // synchronous copies must not overlap another thread's stream capture (engine.h)
    Q3_HIP(hipDeviceSynchronize()); // the stream's pending decodes must have landed
    size_t off = 0;
    auto one = [&](const Ext& e) { if (!e.H) return; const size_t n = (size_t)e.H * e.C; Q3_HIP(hipMemcpy(out + off, e.hist.p + (size_t)s * n, n * 4, hipMemcpyDeviceToHost)); off += n; };
    one(m.z_ext);
    for (size_t l = 0; l < m.k_ext.size(); l++) { one(m.k_ext[l]); one(m.v_ext[l]); }
    for (auto& e : m.dw_ext) one(e);
    one(m.convin_ext);
    for (size_t b = 0; b < m.ct_ext.size(); b++) { one(m.ct_ext[b]); for (auto& e : m.ru_ext[b]) one(e); }
    one(m.out_ext);
    out[off] = (float)m.kv_len[s]; out[off + 1] = (float)m.n_seen[s];
}
void CodecDecoder::state_import(int s, const float* in, size_t n_floats) {
    Impl& m = *impl_;
    Q3_CHECK(s >= 0 && s < m.n_streams && in, "stream out of range");
    Q3_CHECK(n_floats == state_floats(), "state_import: blob holds " + std::to_string(n_floats) + " floats, this decoder's state has " + std::to_string(state_floats()));
    // the trailer drives device-side indexing of the KV / history buffers on the next decode: validate before anything is committed
    const float kvf = in[n_floats - 2], seenf = in[n_floats - 1];
    Q3_CHECK(kvf == kvf && seenf == seenf && kvf >= 0.0f && kvf <= (float)(m.window - 1) && kvf == (float)(int)kvf, "state_import: cached-position count in the blob is not an integer in [0, window)");
    Q3_CHECK(seenf >= 0.0f && seenf < 9.0e15f && seenf == (float)(long long)seenf, "state_import: frame counter in the blob is negative or not finite");
    std::lock_guard<std::mutex> cap(capture_mutex());
    Q3_HIP(hipDeviceSynchronize());
    size_t off = 0;
    auto one = [&](Ext& e) { if (!e.H) return; const size_t n = (size_t)e.H * e.C; Q3_HIP(hipMemcpy(e.hist.p + (size_t)s * n, in + off, n * 4, hipMemcpyHostToDevice)); off += n; };
    one(m.z_ext);
    for (size_t l = 0; l < m.k_ext.size(); l++) { one(m.k_ext[l]); one(m.v_ext[l]); }
    for (auto& e : m.dw_ext) one(e);
    one(m.convin_ext);
    for (size_t b = 0; b < m.ct_ext.size(); b++) { one(m.ct_ext[b]); for (auto& e : m.ru_ext[b]) one(e); }
    one(m.out_ext);
    m.kv_len[s] = (int)in[off]; m.n_seen[s] = (long long)in[off + 1];
}

void CodecDecoder::reset_async(hipStream_t st, int s) {
    Impl& m = *impl_;
    Q3_CHECK(s >= 0 && s < m.n_streams, "stream out of range");
    HistTable tab{};
    size_t mx = 0;
    auto flush = [&]() {
        if (tab.n) hipLaunchKernelGGL(k_hist_zero, dim3((unsigned)((mx + 255) / 256), 1, tab.n), dim3(256), 0, st, tab, s);
        tab.n = 0; mx = 0;
    };
    for (Ext* e : m.all_ext) {
        if (!e->H) continue;
        HistDesc d{}; d.hist = e->hist.p; d.H = e->H; d.C = e->C;
        tab.d[tab.n++] = d;
        mx = std::max(mx, (size_t)e->H * e->C);
        if (tab.n == 48) flush();
    }
    flush();
    m.kv_len[s] = 0; m.n_seen[s] = 0;
}

int CodecDecoder::decode(hipStream_t st, int s, const int64_t* codes, int n_frames, bool is_last, float* pcm) {
    const int T = decode_async(st, s, codes, n_frames, is_last, pcm, 0);
    Q3_HIP(hipStreamSynchronize(st));
    return T;
}
int CodecDecoder::decode_async(hipStream_t st, int s, const int64_t* codes, int n_frames, bool is_last, float* pcm, int lane) {
    (void)is_last; // causal stack: nothing is held back (valid_samples == everything)
    return decode_group_async(st, 1, &s, codes, n_frames, &pcm, lane);
}

// One pass for G streams that each contribute n_frames new frames: every GEMM runs over G*T rows, so the 733 MB of
// codec weights are streamed once per group instead of once per stream, and one set of launches serves the group.
int CodecDecoder::decode_group_async(hipStream_t st, int G, const int* streams, const int64_t* codes, int n_frames, float* const* pcm, int lane) {
    Impl& m = *impl_;
    Q3_CHECK(lane >= 0 && lane < (int)m.lanes.size(), "codec lane out of range");
    Q3_CHECK(G >= 1 && G <= m.gmax, "group size out of range");
    m.S = &m.lanes[lane];
    if (n_frames <= 0) return 0;
    Q3_CHECK(n_frames <= m.max_frames, "too many frames per decode call");
    for (int g = 0; g < G; g++) {
        Q3_CHECK(streams[g] >= 0 && streams[g] < m.n_streams, "stream out of range");
        for (int h = 0; h < g; h++) Q3_CHECK(streams[h] != streams[g], "a stream may appear once per group");
    }
    const int H = m.hidden, T0 = n_frames, R0 = G * T0, W = m.window;
    if (m.S->ring_idx == m.ring) { Q3_HIP(hipStreamSynchronize(st)); m.S->ring_idx = 0; } // all earlier uploads have been consumed
    int64_t* hin = m.S->h_in + (size_t)(m.S->ring_idx++) * m.in_slot();
    const size_t ncodes = (size_t)R0 * m.n_q;
    std::copy(codes, codes + ncodes, hin);
    for (int g = 0; g < G; g++) { hin[ncodes + g] = streams[g]; hin[ncodes + G + g] = m.kv_len[streams[g]]; hin[ncodes + 2 * G + g] = m.n_seen[streams[g]]; }
    Q3_HIP(hipMemcpyAsync(m.S->d_in.p, hin, (ncodes + 3 * (size_t)G) * 8, hipMemcpyHostToDevice, st));
    const int64_t* d_codes = m.S->d_in.p;
    const int64_t* meta = m.S->d_in.p + ncodes;
    {   // every streaming conv of the pass with its row count per stream
        m.table.n = 0;
        m.plan_hist(m.z_ext, T0);
        for (int l = 0; l < m.n_layers; l++) { m.plan_hist(m.k_ext[l], T0); m.plan_hist(m.v_ext[l], T0); }
        int Tp = T0;
        for (int i = 0; i < m.n_up; i++) { Tp *= m.up_ratios[i]; m.plan_hist(m.dw_ext[i], Tp); }
        m.plan_hist(m.convin_ext, Tp);
        for (int b = 0; b < m.n_dec; b++) {
            m.plan_hist(m.ct_ext[b], Tp);
            Tp *= m.dec_rates[b];
            for (int u = 0; u < 3; u++) m.plan_hist(m.ru_ext[b][u], Tp);
        }
        m.plan_hist(m.out_ext, Tp);
        m.move_hist(st, G, meta, false);
    }
    // 1. RVQ sum -> z_ext current rows ; 2. pre_conv
    hipLaunchKernelGGL(k_rvq_sum, dim3((m.cb_dim + 255) / 256, R0), dim3(256), 0, st, d_codes, m.d_cb_ptrs.p, m.n_q, m.cb_size, m.cb_dim,
                       m.work(m.z_ext), m.cb_dim, m.cur_map(m.z_ext, T0));
    m.run_conv(st, m.pre_conv, m.ext_base(m.z_ext, T0), m.cb_dim, R0, m.plain(m.S->h.p), H, H);
    // 3. transformer
    for (int l = 0; l < m.n_layers; l++) {
        auto& L = m.tf[l];
        Ext &ke = m.k_ext[l], &ve = m.v_ext[l];
        hipLaunchKernelGGL(k_rmsnorm_rows, dim3(R0), dim3(256), 0, st, m.S->h.p, H, L.attn_norm.p, H, m.eps, m.S->xn.p, H);
        m.run_conv(st, L.wqkv, m.plain(m.S->xn.p), H, R0, m.plain(m.S->qkv.p), 3 * H, 3 * H);
        hipLaunchKernelGGL(k_codec_rope_append, dim3((H / 2 + 255) / 256, R0), dim3(256), 0, st, m.S->qkv.p, 3 * H, H, m.head_dim, m.rope_c.p, m.rope_s.p,
                           meta, G, T0, m.max_pos, m.work(ke), m.work(ve), m.cur_map(ke, T0));
        hipLaunchKernelGGL(k_codec_attn, dim3(m.n_heads, R0), dim3(64), 0, st, m.S->qkv.p, 3 * H, H, m.head_dim, m.work(ke), m.work(ve), meta, G, T0, W,
                           m.S->att.p, H);
        m.run_conv(st, L.wo, m.plain(m.S->att.p), H, R0, m.plain(m.S->h.p), H, H, EPI_RES_SCALE, m.plain(m.S->h.p), H, L.ls_attn.p);
        hipLaunchKernelGGL(k_rmsnorm_rows, dim3(R0), dim3(256), 0, st, m.S->h.p, H, L.ffn_norm.p, H, m.eps, m.S->xn.p, H);
        m.run_conv(st, L.wgu, m.plain(m.S->xn.p), H, R0, m.plain(m.S->gu.p), 2 * m.ffn, 2 * m.ffn);
        hipLaunchKernelGGL(k_swiglu_rows, dim3((m.ffn + 255) / 256, R0), dim3(256), 0, st, m.S->gu.p, m.ffn, m.S->act.p);
        m.run_conv(st, L.wdown, m.plain(m.S->act.p), m.ffn, R0, m.plain(m.S->h.p), H, H, EPI_RES_SCALE, m.plain(m.S->h.p), H, L.ls_ffn.p);
    }
    for (int g = 0; g < G; g++) {
        const int s = streams[g], tot = m.kv_len[s] + T0;
        m.kv_len[s] = tot < W - 1 ? tot : W - 1;
        m.n_seen[s] += T0;
    }
    // 4. final norm -> t1 [R0][H]
    hipLaunchKernelGGL(k_rmsnorm_rows, dim3(R0), dim3(256), 0, st, m.S->h.p, H, m.tf_norm.p, H, m.eps, m.S->t1.p, H);
    // 5. upsample stages: x in t1 (plain rows; T = rows per stream)
    int T = T0;
    float* x = m.S->t1.p;
    for (int i = 0; i < m.n_up; i++) {
        auto& U = m.up[i];
        const int f = m.up_ratios[i];
        Ext& e = m.dw_ext[i];
        // transposed conv: GEMM row (g,t) writes f consecutive H-wide rows of segment g: [T][f*H] == [T*f][H]
        m.run_conv(st, U.ct, m.plain(x), H, G * T, Impl::Loc{m.work(e) + (size_t)e.H * e.C, T, e.H}, f * H, H);
        T *= f;
        hipLaunchKernelGGL(k_dwconv7, dim3((H + 255) / 256, G * T), dim3(256), 0, st, m.work(e), H, U.dw_w.p, U.dw_b.p, m.S->t2.p, m.base_map(e, T));
        hipLaunchKernelGGL(k_layernorm_rows, dim3(G * T), dim3(256), 0, st, m.S->t2.p, H, U.ln_w.p, U.ln_b.p, H, 1e-6f, m.S->t2.p, H);
        float* m1 = m.S->t1.p; // [G*T][4H]
        m.run_conv(st, U.pw1, m.plain(m.S->t2.p), H, G * T, m.plain(m1), 4 * H, 4 * H, EPI_GELU);
        // y_new = y + gamma * (pw2(m1) + b): write into t2 (y lives in the ext buffer)
        m.run_conv(st, U.pw2, m.plain(m1), 4 * H, G * T, m.plain(m.S->t2.p), H, H, EPI_RES_SCALE, m.ext_cur(e, T), H, U.gamma.p);
        // next stage input must not alias its own output buffers: move to t1
        m.copy_rows(st, m.S->t2.p, m.S->t1.p, G * T, H, plain_map());
        x = m.S->t1.p;
    }
    // 6. conv_in
    m.copy_rows(st, x, m.work(m.convin_ext), G * T, H, m.cur_map(m.convin_ext, T));
    m.run_conv(st, m.conv_in, m.ext_base(m.convin_ext, T), H, G * T, m.plain(m.S->t1.p), m.dec_dim, m.dec_dim);
    float* d = m.S->t1.p; // [G*T][ch]
    // 7. decoder blocks
    for (int b = 0; b < m.n_dec; b++) {
        auto& B = m.blk[b];
        Ext& ce = m.ct_ext[b];
        m.snake(st, B.snake, d, m.work(ce), G * T, m.cur_map(ce, T), m.presplit);
        float* y = (d == m.S->t1.p) ? m.S->t2.p : m.S->t1.p; // [G*T*r][co]
        m.run_conv(st, B.ct, m.ext_base(ce, T), B.cin, G * T, m.plain(y), B.rate * B.cout, B.cout, EPI_NONE, Impl::Loc{nullptr, SEG_NONE, 0}, 0, nullptr, nullptr, m.presplit);
        T *= B.rate;
        for (int u = 0; u < 3; u++) {
            Ext& re = m.ru_ext[b][u];
            auto& R = B.ru[u];
            m.snake(st, R.s1, y, m.work(re), G * T, m.cur_map(re, T), m.presplit);
            float* c1o = d; // the block input buffer is free now: reuse as scratch [G*T][co]
            m.run_conv(st, R.c1, m.ext_base(re, T), B.cout, G * T, m.plain(c1o), B.cout, B.cout, EPI_SNAKE, Impl::Loc{nullptr, SEG_NONE, 0}, 0, nullptr, &R.s2, m.presplit, m.presplit); // conv1 + snake2 fused
            m.run_conv(st, R.c2, m.plain(c1o), B.cout, G * T, m.plain(y), B.cout, B.cout, EPI_RES, m.plain(y), B.cout, nullptr, nullptr, m.presplit);
        }
        d = y;
    }
    // 8. output conv
    m.snake(st, m.snake_out, d, m.work(m.out_ext), G * T, m.cur_map(m.out_ext, T));
    if (m.out_ext.C == 96 && T % 128 == 0) // (T = frames x 1920: every chunk length qualifies at the full decoder width)
        hipLaunchKernelGGL((k_conv_out_win<96>), dim3(G * T / 128), dim3(256), 0, st, m.work(m.out_ext), m.conv_out_w.p, m.conv_out_b, m.S->pcm.p, G * T, m.base_map(m.out_ext, T));
    else
    hipLaunchKernelGGL(k_conv_out_wave, dim3((G * T + 3) / 4), dim3(256), 0, st, m.work(m.out_ext), m.out_ext.C, m.conv_out_w.p, m.conv_out_b, m.S->pcm.p, G * T,
                       m.base_map(m.out_ext, T));
    m.move_hist(st, G, meta, true);
    Q3_LAUNCH_CHECK();
    for (int g = 0; g < G; g++) Q3_HIP(hipMemcpyAsync(pcm[g], m.S->pcm.p + (size_t)g * T, (size_t)T * 4, hipMemcpyDeviceToHost, st));
    return T;
}

} // namespace q3
