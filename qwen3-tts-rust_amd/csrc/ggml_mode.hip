// ggml_mode.hip -- the opt-in "ggml-CPU" arithmetic mode of the transformer on the device (Q3_SPEC=ggml).
//
// The reference's codec tokens come out of llama.cpp b8123's kernels (/root/reference/src/models/llama/mod.rs:442-451, version pin
// src/download.rs:207-221), whose arithmetic differs from include/q3tts_spec.h's in ways that flip a few per cent of greedy tokens on
// synthetic weights: activations quantised to Q8_K 256-blocks for K-quant rows, roundf for Q8_0 activations, block sums
// accumulated without fma in ggml's order, sums of squares and softmax denominators in double.  The test suite holds a CPU restatement of that
// arithmetic; these kernels restate the SAME arithmetic on the device, bit for bit (tests/test_gpu_parity.py::test_ggml_mode_engine_matches_oracle),
// so that the day llama.cpp b8123 and the real weights are at hand the GPU can be put beside the reference at all.
//
// This is a correctness path, not the product's fast path: one thread per (row, token tile) walks the GGUF blocks exactly as stored (a raw
// copy of every matrix is uploaded only when the mode is on), sums run in the reference order, and nothing here is tuned.
#include "ggml_mode.h"
#include "kdev.h"

namespace q3 {

static __device__ __forceinline__ uint16_t ld16(const uint8_t* p) { return (uint16_t)p[0] | ((uint16_t)p[1] << 8); }

// h[tok][0..d) = input row (plain, int32-indexed or argmax-key-indexed table row)
__global__ void k_gg_load_rows(const float* __restrict__ x, int x_stride, const int32_t* __restrict__ idx, int idx_stride,
                               const unsigned long long* __restrict__ idx_keys, int d, float* __restrict__ h) {
    const int tok = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= d) return;
    size_t row = (size_t)tok;
    if (idx_keys) row = (size_t)key_code(idx_keys[(size_t)tok * idx_stride]);
    else if (idx) row = (size_t)idx[(size_t)tok * idx_stride];
    h[(size_t)tok * d + i] = x[row * x_stride + i];
}
__global__ void k_gg_add(float* __restrict__ h, const float* __restrict__ o, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) h[i] = h[i] + o[i];
}
// ggml_compute_forward_rms_norm_f32: sum of squares in double in index order, scale = 1 / sqrtf(mean + eps); y = (x * scale) * g
__global__ void __launch_bounds__(64) k_gg_rmsnorm(const float* __restrict__ x, const float* __restrict__ g, int d, float eps, float* __restrict__ y) {
    __shared__ float scale_s;
    const int tok = blockIdx.x;
    const float* xr = x + (size_t)tok * d;
    if (threadIdx.x == 0) {
        double sum = 0.0;
        for (int i = 0; i < d; i++) sum += (double)(xr[i] * xr[i]);
        const float mean = (float)(sum / (double)d);
        scale_s = 1.0f / q3_sqrtf(mean + eps);
    }
    __syncthreads();
    const float sc = scale_s;
    for (int i = threadIdx.x; i < d; i += 64) y[(size_t)tok * d + i] = (xr[i] * sc) * g[i];
}
// quantize_row_q8_0_ref (d = amax / 127, q = roundf(x / d), f16 scale) per 32 and quantize_row_q8_K_ref (iscale = -127 / max, nearest_int, f32 scale,
// sums of 16) per 256: both for every row, the dot kernels pick by weight type
__global__ void __launch_bounds__(64) k_gg_quant(const float* __restrict__ x, int k, int8_t* __restrict__ q8, uint16_t* __restrict__ d8, int8_t* __restrict__ qk,
                                                 float* __restrict__ dk, int16_t* __restrict__ bs) {
    const int tok = blockIdx.y, b = blockIdx.x * 64 + threadIdx.x; // 32-block index
    const float* xr = x + (size_t)tok * k;
    if (b < k / 32) {
        float amax = 0.0f;
        for (int j = 0; j < 32; j++) { const float v = q3_fabsf(xr[32 * b + j]); if (v > amax) amax = v; }
        const float d = amax / 127.0f;
        const float id = d != 0.0f ? 1.0f / d : 0.0f;
        d8[(size_t)tok * (k / 32) + b] = f2h(d);
        for (int j = 0; j < 32; j++) q8[(size_t)tok * k + 32 * b + j] = (int8_t)__builtin_roundf(xr[32 * b + j] * id);
    }
    if (b < k / 256) {
        const float* xs = xr + 256 * b;
        int8_t* qs = qk + (size_t)tok * k + 256 * b;
        int16_t* bsum = bs + (size_t)tok * (k / 16) + 16 * b;
        float mx = 0.0f, amax = 0.0f;
        for (int j = 0; j < 256; j++) { const float ax = q3_fabsf(xs[j]); if (ax > amax) { amax = ax; mx = xs[j]; } }
        if (amax == 0.0f) {
            dk[(size_t)tok * (k / 256) + b] = 0.0f;
            for (int j = 0; j < 256; j++) qs[j] = 0;
            for (int j = 0; j < 16; j++) bsum[j] = 0;
        } else {
            const float iscale = -127.f / mx;
            for (int j = 0; j < 256; j++) { const int v = q3_nearest_int_ggml(iscale * xs[j]); qs[j] = (int8_t)(v < 127 ? v : 127); }
            for (int j = 0; j < 16; j++) { int sum = 0; for (int ii = 0; ii < 16; ii++) sum += qs[j * 16 + ii]; bsum[j] = (int16_t)sum; }
            dk[(size_t)tok * (k / 256) + b] = 1.0f / iscale;
        }
    }
}

// ---- ggml_vec_dot_*_generic, one (row, token) pair ----
static __device__ float gg_dot_q8_0(const uint8_t* row, int k, const int8_t* xq, const uint16_t* xd) {
    float sumf = 0.0f;
    for (int ib = 0; ib < k / 32; ib++) {
        const uint8_t* blk = row + 34 * ib;
        const int8_t* qs = reinterpret_cast<const int8_t*>(blk + 2);
        int sumi = 0;
        for (int j = 0; j < 32; j++) sumi += (int)qs[j] * (int)xq[32 * ib + j];
        const float sc = h2f(ld16(blk)) * h2f(xd[ib]);
        sumf = sumf + (float)sumi * sc;
    }
    return sumf;
}
static __device__ float gg_dot_q5_K(const uint8_t* row, int k, const int8_t* xq, const float* xdk, const int16_t* xbs) {
    float sums[8], sumf = 0.0f;
    for (int l = 0; l < 8; l++) sums[l] = 0.0f;
    for (int i = 0; i < k / 256; i++) {
        const uint8_t* blk = row + 176 * i;
        const uint8_t* sc12 = blk + 4; const uint8_t* hm = blk + 16; const uint8_t* q4 = blk + 48;
        const int8_t* q8 = xq + 256 * i;
        int aux32[8];
        for (int l = 0; l < 8; l++) aux32[l] = 0;
        int scales[8], mins[8];
        for (int j = 0; j < 8; j++) {
            if (j < 4) { scales[j] = sc12[j] & 63; mins[j] = sc12[j + 4] & 63; }
            else { scales[j] = (sc12[j + 4] & 0xF) | ((sc12[j - 4] >> 6) << 4); mins[j] = (sc12[j + 4] >> 4) | ((sc12[j] >> 6) << 4); }
        }
        int sumi = 0;
        for (int j = 0; j < 16; j++) sumi += (int)xbs[16 * i + j] * mins[j / 2];
        for (int j = 0; j < 8; j++) { // sub-block j: elements 32 j .. 32 j + 31 = nibble (j & 1) of q4[32 (j >> 1) + l], high bit (j) of hm[l]
            const int scale = scales[j];
            for (int gq = 0; gq < 4; gq++)
                for (int l = 0; l < 8; l++) {
                    const int e = 8 * gq + l;
                    const int nib = (j & 1) ? (q4[32 * (j >> 1) + e] >> 4) : (q4[32 * (j >> 1) + e] & 0xF);
                    const int a = nib + ((hm[e] >> j) & 1 ? 16 : 0);
                    const int16_t p = (int16_t)((int)q8[32 * j + e] * a);
                    aux32[l] += scale * (int)p;
                }
        }
        const float d = h2f(ld16(blk)) * xdk[i];
        for (int l = 0; l < 8; l++) sums[l] = sums[l] + d * (float)aux32[l];
        const float dmin = h2f(ld16(blk + 2)) * xdk[i];
        sumf = sumf - dmin * (float)sumi;
    }
    for (int l = 0; l < 8; l++) sumf = sumf + sums[l];
    return sumf;
}
static __device__ float gg_dot_q6_K(const uint8_t* row, int k, const int8_t* xq, const float* xdk) {
    float sums[8], sumf = 0.0f;
    for (int l = 0; l < 8; l++) sums[l] = 0.0f;
    for (int i = 0; i < k / 256; i++) {
        const uint8_t* blk = row + 210 * i;
        const int8_t* sc = reinterpret_cast<const int8_t*>(blk + 192);
        const int8_t* q8 = xq + 256 * i;
        int aux32[8];
        for (int l = 0; l < 8; l++) aux32[l] = 0;
        for (int j = 0; j < 16; j++) { // 16-element sub-block j: elements 16 j .. 16 j + 15
            const int scale = sc[j];
            for (int gq = 0; gq < 2; gq++)
                for (int l = 0; l < 8; l++) {
                    const int e = 16 * j + 8 * gq + l;            // element of the super-block
                    const int half = e >> 7, r = e & 127, grp = r >> 5, ll = r & 31;
                    const uint8_t* q4 = blk + 64 * half; const uint8_t* qh = blk + 128 + 32 * half;
                    const int lo = (grp & 1) ? q4[ll + 32] : q4[ll];
                    const int nib = (grp >= 2) ? (lo >> 4) : (lo & 0xF);
                    const int a = (int)(int8_t)(nib | (((qh[ll] >> (2 * grp)) & 3) << 4)) - 32;
                    const int16_t p = (int16_t)((int)q8[e] * a);
                    aux32[l] += scale * (int)p;
                }
        }
        const float d = h2f(ld16(blk + 208)) * xdk[i];
        for (int l = 0; l < 8; l++) sums[l] = sums[l] + d * (float)aux32[l];
    }
    for (int l = 0; l < 8; l++) sumf = sumf + sums[l];
    return sumf;
}
// out[tok][r] = W[row0 + r] . x[tok]: one thread per (row, token)
__global__ void __launch_bounds__(64) k_gg_matvec(GgMat w, int row0, int nrows, GgAct a, float* __restrict__ out, int out_stride, int ntok) {
    const int r = blockIdx.x * 64 + threadIdx.x, tok = blockIdx.y;
    if (r >= nrows || tok >= ntok) return;
    const uint8_t* row = w.p + (size_t)(row0 + r) * w.row_bytes;
    const int k = w.k;
    float y;
    if (w.type == Q3_T_Q8_0) y = gg_dot_q8_0(row, k, a.q8 + (size_t)tok * k, a.d8 + (size_t)tok * (k / 32));
    else if (w.type == Q3_T_Q5_K) y = gg_dot_q5_K(row, k, a.qk + (size_t)tok * k, a.dk + (size_t)tok * (k / 256), a.bs + (size_t)tok * (k / 16));
    else y = gg_dot_q6_K(row, k, a.qk + (size_t)tok * k, a.dk + (size_t)tok * (k / 256));
    out[(size_t)tok * out_stride + r] = y;
}

// per-head RMSNorm (double sum over 128 in order) + NeoX (M-)RoPE from the spec's tables; q rotated in place, k / v appended to the f16 cache
__global__ void __launch_bounds__(64) k_gg_qk_rope_append(float* __restrict__ qkv, int stride, int n_head, int n_kv, const float* __restrict__ q_norm_w,
                                                          const float* __restrict__ k_norm_w, float eps, const float* __restrict__ rope_cos,
                                                          const float* __restrict__ rope_sin, int n_ctx, const int32_t* __restrict__ mrope_sec,
                                                          TokMeta tm, KvCache kv, int layer) {
    __shared__ float scale_s;
    const int hh = blockIdx.x, tok = blockIdx.y, lane = threadIdx.x;
    const int seq = tm.seq_of(tok), slot = tm.slot_of(tok);
    const int page = kv.page_of(seq, slot >> 6), ps = slot & 63;
    if (hh < n_head + n_kv) {
        float* vec = qkv + (size_t)tok * stride + (size_t)hh * 128;
        const float* wn = hh < n_head ? q_norm_w : k_norm_w;
        if (lane == 0) {
            double sum = 0.0;
            for (int i = 0; i < 128; i++) sum += (double)(vec[i] * vec[i]);
            const float mean = (float)(sum / 128.0);
            scale_s = 1.0f / q3_sqrtf(mean + eps);
        }
        __syncthreads();
        const float sc = scale_s;
        const float y1 = (vec[lane] * sc) * wn[lane], y2 = (vec[lane + 64] * sc) * wn[lane + 64];
        int32_t sec[4] = {mrope_sec[0], mrope_sec[1], mrope_sec[2], mrope_sec[3]};
        int pp = tm.pos_of(tok, q3_mrope_stream(lane, sec));
        if (pp < 0) pp = 0;
        if (pp > n_ctx - 1) pp = n_ctx - 1;
        float o1, o2;
        q3_rope_pair(y1, y2, rope_cos[(size_t)pp * 64 + lane], rope_sin[(size_t)pp * 64 + lane], &o1, &o2);
        if (hh < n_head) { vec[lane] = o1; vec[lane + 64] = o2; }
        else {
            uint16_t* Kw = kv.k + (size_t)page * kv.page_stride() + (size_t)layer * kv.layer_stride() + (size_t)(hh - n_head) * 8192;
            Kw[((lane >> 3) * 64 + ps) * 8 + (lane & 7)] = f2h(o1);
            Kw[(((lane + 64) >> 3) * 64 + ps) * 8 + (lane & 7)] = f2h(o2);
        }
    } else {
        const int kvh = hh - n_head - n_kv;
        const float* vec = qkv + (size_t)tok * stride + (size_t)(n_head + n_kv + kvh) * 128;
        uint16_t* Vw = kv.v + (size_t)page * kv.page_stride() + (size_t)layer * kv.layer_stride() + (size_t)kvh * 8192;
        Vw[ps * 128 + lane] = f2h(vec[lane]); Vw[ps * 128 + lane + 64] = f2h(vec[lane + 64]);
    }
}
// one query head against its n cached f16 positions (the CPU restatement of the mode does the same): scores = plain dot in index order * 1/sqrt(128), expf, softmax sum in double
// in position order, PV in position order, times (float)(1 / sum)
__global__ void __launch_bounds__(64) k_gg_attention(const float* __restrict__ qkv, int stride, int n_head, int n_kv, TokMeta tm, KvCache kv, int layer,
                                                     float* __restrict__ att, float* __restrict__ scores /* [ntok][n_head][n_ctx] */, int n_ctx) {
    __shared__ float mx_s, inv_s;
    const int h = blockIdx.x, tok = blockIdx.y, lane = threadIdx.x;
    const int seq = tm.seq_of(tok), n = tm.slot_of(tok) + 1, kvh = h / (n_head / n_kv);
    const float* q = qkv + (size_t)tok * stride + (size_t)h * 128;
    float* s = scores + ((size_t)tok * n_head + h) * n_ctx;
    const size_t head_off = (size_t)layer * kv.layer_stride() + (size_t)kvh * 8192;
    const float scale = 0.08838834764831845f;
    float mloc = -INFINITY;
    for (int j = lane; j < n; j += 64) {
        const uint16_t* Kb = kv.k + (size_t)kv.page_of(seq, j >> 6) * kv.page_stride() + head_off;
        float acc = 0.0f;
        for (int d = 0; d < 128; d++) acc = acc + q[d] * h2f(Kb[((d >> 3) * 64 + (j & 63)) * 8 + (d & 7)]);
        s[j] = acc * scale;
        mloc = fmaxf(mloc, s[j]);
    }
    mloc = wave_max_bfly(mloc);
    __syncthreads();
    for (int j = lane; j < n; j += 64) s[j] = q3_expf_ggml(s[j] - mloc);
    __syncthreads();
    if (lane == 0) {
        double sum = 0.0;
        for (int j = 0; j < n; j++) sum += (double)s[j];
        inv_s = (float)(1.0 / sum);
    }
    __syncthreads();
    const float inv = inv_s;
    for (int d = lane; d < 128; d += 64) {
        float acc = 0.0f;
        for (int j = 0; j < n; j++) {
            const uint16_t* Vb = kv.v + (size_t)kv.page_of(seq, j >> 6) * kv.page_stride() + head_off;
            acc = acc + s[j] * h2f(Vb[(j & 63) * 128 + d]);
        }
        att[(size_t)tok * (n_head * 128) + (size_t)h * 128 + d] = acc * inv;
    }
    (void)mx_s;
}
__global__ void k_gg_swiglu(const float* __restrict__ g, const float* __restrict__ u, float* __restrict__ y, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const float gt = g[i]; y[i] = (gt / (1.0f + q3_expf_ggml(-gt))) * u[i]; }
}

// ---- launchers ----
void gg_load_rows(hipStream_t st, const float* x, int x_stride, const int32_t* idx, int idx_stride, const unsigned long long* idx_keys, int d, float* h, int ntok) {
    hipLaunchKernelGGL(k_gg_load_rows, dim3((d + 255) / 256, ntok), dim3(256), 0, st, x, x_stride, idx, idx_stride, idx_keys, d, h);
}
void gg_add(hipStream_t st, float* h, const float* o, size_t n) { hipLaunchKernelGGL(k_gg_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, h, o, n); }
void gg_rmsnorm(hipStream_t st, const float* x, const float* g, int d, float eps, float* y, int ntok) { hipLaunchKernelGGL(k_gg_rmsnorm, dim3(ntok), dim3(64), 0, st, x, g, d, eps, y); }
void gg_quant(hipStream_t st, const float* x, int k, const GgAct& a, int ntok) {
    hipLaunchKernelGGL(k_gg_quant, dim3((k / 32 + 63) / 64, ntok), dim3(64), 0, st, x, k, a.q8, a.d8, a.qk, a.dk, a.bs);
}
void gg_matvec(hipStream_t st, const GgMat& w, int row0, int nrows, const GgAct& a, float* out, int out_stride, int ntok) {
    Q3_CHECK(w.type == Q3_T_Q8_0 || w.type == Q3_T_Q5_K || w.type == Q3_T_Q6_K, "ggml mode serves Q8_0 / Q5_K / Q6_K matrices");
    Q3_CHECK(row0 >= 0 && nrows >= 0 && row0 + nrows <= w.n, "ggml-mode matvec row range");
    if (nrows == 0 || ntok == 0) return;
    hipLaunchKernelGGL(k_gg_matvec, dim3((nrows + 63) / 64, ntok), dim3(64), 0, st, w, row0, nrows, a, out, out_stride, ntok);
}
void gg_qk_rope_append(hipStream_t st, float* qkv, int stride, int n_head, int n_kv, const float* q_norm_w, const float* k_norm_w, float eps, const float* rope_cos,
                       const float* rope_sin, int n_ctx, const int32_t* mrope_sec, const TokMeta& tm, const KvCache& kv, int layer, int ntok) {
    hipLaunchKernelGGL(k_gg_qk_rope_append, dim3(n_head + 2 * n_kv, ntok), dim3(64), 0, st, qkv, stride, n_head, n_kv, q_norm_w, k_norm_w, eps, rope_cos, rope_sin, n_ctx,
                       mrope_sec, tm, kv, layer);
}
void gg_attention(hipStream_t st, const float* qkv, int stride, int n_head, int n_kv, const TokMeta& tm, const KvCache& kv, int layer, float* att, float* scores,
                  int n_ctx, int ntok) {
    hipLaunchKernelGGL(k_gg_attention, dim3(n_head, ntok), dim3(64), 0, st, qkv, stride, n_head, n_kv, tm, kv, layer, att, scores, n_ctx);
}
void gg_swiglu(hipStream_t st, const float* g, const float* u, float* y, size_t n) { hipLaunchKernelGGL(k_gg_swiglu, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g, u, y, n); }

} // namespace q3
