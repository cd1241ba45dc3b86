/*
 * q3o_ggml.c -- ORACLE (test infrastructure): "ggml-CPU" arithmetic mode.
 *
 * The reference's token ids come out of llama.cpp b8123's CPU kernels (/root/reference/src/models/llama/mod.rs:442-451; the
 * binary is downloaded at run time, src/download.rs:207-221, and is not in this image).  include/q3tts_spec.h fixes ONE arithmetic
 * for oracle <-> HIP bit parity; this file restates, from the public ggml sources [EXT], the arithmetic llama.cpp's portable C
 * ("generic") CPU path uses for the same graph, so that the distance between the two -- and therefore how often greedy tokens can
 * differ from a real llama.cpp run -- becomes a measured quantity instead of an assumption (tests/test_ggml_mode_cpu.py).
 *
 * What follows ggml [EXT: file / function names of the b8123 tree, restated from memory of the public sources, unverifiable here]:
 *   - activation quantisation for Q8_0 weights: quantize_row_q8_0_ref (ggml/src/ggml-quants.c): d = amax / 127, q = roundf(x / d)
 *     (round half AWAY from zero -- the spec uses rintf, half to even), scale stored as f16;
 *   - activation quantisation for Q5_K / Q6_K weights: quantize_row_q8_K_ref: per 256 elements, iscale = -127 / max (max = the element
 *     of largest magnitude, sign kept), q = nearest_int(iscale * x) clipped to 127, f32 scale d = 1 / iscale, bsums[16] = sums of 16;
 *   - dot products: ggml_vec_dot_q8_0_q8_0_generic (sumf += sumi * (d_x * d_y), block by block, no fma), ggml_vec_dot_q5_K_q8_K_generic
 *     and ggml_vec_dot_q6_K_q8_K_generic (ggml/src/ggml-cpu/quants.c: eight int32 lane accumulators per super-block, sums[l] += d *
 *     aux32[l], the dmin * sum(bsums * mins) term subtracted per super-block, the 8 lanes added at the end);
 *   - float weights: activations converted to the weight type (f16 / bf16) and products accumulated in double (ggml_vec_dot_f16 /
 *     _bf16 / _f32 generic forms use ggml_float = double);
 *   - RMSNorm: sum of squares accumulated in double (ggml_compute_forward_rms_norm_f32), scale = 1/sqrtf(mean + eps);
 *   - softmax / SiLU: expf as glibc's generic routine computes it (q3_expf_ggml in the spec header: ONE definition shared with the HIP ggml-mode kernels;
 *     equal to this container's libm expf on every argument tried), softmax sum in double (ggml_vec_soft_max_f32), SiLU x / (1 + expf(-x)).
 * What stays on the spec: RoPE tables (double-precision cos/sin here; ggml iterates theta in f32), the f16 KV cache, attention's
 * score / PV summation order (llama.cpp's CPU flash-attention kernel blocks differently), residual adds.  SIMD builds of ggml (AVX2,
 * AVX-512, NEON) reorder the same sums again, so even this mode is "a llama.cpp", not "the llama.cpp" -- which is the point of
 * measuring margins rather than claiming bit parity with an absent binary.
 */
#include "q3o.h"
#include <stdlib.h>

static int g_mode = -1; /* -1 = read Q3_SPEC on first use */
int q3o_arith_mode(void) {
    if (g_mode < 0) { const char* e = getenv("Q3_SPEC"); g_mode = (e && strcmp(e, "ggml") == 0) ? 1 : 0; }
    return g_mode;
}
void q3o_set_arith_mode(int mode) { g_mode = mode ? 1 : 0; }

static inline uint16_t ld16(const void* p) { uint16_t v; memcpy(&v, p, 2); return v; }
static inline int nearest_int(float fval) { return q3_nearest_int_ggml(fval); }

/* ---- quantize_row_q8_0_ref ---- */
static void quant_q8_0_ggml(const float* x, int64_t k, int8_t* q, uint16_t* d16) {
    for (int64_t b = 0; b < k / 32; b++) {
        float amax = 0.0f;
        for (int j = 0; j < 32; j++) { float v = fabsf(x[32 * b + j]); if (v > amax) amax = v; }
        const float d = amax / ((1 << 7) - 1);
        const float id = d ? 1.0f / d : 0.0f;
        d16[b] = q3_f32_to_f16(d);
        for (int j = 0; j < 32; j++) q[32 * b + j] = (int8_t)roundf(x[32 * b + j] * id);
    }
}
/* ---- quantize_row_q8_K_ref ---- */
typedef struct { float d; int8_t qs[256]; int16_t bsums[16]; } blk_q8_K;
static void quant_q8_K_ggml(const float* x, int64_t k, blk_q8_K* y) {
    for (int64_t i = 0; i < k / 256; i++) {
        float max = 0, amax = 0;
        for (int j = 0; j < 256; j++) { float ax = fabsf(x[j]); if (ax > amax) { amax = ax; max = x[j]; } }
        if (!amax) { y[i].d = 0; memset(y[i].qs, 0, 256); memset(y[i].bsums, 0, sizeof(y[i].bsums)); x += 256; continue; }
        const float iscale = -127.f / max;
        for (int j = 0; j < 256; j++) { int v = nearest_int(iscale * x[j]); y[i].qs[j] = (int8_t)(v < 127 ? v : 127); }
        for (int j = 0; j < 16; j++) { int sum = 0; for (int ii = 0; ii < 16; ii++) sum += y[i].qs[j * 16 + ii]; y[i].bsums[j] = (int16_t)sum; }
        y[i].d = 1 / iscale;
        x += 256;
    }
}

/* ---- ggml_vec_dot_q8_0_q8_0_generic ---- */
static float dot_q8_0(const uint8_t* row, int64_t k, const int8_t* xq, const uint16_t* xd) {
    float sumf = 0;
    for (int64_t ib = 0; ib < k / 32; ib++) {
        const uint8_t* blk = row + 34 * ib;
        const int8_t* qs = (const int8_t*)(blk + 2);
        int sumi = 0;
        for (int j = 0; j < 32; j++) sumi += qs[j] * xq[32 * ib + j];
        sumf += sumi * (q3_f16_to_f32(ld16(blk)) * q3_f16_to_f32(xd[ib]));
    }
    return sumf;
}
/* ---- ggml_vec_dot_q5_K_q8_K_generic ---- */
static float dot_q5_K(const uint8_t* row, int64_t k, const blk_q8_K* y) {
    static const uint32_t kmask1 = 0x3f3f3f3f, kmask2 = 0x0f0f0f0f, kmask3 = 0x03030303;
    uint32_t utmp[4];
    const uint8_t* scales = (const uint8_t*)&utmp[0];
    const uint8_t* mins = (const uint8_t*)&utmp[2];
    int8_t aux8[256]; int16_t aux16[8]; float sums[8]; int32_t aux32[8];
    memset(sums, 0, sizeof(sums));
    float sumf = 0;
    for (int64_t i = 0; i < k / 256; i++) {
        const uint8_t* blk = row + 176 * i;
        const uint8_t* q4 = blk + 48; const uint8_t* hm = blk + 16;
        const int8_t* q8 = y[i].qs;
        memset(aux32, 0, sizeof(aux32));
        int8_t* a = aux8;
        uint8_t m = 1;
        for (int j = 0; j < 256 / 64; ++j) {
            for (int l = 0; l < 32; ++l) a[l] = (int8_t)(q4[l] & 0xF);
            for (int l = 0; l < 32; ++l) a[l] += (hm[l] & m ? 16 : 0);
            a += 32; m <<= 1;
            for (int l = 0; l < 32; ++l) a[l] = (int8_t)(q4[l] >> 4);
            for (int l = 0; l < 32; ++l) a[l] += (hm[l] & m ? 16 : 0);
            a += 32; m <<= 1;
            q4 += 32;
        }
        memcpy(utmp, blk + 4, 12);
        utmp[3] = ((utmp[2] >> 4) & kmask2) | (((utmp[1] >> 6) & kmask3) << 4);
        const uint32_t uaux = utmp[1] & kmask1;
        utmp[1] = (utmp[2] & kmask2) | (((utmp[0] >> 6) & kmask3) << 4);
        utmp[2] = uaux;
        utmp[0] &= kmask1;
        int sumi = 0;
        for (int j = 0; j < 16; ++j) sumi += y[i].bsums[j] * mins[j / 2];
        a = aux8;
        int is = 0;
        for (int j = 0; j < 256 / 32; ++j) {
            int32_t scale = scales[is++];
            for (int g = 0; g < 4; g++) {
                for (int l = 0; l < 8; ++l) aux16[l] = (int16_t)(q8[l] * a[l]);
                for (int l = 0; l < 8; ++l) aux32[l] += scale * aux16[l];
                q8 += 8; a += 8;
            }
        }
        const float d = q3_f16_to_f32(ld16(blk)) * y[i].d;
        for (int l = 0; l < 8; ++l) sums[l] += d * aux32[l];
        const float dmin = q3_f16_to_f32(ld16(blk + 2)) * y[i].d;
        sumf -= dmin * sumi;
    }
    for (int l = 0; l < 8; ++l) sumf += sums[l];
    return sumf;
}
/* ---- ggml_vec_dot_q6_K_q8_K_generic ---- */
static float dot_q6_K(const uint8_t* row, int64_t k, const blk_q8_K* y) {
    int8_t aux8[256]; int16_t aux16[8]; float sums[8]; int32_t aux32[8];
    memset(sums, 0, sizeof(sums));
    float sumf = 0;
    for (int64_t i = 0; i < k / 256; i++) {
        const uint8_t* blk = row + 210 * i;
        const uint8_t* q4 = blk; const uint8_t* qh = blk + 128;
        const int8_t* sc = (const int8_t*)(blk + 192);
        const int8_t* q8 = y[i].qs;
        memset(aux32, 0, sizeof(aux32));
        int8_t* a = aux8;
        for (int j = 0; j < 256; j += 128) {
            for (int l = 0; l < 32; ++l) {
                a[l + 0] = (int8_t)((q4[l + 0] & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32;
                a[l + 32] = (int8_t)((q4[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32;
                a[l + 64] = (int8_t)((q4[l + 0] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32;
                a[l + 96] = (int8_t)((q4[l + 32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32;
            }
            a += 128; q4 += 64; qh += 32;
        }
        a = aux8;
        int is = 0;
        for (int j = 0; j < 256 / 16; ++j) {
            int scale = sc[is++];
            for (int g = 0; g < 2; g++) {
                for (int l = 0; l < 8; ++l) aux16[l] = (int16_t)(q8[l] * a[l]);
                for (int l = 0; l < 8; ++l) aux32[l] += scale * aux16[l];
                q8 += 8; a += 8;
            }
        }
        const float d = q3_f16_to_f32(ld16(blk + 208)) * y[i].d;
        for (int l = 0; l < 8; ++l) sums[l] += d * aux32[l];
    }
    for (int l = 0; l < 8; ++l) sumf += sums[l];
    return sumf;
}

/* y[n] = W[n][k] . x  the way ggml's CPU backend would: activations converted once per call to the weight type's vec_dot_type */
void q3o_matvec_ggml(int type, const void* w, int64_t n, int64_t k, const float* xf, float* y) {
    const size_t rb = q3o_type_row_bytes(type, k);
    const uint8_t* base = (const uint8_t*)w;
    if (type == Q3_T_Q8_0) {
        int8_t* q = (int8_t*)malloc((size_t)k); uint16_t* d = (uint16_t*)malloc((size_t)(k / 32) * 2);
        quant_q8_0_ggml(xf, k, q, d);
#pragma omp parallel for schedule(static)
        for (int64_t r = 0; r < n; r++) y[r] = dot_q8_0(base + rb * (size_t)r, k, q, d);
        free(q); free(d);
    } else if (type == Q3_T_Q5_K || type == Q3_T_Q6_K) {
        blk_q8_K* q = (blk_q8_K*)malloc(sizeof(blk_q8_K) * (size_t)(k / 256));
        quant_q8_K_ggml(xf, k, q);
#pragma omp parallel for schedule(static)
        for (int64_t r = 0; r < n; r++) y[r] = type == Q3_T_Q5_K ? dot_q5_K(base + rb * (size_t)r, k, q) : dot_q6_K(base + rb * (size_t)r, k, q);
        free(q);
    } else { /* f32: double accumulation of f32 products; f16 / bf16: activations rounded to the weight type first */
        float* xr = (float*)malloc((size_t)k * 4);
        for (int64_t i = 0; i < k; i++)
            xr[i] = type == Q3_T_F16 ? q3_f16_to_f32(q3_f32_to_f16(xf[i])) : type == Q3_T_BF16 ? q3_bf16_to_f32(q3_f32_to_bf16(xf[i])) : xf[i];
#pragma omp parallel for schedule(static)
        for (int64_t r = 0; r < n; r++) {
            const uint8_t* row = base + rb * (size_t)r;
            double sumf = 0.0;
            for (int64_t i = 0; i < k; i++) {
                const float wv = type == Q3_T_F32 ? ((const float*)row)[i] : type == Q3_T_F16 ? q3_f16_to_f32(ld16(row + 2 * i)) : q3_bf16_to_f32(ld16(row + 2 * i));
                sumf += (double)(wv * xr[i]);
            }
            y[r] = (float)sumf;
        }
        free(xr);
    }
}

void q3o_rmsnorm_ggml(const float* x, const float* g, int64_t d, float eps, float* y) {
    double sum = 0.0;
    for (int64_t i = 0; i < d; i++) sum += (double)(x[i] * x[i]);
    const float mean = (float)(sum / (double)d);
    const float scale = 1.0f / sqrtf(mean + eps);
    for (int64_t i = 0; i < d; i++) y[i] = (x[i] * scale) * g[i];
}
void q3o_headnorm128_ggml(const float* x, const float* g, float eps, float* y) { q3o_rmsnorm_ggml(x, g, 128, eps, y); }
float q3o_swiglu_ggml(float gt, float up) { return (gt / (1.0f + q3_expf_ggml(-gt))) * up; }

/* one query head against n cached f16 positions: scores in order, libm expf, softmax sum in double, PV in order */
void q3o_attn_head_ggml(const float* q, const uint16_t* K, const uint16_t* V, size_t stride, int n, float* out) {
    const float scale = 0.08838834764831845f;
    float* s = (float*)malloc((size_t)n * 4);
    float mx = -INFINITY;
    for (int j = 0; j < n; j++) {
        const uint16_t* kr = K + (size_t)j * stride;
        float acc = 0.0f;
        for (int d = 0; d < 128; d++) acc += q[d] * q3_f16_to_f32(kr[d]);
        s[j] = acc * scale;
        if (s[j] > mx) mx = s[j];
    }
    double sum = 0.0;
    for (int j = 0; j < n; j++) { s[j] = q3_expf_ggml(s[j] - mx); sum += (double)s[j]; }
    const float inv = (float)(1.0 / sum);
    for (int d = 0; d < 128; d++) {
        float acc = 0.0f;
        for (int j = 0; j < n; j++) acc += s[j] * q3_f16_to_f32(V[(size_t)j * stride + d]);
        out[d] = acc * inv;
    }
    free(s);
}
