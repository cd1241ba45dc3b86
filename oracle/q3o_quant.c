/*
 * q3o_quant.c -- ORACLE (test infrastructure): ggml block formats + spec S2/S3/S4 arithmetic.
 *
 * Block formats restated from the public ggml spec [EXT] (llama.cpp b8123 is pinned only by URL at
 * /root/reference/src/download.rs:207-221): Q8_0 {f16 d; i8 qs[32]}, Q5_K {f16 d, dmin; u8 scales[12];
 * u8 qh[32]; u8 qs[128]}, Q6_K {u8 ql[128]; u8 qh[64]; i8 scales[16]; f16 d}.
 * The dot-product arithmetic is include/q3tts_spec.h's (S3): exact integer block dots, one f32 fma per
 * block, chained over the 8 blocks of a 256-element segment, segments added in order inside a
 * 2048-element super-segment, super-segments added in order.
 */
#include "q3o.h"
#include <stdlib.h>

static inline uint16_t ld16(const void* p) { uint16_t v; memcpy(&v, p, 2); return v; }

static inline void q5k_scale_min(int j, const uint8_t* q, int* sc, int* m) {
    if (j < 4) { *sc = q[j] & 63; *m = q[j + 4] & 63; }
    else { *sc = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4); *m = (q[j + 4] >> 4) | ((q[j] >> 6) << 4); }
}
/* quant value (0..31) of element l of sub-block j of a Q5_K super-block */
static inline int q5k_q(const uint8_t* qs, const uint8_t* qh, int j, int l) {
    int jj = j >> 1, hi = j & 1;
    int nib = hi ? (qs[32 * jj + l] >> 4) : (qs[32 * jj + l] & 0xF);
    int hb = (qh[l] >> (2 * jj + hi)) & 1;
    return nib + 16 * hb;
}
/* quant value (-32..31) of element l (0..31) of 32-block j (0..7) of a Q6_K super-block */
static inline int q6k_q(const uint8_t* ql, const uint8_t* qh, int j, int l) {
    int half = j >> 2, grp = j & 3;
    const uint8_t* L = ql + 64 * half;
    const uint8_t* H = qh + 32 * half;
    int q;
    switch (grp) {
        case 0: q = (L[l] & 0xF) | (((H[l] >> 0) & 3) << 4); break;
        case 1: q = (L[l + 32] & 0xF) | (((H[l] >> 2) & 3) << 4); break;
        case 2: q = (L[l] >> 4) | (((H[l] >> 4) & 3) << 4); break;
        default: q = (L[l + 32] >> 4) | (((H[l] >> 6) & 3) << 4); break;
    }
    return q - 32;
}

void q3o_dequant_row(int type, const void* row, int64_t k, float* out) {
    const uint8_t* p = (const uint8_t*)row;
    switch (type) {
        case Q3_T_F32: memcpy(out, row, (size_t)k * 4); break;
        case Q3_T_F16: for (int64_t i = 0; i < k; i++) out[i] = q3_f16_to_f32(ld16(p + 2 * i)); break;
        case Q3_T_BF16: for (int64_t i = 0; i < k; i++) out[i] = q3_bf16_to_f32(ld16(p + 2 * i)); break;
        case Q3_T_Q8_0:
            for (int64_t b = 0; b < k / 32; b++) {
                float d = q3_f16_to_f32(ld16(p + 34 * b));
                const int8_t* qs = (const int8_t*)(p + 34 * b + 2);
                for (int i = 0; i < 32; i++) out[32 * b + i] = d * (float)qs[i];
            }
            break;
        case Q3_T_Q5_K:
            for (int64_t s = 0; s < k / 256; s++) {
                const uint8_t* blk = p + 176 * s;
                float d = q3_f16_to_f32(ld16(blk)), dmin = q3_f16_to_f32(ld16(blk + 2));
                const uint8_t *scales = blk + 4, *qh = blk + 16, *qs = blk + 48;
                for (int j = 0; j < 8; j++) {
                    int sc, m;
                    q5k_scale_min(j, scales, &sc, &m);
                    float d1 = d * (float)sc, m1 = dmin * (float)m;
                    for (int l = 0; l < 32; l++) out[256 * s + 32 * j + l] = d1 * (float)q5k_q(qs, qh, j, l) - m1;
                }
            }
            break;
        case Q3_T_Q6_K:
            for (int64_t s = 0; s < k / 256; s++) {
                const uint8_t* blk = p + 210 * s;
                const uint8_t *ql = blk, *qh = blk + 128;
                const int8_t* sc = (const int8_t*)(blk + 192);
                float d = q3_f16_to_f32(ld16(blk + 208));
                for (int j = 0; j < 8; j++)
                    for (int l = 0; l < 32; l++)
                        out[256 * s + 32 * j + l] = d * (float)sc[2 * j + l / 16] * (float)q6k_q(ql, qh, j, l);
            }
            break;
        default: memset(out, 0, (size_t)k * 4);
    }
}

void q3o_quant_act(const float* x, int64_t k, int8_t* q, uint16_t* d) {
    for (int64_t b = 0; b < k / 32; b++) d[b] = q3_quant_block32(x + 32 * b, q + 32 * b);
}

/* ---- one 256-element segment -> its f32 chain value (spec S3) ---- */
static float seg_q8_0(const uint8_t* seg, const int8_t* xq, const uint16_t* xd) {
    float acc = 0.0f;
    for (int j = 0; j < 8; j++) {
        const uint8_t* blk = seg + 34 * j;
        const int8_t* qs = (const int8_t*)(blk + 2);
        int32_t isum = 0;
        for (int i = 0; i < 32; i++) isum += (int32_t)qs[i] * (int32_t)xq[32 * j + i];
        float sc = q3_f16_to_f32(ld16(blk)) * q3_f16_to_f32(xd[j]);
        acc = q3_fmaf((float)isum, sc, acc);
    }
    return acc;
}
static float seg_q5_k(const uint8_t* blk, const int8_t* xq, const uint16_t* xd) {
    float d = q3_f16_to_f32(ld16(blk)), dmin = q3_f16_to_f32(ld16(blk + 2));
    const uint8_t *scales = blk + 4, *qh = blk + 16, *qs = blk + 48;
    float acc = 0.0f;
    for (int j = 0; j < 8; j++) {
        int sc, m;
        q5k_scale_min(j, scales, &sc, &m);
        int32_t s1 = 0, s2 = 0;
        for (int l = 0; l < 32; l++) {
            int x = xq[32 * j + l];
            s1 += q5k_q(qs, qh, j, l) * x;
            s2 += x;
        }
        int32_t i1 = sc * s1, i2 = m * s2;
        float a = d * (float)i1;
        float a2 = dmin * (float)i2;
        float diff = a - a2;
        acc = q3_fmaf(diff, q3_f16_to_f32(xd[j]), acc);
    }
    return acc;
}
static float seg_q6_k(const uint8_t* blk, const int8_t* xq, const uint16_t* xd) {
    const uint8_t *ql = blk, *qh = blk + 128;
    const int8_t* sc = (const int8_t*)(blk + 192);
    float d = q3_f16_to_f32(ld16(blk + 208));
    float acc = 0.0f;
    for (int j = 0; j < 8; j++) {
        int32_t s1 = 0, s2 = 0;
        for (int l = 0; l < 16; l++) s1 += q6k_q(ql, qh, j, l) * (int)xq[32 * j + l];
        for (int l = 16; l < 32; l++) s2 += q6k_q(ql, qh, j, l) * (int)xq[32 * j + l];
        int32_t i = (int32_t)sc[2 * j] * s1 + (int32_t)sc[2 * j + 1] * s2;
        float a = d * (float)i;
        acc = q3_fmaf(a, q3_f16_to_f32(xd[j]), acc);
    }
    return acc;
}
/* 16/32-bit float weights: activations stay f32; block of 32 = four 8-element fma chains combined
 * (c0+c1)+(c2+c3); blocks added in order inside the segment. */
static float seg_float(int type, const uint8_t* seg, const float* xf) {
    float acc = 0.0f;
    for (int j = 0; j < 8; j++) {
        float c[4];
        for (int u = 0; u < 4; u++) {
            float a = 0.0f;
            for (int i = 0; i < 8; i++) {
                int e = 32 * j + 8 * u + i;
                float w = type == Q3_T_F32 ? ((const float*)seg)[e]
                        : type == Q3_T_F16 ? q3_f16_to_f32(ld16(seg + 2 * e)) : q3_bf16_to_f32(ld16(seg + 2 * e));
                a = q3_fmaf(w, xf[e], a);
            }
            c[u] = a;
        }
        float bt = (c[0] + c[1]) + (c[2] + c[3]);
        acc = acc + bt;
    }
    return acc;
}

static float row_dot(int type, const uint8_t* row, int64_t k, const int8_t* xq, const uint16_t* xd, const float* xf) {
    int64_t nseg = k / Q3_SEG;
    size_t seg_bytes = q3o_type_row_bytes(type, Q3_SEG);
    float y = 0.0f;
    for (int64_t ss = 0; ss < nseg; ss += Q3_SSEG_SEGS) {
        float S = 0.0f;
        int64_t end = ss + Q3_SSEG_SEGS < nseg ? ss + Q3_SSEG_SEGS : nseg;
        for (int64_t s = ss; s < end; s++) {
            const uint8_t* seg = row + seg_bytes * (size_t)s;
            float v;
            switch (type) {
                case Q3_T_Q8_0: v = seg_q8_0(seg, xq + 256 * s, xd + 8 * s); break;
                case Q3_T_Q5_K: v = seg_q5_k(seg, xq + 256 * s, xd + 8 * s); break;
                case Q3_T_Q6_K: v = seg_q6_k(seg, xq + 256 * s, xd + 8 * s); break;
                default: v = seg_float(type, seg, xf + 256 * s); break;
            }
            S = (s == ss) ? v : S + v;
        }
        y = (ss == 0) ? S : y + S;
    }
    return y;
}

void q3o_matvec(int type, const void* w, int64_t n, int64_t k, const int8_t* xq, const uint16_t* xd,
                const float* xf, float* y) {
    size_t rb = q3o_type_row_bytes(type, k);
    const uint8_t* base = (const uint8_t*)w;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; r++) y[r] = row_dot(type, base + rb * (size_t)r, k, xq, xd, xf);
}

/* spec S4: 64 lane partials (lane l owns elements 256c+4l..+3 of every 256-chunk c, fma chain in
 * index order), then xor-butterfly 32,16,8,4,2,1 */
float q3o_sumsq_vec(const float* x, int64_t d) {
    float p[64], t[64];
    for (int l = 0; l < 64; l++) {
        float a = 0.0f;
        for (int64_t c = 0; c < d / 256; c++)
            for (int i = 0; i < 4; i++) { float v = x[256 * c + 4 * l + i]; a = q3_fmaf(v, v, a); }
        p[l] = a;
    }
    for (int s = 32; s >= 1; s >>= 1) {
        for (int l = 0; l < 64; l++) t[l] = p[l] + p[l ^ s];
        memcpy(p, t, sizeof(p));
    }
    return p[0];
}

void q3o_rmsnorm(const float* x, const float* g, int64_t d, float eps, float* y) {
    float ss = q3o_sumsq_vec(x, d);
    float mean = ss / (float)d;
    float scale = 1.0f / q3_sqrtf(mean + eps);
    for (int64_t i = 0; i < d; i++) y[i] = (x[i] * scale) * g[i];
}

/* per-head RMSNorm over 128 dims: lane l owns (x[l], x[l+64]) -- the NeoX RoPE pair */
void q3o_headnorm128(const float* x, const float* g, float eps, float* y) {
    float p[64], t[64];
    for (int l = 0; l < 64; l++) {
        float a = x[l] * x[l];
        p[l] = q3_fmaf(x[l + 64], x[l + 64], a);
    }
    for (int s = 32; s >= 1; s >>= 1) {
        for (int l = 0; l < 64; l++) t[l] = p[l] + p[l ^ s];
        memcpy(p, t, sizeof(p));
    }
    float mean = p[0] / 128.0f;
    float scale = 1.0f / q3_sqrtf(mean + eps);
    for (int i = 0; i < 128; i++) y[i] = (x[i] * scale) * g[i];
}
