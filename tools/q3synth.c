/*
 * q3synth.c -- deterministic synthetic-model writer ("Q3TTS-1.7B-synth").
 *
 * No real Qwen3-TTS weights exist offline (the reference downloads them at run time,
 * /root/reference/src/download.rs:66-87), so benchmarks and parity tests run on seeded random weights of
 * the same architecture, written in the same container format the reference loads:
 *   <out>/<gguf|gguf_q8_0|gguf_q5_k_m|gguf_bf16>/{qwen3_assets,qwen3_tts_talker,qwen3_tts_predictor}.gguf
 *   <out>/onnx/q3tts_codec.gguf        (stand-in for qwen3_tts_decoder.onnx)
 * (directory names: /root/reference/src/tts/engine.rs:91-95,123-124).
 *
 * GGUF v3 container + ggml block encodings per the public spec [EXT].  The quantisers here are this
 * repo's own simple reference encoders (any valid encoding is a valid model).
 * Neither the oracle nor the product links this file; both only read the files it writes.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <math.h>
#include <sys/stat.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/q3tts_spec.h"

/* ---------- counter-based RNG: approx N(0,1) from one 64-bit hash (Irwin-Hall of 4x16 bit) ---------- */
static inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline float gauss(uint64_t seed, uint64_t idx) {
    uint64_t h = mix64(seed * 0xD1342543DE82EF95ULL + idx);
    int32_t s = (int32_t)(h & 0xFFFF) + (int32_t)((h >> 16) & 0xFFFF) + (int32_t)((h >> 32) & 0xFFFF) + (int32_t)(h >> 48);
    return ((float)s - 131070.0f) * (1.0f / 37837.2f); /* var of sum = 4*(65536^2/12) */
}

/* ---------- ggml block encoders ---------- */
static void enc_q8_0(const float* x, int64_t k, uint8_t* out) {
    for (int64_t b = 0; b < k / 32; b++) {
        float amax = 0;
        for (int i = 0; i < 32; i++) { float a = fabsf(x[32 * b + i]); if (a > amax) amax = a; }
        float d = amax / 127.0f, id = d ? 1.0f / d : 0.0f;
        uint16_t dh = q3_f32_to_f16(d);
        memcpy(out + 34 * b, &dh, 2);
        for (int i = 0; i < 32; i++) out[34 * b + 2 + i] = (uint8_t)(int8_t)lrintf(x[32 * b + i] * id);
    }
}
static void enc_q5_k(const float* x, int64_t k, uint8_t* out) {
    for (int64_t s = 0; s < k / 256; s++) {
        const float* xs = x + 256 * s;
        uint8_t* blk = out + 176 * s;
        memset(blk, 0, 176);
        float scale[8], mn[8], maxscale = 0, maxmin = 0;
        for (int j = 0; j < 8; j++) {
            float lo = 0, hi = 0;
            for (int l = 0; l < 32; l++) { float v = xs[32 * j + l]; if (v < lo) lo = v; if (v > hi) hi = v; }
            scale[j] = (hi - lo) / 31.0f; mn[j] = -lo;
            if (scale[j] > maxscale) maxscale = scale[j];
            if (mn[j] > maxmin) maxmin = mn[j];
        }
        uint16_t dh = q3_f32_to_f16(maxscale / 63.0f), dmh = q3_f32_to_f16(maxmin / 63.0f);
        float d = q3_f16_to_f32(dh), dmin = q3_f16_to_f32(dmh);
        int sc[8], mq[8];
        for (int j = 0; j < 8; j++) {
            sc[j] = d > 0 ? (int)lrintf(scale[j] / d) : 0; if (sc[j] > 63) sc[j] = 63;
            mq[j] = dmin > 0 ? (int)lrintf(mn[j] / dmin) : 0; if (mq[j] > 63) mq[j] = 63;
        }
        memcpy(blk, &dh, 2); memcpy(blk + 2, &dmh, 2);
        uint8_t* scales = blk + 4; uint8_t* qh = blk + 16; uint8_t* qs = blk + 48;
        for (int j = 0; j < 4; j++) {
            scales[j] = (uint8_t)(sc[j] | ((sc[j + 4] >> 4) << 6));
            scales[j + 4] = (uint8_t)(mq[j] | ((mq[j + 4] >> 4) << 6));
            scales[j + 8] = (uint8_t)((sc[j + 4] & 0xF) | ((mq[j + 4] & 0xF) << 4));
        }
        int q[256];
        for (int j = 0; j < 8; j++) {
            float de = d * (float)sc[j], me = dmin * (float)mq[j];
            for (int l = 0; l < 32; l++) {
                int v = de > 0 ? (int)lrintf((xs[32 * j + l] + me) / de) : 0;
                q[32 * j + l] = v < 0 ? 0 : (v > 31 ? 31 : v);
            }
        }
        for (int jj = 0; jj < 4; jj++)
            for (int l = 0; l < 32; l++) {
                int a = q[64 * jj + l], b = q[64 * jj + 32 + l];
                qs[32 * jj + l] = (uint8_t)((a & 0xF) | ((b & 0xF) << 4));
                qh[l] |= (uint8_t)(((a >> 4) << (2 * jj)) | ((b >> 4) << (2 * jj + 1)));
            }
    }
}
static void enc_q6_k(const float* x, int64_t k, uint8_t* out) {
    for (int64_t s = 0; s < k / 256; s++) {
        const float* xs = x + 256 * s;
        uint8_t* blk = out + 210 * s;
        memset(blk, 0, 210);
        float scale[16], maxabs = 0;
        for (int j = 0; j < 16; j++) {
            float amax = 0;
            for (int l = 0; l < 16; l++) { float a = fabsf(xs[16 * j + l]); if (a > amax) amax = a; }
            scale[j] = amax / 31.0f;
            if (scale[j] > maxabs) maxabs = scale[j];
        }
        uint16_t dh = q3_f32_to_f16(maxabs / 127.0f);
        float d = q3_f16_to_f32(dh);
        int8_t* sc = (int8_t*)(blk + 192);
        memcpy(blk + 208, &dh, 2);
        int q[256];
        for (int j = 0; j < 16; j++) {
            int si = d > 0 ? (int)lrintf(scale[j] / d) : 0; if (si > 127) si = 127;
            sc[j] = (int8_t)si;
            float de = d * (float)si;
            for (int l = 0; l < 16; l++) {
                int v = de > 0 ? (int)lrintf(xs[16 * j + l] / de) : 0;
                v = v < -32 ? -32 : (v > 31 ? 31 : v);
                q[16 * j + l] = v + 32;
            }
        }
        uint8_t* ql = blk; uint8_t* qh = blk + 128;
        for (int n = 0; n < 2; n++)
            for (int l = 0; l < 32; l++) {
                int q1 = q[128 * n + l], q2 = q[128 * n + l + 32], q3 = q[128 * n + l + 64], q4 = q[128 * n + l + 96];
                ql[64 * n + l] = (uint8_t)((q1 & 0xF) | ((q3 & 0xF) << 4));
                ql[64 * n + l + 32] = (uint8_t)((q2 & 0xF) | ((q4 & 0xF) << 4));
                qh[32 * n + l] = (uint8_t)((q1 >> 4) | ((q2 >> 4) << 2) | ((q3 >> 4) << 4) | ((q4 >> 4) << 6));
            }
    }
}
static size_t row_bytes(int type, int64_t k) {
    switch (type) {
        case Q3_T_F32: return (size_t)k * 4;
        case Q3_T_F16: case Q3_T_BF16: return (size_t)k * 2;
        case Q3_T_Q8_0: return (size_t)(k / 32) * 34;
        case Q3_T_Q5_K: return (size_t)(k / 256) * 176;
        case Q3_T_Q6_K: return (size_t)(k / 256) * 210;
    }
    return 0;
}
static void enc_row(int type, const float* x, int64_t k, uint8_t* out) {
    switch (type) {
        case Q3_T_F32: memcpy(out, x, (size_t)k * 4); break;
        case Q3_T_F16: for (int64_t i = 0; i < k; i++) { uint16_t h = q3_f32_to_f16(x[i]); memcpy(out + 2 * i, &h, 2); } break;
        case Q3_T_BF16: for (int64_t i = 0; i < k; i++) { uint16_t h = q3_f32_to_bf16(x[i]); memcpy(out + 2 * i, &h, 2); } break;
        case Q3_T_Q8_0: enc_q8_0(x, k, out); break;
        case Q3_T_Q5_K: enc_q5_k(x, k, out); break;
        case Q3_T_Q6_K: enc_q6_k(x, k, out); break;
    }
}

/* ---------- GGUF writer ---------- */
typedef struct { char name[128]; int n_dims; int64_t ne[4]; int type; float std, mean; uint64_t seed; uint64_t offset; size_t nbytes; } tdesc;
typedef struct { char key[128]; int type; uint64_t u; double f; char str[64]; int32_t arr[8]; int arr_n; } kvdesc;
typedef struct { tdesc* t; int nt, ct; kvdesc* kv; int nkv, ckv; uint64_t seed_base; } gw;

static void gw_init(gw* g, uint64_t seed) { memset(g, 0, sizeof(*g)); g->seed_base = seed; }
static kvdesc* gw_kv(gw* g, const char* key) {
    if (g->nkv == g->ckv) { g->ckv = g->ckv ? 2 * g->ckv : 32; g->kv = (kvdesc*)realloc(g->kv, (size_t)g->ckv * sizeof(kvdesc)); }
    kvdesc* k = &g->kv[g->nkv++];
    memset(k, 0, sizeof(*k));
    snprintf(k->key, sizeof(k->key), "%s", key);
    return k;
}
static void kv_u32(gw* g, const char* key, uint32_t v) { kvdesc* k = gw_kv(g, key); k->type = 4; k->u = v; }
static void kv_f32(gw* g, const char* key, float v) { kvdesc* k = gw_kv(g, key); k->type = 6; k->f = v; }
static void kv_str(gw* g, const char* key, const char* v) { kvdesc* k = gw_kv(g, key); k->type = 8; snprintf(k->str, sizeof(k->str), "%s", v); }
static void kv_arr_i32(gw* g, const char* key, const int32_t* v, int n) { kvdesc* k = gw_kv(g, key); k->type = 9; k->arr_n = n; memcpy(k->arr, v, (size_t)n * 4); }
/* tensor [ne1][ne0] (ne0 fastest) ~ mean + std*N(0,1) */
static void gw_tensor(gw* g, const char* name, int type, int64_t ne0, int64_t ne1, int64_t ne2, float std, float mean) {
    if (g->nt == g->ct) { g->ct = g->ct ? 2 * g->ct : 64; g->t = (tdesc*)realloc(g->t, (size_t)g->ct * sizeof(tdesc)); }
    tdesc* t = &g->t[g->nt];
    memset(t, 0, sizeof(*t));
    snprintf(t->name, sizeof(t->name), "%s", name);
    t->n_dims = ne2 > 1 ? 3 : (ne1 > 1 ? 2 : 1);
    t->ne[0] = ne0; t->ne[1] = ne1; t->ne[2] = ne2; t->ne[3] = 1;
    t->type = type; t->std = std; t->mean = mean;
    t->seed = g->seed_base + (uint64_t)g->nt; /* seed = base + tensor index */
    t->nbytes = row_bytes(type, ne0) * (size_t)(ne1 * ne2);
    g->nt++;
}
static void wr(FILE* f, const void* p, size_t n) { if (fwrite(p, 1, n, f) != n) { perror("fwrite"); exit(1); } }
static void wr_str(FILE* f, const char* s) { uint64_t n = strlen(s); wr(f, &n, 8); wr(f, s, n); }

static int gw_write(gw* g, const char* path) {
    FILE* f = fopen(path, "wb");
    if (!f) { perror(path); return -1; }
    uint32_t ver = 3;
    uint64_t nt = (uint64_t)g->nt, nkv = (uint64_t)g->nkv;
    wr(f, "GGUF", 4); wr(f, &ver, 4); wr(f, &nt, 8); wr(f, &nkv, 8);
    for (int i = 0; i < g->nkv; i++) {
        kvdesc* k = &g->kv[i];
        wr_str(f, k->key);
        uint32_t ty = (uint32_t)k->type;
        wr(f, &ty, 4);
        if (k->type == 4) { uint32_t v = (uint32_t)k->u; wr(f, &v, 4); }
        else if (k->type == 6) { float v = (float)k->f; wr(f, &v, 4); }
        else if (k->type == 8) wr_str(f, k->str);
        else if (k->type == 9) { uint32_t at = 5; uint64_t an = (uint64_t)k->arr_n; wr(f, &at, 4); wr(f, &an, 8); wr(f, k->arr, (size_t)k->arr_n * 4); }
    }
    uint64_t off = 0;
    for (int i = 0; i < g->nt; i++) {
        tdesc* t = &g->t[i];
        t->offset = off;
        off += (t->nbytes + 31) & ~(uint64_t)31;
        wr_str(f, t->name);
        uint32_t nd = (uint32_t)t->n_dims, ty = (uint32_t)t->type;
        wr(f, &nd, 4);
        for (int d = 0; d < t->n_dims; d++) { uint64_t e = (uint64_t)t->ne[d]; wr(f, &e, 8); }
        wr(f, &ty, 4); wr(f, &t->offset, 8);
    }
    long pos = ftell(f);
    static const uint8_t zeros[32] = { 0 };
    wr(f, zeros, (size_t)((32 - (pos % 32)) % 32));
    for (int i = 0; i < g->nt; i++) {
        tdesc* t = &g->t[i];
        int64_t k = t->ne[0], rows = t->ne[1] * t->ne[2];
        if (t->type == Q3_T_F32 || t->type == Q3_T_F16 || t->type == Q3_T_BF16) {
            /* unquantised: row structure is irrelevant, use long pseudo-rows (few, large parallel regions) */
            int64_t total = k * rows;
            int64_t pk = 4096;
            while (total % pk) pk >>= 1;
            k = pk; rows = total / pk;
        }
        size_t rb = row_bytes(t->type, k);
        const int64_t chunk = (int64_t)(((size_t)64 << 20) / (rb ? rb : 1)) + 1; /* ~64 MiB of output per parallel region */
        uint8_t* buf = (uint8_t*)malloc(rb * (size_t)(rows < chunk ? rows : chunk));
        for (int64_t r0 = 0; r0 < rows; r0 += chunk) {
            int64_t rn = rows - r0 < chunk ? rows - r0 : chunk;
#pragma omp parallel
            {
                float* x = (float*)malloc((size_t)k * 4);
#pragma omp for schedule(static)
                for (int64_t r = 0; r < rn; r++) {
                    uint64_t base = (uint64_t)(r0 + r) * (uint64_t)k;
                    for (int64_t i = 0; i < k; i++) x[i] = t->mean + t->std * gauss(t->seed, base + (uint64_t)i);
                    enc_row(t->type, x, k, buf + rb * (size_t)r);
                }
                free(x);
            }
            wr(f, buf, rb * (size_t)rn);
        }
        free(buf);
        size_t pad = ((t->nbytes + 31) & ~(size_t)31) - t->nbytes;
        wr(f, zeros, pad);
    }
    fclose(f);
    free(g->t); free(g->kv);
    return 0;
}

/* ---------- model descriptions ---------- */
typedef struct { int n_embd, n_layer, n_head, n_head_kv, n_ff, n_vocab; int mrope; } tfcfg;
typedef struct { int hidden, cb_dim, n_layers, n_heads, head_dim, ffn, window, dec_dim, n_up, n_dec; int up[4], rates[8]; } ccfg;

static int use_more_bits(int i, int n) { return i < n / 8 || i >= 7 * n / 8 || (i - n / 8) % 3 == 2; } /* llama.cpp Q5_K_M mix [EXT] */

static int wtype(const char* quant, const char* kind, int layer, int n_layer) {
    if (!strcmp(quant, "q8_0")) return Q3_T_Q8_0;
    if (!strcmp(quant, "bf16")) return Q3_T_BF16;
    if (!strcmp(quant, "f16")) return Q3_T_F16;
    if (!strcmp(quant, "f32")) return Q3_T_F32;
    if (!strcmp(quant, "q5_k_m")) {
        if (!strcmp(kind, "output")) return Q3_T_Q6_K;
        if ((!strcmp(kind, "attn_v") || !strcmp(kind, "ffn_down")) && use_more_bits(layer, n_layer)) return Q3_T_Q6_K;
        return Q3_T_Q5_K;
    }
    fprintf(stderr, "unknown quant %s\n", quant);
    exit(2);
}

static void write_tf(const char* path, const char* arch, const tfcfg* c, const char* quant, uint64_t seed, int with_tok_embd) {
    gw g;
    gw_init(&g, seed);
    char key[160], nm[160];
    kv_str(&g, "general.architecture", arch);
    kv_str(&g, "general.name", "Q3TTS-1.7B-synth");
    kv_u32(&g, "general.alignment", 32);
#define K(sfx) (snprintf(key, sizeof(key), "%s.%s", arch, sfx), key)
    kv_u32(&g, K("embedding_length"), (uint32_t)c->n_embd);
    kv_u32(&g, K("block_count"), (uint32_t)c->n_layer);
    kv_u32(&g, K("attention.head_count"), (uint32_t)c->n_head);
    kv_u32(&g, K("attention.head_count_kv"), (uint32_t)c->n_head_kv);
    kv_u32(&g, K("attention.key_length"), 128);
    kv_u32(&g, K("attention.value_length"), 128);
    kv_u32(&g, K("feed_forward_length"), (uint32_t)c->n_ff);
    kv_u32(&g, K("context_length"), 32768);
    kv_u32(&g, K("vocab_size"), (uint32_t)c->n_vocab);
    kv_f32(&g, K("attention.layer_norm_rms_epsilon"), 1e-6f);
    kv_f32(&g, K("rope.freq_base"), 1000000.0f);
    if (c->mrope) { int32_t sec[4] = { 24, 20, 20, 0 }; kv_arr_i32(&g, K("rope.dimension_sections"), sec, 4); }
#undef K
    const int d = c->n_embd, dq = c->n_head * 128, dkv = c->n_head_kv * 128;
    if (with_tok_embd) gw_tensor(&g, "token_embd.weight", wtype(quant, "token_embd", 0, c->n_layer), d, c->n_vocab, 1, 0.02f, 0);
    for (int l = 0; l < c->n_layer; l++) {
#define TN(s) (snprintf(nm, sizeof(nm), "blk.%d.%s.weight", l, s), nm)
        gw_tensor(&g, TN("attn_norm"), Q3_T_F32, d, 1, 1, 0.1f, 1.0f);
        gw_tensor(&g, TN("attn_q"), wtype(quant, "attn_q", l, c->n_layer), d, dq, 1, 0.02f, 0);
        gw_tensor(&g, TN("attn_k"), wtype(quant, "attn_k", l, c->n_layer), d, dkv, 1, 0.02f, 0);
        gw_tensor(&g, TN("attn_v"), wtype(quant, "attn_v", l, c->n_layer), d, dkv, 1, 0.02f, 0);
        gw_tensor(&g, TN("attn_output"), wtype(quant, "attn_output", l, c->n_layer), dq, d, 1, 0.02f, 0);
        gw_tensor(&g, TN("attn_q_norm"), Q3_T_F32, 128, 1, 1, 0.1f, 1.0f);
        gw_tensor(&g, TN("attn_k_norm"), Q3_T_F32, 128, 1, 1, 0.1f, 1.0f);
        gw_tensor(&g, TN("ffn_norm"), Q3_T_F32, d, 1, 1, 0.1f, 1.0f);
        gw_tensor(&g, TN("ffn_gate"), wtype(quant, "ffn_gate", l, c->n_layer), d, c->n_ff, 1, 0.02f, 0);
        gw_tensor(&g, TN("ffn_up"), wtype(quant, "ffn_up", l, c->n_layer), d, c->n_ff, 1, 0.02f, 0);
        gw_tensor(&g, TN("ffn_down"), wtype(quant, "ffn_down", l, c->n_layer), c->n_ff, d, 1, 0.02f, 0);
#undef TN
    }
    gw_tensor(&g, "output_norm.weight", Q3_T_F32, d, 1, 1, 0.1f, 1.0f);
    gw_tensor(&g, "output.weight", wtype(quant, "output", 0, c->n_layer), d, c->n_vocab, 1, 0.02f, 0);
    if (gw_write(&g, path)) exit(1);
}

static void write_assets(const char* path, int proj_out, int text_rows, uint64_t seed) {
    gw g;
    gw_init(&g, seed);
    kv_str(&g, "general.architecture", "qwen3-tts-assets"); /* no array KVs: assets_manager.rs:93-97 rejects them */
    gw_tensor(&g, "proj.weight", Q3_T_F32, 2048, proj_out, 1, 0.02f, 0);
    gw_tensor(&g, "proj.bias", Q3_T_F32, proj_out, 1, 1, 0.02f, 0);
    gw_tensor(&g, "text_embd", Q3_T_F32, 2048, text_rows, 1, 0.05f, 0);
    for (int q = 0; q < 16; q++) {
        char nm[32];
        snprintf(nm, sizeof(nm), "codec_embd.%d", q);
        gw_tensor(&g, nm, Q3_T_F32, 2048, q == 0 ? 3072 : 2048, 1, 0.05f, 0);
    }
    if (gw_write(&g, path)) exit(1);
}

static void write_codec(const char* path, const ccfg* c, uint64_t seed) {
    gw g;
    gw_init(&g, seed);
    char nm[160];
    kv_str(&g, "general.architecture", "q3tts-codec-synth");
    kv_u32(&g, "codec.n_codebooks", 16); kv_u32(&g, "codec.codebook_size", 2048);
    kv_u32(&g, "codec.codebook_dim", (uint32_t)c->cb_dim); kv_u32(&g, "codec.hidden", (uint32_t)c->hidden);
    kv_u32(&g, "codec.n_layers", (uint32_t)c->n_layers); kv_u32(&g, "codec.n_heads", (uint32_t)c->n_heads);
    kv_u32(&g, "codec.head_dim", (uint32_t)c->head_dim); kv_u32(&g, "codec.ffn", (uint32_t)c->ffn);
    kv_u32(&g, "codec.window", (uint32_t)c->window); kv_u32(&g, "codec.dec_dim", (uint32_t)c->dec_dim);
    kv_u32(&g, "codec.n_up", (uint32_t)c->n_up); kv_u32(&g, "codec.n_dec", (uint32_t)c->n_dec);
    kv_f32(&g, "codec.rope_base", 10000.0f); kv_f32(&g, "codec.eps", 1e-5f);
    for (int i = 0; i < c->n_up; i++) { snprintf(nm, sizeof(nm), "codec.up_ratio.%d", i); kv_u32(&g, nm, (uint32_t)c->up[i]); }
    for (int i = 0; i < c->n_dec; i++) { snprintf(nm, sizeof(nm), "codec.dec_rate.%d", i); kv_u32(&g, nm, (uint32_t)c->rates[i]); }
    const int H = c->hidden, A = c->n_heads * c->head_dim;
#define TT(std_, mean_, ne0, ne1, ne2, ...) (snprintf(nm, sizeof(nm), __VA_ARGS__), gw_tensor(&g, nm, Q3_T_F32, ne0, ne1, ne2, std_, mean_))
    for (int q = 0; q < 16; q++) TT(0.25f, 0, c->cb_dim, 2048, 1, "codec.codebook.%d", q);
    TT(1.0f / sqrtf(3.0f * c->cb_dim), 0, 3, c->cb_dim, H, "codec.pre_conv.weight");
    TT(0.02f, 0, H, 1, 1, "codec.pre_conv.bias");
    for (int l = 0; l < c->n_layers; l++) {
        TT(0.1f, 1.0f, H, 1, 1, "codec.tf.%d.attn_norm", l);
        TT(1.0f / sqrtf((float)H), 0, H, A, 1, "codec.tf.%d.wq", l);
        TT(1.0f / sqrtf((float)H), 0, H, A, 1, "codec.tf.%d.wk", l);
        TT(1.0f / sqrtf((float)H), 0, H, A, 1, "codec.tf.%d.wv", l);
        TT(1.0f / sqrtf((float)A), 0, A, H, 1, "codec.tf.%d.wo", l);
        TT(0.02f, 0.1f, H, 1, 1, "codec.tf.%d.ls_attn", l);
        TT(0.1f, 1.0f, H, 1, 1, "codec.tf.%d.ffn_norm", l);
        TT(1.0f / sqrtf((float)H), 0, H, c->ffn, 1, "codec.tf.%d.w_gate", l);
        TT(1.0f / sqrtf((float)H), 0, H, c->ffn, 1, "codec.tf.%d.w_up", l);
        TT(1.0f / sqrtf((float)c->ffn), 0, c->ffn, H, 1, "codec.tf.%d.w_down", l);
        TT(0.02f, 0.1f, H, 1, 1, "codec.tf.%d.ls_ffn", l);
    }
    TT(0.1f, 1.0f, H, 1, 1, "codec.tf.norm");
    for (int i = 0; i < c->n_up; i++) {
        int f = c->up[i];
        TT(1.0f / sqrtf((float)H), 0, f, H, H, "codec.up.%d.convt.weight", i); /* [cin][cout][k] */
        TT(0.02f, 0, H, 1, 1, "codec.up.%d.convt.bias", i);
        TT(0.4f, 0, 7, H, 1, "codec.up.%d.dw.weight", i);                       /* [C][7] */
        TT(0.02f, 0, H, 1, 1, "codec.up.%d.dw.bias", i);
        TT(0.1f, 1.0f, H, 1, 1, "codec.up.%d.ln.weight", i);
        TT(0.02f, 0, H, 1, 1, "codec.up.%d.ln.bias", i);
        TT(1.0f / sqrtf((float)H), 0, H, 4 * H, 1, "codec.up.%d.pw1.weight", i);
        TT(0.02f, 0, 4 * H, 1, 1, "codec.up.%d.pw1.bias", i);
        TT(1.0f / sqrtf(4.0f * H), 0, 4 * H, H, 1, "codec.up.%d.pw2.weight", i);
        TT(0.02f, 0, H, 1, 1, "codec.up.%d.pw2.bias", i);
        TT(0.02f, 0.2f, H, 1, 1, "codec.up.%d.gamma", i);
    }
    TT(1.0f / sqrtf(7.0f * H), 0, 7, H, c->dec_dim, "codec.dec.conv_in.weight");
    TT(0.02f, 0, c->dec_dim, 1, 1, "codec.dec.conv_in.bias");
    int ch = c->dec_dim;
    for (int b = 0; b < c->n_dec; b++) {
        int r = c->rates[b], co = ch / 2;
        TT(0.1f, 0, ch, 1, 1, "codec.dec.%d.snake.alpha", b);
        TT(0.1f, 0, ch, 1, 1, "codec.dec.%d.snake.beta", b);
        TT(1.0f / sqrtf(2.0f * ch), 0, 2 * r, co, ch, "codec.dec.%d.convt.weight", b); /* [cin][cout][k] */
        TT(0.02f, 0, co, 1, 1, "codec.dec.%d.convt.bias", b);
        for (int u = 0; u < 3; u++) {
            TT(0.1f, 0, co, 1, 1, "codec.dec.%d.ru.%d.snake1.alpha", b, u);
            TT(0.1f, 0, co, 1, 1, "codec.dec.%d.ru.%d.snake1.beta", b, u);
            TT(0.5f / sqrtf(7.0f * co), 0, 7, co, co, "codec.dec.%d.ru.%d.conv1.weight", b, u);
            TT(0.02f, 0, co, 1, 1, "codec.dec.%d.ru.%d.conv1.bias", b, u);
            TT(0.1f, 0, co, 1, 1, "codec.dec.%d.ru.%d.snake2.alpha", b, u);
            TT(0.1f, 0, co, 1, 1, "codec.dec.%d.ru.%d.snake2.beta", b, u);
            TT(0.5f / sqrtf((float)co), 0, 1, co, co, "codec.dec.%d.ru.%d.conv2.weight", b, u);
            TT(0.02f, 0, co, 1, 1, "codec.dec.%d.ru.%d.conv2.bias", b, u);
        }
        ch = co;
    }
    TT(0.1f, 0, ch, 1, 1, "codec.dec.snake_out.alpha");
    TT(0.1f, 0, ch, 1, 1, "codec.dec.snake_out.beta");
    TT(0.1f / sqrtf(7.0f * ch), 0, 7, ch, 1, "codec.dec.conv_out.weight");
    TT(0.01f, 0, 1, 1, 1, "codec.dec.conv_out.bias");
#undef TT
    if (gw_write(&g, path)) exit(1);
}

int main(int argc, char** argv) {
    const char* out = NULL; const char* preset = "full"; const char* quant = "q8_0";
    uint64_t seed = 1234; int text_rows = 4096; int what = 7; /* bit0 AR models, bit1 assets, bit2 codec */
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--out") && i + 1 < argc) out = argv[++i];
        else if (!strcmp(argv[i], "--preset") && i + 1 < argc) preset = argv[++i];
        else if (!strcmp(argv[i], "--quant") && i + 1 < argc) quant = argv[++i];
        else if (!strcmp(argv[i], "--seed") && i + 1 < argc) seed = strtoull(argv[++i], NULL, 10);
        else if (!strcmp(argv[i], "--text-rows") && i + 1 < argc) text_rows = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--what") && i + 1 < argc) what = atoi(argv[++i]);
        else { fprintf(stderr, "usage: q3synth --out DIR [--preset full|tiny] [--quant q8_0|q5_k_m|bf16|f16|f32] [--seed N] [--text-rows N] [--what mask]\n"); return 2; }
    }
    if (!out) { fprintf(stderr, "--out required\n"); return 2; }
#ifdef _OPENMP
    if (!getenv("OMP_NUM_THREADS")) { int n = omp_get_num_procs(); omp_set_num_threads(n > 8 ? 8 : n); } /* cgroup CPU shares are smaller than the core count */
#endif
    tfcfg talker, pred; ccfg codec;
    if (!strcmp(preset, "full")) { /* SURVEY 8d "Q3TTS-1.7B-synth" [EXT dims] */
        talker = (tfcfg){ 2048, 28, 16, 8, 6144, 3072, 1 };
        pred = (tfcfg){ 1024, 5, 16, 8, 3072, 30720, 0 };
        codec = (ccfg){ 1024, 512, 8, 16, 64, 3072, 72, 1536, 2, 4, { 2, 2 }, { 8, 5, 4, 3 } };
    } else if (!strcmp(preset, "tiny")) {
        talker = (tfcfg){ 2048, 2, 2, 1, 512, 3072, 1 };
        pred = (tfcfg){ 256, 2, 2, 1, 512, 30720, 0 };
        codec = (ccfg){ 64, 32, 2, 2, 32, 128, 8, 256, 2, 4, { 2, 2 }, { 8, 5, 4, 3 } };
    } else { fprintf(stderr, "unknown preset\n"); return 2; }
    const char* qdir = !strcmp(quant, "q8_0") ? "gguf_q8_0" : !strcmp(quant, "q5_k_m") ? "gguf_q5_k_m" : !strcmp(quant, "f32") ? "gguf" : !strcmp(quant, "bf16") ? "gguf_bf16" : "gguf_f16";
    char p[1024];
    mkdir(out, 0755);
    snprintf(p, sizeof(p), "%s/%s", out, qdir); mkdir(p, 0755);
    if (what & 1) {
        snprintf(p, sizeof(p), "%s/%s/qwen3_tts_talker.gguf", out, qdir);
        write_tf(p, "qwen3-tts-talker", &talker, quant, seed, 1);
        snprintf(p, sizeof(p), "%s/%s/qwen3_tts_predictor.gguf", out, qdir);
        write_tf(p, "qwen3-tts-predictor", &pred, quant, seed + 1000, 0);
    }
    if (what & 2) {
        snprintf(p, sizeof(p), "%s/%s/qwen3_assets.gguf", out, qdir);
        write_assets(p, pred.n_embd, text_rows, seed + 2000);
    }
    if (what & 4) {
        snprintf(p, sizeof(p), "%s/onnx", out); mkdir(p, 0755);
        snprintf(p, sizeof(p), "%s/onnx/q3tts_codec.gguf", out);
        write_codec(p, &codec, seed + 3000);
    }
    return 0;
}
