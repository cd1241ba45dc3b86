/*
 * q3o_model.c -- ORACLE (test infrastructure): Qwen3-style decoder-only transformer on an EMBEDDING
 * input, i.e. what the reference obtains from llama_decode(batch.embd) + llama_get_logits /
 * llama_get_embeddings (/root/reference/src/models/llama/mod.rs:442-477; call sites
 * src/tts/engine.rs:460-462,580-582,607-609,637-639).
 *
 * Block structure [EXT, llama.cpp qwen3 graph == transformers Qwen3DecoderLayer]: RMSNorm -> q/k/v ->
 * per-head q/k RMSNorm -> NeoX (M-)RoPE -> causal GQA attention over an f16 KV cache -> o-proj ->
 * residual; RMSNorm -> SwiGLU FFN -> residual; final RMSNorm ("embeddings") -> output matrix (logits).
 * Arithmetic: include/q3tts_spec.h (S2-S9).
 */
#include "q3o.h"
#include <stdio.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int kv_int(const q3o_gguf* g, const char* arch, const char* suffix, int def) {
    char key[192];
    snprintf(key, sizeof(key), "%s.%s", arch, suffix);
    const q3o_gguf_kv* kv = q3o_gguf_kv_find(g, key);
    if (!kv) return def;
    switch (kv->type) {
        case 0: case 2: case 4: case 10: case 7: return (int)kv->v.u;
        case 1: case 3: case 5: case 11: return (int)kv->v.i;
        case 6: case 12: return (int)kv->v.f;
        default: return def;
    }
}
static float kv_float(const q3o_gguf* g, const char* arch, const char* suffix, float def) {
    char key[192];
    snprintf(key, sizeof(key), "%s.%s", arch, suffix);
    const q3o_gguf_kv* kv = q3o_gguf_kv_find(g, key);
    if (!kv) return def;
    if (kv->type == 6 || kv->type == 12) return (float)kv->v.f;
    return def;
}

static const q3o_gguf_tensor* need(q3o_gguf* g, const char* name, char* err, size_t errlen) {
    const q3o_gguf_tensor* t = q3o_gguf_find(g, name);
    if (!t && err[0] == 0) snprintf(err, errlen, "missing tensor %s", name);
    return t;
}

q3o_model* q3o_model_load(const char* path, int n_ctx, char* err, size_t errlen) {
    err[0] = 0;
    q3o_gguf* g = q3o_gguf_open(path, err, errlen);
    if (!g) return NULL;
    q3o_model* m = (q3o_model*)calloc(1, sizeof(*m));
    m->g = g;
    const q3o_gguf_kv* a = q3o_gguf_kv_find(g, "general.architecture");
    snprintf(m->arch, sizeof(m->arch), "%s", (a && a->str) ? a->str : "qwen3");
    m->n_embd = kv_int(g, m->arch, "embedding_length", 0);
    m->n_layer = kv_int(g, m->arch, "block_count", 0);
    m->n_head = kv_int(g, m->arch, "attention.head_count", 0);
    m->n_head_kv = kv_int(g, m->arch, "attention.head_count_kv", m->n_head);
    m->head_dim = kv_int(g, m->arch, "attention.key_length", m->n_head ? m->n_embd / m->n_head : 0);
    m->n_ff = kv_int(g, m->arch, "feed_forward_length", 0);
    m->eps = kv_float(g, m->arch, "attention.layer_norm_rms_epsilon", 1e-6f);
    m->rope_base = kv_float(g, m->arch, "rope.freq_base", 1000000.0f);
    {
        char key[192];
        snprintf(key, sizeof(key), "%s.rope.dimension_sections", m->arch);
        const q3o_gguf_kv* s = q3o_gguf_kv_find(g, key);
        if (s && s->type == 9 && s->arr && (s->arr_type == 5 || s->arr_type == 4))
            for (uint64_t i = 0; i < s->arr_n && i < 4; i++) m->mrope_sec[i] = ((int32_t*)s->arr)[i];
    }
    m->n_ctx = n_ctx;
    if (m->n_embd <= 0 || m->n_layer <= 0 || m->n_head <= 0 || m->head_dim != Q3_HEAD_DIM ||
        m->n_embd % Q3_SEG || m->n_ff % Q3_SEG || m->n_head % m->n_head_kv) {
        snprintf(err, errlen, "unsupported hparams embd=%d layer=%d head=%d/%d hd=%d ff=%d", m->n_embd,
                 m->n_layer, m->n_head, m->n_head_kv, m->head_dim, m->n_ff);
        q3o_model_free(m);
        return NULL;
    }
    m->layers = (q3o_layer*)calloc((size_t)m->n_layer, sizeof(q3o_layer));
    char nm[160];
    for (int l = 0; l < m->n_layer; l++) {
        q3o_layer* L = &m->layers[l];
#define T(field, suffix) snprintf(nm, sizeof(nm), "blk.%d." suffix ".weight", l); L->field = need(g, nm, err, errlen)
        T(attn_norm, "attn_norm"); T(wq, "attn_q"); T(wk, "attn_k"); T(wv, "attn_v"); T(wo, "attn_output");
        T(q_norm, "attn_q_norm"); T(k_norm, "attn_k_norm"); T(ffn_norm, "ffn_norm");
        T(w_gate, "ffn_gate"); T(w_up, "ffn_up"); T(w_down, "ffn_down");
#undef T
    }
    m->output_norm = need(g, "output_norm.weight", err, errlen);
    m->output = need(g, "output.weight", err, errlen);
    if (err[0]) { q3o_model_free(m); return NULL; }
    m->n_vocab = (int)m->output->ne[1];
    size_t kvn = (size_t)m->n_layer * (size_t)n_ctx * (size_t)m->n_head_kv * Q3_HEAD_DIM;
    m->kcache = (uint16_t*)calloc(kvn, 2);
    m->vcache = (uint16_t*)calloc(kvn, 2);
    m->rope_cos = (float*)malloc((size_t)n_ctx * 64 * 4);
    m->rope_sin = (float*)malloc((size_t)n_ctx * 64 * 4);
    for (int p = 0; p < n_ctx; p++)
        for (int i = 0; i < 64; i++) {
            double inv = pow((double)m->rope_base, -(double)i / 64.0);
            double ang = (double)p * inv;
            m->rope_cos[p * 64 + i] = (float)cos(ang);
            m->rope_sin[p * 64 + i] = (float)sin(ang);
        }
    m->n_threads = 0;
    return m;
}

void q3o_model_free(q3o_model* m) {
    if (!m) return;
    free(m->layers); free(m->kcache); free(m->vcache); free(m->rope_cos); free(m->rope_sin);
    q3o_gguf_close(m->g);
    free(m);
}

void q3o_model_clear_kv(q3o_model* m) { m->n_past = 0; } /* llama_memory_seq_rm(mem,-1,0,-1), llama/mod.rs:482 */

static void mv(const q3o_gguf_tensor* w, const int8_t* xq, const uint16_t* xd, const float* xf, float* y) {
    if (q3o_arith_mode()) { q3o_matvec_ggml(w->type, w->data, w->ne[1], w->ne[0], xf, y); return; } /* Q3_SPEC=ggml: q3o_ggml.c */
    q3o_matvec(w->type, w->data, w->ne[1], w->ne[0], xq, xd, xf, y);
}
static void rmsn(const float* x, const float* g, int64_t d, float eps, float* y) { if (q3o_arith_mode()) q3o_rmsnorm_ggml(x, g, d, eps, y); else q3o_rmsnorm(x, g, d, eps, y); }
static int is_float_type(int t) { return t == Q3_T_F32 || t == Q3_T_F16 || t == Q3_T_BF16; }

/* spec S7: one query head against n cached positions. K/V: f16, element (pos,d) at base[pos*stride+d] */
static void attn_head(const float* q, const uint16_t* K, const uint16_t* V, size_t stride, int n, float* out) {
    const float scale = 0.08838834764831845f; /* 1/sqrt(128) */
    float M = 0, L = 0, O[128];
    float s[Q3_ATT_CHUNK], p[Q3_ATT_CHUNK];
    for (int c0 = 0; c0 < n; c0 += Q3_ATT_CHUNK) {
        int cn = n - c0 < Q3_ATT_CHUNK ? n - c0 : Q3_ATT_CHUNK;
        float mc = 0;
        for (int j = 0; j < cn; j++) {
            const uint16_t* kr = K + (size_t)(c0 + j) * stride;
            float acc = 0.0f;
            for (int d = 0; d < 128; d++) acc = q3_fmaf(q[d], q3_f16_to_f32(kr[d]), acc);
            s[j] = acc * scale;
            if (j == 0 || s[j] > mc) mc = s[j];
        }
        for (int j = 0; j < cn; j++) p[j] = q3_expf(s[j] - mc);
        float part[64], tmp[64];
        for (int l = 0; l < 64; l++) {
            float a = 0.0f;
            for (int t = 0; t < 4; t++) { int j = l + 64 * t; if (j < cn) a = a + p[j]; }
            part[l] = a;
        }
        for (int sft = 32; sft >= 1; sft >>= 1) {
            for (int l = 0; l < 64; l++) tmp[l] = part[l] + part[l ^ sft];
            memcpy(part, tmp, sizeof(part));
        }
        float lc = part[0];
        float oc[128];
        for (int d = 0; d < 128; d++) { /* 16 interleaved chains r = j mod 16, fixed combine tree */
            float S[16] = { 0 };
            for (int j = 0; j < cn; j++) {
                float v = q3_f16_to_f32(V[(size_t)(c0 + j) * stride + d]);
                S[j & 15] = q3_fmaf(p[j], v, S[j & 15]);
            }
            float T4[4];
            for (int w = 0; w < 4; w++) T4[w] = (S[4 * w] + S[4 * w + 1]) + (S[4 * w + 2] + S[4 * w + 3]);
            oc[d] = (T4[0] + T4[1]) + (T4[2] + T4[3]);
        }
        if (c0 == 0) {
            M = mc; L = lc; memcpy(O, oc, sizeof(O));
        } else {
            float mn = M > mc ? M : mc;
            float a = q3_expf(M - mn), b = q3_expf(mc - mn);
            float t = lc * b;
            L = q3_fmaf(L, a, t);
            for (int d = 0; d < 128; d++) { float u = oc[d] * b; O[d] = q3_fmaf(O[d], a, u); }
            M = mn;
        }
    }
    for (int d = 0; d < 128; d++) out[d] = O[d] / L;
}

int q3o_model_eval(q3o_model* m, const float* x, const int32_t pos[4], float* hidden_out,
                   float* logits_out, int row0, int row1) {
    if (m->n_past >= m->n_ctx) return -1;
#ifdef _OPENMP
    if (m->n_threads > 0) omp_set_num_threads(m->n_threads);
#endif
    const int d = m->n_embd, nh = m->n_head, nkv = m->n_head_kv, ff = m->n_ff;
    const int dq = nh * 128, dkv = nkv * 128;
    int maxd = d > ff ? d : ff;
    if (dq > maxd) maxd = dq;
    float* h = (float*)malloc((size_t)d * 4);
    float* xn = (float*)malloc((size_t)maxd * 4);
    int8_t* xq = (int8_t*)malloc((size_t)maxd);
    uint16_t* xd = (uint16_t*)malloc((size_t)(maxd / 32) * 2);
    float* q = (float*)malloc((size_t)dq * 4);
    float* k = (float*)malloc((size_t)dkv * 4);
    float* v = (float*)malloc((size_t)dkv * 4);
    float* att = (float*)malloc((size_t)dq * 4);
    float* o = (float*)malloc((size_t)d * 4);
    float* gt = (float*)malloc((size_t)ff * 4);
    float* up = (float*)malloc((size_t)ff * 4);
    memcpy(h, x, (size_t)d * 4);
    const int slot = m->n_past;
    const size_t stride = (size_t)nkv * 128;
    for (int l = 0; l < m->n_layer; l++) {
        const q3o_layer* L = &m->layers[l];
        rmsn(h, (const float*)L->attn_norm->data, d, m->eps, xn);
        q3o_quant_act(xn, d, xq, xd);
        mv(L->wq, xq, xd, xn, q); mv(L->wk, xq, xd, xn, k); mv(L->wv, xq, xd, xn, v);
        uint16_t* Kl = m->kcache + (size_t)l * m->n_ctx * stride;
        uint16_t* Vl = m->vcache + (size_t)l * m->n_ctx * stride;
        float tmp[128];
        for (int hh = 0; hh < nh + nkv; hh++) {
            float* vec = hh < nh ? q + 128 * hh : k + 128 * (hh - nh);
            if (q3o_arith_mode()) q3o_headnorm128_ggml(vec, (const float*)(hh < nh ? L->q_norm->data : L->k_norm->data), m->eps, tmp);
            else q3o_headnorm128(vec, (const float*)(hh < nh ? L->q_norm->data : L->k_norm->data), m->eps, tmp);
            for (int i = 0; i < 64; i++) {
                int32_t pp = pos[q3_mrope_stream(i, m->mrope_sec)];
                if (pp < 0) pp = 0;
                if (pp >= m->n_ctx) pp = m->n_ctx - 1;
                q3_rope_pair(tmp[i], tmp[i + 64], m->rope_cos[pp * 64 + i], m->rope_sin[pp * 64 + i], &vec[i], &vec[i + 64]);
            }
        }
        for (int i = 0; i < dkv; i++) {
            Kl[(size_t)slot * stride + i] = q3_f32_to_f16(k[i]);
            Vl[(size_t)slot * stride + i] = q3_f32_to_f16(v[i]);
        }
        const int grp = nh / nkv;
#pragma omp parallel for schedule(static)
        for (int hh = 0; hh < nh; hh++) {
            int kvh = hh / grp;
            if (q3o_arith_mode()) q3o_attn_head_ggml(q + 128 * hh, Kl + 128 * kvh, Vl + 128 * kvh, stride, slot + 1, att + 128 * hh);
            else attn_head(q + 128 * hh, Kl + 128 * kvh, Vl + 128 * kvh, stride, slot + 1, att + 128 * hh);
        }
        q3o_quant_act(att, dq, xq, xd);
        mv(L->wo, xq, xd, att, o);
        for (int i = 0; i < d; i++) h[i] = h[i] + o[i];
        rmsn(h, (const float*)L->ffn_norm->data, d, m->eps, xn);
        q3o_quant_act(xn, d, xq, xd);
        mv(L->w_gate, xq, xd, xn, gt); mv(L->w_up, xq, xd, xn, up);
        if (q3o_arith_mode()) { for (int i = 0; i < ff; i++) gt[i] = q3o_swiglu_ggml(gt[i], up[i]); }
        else for (int i = 0; i < ff; i++) gt[i] = q3_swiglu(gt[i], up[i]);
        q3o_quant_act(gt, ff, xq, xd);
        mv(L->w_down, xq, xd, gt, o);
        for (int i = 0; i < d; i++) h[i] = h[i] + o[i];
    }
    m->n_past++;
    rmsn(h, (const float*)m->output_norm->data, d, m->eps, xn);
    if (hidden_out) memcpy(hidden_out, xn, (size_t)d * 4);
    if (logits_out && row1 > row0) {
        q3o_quant_act(xn, d, xq, xd);
        size_t rb = q3o_type_row_bytes(m->output->type, d);
        (void)is_float_type;
        if (q3o_arith_mode()) q3o_matvec_ggml(m->output->type, (const uint8_t*)m->output->data + rb * (size_t)row0, row1 - row0, d, xn, logits_out);
        else q3o_matvec(m->output->type, (const uint8_t*)m->output->data + rb * (size_t)row0, row1 - row0, d, xq, xd, xn, logits_out);
    }
    free(h); free(xn); free(xq); free(xd); free(q); free(k); free(v); free(att); free(o); free(gt); free(up);
    return 0;
}
