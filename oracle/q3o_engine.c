/*
 * q3o_engine.c -- ORACLE (test infrastructure): assets, sampler, prompt builder, chunker and the
 * generation loop, restating the SOURCE-PINNED parts of the reference:
 *   assets      /root/reference/src/assets_manager.rs:212-249 (tensors), 383-399 (project), 419-460 (gathers)
 *   sampler     /root/reference/src/models/llama/mod.rs:639-776
 *   prompt      /root/reference/src/tts/prompt.rs:28-118, 141-277
 *   chunker     /root/reference/src/tts/engine.rs:495-543
 *   loop        /root/reference/src/tts/engine.rs:445-656
 */
#include "q3o.h"
#include <stdio.h>
#include <stdlib.h>

/* ============================ assets ============================ */
q3o_assets* q3o_assets_load(const char* path, char* err, size_t errlen) {
    q3o_gguf* g = q3o_gguf_open(path, err, errlen);
    if (!g) return NULL;
    q3o_assets* a = (q3o_assets*)calloc(1, sizeof(*a));
    a->g = g;
    const q3o_gguf_tensor* pw = q3o_gguf_find(g, "proj.weight");
    const q3o_gguf_tensor* pb = q3o_gguf_find(g, "proj.bias");
    if (!pw) { snprintf(err, errlen, "proj.weight (tensor) missing"); q3o_assets_free(a); return NULL; }
    if (!pb) { snprintf(err, errlen, "proj.bias (tensor) missing"); q3o_assets_free(a); return NULL; }
    if (pw->type != Q3_T_F32 || pb->type != Q3_T_F32) { /* assets_manager.rs:163-167 */
        snprintf(err, errlen, "Unsupported tensor type (expected F32)"); q3o_assets_free(a); return NULL;
    }
    a->proj_w = (const float*)pw->data;
    a->proj_b = (const float*)pb->data;
    a->proj_out = pb->ne[0];
    a->proj_in = pw->ne[0] * pw->ne[1] / a->proj_out;
    const q3o_gguf_tensor* tt = q3o_gguf_find(g, "text_embd"); /* optional: :226-233 */
    if (tt) {
        if (tt->type != Q3_T_F32) { snprintf(err, errlen, "Unsupported tensor type (expected F32)"); q3o_assets_free(a); return NULL; }
        a->text_table = (const float*)tt->data;
        a->text_rows = (tt->ne[0] * tt->ne[1]) / 2048;
    }
    for (int i = 0; i < 16; i++) { /* :235-241: tables pushed in order, missing ones skipped */
        char nm[32];
        snprintf(nm, sizeof(nm), "codec_embd.%d", i);
        const q3o_gguf_tensor* t = q3o_gguf_find(g, nm);
        if (t) {
            if (t->type != Q3_T_F32) { snprintf(err, errlen, "Unsupported tensor type (expected F32)"); q3o_assets_free(a); return NULL; }
            a->codec[a->n_codec] = (const float*)t->data;
            a->codec_rows[a->n_codec] = (t->ne[0] * t->ne[1]) / 2048;
            a->n_codec++;
        }
    }
    /* :244-249 */
    if (a->text_rows * 2048 >= (int64_t)(151671 + 1) * 2048) memcpy(a->tts_pad, a->text_table + (size_t)151671 * 2048, 2048 * 4);
    else memset(a->tts_pad, 0, sizeof(a->tts_pad));
    return a;
}
void q3o_assets_free(q3o_assets* a) { if (!a) return; q3o_gguf_close(a->g); free(a); }

/* assets_manager.rs:383-399: sum = bias; for i ascending: sum += h[i]*W[o*n_in+i]  (mul then add, f32) */
void q3o_project(const q3o_assets* a, const float* hidden, int64_t n_in, float* out) {
#pragma omp parallel for schedule(static)
    for (int64_t o = 0; o < a->proj_out; o++) {
        float sum = a->proj_b[o];
        const float* w = a->proj_w + (size_t)o * (size_t)n_in;
        for (int64_t i = 0; i < n_in; i++) { float t = hidden[i] * w[i]; sum = sum + t; }
        out[o] = sum;
    }
}
/* assets_manager.rs:419-437 */
void q3o_codec_embedding(const q3o_assets* a, int q, int32_t code, float* out) {
    if (q >= 0 && q < a->n_codec) {
        int64_t c = code < 0 ? 0 : code;
        if ((c + 1) * 2048 <= a->codec_rows[q] * 2048) { memcpy(out, a->codec[q] + (size_t)c * 2048, 2048 * 4); return; }
    }
    memset(out, 0, 2048 * 4);
}
/* assets_manager.rs:444-460 */
void q3o_text_embedding(const q3o_assets* a, int64_t token, float* out) {
    if (token >= 0 && (token + 1) * 2048 <= a->text_rows * 2048) { memcpy(out, a->text_table + (size_t)token * 2048, 2048 * 4); return; }
    for (int i = 0; i < 2048; i++) {
        float v = (float)(uint64_t)((uint64_t)token * 17u + (uint64_t)i);
        out[i] = fmodf(v, 2.0f) - 1.0f;
    }
}

/* ============================ sampler ============================ */
static uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
static void chacha12_block(const uint32_t key[8], uint64_t counter, uint32_t out[16]) {
    uint32_t s[16] = { 0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                       key[4], key[5], key[6], key[7], (uint32_t)counter, (uint32_t)(counter >> 32), 0, 0 };
    uint32_t x[16];
    memcpy(x, s, sizeof(x));
#define QR(a, b, c, d) x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 12); \
                       x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 8);  x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 7)
    for (int r = 0; r < 6; r++) {
        QR(0, 4, 8, 12); QR(1, 5, 9, 13); QR(2, 6, 10, 14); QR(3, 7, 11, 15);
        QR(0, 5, 10, 15); QR(1, 6, 11, 12); QR(2, 7, 8, 13); QR(3, 4, 9, 14);
    }
#undef QR
    for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}
/* rand_core SeedableRng::seed_from_u64 (PCG32 expansion) + ChaCha12 (rand 0.10 StdRng) [EXT, unpinned] */
void q3o_rng_seed(q3o_rng* r, uint64_t state) {
    for (int i = 0; i < 8; i++) {
        state = state * 6364136223846793005ULL + 11634580027462260723ULL;
        uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27);
        uint32_t rot = (uint32_t)(state >> 59);
        r->key[i] = (xs >> rot) | (xs << ((32 - rot) & 31));
    }
    r->counter = 0;
    r->idx = 16;
}
uint32_t q3o_rng_next_u32(q3o_rng* r) {
    if (r->idx >= 16) { chacha12_block(r->key, r->counter++, r->buf); r->idx = 0; }
    return r->buf[r->idx++];
}
void q3o_sampler_init(q3o_sampler* s, float temperature, int top_k, float top_p, uint64_t seed) {
    s->temperature = temperature; s->top_k = top_k; s->top_p = top_p;
    q3o_rng_seed(&s->rng, seed);
}

typedef struct { int idx; float v; } cand_t;
static void stable_sort_desc(cand_t* a, int n) { /* merge sort, stable; NaN compares Equal (mod.rs:708) */
    if (n < 2) return;
    cand_t* tmp = (cand_t*)malloc((size_t)n * sizeof(cand_t));
    for (int w = 1; w < n; w *= 2) {
        for (int lo = 0; lo < n; lo += 2 * w) {
            int mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            int i = lo, j = mid, k = lo;
            while (i < mid && j < hi) { if (a[j].v > a[i].v) tmp[k++] = a[j++]; else tmp[k++] = a[i++]; }
            while (i < mid) tmp[k++] = a[i++];
            while (j < hi) tmp[k++] = a[j++];
        }
        memcpy(a, tmp, (size_t)n * sizeof(cand_t));
    }
    free(tmp);
}

int32_t q3o_sample(q3o_sampler* s, const float* logits, int n_vocab, int start, int end) {
    if (end > n_vocab) end = n_vocab;
    if (s->temperature <= 0.0f) { /* mod.rs:690-701: first max, strict > */
        float max_val = -INFINITY;
        int max_idx = start;
        for (int i = start; i < end; i++) if (logits[i] > max_val) { max_val = logits[i]; max_idx = i; }
        return max_idx;
    }
    int n = end > start ? end - start : 0;
    if (n == 0) return start;
    cand_t* c = (cand_t*)malloc((size_t)n * sizeof(cand_t));
    for (int i = 0; i < n; i++) { c[i].idx = start + i; c[i].v = logits[start + i]; }
    stable_sort_desc(c, n);
    if (s->top_k > 0 && s->top_k < n) n = s->top_k;
    float max_logit = c[0].v;
    float sum = 0.0f;
    for (int i = 0; i < n; i++) { float sc = (c[i].v - max_logit) / s->temperature; c[i].v = q3_expf(sc); sum += c[i].v; }
    if (sum > 0.0f) for (int i = 0; i < n; i++) c[i].v /= sum;
    if (s->top_p < 1.0f) {
        float cum = 0.0f;
        int cut = n;
        for (int i = 0; i < n; i++) { cum += c[i].v; if (cum >= s->top_p) { cut = i + 1; break; } }
        n = cut;
        float ns = 0.0f;
        for (int i = 0; i < n; i++) ns += c[i].v;
        if (ns > 0.0f) for (int i = 0; i < n; i++) c[i].v /= ns;
    }
    float r = (float)q3o_rng_next_u32(&s->rng) / 4294967296.0f; /* u32::MAX as f32 == 2^32 */
    float cum = 0.0f;
    int32_t res = c[0].idx;
    int found = 0;
    for (int i = 0; i < n; i++) { cum += c[i].v; if (r < cum) { res = c[i].idx; found = 1; break; } }
    if (!found) res = c[0].idx;
    free(c);
    return res;
}

/* ============================ prompt builder ============================ */
static void add2(const float* a, const float* b, float* o) { for (int i = 0; i < 2048; i++) o[i] = a[i] + b[i]; }

int q3o_build_core(const q3o_assets* a, const int32_t* text_ids, int n_text, int has_lang, int lang_id,
                   int has_spk_id, int spk_id, const float* spk_emb, const int32_t* instr_ids,
                   int n_instr, const float* mid, int n_mid, float* out, int max_rows) {
    int n = 0;
    float e[2048], marker[2048], pad0[2048], t[2048];
#define ROW() (n < max_rows ? out + (size_t)(n++) * 2048 : NULL)
#define PUSH_TEXT(id) do { float* r_ = ROW(); if (!r_) return -1; q3o_text_embedding(a, (id), r_); } while (0)
#define PUSH_MARK_CODEC(code) do { float* r_ = ROW(); if (!r_) return -1; q3o_codec_embedding(a, 0, (code), e); add2(marker, e, r_); } while (0)
    if (instr_ids) { /* prompt.rs:154-169 */
        PUSH_TEXT(151644); PUSH_TEXT(872); PUSH_TEXT(198);
        for (int i = 0; i < n_instr; i++) PUSH_TEXT(instr_ids[i]);
        PUSH_TEXT(151645); PUSH_TEXT(198);
    }
    PUSH_TEXT(151644); PUSH_TEXT(77091); PUSH_TEXT(198); /* :173-175 */
    q3o_text_embedding(a, Q3_TEXT_AUDIO_MARKER, marker);
    if (has_lang) { /* :180-191 */
        PUSH_MARK_CODEC(Q3_CODEC_THINK); PUSH_MARK_CODEC(Q3_CODEC_THINK_BOS); PUSH_MARK_CODEC(lang_id); PUSH_MARK_CODEC(Q3_CODEC_THINK_EOS);
    } else { /* :192-204 */
        PUSH_MARK_CODEC(Q3_CODEC_NOTHINK); PUSH_MARK_CODEC(Q3_CODEC_THINK_BOS); PUSH_MARK_CODEC(Q3_CODEC_THINK_EOS);
    }
    if (has_spk_id) { PUSH_MARK_CODEC(spk_id); } /* :207-214 */
    else if (spk_emb) { float* r = ROW(); if (!r) return -1; add2(marker, spk_emb, r); } /* :215-222 */
    for (int i = 0; i < n_mid; i++) { float* r = ROW(); if (!r) return -1; memcpy(r, mid + (size_t)i * 2048, 2048 * 4); } /* :225-227 */
    q3o_codec_embedding(a, 0, Q3_CODEC_PAD, pad0); /* :232 */
    { float* r = ROW(); if (!r) return -1; q3o_text_embedding(a, Q3_TEXT_BOS, t); add2(t, pad0, r); } /* :233-239 */
    for (int i = 0; i < n_text; i++) { float* r = ROW(); if (!r) return -1; q3o_text_embedding(a, text_ids[i], t); add2(t, pad0, r); } /* :241-245 */
    { float* r = ROW(); if (!r) return -1; q3o_text_embedding(a, Q3_TEXT_EOS, t); add2(t, pad0, r); } /* :248-254 */
    PUSH_MARK_CODEC(Q3_CODEC_BOS); /* :258-264 */
    return n;
#undef ROW
#undef PUSH_TEXT
#undef PUSH_MARK_CODEC
}

int q3o_build_clone(const q3o_assets* a, const int32_t* text_ids, int n_text, const int32_t* ref_codes,
                    int n_ref_codes, const int32_t* ref_text_ids, int n_ref_text, const float* spk_emb,
                    int lang_id, const int32_t* instr_ids, int n_instr, float* out, int max_rows) {
    int n_steps = n_ref_codes / 16; /* prompt.rs:79 */
    int n_mid = (n_ref_text + 2) + 1 + n_steps + 1;
    float* mid = (float*)malloc((size_t)n_mid * 2048 * 4);
    float pad[2048], t[2048], marker[2048], e[2048];
    int m = 0;
    q3o_codec_embedding(a, 0, Q3_CODEC_PAD, pad); /* :47 */
    for (int i = 0; i < n_ref_text + 2; i++) { /* :41-58 */
        int64_t tid = i == 0 ? Q3_TEXT_BOS : (i == n_ref_text + 1 ? Q3_TEXT_EOS : ref_text_ids[i - 1]);
        q3o_text_embedding(a, tid, t);
        add2(t, pad, mid + (size_t)(m++) * 2048);
    }
    q3o_text_embedding(a, Q3_TEXT_AUDIO_MARKER, marker); /* :67 */
    q3o_codec_embedding(a, 0, Q3_CODEC_AUDIO_START, e);  /* :68 */
    add2(marker, e, mid + (size_t)(m++) * 2048);
    for (int s = 0; s < n_steps; s++) { /* :80-96 */
        float sum[2048];
        for (int i = 0; i < 2048; i++) sum[i] = 0.0f;
        for (int q = 0; q < 16; q++) {
            q3o_codec_embedding(a, q, ref_codes[s * 16 + q], e);
            for (int i = 0; i < 2048; i++) sum[i] += e[i];
        }
        add2(marker, sum, mid + (size_t)(m++) * 2048);
    }
    add2(marker, pad, mid + (size_t)(m++) * 2048); /* :100-106 */
    int n = q3o_build_core(a, text_ids, n_text, 1, lang_id, 0, 0, spk_emb, instr_ids, n_instr, mid, m, out, max_rows);
    free(mid);
    return n;
}

/* ============================ chunker ============================ */
int q3o_chunker_push(q3o_chunker* c, const int64_t* codes, int n, int is_final, q3o_decode_cb cb, void* user) {
    if (c->len + n > (int)(sizeof(c->buf) / sizeof(c->buf[0]))) return -1;
    for (int i = 0; i < n; i++) c->buf[c->len++] = codes[i];
    if (c->len >= Q3_CHUNK_CODES || is_final) { /* engine.rs:510 */
        int valid = (c->len / 16) * 16;         /* :512 */
        if (valid > 0) {
            int64_t safe[4096];
            for (int i = 0; i < valid; i++) { int64_t v = c->buf[i]; safe[i] = v < 0 ? 0 : (v > 2047 ? 2047 : v); } /* :515-519 */
            if (c->n_calls < 1024) { c->call_frames[c->n_calls] = valid / 16; c->call_final[c->n_calls] = is_final; }
            c->n_calls++;
            if (cb) cb(user, safe, valid, is_final);
            int remaining = c->len - valid;
            if (remaining > 0 && !is_final) { memmove(c->buf, c->buf + valid, (size_t)remaining * sizeof(int64_t)); c->len = remaining; } /* :528-533 */
            else c->len = 0;
        } else {
            c->len = 0; /* :535 */
        }
    }
    return 0;
}

/* ============================ loop ============================ */
q3o_engine* q3o_engine_create(const char* dir, const char* codec_path, int n_threads, char* err, size_t errlen) {
    char p[1024];
    q3o_engine* e = (q3o_engine*)calloc(1, sizeof(*e));
    snprintf(p, sizeof(p), "%s/qwen3_assets.gguf", dir);
    e->assets = q3o_assets_load(p, err, errlen);
    if (!e->assets) goto fail;
    snprintf(p, sizeof(p), "%s/qwen3_tts_talker.gguf", dir);
    e->talker = q3o_model_load(p, Q3_TALKER_NCTX, err, errlen); /* engine.rs:133 */
    if (!e->talker) goto fail;
    snprintf(p, sizeof(p), "%s/qwen3_tts_predictor.gguf", dir);
    e->predictor = q3o_model_load(p, Q3_PRED_NCTX, err, errlen); /* engine.rs:136 */
    if (!e->predictor) goto fail;
    if (codec_path && codec_path[0]) {
        e->codec = q3o_codec_load(codec_path, err, errlen);
        if (!e->codec) goto fail;
    }
    e->max_steps = Q3_DEFAULT_MAX_STEPS;
    e->temperature = 0.7f; e->top_k = 40; e->top_p = 0.9f; e->seed = 42; /* engine.rs:25-34 */
    e->n_threads = n_threads;
    e->talker->n_threads = n_threads;
    e->predictor->n_threads = n_threads;
    return e;
fail:
    q3o_engine_free(e);
    return NULL;
}
void q3o_engine_free(q3o_engine* e) {
    if (!e) return;
    q3o_assets_free(e->assets); q3o_model_free(e->talker); q3o_model_free(e->predictor);
    if (e->codec) q3o_codec_free(e->codec);
    free(e);
}

typedef struct { q3o_codec* codec; float* pcm; int max_pcm; int n_pcm; } dec_ctx;
static void dec_cb(void* user, const int64_t* codes, int n_codes, int is_final) {
    dec_ctx* d = (dec_ctx*)user;
    if (!d->codec || !d->pcm) return;
    int got = q3o_codec_decode(d->codec, codes, n_codes / 16, is_final, d->pcm + d->n_pcm, d->max_pcm - d->n_pcm);
    if (got > 0) d->n_pcm += got;
}

int q3o_engine_generate(q3o_engine* e, const float* prompt, int n_prompt, int32_t* codes_out,
                        float* pcm_out, int max_pcm, int* n_pcm_out) {
    q3o_model *T = e->talker, *P = e->predictor;
    const int dT = T->n_embd, dP = P->n_embd;
    if (n_prompt <= 0 || dT != 2048) return -1;
    float* t_hidden = (float*)malloc((size_t)dT * 4);
    float* t_logits = (float*)malloc((size_t)T->n_vocab * 4);
    float* p_logits = (float*)malloc(2048 * 4);
    float* in_p = (float*)malloc((size_t)dP * 4);
    float emb[2048], feedback[2048], step_emb[16][2048];
    q3o_model_clear_kv(T);
    int t_end = T->n_vocab < Q3_SAMPLE_END ? T->n_vocab : Q3_SAMPLE_END;
    for (int t = 0; t < n_prompt; t++) { /* prefill, engine.rs:456-462; positions engine.rs:306-314 */
        int32_t pos[4] = { t, t, t, 0 };
        int last = t == n_prompt - 1;
        if (q3o_model_eval(T, prompt + (size_t)t * 2048, pos, last ? t_hidden : NULL, last ? t_logits : NULL, 0, t_end)) return -2;
    }
    q3o_sampler ts;
    q3o_sampler_init(&ts, e->temperature, e->top_k, e->top_p, e->seed); /* engine.rs:479-485 */
    q3o_chunker ch;
    memset(&ch, 0, sizeof(ch));
    if (e->codec) q3o_codec_reset(e->codec);
    dec_ctx dc = { e->codec, pcm_out, max_pcm, 0 };
    int cur_pos = n_prompt, n_frames = 0;
    for (int step = 0; step < e->max_steps; step++) {
        if (e->mask_eos) t_logits[Q3_CODEC_EOS] = -INFINITY;
        int32_t code0 = q3o_sample(&ts, t_logits, T->n_vocab, 0, Q3_SAMPLE_END); /* :555 */
        float pmargin = INFINITY;
        if (e->margins && 2 * n_frames + 1 < e->margins_cap) {
            float m1 = -INFINITY, m2 = -INFINITY;
            for (int i = 0; i < t_end; i++) { const float v = t_logits[i]; if (v > m1) { m2 = m1; m1 = v; } else if (v > m2) m2 = v; }
            e->margins[2 * n_frames] = m1 - m2;
        }
        if (e->own_codes) e->own_codes[(size_t)n_frames * 16] = code0;
        if (e->forced && n_frames < e->forced_frames) code0 = e->forced[(size_t)n_frames * 16];
        if (code0 == Q3_CODEC_EOS || code0 == Q3_TEXT_EOS) break;                /* :558 */
        int32_t* fc = codes_out + (size_t)n_frames * 16;
        fc[0] = code0;
        q3o_project(e->assets, t_hidden, dT, in_p); /* :568 */
        q3o_model_clear_kv(P);                      /* :575 */
        int32_t pos0[4] = { 0, 0, 0, 0 }, pos1[4] = { 1, 1, 1, 1 };
        if (q3o_model_eval(P, in_p, pos0, NULL, NULL, 0, 0)) return -3;
        q3o_codec_embedding(e->assets, 0, code0, step_emb[0]); /* :585 */
        q3o_project(e->assets, step_emb[0], 2048, in_p);       /* :569 */
        if (q3o_model_eval(P, in_p, pos1, NULL, p_logits, 0, 2048)) return -3;
        for (int q = 1; q < 16; q++) { /* :587-611 */
            float mx = -INFINITY; int mi = 0; /* greedy predictor sampler, :470, mod.rs:690-701 */
            for (int i = 0; i < 2048; i++) if (p_logits[i] > mx) { mx = p_logits[i]; mi = i; }
            if (e->margins) { float m2 = -INFINITY; for (int i = 0; i < 2048; i++) if (i != mi && p_logits[i] > m2) m2 = p_logits[i]; if (mx - m2 < pmargin) pmargin = mx - m2; }
            if (e->own_codes) e->own_codes[(size_t)n_frames * 16 + q] = mi;
            if (e->forced && n_frames < e->forced_frames) mi = e->forced[(size_t)n_frames * 16 + q];
            fc[q] = mi;
            q3o_codec_embedding(e->assets, q, mi, step_emb[q]);
            if (q < 15) {
                q3o_project(e->assets, step_emb[q], 2048, in_p);
                int32_t pp[4] = { q + 1, q + 1, q + 1, q + 1 };
                if (q3o_model_eval(P, in_p, pp, NULL, p_logits, q * 2048, (q + 1) * 2048)) return -3;
            }
        }
        if (e->margins && 2 * n_frames + 1 < e->margins_cap) e->margins[2 * n_frames + 1] = pmargin;
        n_frames++;
        int64_t frame64[16];
        for (int q = 0; q < 16; q++) frame64[q] = fc[q];
        q3o_chunker_push(&ch, frame64, 16, 0, dec_cb, &dc); /* :613-620 */
        for (int i = 0; i < 2048; i++) feedback[i] = 0.0f;  /* :622-630 */
        for (int q = 0; q < 16; q++) for (int i = 0; i < 2048; i++) feedback[i] += step_emb[q][i];
        for (int i = 0; i < 2048; i++) feedback[i] += e->assets->tts_pad[i];
        int32_t tp[4] = { cur_pos, cur_pos, cur_pos, 0 }; /* :633 */
        (void)emb;
        if (q3o_model_eval(T, feedback, tp, t_hidden, t_logits, 0, t_end)) return -2;
        cur_pos++;
    }
    q3o_chunker_push(&ch, NULL, 0, 1, dec_cb, &dc); /* :644 */
    if (n_pcm_out) *n_pcm_out = dc.n_pcm;
    free(t_hidden); free(t_logits); free(p_logits); free(in_p);
    return n_frames;
}
