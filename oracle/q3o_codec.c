/*
 * q3o_codec.c -- ORACLE (test infrastructure): streaming codec decoder, codes -> 24 kHz PCM.
 *
 * What is pinned by the reference: only the call contract of AudioDecoder::decode
 * (/root/reference/src/models/onnx.rs:342-458: audio_codes i64 [1,N,16], is_last, streaming state,
 * final_wav + valid_samples) and the state shapes (onnx.rs:474-495: pre_conv_history [1,512,T],
 * past_key/value [1,16,T,64] x 8, 1024-channel latent).  The graph itself (qwen3_tts_decoder.onnx) is
 * NOT in /root/reference nor in this image => "parity unpinned" for the PCM.
 *
 * This file defines the stand-in "Q3TTS-codec-synth": the architecture of transformers'
 * Qwen3OmniMoeCode2Wav (modeling_qwen3_omni_moe.py:3180-3697 [EXT analogue]) made exactly streamable:
 * every convolution is causal with explicit history, the transposed convolutions keep the first T*stride
 * outputs (right trim k-stride), attention is a 72-wide sliding window over a per-layer KV history.
 * Streaming in any chunking therefore equals one full-sequence pass (tests check this), and
 * samples/frame = prod(up_ratios)*prod(dec_rates) = 1920 for the default config.
 * Accumulation is in double (this is a float-tolerance oracle: PCM within 1e-4 RMS).
 */
#include "q3o.h"
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>

typedef struct { int cin, cout, k, dil, groups; const float *w, *b; float* hist; int hlen; } conv_t;     /* causal conv */
typedef struct { int cin, cout, k, s; const float *w, *b; float* prev; int nprev; } convt_t;           /* causal transposed conv */
typedef struct { int c; float *ea, *inv_eb; } snake_t;
typedef struct { snake_t s1, s2; conv_t c1, c2; } resunit_t;
typedef struct { snake_t snake; convt_t ct; resunit_t ru[3]; } decblock_t;
typedef struct { convt_t ct; conv_t dw; const float *ln_w, *ln_b, *pw1_w, *pw1_b, *pw2_w, *pw2_b, *gamma; } upstage_t;
typedef struct { const float *attn_norm, *wq, *wk, *wv, *wo, *ls_attn, *ffn_norm, *w_gate, *w_up, *w_down, *ls_ffn; float *kh, *vh; } tflayer_t;

struct q3o_codec {
    q3o_gguf* g;
    int n_q, cb_size, cb_dim, hidden, n_layers, n_heads, head_dim, ffn, window, n_up, dec_dim, n_dec;
    int up_ratios[4], dec_rates[8];
    float rope_base, eps;
    const float* codebook[32];
    conv_t pre_conv;
    tflayer_t* tf; const float* tf_norm;
    int kv_len;     /* positions currently held in history (<= window-1) */
    int64_t n_seen; /* absolute frame index of the next frame */
    upstage_t up[4];
    conv_t conv_in;
    decblock_t dec[8];
    snake_t snake_out; conv_t conv_out;
};

static int kvi(const q3o_gguf* g, const char* key, int def) {
    const q3o_gguf_kv* kv = q3o_gguf_kv_find(g, key);
    if (!kv) return def;
    if (kv->type == 4 || kv->type == 10 || kv->type == 0 || kv->type == 2) return (int)kv->v.u;
    if (kv->type == 5 || kv->type == 11 || kv->type == 1 || kv->type == 3) return (int)kv->v.i;
    return def;
}
static float kvf(const q3o_gguf* g, const char* key, float def) {
    const q3o_gguf_kv* kv = q3o_gguf_kv_find(g, key);
    return (kv && (kv->type == 6 || kv->type == 12)) ? (float)kv->v.f : def;
}
static const float* T(q3o_codec* c, char* err, size_t errlen, const char* fmt, ...) {
    char nm[160];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(nm, sizeof(nm), fmt, ap);
    va_end(ap);
    const q3o_gguf_tensor* t = q3o_gguf_find(c->g, nm);
    if (!t || t->type != Q3_T_F32) { if (!err[0]) snprintf(err, errlen, "codec tensor %s missing or not F32", nm); return NULL; }
    return (const float*)t->data;
}
static void conv_init(conv_t* cv, int cin, int cout, int k, int dil, int groups, const float* w, const float* b) {
    cv->cin = cin; cv->cout = cout; cv->k = k; cv->dil = dil; cv->groups = groups; cv->w = w; cv->b = b;
    cv->hlen = (k - 1) * dil;
    cv->hist = (float*)calloc((size_t)cin * (size_t)(cv->hlen > 0 ? cv->hlen : 1), 4);
}
static void convt_init(convt_t* ct, int cin, int cout, int k, int s, const float* w, const float* b) {
    ct->cin = cin; ct->cout = cout; ct->k = k; ct->s = s; ct->w = w; ct->b = b;
    ct->nprev = (k + s - 1) / s - 1;
    ct->prev = (float*)calloc((size_t)cin * (size_t)(ct->nprev > 0 ? ct->nprev : 1), 4);
}
static void snake_init(snake_t* s, int c, const float* alpha, const float* beta) {
    s->c = c;
    s->ea = (float*)malloc((size_t)c * 4);
    s->inv_eb = (float*)malloc((size_t)c * 4);
    if (!alpha || !beta) return;
    for (int i = 0; i < c; i++) { s->ea[i] = expf(alpha[i]); s->inv_eb[i] = 1.0f / (expf(beta[i]) + 1e-9f); }
}

/* in [cin][T] -> out [cout][T]; x(t<0) comes from hist (zeros at stream start) */
static void conv_run(conv_t* cv, const float* in, int T, float* out) {
    const int H = cv->hlen, W = H + T;
    float* ext = (float*)malloc((size_t)cv->cin * (size_t)W * 4);
    for (int c = 0; c < cv->cin; c++) {
        memcpy(ext + (size_t)c * W, cv->hist + (size_t)c * H, (size_t)H * 4);
        memcpy(ext + (size_t)c * W + H, in + (size_t)c * T, (size_t)T * 4);
    }
    const int cpg_in = cv->cin / cv->groups, cpg_out = cv->cout / cv->groups;
    /* (channel, time) pairs are independent sums: collapsing both loops keeps every core busy in the narrow stages at the end of the stack
       (96 ... 1 output channels over tens of thousands of samples) without changing a single result */
#pragma omp parallel for collapse(2) schedule(static)
    for (int co = 0; co < cv->cout; co++) {
        for (int t = 0; t < T; t++) {
            const int g = co / cpg_out;
            double acc = cv->b ? cv->b[co] : 0.0;
            for (int ci = 0; ci < cpg_in; ci++) {
                const float* x = ext + (size_t)(g * cpg_in + ci) * W + t; /* x[t + j*dil] == input time t-(k-1-j)*dil */
                const float* w = cv->w + ((size_t)co * cpg_in + ci) * cv->k;
                for (int j = 0; j < cv->k; j++) acc += (double)w[j] * (double)x[j * cv->dil];
            }
            out[(size_t)co * T + t] = (float)acc;
        }
    }
    for (int c = 0; c < cv->cin; c++) memcpy(cv->hist + (size_t)c * H, ext + (size_t)c * W + T, (size_t)H * 4);
    free(ext);
}
/* in [cin][T] -> out [cout][T*s]; y[n] = b + sum_ci sum_{t: 0<=n-t*s<k} x[ci][t]*w[ci][co][n-t*s] */
static void convt_run(convt_t* ct, const float* in, int T, float* out) {
    const int P = ct->nprev, W = P + T, s = ct->s, k = ct->k;
    float* ext = (float*)malloc((size_t)ct->cin * (size_t)W * 4);
    for (int c = 0; c < ct->cin; c++) {
        memcpy(ext + (size_t)c * W, ct->prev + (size_t)c * P, (size_t)P * 4);
        memcpy(ext + (size_t)c * W + P, in + (size_t)c * T, (size_t)T * 4);
    }
#pragma omp parallel for collapse(2) schedule(static)
    for (int co = 0; co < ct->cout; co++) {
        for (int n = 0; n < T * s; n++) {
            double acc = ct->b ? ct->b[co] : 0.0;
            int t_hi = n / s;
            for (int t = t_hi; t >= t_hi - P; t--) {
                int j = n - t * s;
                if (j >= k) break;
                for (int ci = 0; ci < ct->cin; ci++)
                    acc += (double)ext[(size_t)ci * W + P + t] * (double)ct->w[((size_t)ci * ct->cout + co) * k + j];
            }
            out[(size_t)co * T * s + n] = (float)acc;
        }
    }
    for (int c = 0; c < ct->cin; c++) memcpy(ct->prev + (size_t)c * P, ext + (size_t)c * W + T, (size_t)P * 4);
    free(ext);
}
static void snake_run(const snake_t* s, float* x, int T) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int c = 0; c < s->c; c++)
        for (int t = 0; t < T; t++) {
            float v = x[(size_t)c * T + t];
            float sn = sinf(v * s->ea[c]);
            x[(size_t)c * T + t] = v + s->inv_eb[c] * (sn * sn);
        }
}
/* y[n][t] = sum_k W[n][k] x[k][t] (+b[n]) ; x,y channel-major [C][T] */
static void linear_ct(const float* W, const float* b, int n_out, int n_in, const float* x, int T, float* y) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < n_out; n++)
        for (int t = 0; t < T; t++) {
            double acc = b ? b[n] : 0.0;
            for (int k = 0; k < n_in; k++) acc += (double)W[(size_t)n * n_in + k] * (double)x[(size_t)k * T + t];
            y[(size_t)n * T + t] = (float)acc;
        }
}
static void rmsnorm_ct(const float* x, const float* g, int C, int T, float eps, float* y) {
    for (int t = 0; t < T; t++) {
        double ss = 0;
        for (int c = 0; c < C; c++) ss += (double)x[(size_t)c * T + t] * x[(size_t)c * T + t];
        float sc = (float)(1.0 / sqrt(ss / C + eps));
        for (int c = 0; c < C; c++) y[(size_t)c * T + t] = x[(size_t)c * T + t] * sc * g[c];
    }
}

q3o_codec* q3o_codec_load(const char* path, char* err, size_t errlen) {
    err[0] = 0;
    q3o_gguf* g = q3o_gguf_open(path, err, errlen);
    if (!g) return NULL;
    q3o_codec* c = (q3o_codec*)calloc(1, sizeof(*c));
    c->g = g;
    c->n_q = kvi(g, "codec.n_codebooks", 16); c->cb_size = kvi(g, "codec.codebook_size", 2048);
    c->cb_dim = kvi(g, "codec.codebook_dim", 512); c->hidden = kvi(g, "codec.hidden", 1024);
    c->n_layers = kvi(g, "codec.n_layers", 8); c->n_heads = kvi(g, "codec.n_heads", 16);
    c->head_dim = kvi(g, "codec.head_dim", 64); c->ffn = kvi(g, "codec.ffn", 3072);
    c->window = kvi(g, "codec.window", 72); c->dec_dim = kvi(g, "codec.dec_dim", 1536);
    c->rope_base = kvf(g, "codec.rope_base", 10000.0f); c->eps = kvf(g, "codec.eps", 1e-5f);
    c->n_up = kvi(g, "codec.n_up", 2); c->n_dec = kvi(g, "codec.n_dec", 4);
    for (int i = 0; i < c->n_up; i++) { char k[64]; snprintf(k, sizeof(k), "codec.up_ratio.%d", i); c->up_ratios[i] = kvi(g, k, 2); }
    static const int def_rates[8] = { 8, 5, 4, 3, 2, 2, 2, 2 };
    for (int i = 0; i < c->n_dec; i++) { char k[64]; snprintf(k, sizeof(k), "codec.dec_rate.%d", i); c->dec_rates[i] = kvi(g, k, def_rates[i]); }
    const int H = c->hidden;
    for (int q = 0; q < c->n_q; q++) c->codebook[q] = T(c, err, errlen, "codec.codebook.%d", q);
    conv_init(&c->pre_conv, c->cb_dim, H, 3, 1, 1, T(c, err, errlen, "codec.pre_conv.weight"), T(c, err, errlen, "codec.pre_conv.bias"));
    c->tf = (tflayer_t*)calloc((size_t)c->n_layers, sizeof(tflayer_t));
    for (int l = 0; l < c->n_layers; l++) {
        tflayer_t* L = &c->tf[l];
        L->attn_norm = T(c, err, errlen, "codec.tf.%d.attn_norm", l); L->wq = T(c, err, errlen, "codec.tf.%d.wq", l);
        L->wk = T(c, err, errlen, "codec.tf.%d.wk", l); L->wv = T(c, err, errlen, "codec.tf.%d.wv", l);
        L->wo = T(c, err, errlen, "codec.tf.%d.wo", l); L->ls_attn = T(c, err, errlen, "codec.tf.%d.ls_attn", l);
        L->ffn_norm = T(c, err, errlen, "codec.tf.%d.ffn_norm", l); L->w_gate = T(c, err, errlen, "codec.tf.%d.w_gate", l);
        L->w_up = T(c, err, errlen, "codec.tf.%d.w_up", l); L->w_down = T(c, err, errlen, "codec.tf.%d.w_down", l);
        L->ls_ffn = T(c, err, errlen, "codec.tf.%d.ls_ffn", l);
        L->kh = (float*)calloc((size_t)c->window * c->n_heads * c->head_dim, 4);
        L->vh = (float*)calloc((size_t)c->window * c->n_heads * c->head_dim, 4);
    }
    c->tf_norm = T(c, err, errlen, "codec.tf.norm");
    for (int i = 0; i < c->n_up; i++) {
        upstage_t* u = &c->up[i];
        int f = c->up_ratios[i];
        convt_init(&u->ct, H, H, f, f, T(c, err, errlen, "codec.up.%d.convt.weight", i), T(c, err, errlen, "codec.up.%d.convt.bias", i));
        conv_init(&u->dw, H, H, 7, 1, H, T(c, err, errlen, "codec.up.%d.dw.weight", i), T(c, err, errlen, "codec.up.%d.dw.bias", i));
        u->ln_w = T(c, err, errlen, "codec.up.%d.ln.weight", i); u->ln_b = T(c, err, errlen, "codec.up.%d.ln.bias", i);
        u->pw1_w = T(c, err, errlen, "codec.up.%d.pw1.weight", i); u->pw1_b = T(c, err, errlen, "codec.up.%d.pw1.bias", i);
        u->pw2_w = T(c, err, errlen, "codec.up.%d.pw2.weight", i); u->pw2_b = T(c, err, errlen, "codec.up.%d.pw2.bias", i);
        u->gamma = T(c, err, errlen, "codec.up.%d.gamma", i);
    }
    conv_init(&c->conv_in, H, c->dec_dim, 7, 1, 1, T(c, err, errlen, "codec.dec.conv_in.weight"), T(c, err, errlen, "codec.dec.conv_in.bias"));
    int ch = c->dec_dim;
    for (int b = 0; b < c->n_dec; b++) {
        decblock_t* B = &c->dec[b];
        int r = c->dec_rates[b], co = ch / 2;
        snake_init(&B->snake, ch, T(c, err, errlen, "codec.dec.%d.snake.alpha", b), T(c, err, errlen, "codec.dec.%d.snake.beta", b));
        convt_init(&B->ct, ch, co, 2 * r, r, T(c, err, errlen, "codec.dec.%d.convt.weight", b), T(c, err, errlen, "codec.dec.%d.convt.bias", b));
        static const int dil[3] = { 1, 3, 9 };
        for (int u = 0; u < 3; u++) {
            resunit_t* R = &B->ru[u];
            snake_init(&R->s1, co, T(c, err, errlen, "codec.dec.%d.ru.%d.snake1.alpha", b, u), T(c, err, errlen, "codec.dec.%d.ru.%d.snake1.beta", b, u));
            conv_init(&R->c1, co, co, 7, dil[u], 1, T(c, err, errlen, "codec.dec.%d.ru.%d.conv1.weight", b, u), T(c, err, errlen, "codec.dec.%d.ru.%d.conv1.bias", b, u));
            snake_init(&R->s2, co, T(c, err, errlen, "codec.dec.%d.ru.%d.snake2.alpha", b, u), T(c, err, errlen, "codec.dec.%d.ru.%d.snake2.beta", b, u));
            conv_init(&R->c2, co, co, 1, 1, 1, T(c, err, errlen, "codec.dec.%d.ru.%d.conv2.weight", b, u), T(c, err, errlen, "codec.dec.%d.ru.%d.conv2.bias", b, u));
        }
        ch = co;
    }
    snake_init(&c->snake_out, ch, T(c, err, errlen, "codec.dec.snake_out.alpha"), T(c, err, errlen, "codec.dec.snake_out.beta"));
    conv_init(&c->conv_out, ch, 1, 7, 1, 1, T(c, err, errlen, "codec.dec.conv_out.weight"), T(c, err, errlen, "codec.dec.conv_out.bias"));
    if (err[0]) { q3o_codec_free(c); return NULL; }
    return c;
}

static void conv_free(conv_t* c) { free(c->hist); }
static void snake_free(snake_t* s) { free(s->ea); free(s->inv_eb); }
void q3o_codec_free(q3o_codec* c) {
    if (!c) return;
    conv_free(&c->pre_conv); conv_free(&c->conv_in); conv_free(&c->conv_out); snake_free(&c->snake_out);
    if (c->tf) for (int l = 0; l < c->n_layers; l++) { free(c->tf[l].kh); free(c->tf[l].vh); }
    free(c->tf);
    for (int i = 0; i < c->n_up; i++) { free(c->up[i].ct.prev); conv_free(&c->up[i].dw); }
    for (int b = 0; b < c->n_dec; b++) {
        snake_free(&c->dec[b].snake); free(c->dec[b].ct.prev);
        for (int u = 0; u < 3; u++) { snake_free(&c->dec[b].ru[u].s1); snake_free(&c->dec[b].ru[u].s2); conv_free(&c->dec[b].ru[u].c1); conv_free(&c->dec[b].ru[u].c2); }
    }
    q3o_gguf_close(c->g);
    free(c);
}
static void conv_reset(conv_t* c) { memset(c->hist, 0, (size_t)c->cin * (size_t)(c->hlen > 0 ? c->hlen : 1) * 4); }
static void convt_reset(convt_t* c) { memset(c->prev, 0, (size_t)c->cin * (size_t)(c->nprev > 0 ? c->nprev : 1) * 4); }
void q3o_codec_reset(q3o_codec* c) { /* AudioDecoder::create_state, onnx.rs:474-495: all-empty state */
    conv_reset(&c->pre_conv); conv_reset(&c->conv_in); conv_reset(&c->conv_out);
    c->kv_len = 0; c->n_seen = 0;
    for (int i = 0; i < c->n_up; i++) { convt_reset(&c->up[i].ct); conv_reset(&c->up[i].dw); }
    for (int b = 0; b < c->n_dec; b++) { convt_reset(&c->dec[b].ct); for (int u = 0; u < 3; u++) { conv_reset(&c->dec[b].ru[u].c1); conv_reset(&c->dec[b].ru[u].c2); } }
}
/* OpenMP team size of the calling thread's next parallel regions (the codec thread of engine.rs:495 runs with the library's default team;
   bench.py's cpu_baseline leg sets it explicitly so that the 4-thread and the all-cores figures are what they say) */
void q3o_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int q3o_codec_samples_per_frame(const q3o_codec* c) {
    int s = 1;
    for (int i = 0; i < c->n_up; i++) s *= c->up_ratios[i];
    for (int i = 0; i < c->n_dec; i++) s *= c->dec_rates[i];
    return s;
}

/* one transformer layer over T new frames; h [H][T] in place */
static void tf_layer(q3o_codec* c, tflayer_t* L, float* h, int T) {
    const int H = c->hidden, nh = c->n_heads, hd = c->head_dim, W = c->window, half = hd / 2;
    float* xn = (float*)malloc((size_t)H * T * 4);
    float* q = (float*)malloc((size_t)nh * hd * T * 4);
    float* k = (float*)malloc((size_t)nh * hd * T * 4);
    float* v = (float*)malloc((size_t)nh * hd * T * 4);
    float* att = (float*)malloc((size_t)nh * hd * T * 4);
    float* o = (float*)malloc((size_t)H * T * 4);
    rmsnorm_ct(h, L->attn_norm, H, T, c->eps, xn);
    linear_ct(L->wq, NULL, nh * hd, H, xn, T, q);
    linear_ct(L->wk, NULL, nh * hd, H, xn, T, k);
    linear_ct(L->wv, NULL, nh * hd, H, xn, T, v);
    for (int t = 0; t < T; t++) { /* NeoX RoPE at absolute position */
        double p = (double)(c->n_seen + t);
        for (int hh = 0; hh < nh; hh++)
            for (int i = 0; i < half; i++) {
                double ang = p * pow((double)c->rope_base, -(double)i / half);
                float cs = (float)cos(ang), sn = (float)sin(ang);
                size_t i1 = (size_t)(hh * hd + i) * T + t, i2 = (size_t)(hh * hd + i + half) * T + t;
                float a = q[i1], b = q[i2];
                q[i1] = a * cs - b * sn; q[i2] = b * cs + a * sn;
                a = k[i1]; b = k[i2];
                k[i1] = a * cs - b * sn; k[i2] = b * cs + a * sn;
            }
    }
    /* keys/values visible: history (kv_len) then the T new ones; position-major [pos][nh*hd] */
    const int Lh = c->kv_len, tot = Lh + T;
    float* K = (float*)malloc((size_t)tot * nh * hd * 4);
    float* V = (float*)malloc((size_t)tot * nh * hd * 4);
    memcpy(K, L->kh, (size_t)Lh * nh * hd * 4);
    memcpy(V, L->vh, (size_t)Lh * nh * hd * 4);
    for (int t = 0; t < T; t++)
        for (int e = 0; e < nh * hd; e++) { K[(size_t)(Lh + t) * nh * hd + e] = k[(size_t)e * T + t]; V[(size_t)(Lh + t) * nh * hd + e] = v[(size_t)e * T + t]; }
    const double scale = 1.0 / sqrt((double)hd);
#pragma omp parallel for schedule(static)
    for (int hh = 0; hh < nh; hh++)
        for (int t = 0; t < T; t++) {
            int j1 = Lh + t, j0 = j1 - (W - 1);
            if (j0 < 0) j0 = 0;
            double s[512], mx = -1e300, den = 0;
            for (int j = j0; j <= j1; j++) {
                double a = 0;
                for (int d = 0; d < hd; d++) a += (double)q[(size_t)(hh * hd + d) * T + t] * K[(size_t)j * nh * hd + hh * hd + d];
                s[j - j0] = a * scale;
                if (s[j - j0] > mx) mx = s[j - j0];
            }
            for (int j = j0; j <= j1; j++) { s[j - j0] = exp(s[j - j0] - mx); den += s[j - j0]; }
            for (int d = 0; d < hd; d++) {
                double a = 0;
                for (int j = j0; j <= j1; j++) a += s[j - j0] * V[(size_t)j * nh * hd + hh * hd + d];
                att[(size_t)(hh * hd + d) * T + t] = (float)(a / den);
            }
        }
    /* keep the last window-1 positions */
    int keep = tot < W - 1 ? tot : W - 1;
    memcpy(L->kh, K + (size_t)(tot - keep) * nh * hd, (size_t)keep * nh * hd * 4);
    memcpy(L->vh, V + (size_t)(tot - keep) * nh * hd, (size_t)keep * nh * hd * 4);
    linear_ct(L->wo, NULL, H, nh * hd, att, T, o);
    for (int cc = 0; cc < H; cc++) for (int t = 0; t < T; t++) h[(size_t)cc * T + t] += L->ls_attn[cc] * o[(size_t)cc * T + t];
    rmsnorm_ct(h, L->ffn_norm, H, T, c->eps, xn);
    float* gt = (float*)malloc((size_t)c->ffn * T * 4);
    float* up = (float*)malloc((size_t)c->ffn * T * 4);
    linear_ct(L->w_gate, NULL, c->ffn, H, xn, T, gt);
    linear_ct(L->w_up, NULL, c->ffn, H, xn, T, up);
    for (size_t i = 0; i < (size_t)c->ffn * T; i++) { float g = gt[i]; gt[i] = (g / (1.0f + expf(-g))) * up[i]; }
    linear_ct(L->w_down, NULL, H, c->ffn, gt, T, o);
    for (int cc = 0; cc < H; cc++) for (int t = 0; t < T; t++) h[(size_t)cc * T + t] += L->ls_ffn[cc] * o[(size_t)cc * T + t];
    free(xn); free(q); free(k); free(v); free(att); free(o); free(K); free(V); free(gt); free(up);
}

int q3o_codec_decode(q3o_codec* c, const int64_t* codes, int n_frames, int is_last, float* pcm, int max_samples) {
    (void)is_last; /* causal stack: nothing is held back, so is_last changes nothing (valid_samples == all) */
    if (n_frames <= 0) return 0;
    const int H = c->hidden;
    int T = n_frames;
    if (T * q3o_codec_samples_per_frame(c) > max_samples) return -1;
    /* 1. RVQ sum */
    float* z = (float*)calloc((size_t)c->cb_dim * T, 4);
    for (int t = 0; t < T; t++)
        for (int q = 0; q < c->n_q; q++) {
            int64_t code = codes[(size_t)t * c->n_q + q];
            if (code < 0) code = 0;
            if (code >= c->cb_size) code = c->cb_size - 1;
            const float* row = c->codebook[q] + (size_t)code * c->cb_dim;
            for (int d = 0; d < c->cb_dim; d++) z[(size_t)d * T + t] += row[d];
        }
    /* 2. pre_conv */
    float* h = (float*)malloc((size_t)H * T * 4);
    conv_run(&c->pre_conv, z, T, h);
    free(z);
    /* 3. transformer */
    for (int l = 0; l < c->n_layers; l++) tf_layer(c, &c->tf[l], h, T);
    {
        int tot = c->kv_len + T;
        c->kv_len = tot < c->window - 1 ? tot : c->window - 1;
        c->n_seen += T;
    }
    float* hn = (float*)malloc((size_t)H * T * 4);
    rmsnorm_ct(h, c->tf_norm, H, T, c->eps, hn);
    free(h);
    float* x = hn;
    /* 4. upsample stages */
    for (int i = 0; i < c->n_up; i++) {
        upstage_t* u = &c->up[i];
        int f = c->up_ratios[i];
        float* y = (float*)malloc((size_t)H * T * f * 4);
        convt_run(&u->ct, x, T, y);
        free(x);
        T *= f;
        float* dwo = (float*)malloc((size_t)H * T * 4);
        conv_run(&u->dw, y, T, dwo);
        for (int t = 0; t < T; t++) { /* LayerNorm over channels, eps 1e-6 */
            double mu = 0, var = 0;
            for (int cc = 0; cc < H; cc++) mu += dwo[(size_t)cc * T + t];
            mu /= H;
            for (int cc = 0; cc < H; cc++) { double dlt = dwo[(size_t)cc * T + t] - mu; var += dlt * dlt; }
            var /= H;
            float inv = (float)(1.0 / sqrt(var + 1e-6));
            for (int cc = 0; cc < H; cc++) dwo[(size_t)cc * T + t] = (float)((dwo[(size_t)cc * T + t] - mu) * inv) * u->ln_w[cc] + u->ln_b[cc];
        }
        float* m1 = (float*)malloc((size_t)4 * H * T * 4);
        linear_ct(u->pw1_w, u->pw1_b, 4 * H, H, dwo, T, m1);
        for (size_t e = 0; e < (size_t)4 * H * T; e++) { float vv = m1[e]; m1[e] = 0.5f * vv * (1.0f + erff(vv * 0.70710678118654752f)); }
        linear_ct(u->pw2_w, u->pw2_b, H, 4 * H, m1, T, dwo);
        for (int cc = 0; cc < H; cc++) for (int t = 0; t < T; t++) y[(size_t)cc * T + t] += u->gamma[cc] * dwo[(size_t)cc * T + t];
        free(dwo); free(m1);
        x = y;
    }
    /* 5. conv decoder */
    int ch = c->dec_dim;
    float* d = (float*)malloc((size_t)ch * T * 4);
    conv_run(&c->conv_in, x, T, d);
    free(x);
    for (int b = 0; b < c->n_dec; b++) {
        decblock_t* B = &c->dec[b];
        int r = c->dec_rates[b], co = ch / 2;
        snake_run(&B->snake, d, T);
        float* y = (float*)malloc((size_t)co * T * r * 4);
        convt_run(&B->ct, d, T, y);
        free(d);
        T *= r;
        float* tmp = (float*)malloc((size_t)co * T * 4);
        float* tmp2 = (float*)malloc((size_t)co * T * 4);
        for (int u = 0; u < 3; u++) {
            resunit_t* R = &B->ru[u];
            memcpy(tmp, y, (size_t)co * T * 4);
            snake_run(&R->s1, tmp, T);
            conv_run(&R->c1, tmp, T, tmp2);
            snake_run(&R->s2, tmp2, T);
            conv_run(&R->c2, tmp2, T, tmp);
            for (size_t e = 0; e < (size_t)co * T; e++) y[e] += tmp[e];
        }
        free(tmp); free(tmp2);
        d = y; ch = co;
    }
    snake_run(&c->snake_out, d, T);
    float* w = (float*)malloc((size_t)T * 4);
    conv_run(&c->conv_out, d, T, w);
    for (int t = 0; t < T; t++) pcm[t] = w[t] < -1.0f ? -1.0f : (w[t] > 1.0f ? 1.0f : w[t]);
    free(w); free(d);
    return T;
}
