/*
 * q3o.h -- CPU ORACLE for the Qwen3-TTS hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This directory restates, in plain C, the algorithm of the reference's hot path
 * (/root/reference/src/tts/engine.rs:445-656 and the files it calls) under the arithmetic
 * specification of include/q3tts_spec.h.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product (qwen3-tts-rust_amd/) never links or calls it.
 *
 * PARITY PINNING STATUS: the reference has no tests and no golden vectors, and its
 * transformer / vocoder arithmetic lives in llama.cpp b8123 + onnxruntime 1.24.2, neither of which is
 * in /root/reference or this image.  Source-pinned parts (loop protocol, sampler, projection order,
 * gathers, feedback sum, prompt layout, chunker, mel, file formats) are restated line-for-line in
 * meaning and pinned by known-answer tests derived from the reference source; the transformer block
 * math is pinned against the locally installed transformers Qwen3 modules (float tolerance), the mel
 * front end against transformers.audio_utils, the codec building blocks against the transformers
 * Code2Wav modules, the sampler's ChaCha12 core against a published known-answer vector; the
 * quantised-matmul rounding (llama.cpp), the exported codec-decoder graph (ONNX), the RNG seed
 * expansion and the tokenizer remain "parity unpinned".
 */
#ifndef Q3O_H
#define Q3O_H
#include <stdint.h>
#include <stddef.h>
#include "../include/q3tts_spec.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------- GGUF container (follows assets_manager.rs:33-148 + public GGUF spec) ------------- */
typedef struct q3o_gguf_tensor {
    char name[128];
    int n_dims;
    int64_t ne[4];     /* ne[0] fastest */
    int type;          /* q3_ggml_type */
    uint64_t offset;   /* relative to data start */
    const uint8_t* data;
    size_t nbytes;
} q3o_gguf_tensor;

typedef struct q3o_gguf_kv {
    char key[128];
    int type;            /* gguf value type 0..12 */
    int arr_type;        /* for arrays */
    uint64_t arr_n;
    union { uint64_t u; int64_t i; double f; } v;   /* scalars widened */
    char* str;           /* strings */
    void* arr;           /* raw array payload (numeric arrays only) */
} q3o_gguf_kv;

typedef struct q3o_gguf {
    uint8_t* map; size_t map_size; int fd;
    uint32_t version;
    uint64_t n_tensors, n_kv;
    q3o_gguf_tensor* tensors;
    q3o_gguf_kv* kv;
    size_t data_start;
} q3o_gguf;

q3o_gguf* q3o_gguf_open(const char* path, char* err, size_t errlen);
void q3o_gguf_close(q3o_gguf* g);
const q3o_gguf_tensor* q3o_gguf_find(const q3o_gguf* g, const char* name);
const q3o_gguf_kv* q3o_gguf_kv_find(const q3o_gguf* g, const char* key);
size_t q3o_type_row_bytes(int type, int64_t k);

/* ---------------- quantised rows ---------------- */
void q3o_dequant_row(int type, const void* row, int64_t k, float* out);
/* spec S2: quantise activations: q[k] int8, d[k/32] f16 */
void q3o_quant_act(const float* x, int64_t k, int8_t* q, uint16_t* d);
/* spec S3: y[n] = dot(W row n, x); W is [n][k] of `type`.  For quantised W uses (xq, xd); for
 * 16/32-bit float W uses xf. */
void q3o_matvec(int type, const void* w, int64_t n, int64_t k, const int8_t* xq, const uint16_t* xd,
                const float* xf, float* y);
/* "ggml-CPU" arithmetic mode (q3o_ggml.c; env Q3_SPEC=ggml or q3o_set_arith_mode(1)): llama.cpp's portable CPU kernels restated [EXT] */
int q3o_arith_mode(void);
void q3o_set_arith_mode(int mode);
void q3o_matvec_ggml(int type, const void* w, int64_t n, int64_t k, const float* xf, float* y);
void q3o_rmsnorm_ggml(const float* x, const float* g, int64_t d, float eps, float* y);
void q3o_headnorm128_ggml(const float* x, const float* g, float eps, float* y);
float q3o_swiglu_ggml(float gt, float up);
void q3o_attn_head_ggml(const float* q, const uint16_t* K, const uint16_t* V, size_t stride, int n, float* out);
float q3o_sumsq_vec(const float* x, int64_t d);
void q3o_rmsnorm(const float* x, const float* g, int64_t d, float eps, float* y);
void q3o_headnorm128(const float* x, const float* g, float eps, float* y);

/* ---------------- transformer (talker / predictor) ---------------- */
typedef struct q3o_layer {
    const q3o_gguf_tensor *attn_norm, *wq, *wk, *wv, *wo, *q_norm, *k_norm, *ffn_norm, *w_gate, *w_up, *w_down;
} q3o_layer;

typedef struct q3o_model {
    q3o_gguf* g;
    char arch[64];
    int n_embd, n_layer, n_head, n_head_kv, head_dim, n_ff, n_vocab, n_ctx;
    float eps, rope_base;
    int32_t mrope_sec[4];
    q3o_layer* layers;
    const q3o_gguf_tensor *output_norm, *output;
    /* state */
    uint16_t* kcache; /* [layer][pos][kvh][128] f16 */
    uint16_t* vcache;
    int n_past;
    float* rope_cos; /* [n_ctx][64] */
    float* rope_sin;
    int n_threads;
} q3o_model;

q3o_model* q3o_model_load(const char* path, int n_ctx, char* err, size_t errlen);
void q3o_model_free(q3o_model* m);
void q3o_model_clear_kv(q3o_model* m);
/* one token: x[n_embd] embedding, pos[4] M-RoPE streams (predictor: pos[0] only, others ignored when
 * sections are zero).  hidden_out[n_embd] = final-norm hidden (may be NULL); logits for rows
 * [row0,row1) of the output matrix written to logits_out[row1-row0] (may be NULL). */
int q3o_model_eval(q3o_model* m, const float* x, const int32_t pos[4], float* hidden_out,
                   float* logits_out, int row0, int row1);

/* ---------------- assets (assets_manager.rs) ---------------- */
typedef struct q3o_assets {
    q3o_gguf* g;
    const float* proj_w; int64_t proj_out, proj_in;   /* [out][in] */
    const float* proj_b;
    const float* text_table; int64_t text_rows;       /* rows of 2048 */
    const float* codec[16]; int64_t codec_rows[16]; int n_codec;
    float tts_pad[2048];
} q3o_assets;

q3o_assets* q3o_assets_load(const char* path, char* err, size_t errlen);
void q3o_assets_free(q3o_assets* a);
void q3o_project(const q3o_assets* a, const float* hidden, int64_t n_in, float* out);  /* :383-399 */
void q3o_codec_embedding(const q3o_assets* a, int q, int32_t code, float* out2048);   /* :419-437 */
void q3o_text_embedding(const q3o_assets* a, int64_t token, float* out2048);         /* :444-460 */

/* ---------------- sampler (llama/mod.rs:666-776) ---------------- */
typedef struct q3o_rng { uint32_t key[8]; uint64_t counter; uint32_t buf[16]; int idx; } q3o_rng;
void q3o_rng_seed(q3o_rng* r, uint64_t seed);   /* rand 0.10 StdRng::seed_from_u64 [EXT] */
uint32_t q3o_rng_next_u32(q3o_rng* r);
typedef struct q3o_sampler { float temperature; int top_k; float top_p; q3o_rng rng; } q3o_sampler;
void q3o_sampler_init(q3o_sampler* s, float temperature, int top_k, float top_p, uint64_t seed);
int32_t q3o_sample(q3o_sampler* s, const float* logits, int n_vocab, int start, int end);

/* ---------------- prompt builder (prompt.rs) ---------------- */
/* returns number of rows written into out (capacity max_rows rows of 2048 f32); <0 on overflow */
int q3o_build_core(const q3o_assets* a, const int32_t* text_ids, int n_text, int has_lang, int lang_id,
                   int has_spk_id, int spk_id, const float* spk_emb, const int32_t* instr_ids,
                   int n_instr, const float* mid, int n_mid, float* out, int max_rows);
int q3o_build_clone(const q3o_assets* a, const int32_t* text_ids, int n_text, const int32_t* ref_codes,
                    int n_ref_codes, const int32_t* ref_text_ids, int n_ref_text, const float* spk_emb,
                    int lang_id, const int32_t* instr_ids, int n_instr, float* out, int max_rows);

/* ---------------- chunker (engine.rs:495-543) ---------------- */
typedef struct q3o_chunker {
    int64_t buf[4096]; int len;
    int n_calls; int call_frames[1024]; int call_final[1024];  /* trace of decoder invocations */
} q3o_chunker;
/* push a message (codes,n,is_final); for each decoder invocation calls cb(user, codes, n_codes, is_final) */
typedef void (*q3o_decode_cb)(void* user, const int64_t* codes, int n_codes, int is_final);
int q3o_chunker_push(q3o_chunker* c, const int64_t* codes, int n, int is_final, q3o_decode_cb cb, void* user);

/* ---------------- codec decoder ("Q3TTS-codec-synth", Code2Wav analogue [EXT]) ---------------- */
typedef struct q3o_codec q3o_codec;
q3o_codec* q3o_codec_load(const char* path, char* err, size_t errlen);
void q3o_codec_free(q3o_codec* c);
void q3o_codec_reset(q3o_codec* c);
int q3o_codec_samples_per_frame(const q3o_codec* c);
void q3o_set_threads(int n); /* OpenMP team size for the calling thread's following parallel regions */
/* streaming decode of n_frames frames (codes [n_frames][16]); returns samples written */
int q3o_codec_decode(q3o_codec* c, const int64_t* codes, int n_frames, int is_last, float* pcm, int max_samples);

/* ---------------- mel (onnx.rs:167-320) ---------------- */
/* returns n_frames; mel_out must hold n_frames*128 floats (call with mel_out NULL to size) */
int q3o_mel(const float* audio, int n, float* mel_out);

/* ---------------- the loop (engine.rs:445-656) ---------------- */
typedef struct q3o_engine {
    q3o_assets* assets;
    q3o_model* talker;
    q3o_model* predictor;
    q3o_codec* codec;      /* may be NULL: codes only */
    int max_steps;
    float temperature; int top_k; float top_p; uint64_t seed;
    int mask_eos;          /* bench/test knob: EOS logit excluded so runs have fixed length */
    int n_threads;
    /* measurement hooks (tests only; NULL = off).  margins[2*f] = top-1 minus top-2 logit of frame f's code_0 (over [0,2160), EOS mask applied),
     * margins[2*f+1] = smallest such gap among the frame's 15 predictor codes.  forced[f*16+q] (teacher forcing) replaces the loop's own pick
     * AFTER it has been recorded in own_codes, so two arithmetic modes can be compared frame by frame along one trajectory. */
    float* margins; int margins_cap;
    const int32_t* forced; int forced_frames;
    int32_t* own_codes;
} q3o_engine;

q3o_engine* q3o_engine_create(const char* model_dir_quant, const char* codec_path, int n_threads, char* err, size_t errlen);
void q3o_engine_free(q3o_engine* e);
/* prompt [n_prompt][2048]; codes_out capacity max_steps*16; pcm_out may be NULL.
 * returns number of frames generated, <0 on error. n_pcm_out receives sample count. */
int q3o_engine_generate(q3o_engine* e, const float* prompt, int n_prompt, int32_t* codes_out,
                        float* pcm_out, int max_pcm, int* n_pcm_out);

#ifdef __cplusplus
}
#endif
#endif
