/*
 * q3tts_llama.h -- "Boundary A": the llama.cpp-named C symbols the UNMODIFIED reference dlopens from
 * <cwd>/runtime/libllama.so (/root/reference/src/models/llama/mod.rs:148-316; every symbol is .expect()-resolved
 * at :241-292, so all 28 must exist).  libq3tts.so exports them; build.sh also installs it as runtime/libllama.so.
 *
 * Struct layouts mirror the #[repr(C)] definitions at llama/mod.rs:7-74 (x86-64 SysV: llama_model_params 72 B,
 * llama_context_params 136 B, llama_batch 56 B; checked by tests/test_abi.py through q3tts_llama_abi_sizes).
 *
 * Semantics honoured: models are loaded from this repo's GGUF layout; llama_decode consumes
 * batch.embd ([n][n_embd] f32) and batch.pos (4 stream-major position arrays of n ints for the M-RoPE talker, n ints
 * otherwise, llama/mod.rs:556-581 + engine.rs:306-318); outputs stay valid until the next llama_decode:
 * llama_get_logits -> [n_out][n_vocab], llama_get_embeddings -> [n_out][n_embd] final-norm hidden, with n_out = all n
 * tokens when params.embeddings is set, else the rows flagged in batch.logits.
 */
#ifndef Q3TTS_LLAMA_H
#define Q3TTS_LLAMA_H
#include <stdint.h>
#include <stddef.h>
#include <stdbool.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct llama_model_params { /* llama/mod.rs:7-27 */
    void* devices; void* tensor_buft_overrides; int32_t n_gpu_layers; int32_t split_mode; int32_t main_gpu; float* tensor_split;
    void* progress_callback; void* progress_callback_user_data; void* kv_overrides;
    bool vocab_only, use_mmap, use_direct_io, use_mlock, check_tensors, use_extra_bufts, no_host, no_alloc;
} llama_model_params;

typedef struct llama_context_params { /* llama/mod.rs:28-63 */
    uint32_t n_ctx, n_batch, n_ubatch, n_seq_max; int32_t n_threads, n_threads_batch, rope_scaling_type, pooling_type, attention_type,
        flash_attn_type; float rope_freq_base, rope_freq_scale, yarn_ext_factor, yarn_attn_factor, yarn_beta_fast, yarn_beta_slow;
    uint32_t yarn_orig_ctx; float defrag_thold; void* cb_eval; void* cb_eval_user_data; int32_t type_k, type_v; void* abort_callback;
    void* abort_callback_data; bool embeddings, offload_kqv, no_perf, op_offload, swa_full, kv_unified; void* samplers; size_t n_samplers;
} llama_context_params;

typedef struct llama_batch { /* llama/mod.rs:64-74 */
    int32_t n_tokens; int32_t* token; float* embd; int32_t* pos; int32_t* n_seq_id; int32_t** seq_id; int8_t* logits;
} llama_batch;

typedef int32_t llama_token;
struct llama_model; struct llama_context; struct llama_vocab; struct llama_sampler; struct llama_memory_i;

void llama_backend_init(void);                                                   /* mod.rs:241,303 */
void llama_backend_free(void);                                                   /* mod.rs:242,318-324 */
llama_model_params llama_model_default_params(void);                             /* mod.rs:243,340 */
struct llama_model* llama_model_load_from_file(const char* path, llama_model_params p); /* mod.rs:246,344 */
void llama_model_free(struct llama_model* m);                                    /* mod.rs:249,372 */
const struct llama_vocab* llama_model_get_vocab(const struct llama_model* m);    /* mod.rs:250,348 */
int32_t llama_model_n_embd(const struct llama_model* m);                         /* mod.rs:253,350 */
int32_t llama_model_n_head(const struct llama_model* m);                         /* mod.rs:254,351 */
int32_t llama_model_n_layer(const struct llama_model* m);                        /* mod.rs:255,352 */
uint32_t llama_n_ctx(const struct llama_context* c);                             /* mod.rs:258 (bound, never called) */
int32_t llama_n_vocab(const struct llama_vocab* v);                              /* mod.rs:259 (bound, never called) */
int32_t llama_vocab_n_tokens(const struct llama_vocab* v);                       /* mod.rs:260,349 */
llama_token llama_vocab_eos(const struct llama_vocab* v);                        /* mod.rs:263,353 */
llama_context_params llama_context_default_params(void);                         /* mod.rs:264,409 */
struct llama_context* llama_init_from_model(struct llama_model* m, llama_context_params p); /* mod.rs:267,432 */
void llama_free(struct llama_context* c);                                        /* mod.rs:270,508 */
llama_batch llama_batch_init(int32_t n_tokens, int32_t embd, int32_t n_seq_max); /* mod.rs:271,536 */
void llama_batch_free(llama_batch b);                                            /* mod.rs:272 (bound, never called) */
int32_t llama_decode(struct llama_context* c, llama_batch b);                    /* mod.rs:273,445 */
float* llama_get_embeddings(struct llama_context* c);                            /* mod.rs:274,455,462 */
float* llama_get_logits(struct llama_context* c);                                /* mod.rs:277,469,474,682 */
struct llama_memory_i* llama_get_memory(struct llama_context* c);                /* mod.rs:278,481 */
void llama_memory_clear(struct llama_memory_i* mem, bool data);                  /* mod.rs:279 (bound, never called) */
bool llama_memory_seq_rm(struct llama_memory_i* mem, int32_t seq, int32_t p0, int32_t p1); /* mod.rs:280,482 */
int32_t llama_memory_seq_pos_max(struct llama_memory_i* mem, int32_t seq);       /* mod.rs:283,494 */
struct llama_sampler* llama_sampler_init_temp(float t);                          /* mod.rs:286 (bound, never called) */
llama_token llama_sampler_sample(struct llama_sampler* s, struct llama_context* c, int32_t idx); /* mod.rs:289 (never called) */
void llama_sampler_free(struct llama_sampler* s);                                /* mod.rs:292 (bound, never called) */
void ggml_backend_load_all(void);                                                /* mod.rs:222-231,302 (optional) */

/* test hook: sizeof / key offsets of the three by-value structs */
void q3tts_llama_abi_sizes(int32_t* out /* [8] */);

#ifdef __cplusplus
}
#endif
#endif
