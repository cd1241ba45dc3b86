/*
 * q3tts.h -- C ABI of the MI355X-native Qwen3-TTS engine ("Boundary B").
 *
 * This is what the reference's `src/tts` layer binds instead of llama.cpp + ONNX Runtime:
 *   q3tts_engine_create      <- TtsEngine::new                (/root/reference/src/tts/engine.rs:84-169)
 *   q3tts_generate_batch     <- TtsEngine::run_inference_stream (engine.rs:445-656), batched over requests
 *   q3tts_prompt_*           <- PromptBuilder::{build_core, build_clone_prompt} (src/tts/prompt.rs:28-277)
 *   q3tts_sampler_*          <- LlamaSampler::{new, greedy, sample} (src/models/llama/mod.rs:627-776)
 *   q3tts_chunker_*          <- decoder-thread chunker (engine.rs:495-543)
 *   q3tts_decoder_*          <- AudioDecoder::{load, create_state, decode} (src/models/onnx.rs:324-496)
 *   q3tts_assets_*           <- Assets::{load, project, get_codec_embedding, get_text_embedding} (src/assets_manager.rs)
 *   q3tts_mel                <- SpeakerEncoder::compute_mel (src/models/onnx.rs:167-320)
 * "Boundary A", the 28 llama_* symbols the unmodified reference dlopens, is declared in q3tts_llama.h.
 *
 * Rules: plain pointers and sizes, caller-owned buffers, int status (0 = ok, like llama_decode), no
 * exceptions/aborts across the ABI; q3tts_last_error() returns the thread's last message.
 * The HIP path is the only path: with no usable GPU the compute entry points return an error.
 */
#ifndef Q3TTS_H
#define Q3TTS_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define Q3TTS_OK 0
#define Q3TTS_ERR -1

const char* q3tts_last_error(void);
int q3tts_version(void);
int q3tts_device_count(void); /* visible HIP devices (0 when none) */

/* ---- SamplerConfig (engine.rs:13-45) ---- */
typedef struct q3tts_sampler_config {
    float temperature; /* default 0.7, <= 0 = greedy */
    int32_t top_k;     /* default 40, 0 = disabled   */
    float top_p;       /* default 0.9, 1.0 = disabled */
    int32_t has_seed;  /* 0 => wall-clock nanoseconds (engine.rs:473-478) */
    uint64_t seed;
} q3tts_sampler_config;
void q3tts_sampler_config_default(q3tts_sampler_config* c);

/* ---- engine ---- */
typedef struct q3tts_engine q3tts_engine;
typedef struct q3tts_engine_params {
    const char* model_dir; /* holds gguf*, onnx/ (engine.rs:91-124) */
    const char* quant;     /* "q8_0" | "q5_k_m" | "none" ... */
    int32_t device;        /* HIP device ordinal */
    int32_t max_batch;     /* lock-stepped sequences per step */
    int32_t max_prompt;    /* <= 1024 (reference's effective cap, llama/mod.rs:567-581) */
    int32_t max_steps;     /* default 512 (engine.rs:152) */
    int32_t load_codec;    /* 0: codes only */
    int32_t use_graph;     /* 1: hipGraph frame replay */
} q3tts_engine_params;
void q3tts_engine_params_default(q3tts_engine_params* p);
int q3tts_engine_create(const q3tts_engine_params* p, q3tts_engine** out);
void q3tts_engine_destroy(q3tts_engine* e);

typedef struct q3tts_request {
    const float* prompt;   /* [n_prompt][2048] f32 embedding rows (PromptData.embd) */
    int32_t n_prompt;
    q3tts_sampler_config sampler;
    int32_t max_steps;
    int32_t mask_eos;      /* test/bench knob: never emit EOS (fixed-length runs) */
    int32_t* codes_out;    /* capacity max_steps*16 */
    float* pcm_out;        /* capacity pcm_capacity floats (may be NULL) */
    int64_t pcm_capacity;
    /* results */
    int32_t n_frames;
    int64_t n_pcm;
    double prefill_ms, first_chunk_ms, total_ms;
} q3tts_request;
/* runs all requests to completion in lock step (request-level batching) */
int q3tts_generate_batch(q3tts_engine* e, q3tts_request* reqs, int32_t n_reqs, int32_t want_pcm);

typedef struct q3tts_stats {
    double frame_loop_ms; int64_t frames; double prefill_ms;
    double gemv_ms; int64_t gemv_launches; double gemv_bytes; /* instrumented leg only */
    double codec_ms; int64_t codec_calls;
    double talker_weight_bytes, predictor_weight_bytes, kv_bytes_per_token;
    double gu_ms; int64_t gu_launches; double gu_bytes;       /* talker gate/up kernel alone (instrumented leg) */
    int64_t sched_steps; double slot_frames;                  /* scheduler: frame groups launched, sum of graph width x frames */
    int64_t graph_frames;                                     /* frame-graph replays (each advances every active slot by one frame) */
} q3tts_stats;
int q3tts_engine_stats(q3tts_engine* e, q3tts_stats* out);
void q3tts_engine_reset_stats(q3tts_engine* e);
void q3tts_engine_set_instrument(q3tts_engine* e, int32_t on);
double q3tts_engine_bytes_per_step(q3tts_engine* e, int32_t batch, double mean_ctx);

/* ---- continuous-batching scheduler (SURVEY 8b "q3tts_submit / q3tts_poll"; BASELINE config 3) ----
 * Requests queue up and are admitted into the engine's max_batch sequence slots as slots free up; every slot advances
 * one frame per graph replay; finished sequences retire on EOS (engine.rs:558-561) or max_steps.  Results are identical
 * to running each request alone (the arithmetic is batch-invariant).  Either call q3tts_sched_start once (background
 * driver thread) or drive it yourself with q3tts_sched_step / q3tts_wait.  q3tts_generate_batch is submit-all + wait-all. */
#define Q3TTS_REQ_QUEUED 0
#define Q3TTS_REQ_RUNNING 1
#define Q3TTS_REQ_DRAINING 2 /* all frames generated, codec still decoding the tail */
#define Q3TTS_REQ_DONE 3
#define Q3TTS_REQ_FAILED (-1)
typedef struct q3tts_req_status {
    int32_t state; int32_t n_frames; int64_t n_pcm; /* frames emitted / PCM samples decoded so far (streamable) */
    double queue_ms, prefill_ms, first_chunk_ms, total_ms;
} q3tts_req_status;
/* r->prompt/n_prompt/sampler/max_steps/mask_eos are read (the prompt is copied); output fields of r are ignored */
int q3tts_submit(q3tts_engine* e, const q3tts_request* r, int32_t want_pcm, int64_t* req_id);
int q3tts_poll(q3tts_engine* e, int64_t req_id, q3tts_req_status* out);
/* streaming read: frames [frame_off, frame_off+max_frames) and PCM [pcm_off, pcm_off+pcm_cap) available right now */
int q3tts_fetch(q3tts_engine* e, int64_t req_id, int32_t* codes_out, int32_t frame_off, int32_t max_frames, float* pcm_out,
                int64_t pcm_off, int64_t pcm_cap, int32_t* got_frames, int64_t* got_pcm);
int q3tts_wait(q3tts_engine* e, int64_t req_id, double timeout_ms /* <0: forever */); /* 0 done, 1 timeout, <0 error */
int q3tts_release(q3tts_engine* e, int64_t req_id);
int q3tts_sched_start(q3tts_engine* e);
int q3tts_sched_stop(q3tts_engine* e);
int q3tts_sched_step(q3tts_engine* e, int32_t* busy);
/* voices: the engine-side analogue of VoiceFile (utils/voice_file.rs:5-22).  ref_codes [n_ref_frames*16] / ref_text_ids may be
 * NULL for preset voices.  In multi-GPU serving the owning rank broadcasts these arrays first (q3tts.dist.broadcast_voice). */
int q3tts_voice_register(q3tts_engine* e, const float* spk_emb2048, const int32_t* ref_codes, int32_t n_ref_codes,
                         const int32_t* ref_text_ids, int32_t n_ref_text, int32_t* voice_id);
/* generate_with_voice_ids (engine.rs:390-435): builds the preset or clone prompt from the voice + token ids, then submits */
int q3tts_submit_text(q3tts_engine* e, int32_t voice_id, const int32_t* text_ids, int32_t n_text, int32_t lang_id,
                      const int32_t* instr_ids, int32_t n_instr, const q3tts_sampler_config* sampler, int32_t max_steps,
                      int32_t mask_eos, int32_t want_pcm, int64_t* req_id);

/* ---- multi-GPU: request sharding, weights replicated per device, no collective on the data path ----
 * q3tts_group_*: ONE process, N devices.  One engine (+ scheduler thread + decoder thread) per entry of device_ids; requests go round-robin
 * (request i -> engine i % n); group request ids carry the engine index.  q3tts_group_voice_register uploads the VoiceFile payload
 * (utils/voice_file.rs:5-22) to the first device and broadcasts it device-to-device -- ncclBroadcast over communicators from
 * ncclCommInitAll (RCCL over xGMI; librccl.so is bound at run time) or, when RCCL is unavailable or a device is listed twice,
 * hipMemcpyPeerAsync -- and every engine registers the voice from its own device's copy.  Register voices through the group only, so
 * that one voice id is valid on every engine.  p->device is ignored. */
typedef struct q3tts_group q3tts_group;
int q3tts_group_create(const q3tts_engine_params* p, const int32_t* device_ids, int32_t n_dev, q3tts_group** out);
void q3tts_group_destroy(q3tts_group* g);
int32_t q3tts_group_size(q3tts_group* g);
q3tts_engine* q3tts_group_engine(q3tts_group* g, int32_t i);   /* borrowed: stats, assets, per-engine calls */
int32_t q3tts_group_uses_rccl(q3tts_group* g);                 /* 1: registrations go through ncclBroadcast, 0: peer copies */
/* What the group's one exchange step actually runs on, so a multi-GPU harness can ASSERT "RCCL, N ranks" instead of assuming it: rccl_ranks = size of
 * the ncclCommInitAll communicator set (0 = peer copies: one device, a device listed twice, or librccl absent), and how many registrations took
 * which path so far.  The first registration also prints one line to stderr naming the path. */
typedef struct q3tts_group_info_t {
    int32_t n_devices, rccl_loaded, rccl_ranks, distinct_devices;
    int64_t registrations_rccl, registrations_peer_copy;
} q3tts_group_info_t;
int q3tts_group_info(q3tts_group* g, q3tts_group_info_t* out);
int q3tts_group_voice_register(q3tts_group* g, const float* spk_emb2048, const int32_t* ref_codes, int32_t n_ref_codes,
                               const int32_t* ref_text_ids, int32_t n_ref_text, int32_t* voice_id);
int q3tts_group_submit(q3tts_group* g, const q3tts_request* r, int32_t want_pcm, int64_t* req_id);
int q3tts_group_submit_text(q3tts_group* g, int32_t voice_id, const int32_t* text_ids, int32_t n_text, int32_t lang_id,
                            const int32_t* instr_ids, int32_t n_instr, const q3tts_sampler_config* sampler, int32_t max_steps,
                            int32_t mask_eos, int32_t want_pcm, int64_t* req_id);
int32_t q3tts_group_device_of(q3tts_group* g, int64_t req_id); /* HIP device a request was sharded to */
int q3tts_group_poll(q3tts_group* g, int64_t req_id, q3tts_req_status* out);
int q3tts_group_fetch(q3tts_group* g, int64_t req_id, int32_t* codes_out, int32_t frame_off, int32_t max_frames, float* pcm_out,
                      int64_t pcm_off, int64_t pcm_cap, int32_t* got_frames, int64_t* got_pcm);
int q3tts_group_wait(q3tts_group* g, int64_t req_id, double timeout_ms);
int q3tts_group_release(q3tts_group* g, int64_t req_id);
int q3tts_group_start(q3tts_group* g);                         /* q3tts_sched_start on every engine */
int q3tts_group_stop(q3tts_group* g);
/* q3tts_comm_*: ONE process per GPU (what `bench.py --gpus N` / torchrun-style launchers use).  Rank 0 makes the 128-byte id and ships it to
 * the other ranks by any means (environment, file, socket); every rank creates its communicator (ncclCommInitRank) and calls
 * q3tts_comm_voice_register collectively: the root passes the voice, the other ranks pass NULL / 0 and receive it; each registers it on
 * its own engine. */
typedef struct q3tts_comm q3tts_comm;
int q3tts_comm_available(void);                                /* 1 when librccl.so could be loaded */
int q3tts_comm_unique_id(uint8_t out_id[128]);
int q3tts_comm_create(const uint8_t id[128], int32_t rank, int32_t world, int32_t device, q3tts_comm** out);
void q3tts_comm_destroy(q3tts_comm* c);
int q3tts_comm_voice_register(q3tts_comm* c, q3tts_engine* e, int32_t root, const float* spk_emb2048, const int32_t* ref_codes,
                              int32_t n_ref_codes, const int32_t* ref_text_ids, int32_t n_ref_text, int32_t* voice_id);
int32_t q3tts_engine_device(q3tts_engine* e);

/* ---- assets + prompt builder (host) ---- */
typedef struct q3tts_assets q3tts_assets;
int q3tts_assets_open(const char* gguf_path, q3tts_assets** out);
void q3tts_assets_close(q3tts_assets* a);
const q3tts_assets* q3tts_engine_assets(q3tts_engine* e);
int q3tts_assets_codec_embedding(const q3tts_assets* a, int32_t q, int32_t code, float* out2048);
int q3tts_assets_text_embedding(const q3tts_assets* a, int64_t token, float* out2048);
int q3tts_assets_tts_pad(const q3tts_assets* a, float* out2048);
/* returns rows written (each 2048 f32) or <0; lang_id/spk_id < 0 mean None; pointers may be NULL */
int q3tts_prompt_build_core(const q3tts_assets* a, const int32_t* text_ids, int32_t n_text, int32_t lang_id, int32_t spk_id,
                            const float* spk_emb2048, const int32_t* instr_ids, int32_t n_instr, const float* mid_rows,
                            int32_t n_mid, float* out, int32_t max_rows);
int q3tts_prompt_build_clone(const q3tts_assets* a, const int32_t* text_ids, int32_t n_text, const int32_t* ref_codes,
                             int32_t n_ref_codes, const int32_t* ref_text_ids, int32_t n_ref_text, const float* spk_emb2048,
                             int32_t lang_id, const int32_t* instr_ids, int32_t n_instr, float* out, int32_t max_rows);

/* ---- sampler (host) ---- */
typedef struct q3tts_sampler q3tts_sampler;
q3tts_sampler* q3tts_sampler_new(float temperature, int32_t top_k, float top_p, uint64_t seed);
void q3tts_sampler_free(q3tts_sampler* s);
int32_t q3tts_sampler_sample(q3tts_sampler* s, const float* logits, int32_t n_vocab, int32_t start, int32_t end);

/* ---- chunker (host) ---- */
typedef void (*q3tts_decode_cb)(void* user, const int64_t* codes, int32_t n_codes, int32_t is_final);
typedef struct q3tts_chunker q3tts_chunker;
q3tts_chunker* q3tts_chunker_new(q3tts_decode_cb cb, void* user);
void q3tts_chunker_free(q3tts_chunker* c);
int q3tts_chunker_push(q3tts_chunker* c, const int64_t* codes, int32_t n, int32_t is_final);

/* ---- streaming codec decoder (device) ---- */
typedef struct q3tts_decoder q3tts_decoder;
int q3tts_decoder_create(const char* codec_gguf, int32_t n_streams, q3tts_decoder** out);
void q3tts_decoder_destroy(q3tts_decoder* d);
/* batched form: one pass decodes up to max_group streams that each contribute the same number of new frames */
int q3tts_decoder_create_ex(const char* codec_gguf, int32_t n_streams, int32_t max_frames_per_call, int32_t max_group, q3tts_decoder** out);
/* streams [G] distinct; codes [G][n_frames][16] i64; wav_out [G][n_frames*samples_per_frame] f32 */
int q3tts_decoder_decode_group(q3tts_decoder* d, int32_t G, const int32_t* streams, const int64_t* codes, int32_t n_frames, float* wav_out);
int q3tts_decoder_samples_per_frame(q3tts_decoder* d);
int q3tts_decoder_reset(q3tts_decoder* d, int32_t stream);
/* codes [n_frames][16] i64; writes up to n_frames*samples_per_frame floats; *valid_samples = prefix to keep */
int q3tts_decoder_decode(q3tts_decoder* d, int32_t stream, const int64_t* codes, int32_t n_frames, int32_t is_last,
                         float* final_wav, int64_t* valid_samples);

/* DecoderState (onnx.rs:461-496) export / import: one stream's streaming state as a flat f32 blob whose layout is a list of named tensors
 * (pre_conv_history, past_key_i / past_value_i, the conv_history pieces, counters) -- to checkpoint a stream, move it to another decoder
 * or device, or inspect it.  q3tts_decoder_state_entry(i) describes entry i and returns the entry count. */
int64_t q3tts_decoder_state_floats(q3tts_decoder* d);
/* n_floats = the caller's buffer length; it must equal q3tts_decoder_state_floats().  _import also validates the blob's trailer (cached-position
 * count within the attention window, frame counter finite and non-negative) before anything is committed, so a blob from another decoder
 * configuration or a corrupted one is an error, not an out-of-range device access on the next decode. */
int q3tts_decoder_state_export(q3tts_decoder* d, int32_t stream, float* out, int64_t n_floats);
int q3tts_decoder_state_import(q3tts_decoder* d, int32_t stream, const float* in, int64_t n_floats);
int32_t q3tts_decoder_state_entry(q3tts_decoder* d, int32_t i, const char** name, int64_t* offset, int32_t* rows, int32_t* cols);

/* ---- mel front end (device) ---- */
int q3tts_mel_frames(int32_t n_samples);
int q3tts_mel(const float* audio, int32_t n_samples, float* mel_out /* [frames][128] */);

/* ---- transformer contexts (shared with the llama shim; also the layer-level parity surface) ---- */
typedef struct q3tts_tf q3tts_tf;
int q3tts_tf_open(const char* gguf_path, int32_t n_ctx, int32_t max_tok, q3tts_tf** out);
void q3tts_tf_close(q3tts_tf* t);
int q3tts_tf_dims(q3tts_tf* t, int32_t* n_embd, int32_t* n_layer, int32_t* n_head, int32_t* n_vocab);
void q3tts_tf_clear(q3tts_tf* t);
/* ntok tokens of one sequence: x [ntok][n_embd]; pos [ntok][4]; appends to the KV cache.
 * hidden_out [ntok][n_embd] (final-norm) and logits_out [ntok][row1-row0] may be NULL. */
int q3tts_tf_eval(q3tts_tf* t, const float* x, const int32_t* pos4, int32_t ntok, float* hidden_out, float* logits_out,
                  int32_t row0, int32_t row1);

/* ---- tokenizer (host code) ----
 * Byte-level BPE over a HuggingFace tokenizer.json: what the reference's Tokenizer::{load, encode, decode} get from the `tokenizers` crate
 * (/root/reference/src/utils/tokenizer.rs:9-38): <model_dir>/tokenizer/tokenizer.json, encode(text, add_special_tokens = false),
 * decode(ids, skip_special_tokens = false).  Qwen2-family files: added tokens, Split(Qwen2 pattern) + ByteLevel, BPE merges. */
typedef struct q3tts_tokenizer q3tts_tokenizer;
int q3tts_tokenizer_open(const char* tokenizer_json, q3tts_tokenizer** out);
void q3tts_tokenizer_close(q3tts_tokenizer* t);
int32_t q3tts_tokenizer_encode(q3tts_tokenizer* t, const char* text_utf8, int32_t* ids, int32_t cap); /* count (may exceed cap), < 0 error */
int64_t q3tts_tokenizer_decode(q3tts_tokenizer* t, const int32_t* ids, int32_t n, char* buf, int64_t cap); /* byte length, < 0 error */
int32_t q3tts_tokenizer_vocab_size(q3tts_tokenizer* t);
int64_t q3tts_text_nfc(const char* utf8, char* out, int64_t cap); /* NFC as the tokenizer's normaliser applies it; returns the size needed (with NUL) */

/* ---- ONNX graph ingestion (host code, no GPU needed) ----
 * A minimal reader of ONNX ModelProto files (protobuf wire format walked by hand): what `ort::Session` parses for the reference's
 * qwen3_tts_decoder.onnx / codec_encoder.onnx / speaker_encoder.onnx (/root/reference/src/models/onnx.rs:97-163, 324-347).  It exposes
 * the node list with attributes, the initialisers (weights, zero-copy into the mapped file), graph inputs / outputs, the op -> HIP
 * kernel table of this engine, and a check of the streaming-decoder I/O contract (onnx.rs:355-455).  tools/q3onnx_dump prints all of it. */
typedef struct q3tts_onnx q3tts_onnx;
int q3tts_onnx_open(const char* path, q3tts_onnx** out);
void q3tts_onnx_close(q3tts_onnx* m);
int q3tts_onnx_counts(q3tts_onnx* m, int32_t* n_nodes, int32_t* n_initializers, int32_t* n_inputs, int32_t* n_outputs);
int64_t q3tts_onnx_summary(q3tts_onnx* m, char* buf, int64_t cap); /* returns the size needed (with NUL) */
int q3tts_onnx_node(q3tts_onnx* m, int32_t i, const char** op_type, const char** name, int32_t* n_in, int32_t* n_out, int32_t* n_attr);
const char* q3tts_onnx_node_input(q3tts_onnx* m, int32_t i, int32_t j);
const char* q3tts_onnx_node_output(q3tts_onnx* m, int32_t i, int32_t j);
int32_t q3tts_onnx_node_attr_ints(q3tts_onnx* m, int32_t i, const char* attr, int64_t* out, int32_t cap); /* count, -1 = absent */
int32_t q3tts_onnx_node_attr_float(q3tts_onnx* m, int32_t i, const char* attr, float* out);
int q3tts_onnx_initializer(q3tts_onnx* m, int32_t i, const char** name, int32_t* dtype, int64_t* dims8, int32_t* ndims, const void** data,
                           int64_t* nbytes);
const char* q3tts_onnx_op_kernel(const char* op_type);            /* HIP kernel serving the op, NULL = none yet */
int q3tts_onnx_decoder_contract(q3tts_onnx* m, char* buf, int64_t cap); /* 0 satisfied, 1 missing (names in buf) */

/* ---- ONNX graph execution on the GPU (second half; row a17) ----
 * What `ort::Session::run` is to the reference's encoders (/root/reference/src/models/onnx.rs:97-121 `AudioEncoder::encode`:
 * "input_values" [1, T] f32 -> "audio_codes" [1, F, 16] i64; :140-163 `SpeakerEncoder::encode`: "mels" [1, n, 128] f32 -> "spk_emb" [1, 2048]):
 * a general interpreter of the operator set such exports use, one HIP launch per node, float tensors resident in HBM, shape arithmetic on the
 * host.  dtype codes are ONNX's (1 = f32, 7 = i64, 9 = bool, fetched as f32 0 / 1).  q3tts_onnx_session_unsupported lists op types of the
 * graph without a kernel (0 = the graph is executable). */
typedef struct q3tts_onnx_session q3tts_onnx_session;
int q3tts_onnx_session_open(const char* path, int32_t device, q3tts_onnx_session** out);
void q3tts_onnx_session_close(q3tts_onnx_session* s);
int32_t q3tts_onnx_session_unsupported(q3tts_onnx_session* s, char* buf, int64_t cap); /* count; names comma-separated in buf */
int q3tts_onnx_session_set_input(q3tts_onnx_session* s, const char* name, int32_t dtype, const void* data, const int64_t* shape, int32_t rank);
int q3tts_onnx_session_run(q3tts_onnx_session* s);
int q3tts_onnx_session_output_info(q3tts_onnx_session* s, const char* name, int32_t* dtype, int32_t* rank, int64_t* shape8);
int q3tts_onnx_session_output(q3tts_onnx_session* s, const char* name, void* dst, int64_t cap_bytes); /* f32 or i64 elements */
int64_t q3tts_onnx_session_launches(q3tts_onnx_session* s);                              /* kernel launches of all runs so far */
int q3tts_onnx_op_executable(const char* op_type);                                       /* 1 when the executor runs the op */
/* AudioDecoder over the executor (/root/reference/src/models/onnx.rs:322-458): the exported streaming decoder graph with its state
 * (pre_conv_history, latent_buffer, conv_history, past_key_i / past_value_i) carried on the device between chunks.  _decode returns the first
 * `valid_samples` samples of `final_wav`; *n_out receives their count.  The chunk's PCM stays in the handle until the next _decode / _reset:
 * when pcm is NULL or cap is too small the call returns 2 (state advanced, nothing lost) and q3tts_onnx_decoder_fetch copies the same chunk
 * out again into a buffer of at least *n_out samples. */
typedef struct q3tts_onnx_decoder q3tts_onnx_decoder;
int q3tts_onnx_decoder_open(const char* path, int32_t device, q3tts_onnx_decoder** out);
void q3tts_onnx_decoder_close(q3tts_onnx_decoder* d);
int q3tts_onnx_decoder_reset(q3tts_onnx_decoder* d);
int q3tts_onnx_decoder_decode(q3tts_onnx_decoder* d, const int64_t* codes, int32_t n_frames, int32_t is_final, float* pcm, int64_t cap, int64_t* n_out);
int q3tts_onnx_decoder_fetch(q3tts_onnx_decoder* d, float* pcm, int64_t cap, int64_t* n_out);

/* ---- kernel-level entry points used by the parity tests (host buffers in/out) ---- */
int q3tts_op_gemv_q8(const void* w_q8_0 /* GGUF Q8_0 rows [n][k/32][34 B] */, int32_t n, int32_t k, const int8_t* xq,
                     const uint16_t* xd, int32_t ntok, float* y /* [ntok][n] */, int32_t lpr);
/* batched layer path (>= 16 tokens): fused gate/up GEMM on the matrix cores + SwiGLU + int8 quantisation; w = [2*ff][k] Q8_0 rows
 * (gate rows first), k = 1024 or 2048; aq [ntok][ff], ad [ntok][ff/32] */
/* K-quant forms of the two entries above: the matrix is built from up to 3 tensors (raw GGUF rows of ggml type 8 = Q8_0, 13 = Q5_K, 14 = Q6_K; row counts
 * multiples of 32) into packed planes exactly as the engine does at load; >= 16 tokens with lpr = 0 take the matrix-core kernel (k_gemm_kq_mfma). */
int q3tts_op_gemv_kq(const void* const* raws, const int32_t* types, const int32_t* rows, int32_t nparts, int32_t k, const int8_t* xq, const uint16_t* xd,
                     int32_t ntok, float* y, int32_t lpr);
int q3tts_op_gateup_kq(const void* gate_raw, const void* up_raw, int32_t type, int32_t ff, int32_t k, const int8_t* xq, const uint16_t* xd, int32_t ntok,
                       int8_t* aq, uint16_t* ad);
int q3tts_op_gateup_q8(const void* w_q8_0, int32_t ff, int32_t k, const int8_t* xq, const uint16_t* xd, int32_t ntok, int8_t* aq,
                       uint16_t* ad);
/* float-weight matmul (spec S3 float form; ggml types 0 = f32, 1 = f16, 30 = bf16): ntok >= 12 runs the matrix-core kernel,
 * fewer tokens the one-wave-per-row GEMV; row0 selects a row range like the head of the code predictor does */
int q3tts_op_matmul_float(const void* w /* [n][k] */, int32_t type, int32_t n, int32_t k, int32_t row0, int32_t nrows,
                          const float* x /* [ntok][k] */, int32_t ntok, float* y /* [ntok][nrows] */);
int q3tts_op_rmsnorm_quant(const float* x, const float* g, int32_t d, int32_t ntok, float eps, int8_t* xq, uint16_t* xd,
                           float* xn);
int q3tts_op_swiglu_quant(const float* gu, int32_t ff, int32_t ntok, int8_t* aq, uint16_t* ad);
int q3tts_op_argmax(const float* logits, int32_t n, int32_t start, int32_t end, int32_t mask_idx, int32_t* out);
/* device sampler (llama/mod.rs:666-776): n_draws consecutive draws of one seeded sequence over logits[0,n), n <= 4096 */
int q3tts_op_sample(const float* logits, int32_t n, float temperature, int32_t top_k, float top_p, uint64_t seed, int32_t mask_idx,
                    int32_t n_draws, int32_t* out);
int q3tts_op_project(const float* x, const float* w /* [n_out][n_in] */, const float* b, int32_t n_in, int32_t n_out, float* y);

#ifdef __cplusplus
}
#endif
#endif
