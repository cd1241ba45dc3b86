// kernels.h -- launch API of the hand-written gfx950 kernels for the AR decode path.
// Every kernel implements the arithmetic of include/q3tts_spec.h exactly (bit-identical to oracle/).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace q3 {

// Q8_0 matrix repacked for coalesced 1-KiB wave loads (see weights.cpp / DESIGN.md "HBM layout"):
//   qs : [N/32][K/32][2 halves][32 rows][16 B]   int8 quants, one 1-KiB tile per (row-group, block)
//   sc : [N/32][K/256][32 rows][8]               f16 block scales, one 16-B vector per (row, segment)
struct Q8Mat {
    const uint8_t* qs = nullptr;
    const uint16_t* sc = nullptr;
    int N = 0;      // logical rows
    int Npad = 0;   // rows padded to 32
    int K = 0;
    // K-quant rows (Q5_K / Q6_K) stay PACKED in HBM and are unpacked in registers in front of the int8 dot products.  A matrix with K-quant
    // row groups carries the ggml type and the byte offset (in 16-B units) of every 32-row group; per (row group, 256-segment):
    //   nibble plane   [8 blocks][2 halves][32 rows][8 B]   low 4 bits; byte b of word w holds weights 8w+b (low nibble) and 8w+4+b (high nibble) of the half
    //   high plane     Q5_K: [32 rows][8 blocks][4 B]        bit 8b+J of the word = bit 4 of weight 4J+b of the block
    //                  Q6_K: [32 rows][8 blocks][2][4 B]     the same for bit 4 (word 0) and bit 5 (word 1); values are stored +32 (unsigned 6 bit)
    // = 0.625 / 0.75 B per weight (+ 0.125 of scales and metadata), against 1.125 for the int8 planes round 1 expanded them to.
    // Metadata: one 16-B vector per (row, segment): Q5_K = {sc[8], m[8]} u8 with d,dmin in sc[0..1]; Q6_K = 16 i8 sub-block scales with d in sc[0].
    const uint8_t* rg_type = nullptr;   // [Npad/32], null => all Q8_0 (uniform 1-KiB tiles)
    const uint32_t* rg_off = nullptr;   // [Npad/32] offset of the row group's quants from qs in 16-B units (K-quant matrices only)
    const uint8_t* meta = nullptr;      // [N/32][K/256][32 rows][16 B]
    size_t qbytes = 0;                  // bytes of the quant planes as stored (0 => Npad*K)
    // The same map by tensor, in the kernel arguments (a fused matrix is at most 3 tensors: q, k, v): row groups [part_rg0[i], part_rg0[i+1])
    // have type part_type[i] and start at qs + 16 * part_off[i].  Kernels take type and address from here -- scalar loads of the argument
    // block -- because reading rg_type / rg_off first would put a dependent global load (~1.5 us) in front of the whole weight stream.
    // (plain fields, not arrays: selecting from an argument array makes the compiler copy the arguments to scratch and index them there)
    int nparts = 0; int p1_rg0 = 0x7fffffff, p2_rg0 = 0x7fffffff; int p0_type = 0, p1_type = 0, p2_type = 0; uint32_t p0_off = 0, p1_off = 0, p2_off = 0;
    size_t bytes() const { return (qbytes ? qbytes : (size_t)Npad * K) + (size_t)Npad * (K / 32) * 2 + (meta ? (size_t)Npad * (K / 256) * 16 : 0); }
};
// bytes of one 32-row group of K columns in HBM, by ggml type (8 = Q8_0, 13 = Q5_K, 14 = Q6_K)
inline size_t q3_rowgroup_bytes(int type, int K) { return type == 13 ? (size_t)(K / 256) * 5120 : type == 14 ? (size_t)(K / 256) * 6144 : (size_t)(K / 32) * 1024; }

// Per-token routing for batched steps (one entry per token row of the activation matrix)
struct TokMeta {
    const int32_t* seq;    // [ntok] sequence slot (page-table row)
    const int32_t* slot;   // [ntok] cache position this token is written to (= #cached before it)
    const int32_t* pos;    // [ntok][4] M-RoPE position streams (engine.rs:306-314)
    // Uniform form (the code predictor's passes: token t is sequence t, every token sits at the same position p in all four streams,
    // engine.rs:316-318): a value >= 0 replaces the three arrays, which removes two dependent global loads from the attention kernels.
    int uniform_pos = -1;
    __device__ __forceinline__ int seq_of(int tok) const { return uniform_pos >= 0 ? tok : seq[tok]; }
    __device__ __forceinline__ int slot_of(int tok) const { return uniform_pos >= 0 ? uniform_pos : slot[tok]; }
    __device__ __forceinline__ int pos_of(int tok, int stream) const { return uniform_pos >= 0 ? uniform_pos : pos[(size_t)tok * 4 + stream]; }
};

struct KvCache {           // paged f16 KV cache (pages of 64 positions)
    uint16_t* k;           // [page][layer][kvh][16 dchunks][64 pos][8]
    uint16_t* v;           // [page][layer][kvh][64 pos][128]
    const int32_t* page_table; // [n_seq][max_pages]; null = identity (sequence s owns page s, single-page contexts)
    int max_pages;
    int n_layer, n_kv;
    __host__ __device__ size_t page_stride() const { return (size_t)n_layer * n_kv * 8192; } // halfs per page (per K or V)
    __host__ __device__ size_t layer_stride() const { return (size_t)n_kv * 8192; }
    __device__ __forceinline__ int page_of(int seq, int logical) const { return page_table ? page_table[(size_t)seq * max_pages + logical] : seq; }
};

// out[(sseg*ntok + tok)*out_stride + r] = super-segment partial of row (row0+r) . x[tok]      (spec S3)
// batched steps: gate/up GEMM on the matrix cores with the SwiGLU + quantisation epilogue; false = shape not supported (caller falls back)
bool launch_gateup_mfma(hipStream_t st, const Q8Mat& wgu, int ff, const int8_t* xq, const uint16_t* xd, int8_t* aq, uint16_t* ad, int ntok);
void launch_gemv_q8(hipStream_t st, const Q8Mat& w, int row0, int nrows, const int8_t* xq, const uint16_t* xd,
                    float* out, int out_stride, int ntok, int lpr_hint = 0);

// h = h_in (+ sum of nparts partial slabs, in order); optional store h_out; xn = rmsnorm(h)*g (spec S4);
// quantise to int8 blocks (spec S2).  h_in may be indirect: row idx[tok] of a table (idx != nullptr).
struct NormArgs {
    const float* h_in; int h_stride;            // [ntok][h_stride]
    const int32_t* idx; int idx_stride;         // optional: h row = h_in + idx[tok*idx_stride]*h_stride
    const unsigned long long* idx_keys;         // optional: same, row selected by an argmax key (low word = ~index)
    const float* parts; int nparts; int parts_stride; // [p][ntok][parts_stride]
    float* h_out;                               // optional [ntok][d]
    const float* g; float eps; int d;
    int8_t* xq; uint16_t* xd;                   // [ntok][d], [ntok][d/32]
    float* xn_out;                              // optional [ntok][d] (final-norm hidden)
};
void launch_rmsnorm_quant(hipStream_t st, const NormArgs& a, int ntok);

// per-head q/k RMSNorm + NeoX M-RoPE; q -> qrot f32; k,v -> f16 paged cache (spec S4',S5,S6)
void launch_qk_rope_append(hipStream_t st, const float* qkv, int qkv_stride, const float* parts_unused, int n_head, int n_kv,
                           const float* q_norm_w, const float* k_norm_w, float eps, const float* rope_cos,
                           const float* rope_sin, int n_ctx, const int32_t* mrope_sec /*device [4]*/, const TokMeta& tm,
                           const KvCache& kv, int layer, float* qrot, int ntok);

// causal GQA attention over the paged cache (spec S7) + int8 quantisation of the output
void launch_attention(hipStream_t st, const float* qrot, int n_head, int n_kv, const TokMeta& tm, const KvCache& kv,
                      int layer, float* att /*optional [ntok][n_head*128]*/, int8_t* aq, uint16_t* ad, int ntok);

// a = silu(g)*u over gate/up slabs, quantise (spec S8,S2). gu layout: [ntok][2*ff] with gate first.
void launch_swiglu_quant(hipStream_t st, const float* gu, int ff, int8_t* aq, uint16_t* ad, int ntok);

// first-max argmax over logits[tok][start..end) with strict '>' (llama/mod.rs:690-701); mask index -> -inf
void launch_argmax(hipStream_t st, const float* logits, int stride, int start, int end, const int32_t* mask_per_tok,
                   int32_t* out, int out_stride, int add, int ntok);

// assets_manager.rs:383-399 : out[o] = b[o] + sum_i x[i]*Wt[i][o]  (mul then add, ascending i). Wt is [n_in][n_out].
void launch_project(hipStream_t st, const float* x, int x_stride, const float* Wt, const float* b, int n_in, int n_out,
                    float* out, int out_stride, int ntok);
// batched form over table rows, used once at load to pre-project the codec tables (same arithmetic)
void launch_project_table(hipStream_t st, const float* table, int64_t rows, const float* Wt, const float* b, int n_in,
                          int n_out, float* out);

void launch_argmax_keys(hipStream_t st, const float* logits, int stride, int n, const int32_t* mask_per_tok, unsigned long long* keys,
                        int key_stride, int ntok);


void launch_copy_f32(hipStream_t st, const float* src, float* dst, size_t n);
// per-device kernel attribute opt-ins (dynamic LDS above 64 KiB); call once per device outside any stream capture
void init_kernel_attributes();

// ---- 16/32-bit float weights (bf16 / f16 / f32 files): f32 activations, no activation quantisation (spec S3 float form:
// 8-element fma sub-chains, block = (c0+c1)+(c2+c3), blocks added in order inside a segment, segments / super-segments in order)
// w: row-major [N][K]; wt: the same elements tiled [ceil(N/64)][K/8][64][8] for k_gemm_float_mfma (rows past N are zero)
struct FMat { const void* w = nullptr; const void* wt = nullptr; int type = 0; int N = 0, K = 0; size_t bytes() const { return (size_t)N * K * (type == 0 ? 4 : 2); } };
void launch_gemv_float(hipStream_t st, const FMat& w, int row0, int nrows, const float* x, int x_stride, float* out, int out_stride, int ntok);
bool launch_gateup_float(hipStream_t st, const FMat& wgu, int ff, const float* x, int x_stride, float* act, int ntok);
void launch_tile_float(hipStream_t st, const void* w, void* wt, int type, int N, int K);
void launch_swiglu_f32(hipStream_t st, const float* gu, int ff, float* out, int ntok);

// ------------------------------- fused decode-step kernels (kernels_fused.hip) -------------------------------
typedef unsigned long long q3_u64;
// prologue of the fused GEMVs: h = h_in (+ parts); RMSNorm; quantise (spec S9,S4,S2)
struct NormPro {
    const float* h_in; int h_stride;
    const q3_u64* idx_keys; int idx_stride;     // optional: row = h_in + code(idx_keys[tok*idx_stride]) * h_stride
    const float* parts; int nparts; int parts_stride; size_t parts_slab; // slab p at parts + p*parts_slab, row tok at + tok*parts_stride
    float* h_out;                               // optional [ntok][d]; MUST NOT alias h_in (other workgroups still read it)
    const float* g; float eps;
    float* xn_out;                              // optional [ntok][d]
};
// epilogue of the head GEMV: first-max argmax through one 64-bit atomicMax per workgroup and token
struct ArgmaxEpi {
    q3_u64* keys; int key_stride;               // keys[tok*key_stride], pre-armed with pack_key(-inf, 0)
    const int32_t* mask_per_tok;                // optional: index excluded from the argmax (EOS masking)
    int idx_add;                                // added to the row index (0: slice-relative codes)
};
void launch_gemv_q8_norm(hipStream_t st, const Q8Mat& w, int row0, int nrows, const NormPro& a, float* out, int out_stride,
                         int ntok, const ArgmaxEpi* am);
void launch_rmsnorm_quant_wg(hipStream_t st, const NormPro& a, int d, int8_t* xq, uint16_t* xd, int ntok);
void launch_gateup_swiglu(hipStream_t st, const Q8Mat& wgu, int ff, const NormPro& a, int8_t* aq, uint16_t* ad, int ntok);
void launch_attention_fused(hipStream_t st, const float* qkv, int qkv_stride, int n_head, int n_kv, const float* q_norm_w,
                            const float* k_norm_w, float eps, const float* rope_cos, const float* rope_sin, int n_ctx,
                            const int32_t* mrope_sec, const TokMeta& tm, const KvCache& kv, int layer, int8_t* aq, uint16_t* ad,
                            int ntok);
void launch_attention_short(hipStream_t st, const float* qkv, int qkv_stride, int n_head, int n_kv, const float* q_norm_w,
                            const float* k_norm_w, float eps, const float* rope_cos, const float* rope_sin, int n_ctx,
                            const int32_t* mrope_sec, const TokMeta& tm, const KvCache& kv, int layer, int8_t* aq, uint16_t* ad, int ntok,
                            float* att = nullptr /*optional f32 rows [ntok][n_head*128]*/);
void launch_project_blk(hipStream_t st, const float* x, int x_stride, const float* Wblk /*[n_out/16][n_in][16]*/, const float* b,
                        int n_in, int n_out, float* out, int out_stride, int ntok);
void launch_feedback_keys(hipStream_t st, const float* const* tables, const int64_t* table_rows, const q3_u64* keys, int key_stride,
                          const float* tts_pad, float* out, int ntok);
void launch_gather_rows_keys(hipStream_t st, const float* table, int64_t rows, const q3_u64* keys, int key_stride, int row_len,
                             float* dst, int ntok);
struct AdvanceKeysArgs {
    int B; int32_t *finished, *n_frames; const int32_t* max_frames; q3_u64* keys; q3_u64* next_key0; int32_t* hist; int hist_stride;
    int32_t *t_slot, *t_pos;
};
void launch_advance_keys(hipStream_t st, const AdvanceKeysArgs& a);

// on-device LlamaSampler::sample (sampler.hip): one workgroup per sequence over logits[b*stride + [0,n)), n <= 4096
struct SampleArgs {
    const float* logits; int stride; int n;
    const float* temperature; const int32_t* top_k; const float* top_p; // per sequence (SamplerConfig, engine.rs:13-45)
    const int32_t* mask_idx;                                            // optional per sequence: index forced to -inf
    const uint32_t* rng_key; uint32_t* draws;                           // [B][8] ChaCha12 key, [B] u32 words consumed so far
    q3_u64* out_key; int out_stride;                                    // receives pack_key(0, token)
};
void launch_sample(hipStream_t st, const SampleArgs& a, int B);

} // namespace q3
