// transformer.h -- device-resident Qwen3-style decoder (talker / code predictor) driven from embeddings.
// Replaces the llama.cpp objects behind LlamaModel / LlamaContext (/root/reference/src/models/llama/mod.rs:326-513).
#pragma once
#include "q3_common.h"
#include "gguf.h"
#include "kernels.h"
#include "ggml_mode.h"
#include <map>
#include <memory>

namespace q3 {

struct TfHparams {
    std::string arch;
    int n_embd = 0, n_layer = 0, n_head = 0, n_kv = 0, n_ff = 0, n_vocab = 0;
    float eps = 1e-6f, rope_base = 1e6f;
    int32_t mrope_sec[4] = {0, 0, 0, 0};
};

class Transformer {
public:
    Transformer(const std::string& gguf_path, int n_ctx, int max_tok);
    // A second execution context over the SAME device weights (read-only: repacked matrices, norms, RoPE tables are shared through a
    // reference-counted store); only the activation workspace is new.  Used for the asynchronous prefill lane: no second 1.5 GB copy.
    Transformer(const Transformer& weights_of, int max_tok);
    const TfHparams& hp() const { return hp_; }
    int n_ctx() const { return n_ctx_; }
    int max_tok() const { return max_tok_; }
    size_t weight_bytes() const { return weight_bytes_; }       // repacked matmul weights (algorithmic bytes/step)
    size_t layer_weight_bytes() const { return layer_weight_bytes_; }
    size_t head_bytes_per_row() const { return (size_t)hp_.n_embd + (size_t)(hp_.n_embd / 32) * 2; }

    struct Input {
        const float* x = nullptr; int x_stride = 0;   // [ntok][x_stride] embeddings ...
        const int32_t* idx = nullptr; int idx_stride = 0; // ... or table rows: x + idx[tok*idx_stride]*x_stride (unfused path)
        const q3_u64* idx_keys = nullptr;             // ... or table rows selected by argmax keys (fused path)
    };
    // fused = 5 launches/layer (decode steps: every token from a different sequence, or a single sequence's
    // tokens when `same_seq_tokens` is set, in which case attention stays unfused)
    bool fused = true;
    // Runs all layers for ntok tokens.  After return (on stream): xq_/xd_ hold the quantised final-norm hidden of
    // every token; hidden_out (optional) gets the f32 final-norm hidden [ntok][n_embd].
    void forward(hipStream_t st, const Input& in, int ntok, const TokMeta& tm, const KvCache& kv, float* hidden_out);
    // logits[tok_count][nrows] = output rows [row0,row0+nrows) . final hidden of tokens [tok0, tok0+tok_count)
    void head(hipStream_t st, int tok0, int tok_count, int row0, int nrows, float* logits, int logits_stride,
              const ArgmaxEpi* am = nullptr, int nrows_valid = -1, float* hidden_out = nullptr);
    void set_same_seq_tokens(bool v) { same_seq_ = v; }
    // every cached context of this model stays <= 64 positions (the code predictor): fold attention into the o-proj launch
    // sequences never exceed 64 positions (one KV page): steps of at least `min_tok` tokens use the single-wave attention kernel
    void set_short_attention(int min_tok) { short_attn_min_ = min_tok; }
    // largest token count that still takes the 5-launch fused layer path (its GEMVs re-read the weights once per 4-8 token tile,
    // which is fine for a model whose weights stay cache-resident, i.e. the code predictor)
    void set_fused_max_tokens(int n) { fused_max_tok_ = n; }

    // Q3_SPEC=ggml at construction: the opt-in "ggml-CPU" arithmetic mode (ggml_mode.hip) -- every forward() / head() then runs llama.cpp's portable
    // arithmetic (as the test suite's CPU restatement of the mode has it), on raw copies of the GGUF matrices; slow by design, bit-exact with that restatement
    bool ggml_mode() const { return ggml_mode_; }

    LaunchTimer* timer = nullptr;    // optional per-GEMV-launch event timing (instrumented bench leg): whole GEMV family
    LaunchTimer* timer_gu = nullptr; // ... and the gate/up kernel alone (the dominant launch by bytes)
    // op-level access for parity tests
    const Q8Mat& mat_qkv(int l) const { return layers_[l].wqkv; }
    const Q8Mat& mat_out() const { return output_; }

private:
    void head_impl(hipStream_t st, int tok0, int tok_count, int row0, int nrows, float* logits, int logits_stride, const ArgmaxEpi* am,
                   int nrows_valid, float* hidden_out);
    struct Layer { Q8Mat wqkv, wo, wgu, wdown; FMat fqkv, fo, fgu, fdown; float *attn_norm, *q_norm, *k_norm, *ffn_norm; };
    FMat make_fmat(const Gguf& g, const std::vector<std::string>& names, int K_expect);
    void forward_float(hipStream_t st, const Input& in, int ntok, const TokMeta& tm, const KvCache& kv, float* hidden_out);
    bool float_mode_ = false; FMat foutput_; DevBuf<float> xnf_, attf_, actf_;
    float* scratch_logits(int) { return scratch_.p; }
    DevBuf<float> scratch_, hid_, big_logits_;
    void gemv(hipStream_t st, const Q8Mat& w, int row0, int nrows, const int8_t* xq, const uint16_t* xd, float* out,
              int out_stride, int ntok);
    Q8Mat make_mat(const Gguf& g, const std::vector<std::pair<std::string, int>>& rows /* tensor, first row */, int N, int K);
    void load_into(const Gguf& g, const std::string& name, Q8Mat& dst, int row_off, int K_expect);
    float* load_f32(const Gguf& g, const std::string& name, int64_t n_expect);
    TfHparams hp_;
    int n_ctx_ = 0, max_tok_ = 0;
    std::vector<Layer> layers_;
    Q8Mat output_;
    float* output_norm_ = nullptr;
    struct WeightStore { // device memory shared by every context of one model
        std::vector<DevBuf<uint8_t>> blobs; DevBuf<float> rope_cos, rope_sin; DevBuf<int32_t> d_mrope;
        std::map<const uint8_t*, uint8_t*> mat_meta, mat_types; std::map<const uint8_t*, uint32_t*> mat_off;
    };
    std::shared_ptr<WeightStore> ws_;
    std::vector<DevBuf<uint8_t>>& blobs_ = ws_init()->blobs;
    size_t weight_bytes_ = 0, layer_weight_bytes_ = 0;
    DevBuf<float>& rope_cos_ = ws_->rope_cos; DevBuf<float>& rope_sin_ = ws_->rope_sin;
    DevBuf<int32_t>& d_mrope_ = ws_->d_mrope;
    WeightStore* ws_init() { if (!ws_) ws_ = std::make_shared<WeightStore>(); return ws_.get(); }
    void alloc_workspace();
    // activations
    DevBuf<float> h_, h2_, parts_o_, parts_d_, qkv_, qrot_, gu_;
    bool same_seq_ = false; int short_attn_min_ = 0; int fused_max_tok_ = 8; bool last_fused_ = false; int last_ntok_ = 0; bool all_q8_ = true;
    std::map<const uint8_t*, uint8_t*>& mat_meta_ = ws_->mat_meta; std::map<const uint8_t*, uint8_t*>& mat_types_ = ws_->mat_types; std::map<const uint8_t*, uint32_t*>& mat_off_ = ws_->mat_off;
    DevBuf<int8_t> xq_, aq_, fq_;
    DevBuf<uint16_t> xd_, ad_, fd_;
    int nparts_d_ = 1;
    // ---- ggml-arithmetic mode ----
    struct GgLayer { GgMat wq, wk, wv, wo, gate, up, down; };
    bool ggml_mode_ = false;
    std::vector<GgLayer> gg_layers_;
    GgMat gg_output_;
    GgMat gg_load(const Gguf& g, const std::string& name, int n_expect, int k_expect);
    void forward_ggml(hipStream_t st, const Input& in, int ntok, const TokMeta& tm, const KvCache& kv, float* hidden_out);
    void gg_linear(hipStream_t st, const GgMat& w, int row0, int nrows, const float* x, float* out, int out_stride, int ntok); // quantise x for w, then the dots
    GgAct gg_act_{};
    DevBuf<int8_t> gg_q8_, gg_qk_; DevBuf<uint16_t> gg_d8_; DevBuf<float> gg_dk_, gg_xn_, gg_att_, gg_scores_, gg_act_f_; DevBuf<int16_t> gg_bs_;
};

// standalone repack of host GGUF Q8_0 rows into a device Q8Mat (storage owns the memory); used by parity ops
Q8Mat q8mat_from_host(const void* raw_q8_0, int n, int k, DevBuf<uint8_t>& storage);
// the same for a K-quant matrix built from up to 3 tensors (GGUF rows of type Q8_0 / Q5_K / Q6_K, row counts multiples of 32): packed planes + tensor map,
// exactly what Transformer::make_mat produces (op-level parity tests)
struct KqPart { const void* raw; int type; int rows; };
Q8Mat kqmat_from_host(const KqPart* parts, int nparts, int k, DevBuf<uint8_t>& storage);

// simple paged KV pool (pages of 64 positions) -- "paged KV cache sized for 288 GB HBM3E"
class KvPool {
public:
    KvPool(int n_layer, int n_kv, int n_pages, int n_seq, int max_pages_per_seq);
    ~KvPool();
    KvPool(const KvPool&) = delete; KvPool& operator=(const KvPool&) = delete;
    KvCache view() const;
    // host-side page accounting
    int alloc_page();
    void free_page(int p);
    void assign(int seq, int logical_page, int physical_page); // updates host + device table
    void ensure(int seq, int n_positions);                      // make sure pages for [0,n_positions) exist
    void ensure(int seq, int n_positions, hipStream_t st);      // same; the table row is uploaded asynchronously on `st`
    void release(int seq);
    int pages_free() const { return (int)free_.size(); }
    int page(int seq, int logical_page) const { return table_[(size_t)seq * max_pages_ + logical_page]; }
    size_t bytes() const { return (k_.n + v_.n) * 2; }
private:
    int n_layer_, n_kv_, n_pages_, n_seq_, max_pages_;
    DevBuf<uint16_t> k_, v_;
    DevBuf<int32_t> d_table_;
    int32_t* table_ = nullptr; // pinned host mirror of the page table
    std::vector<int> free_;
    std::vector<int> used_pages_; // per seq count
};

} // namespace q3
