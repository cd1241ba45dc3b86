// llama_shim.cpp -- Boundary A: llama.cpp-ABI entry points over the HIP transformer (see include/q3tts_llama.h).
#include "../../include/q3tts_llama.h"
#include "tfctx.h"
#include <cstddef>
#include <mutex>

using namespace q3;

struct llama_vocab { int n_tokens; };
struct llama_model { std::shared_ptr<Transformer> tf; llama_vocab vocab; std::string path; };
struct llama_memory_i { llama_context* ctx; };
struct llama_context {
    llama_model* model; std::unique_ptr<TfContext> c; llama_context_params params; llama_memory_i mem;
    std::vector<float> logits, embd; int n_out = 0;
};
struct llama_sampler { float t; };

static std::mutex g_mu;
// capacity (ints) of the pos array of every batch handed out by llama_batch_init: llama_batch itself does not carry it, the caller writes at
// most n_tokens_max ints (llama/mod.rs:567-581) but the M-RoPE layout READS 4 * n_tokens of them (reference quirk 3)
#include <map>
static std::map<const int32_t*, int> g_pos_cap;
#define SHIM_TRY try {
#define SHIM_CATCH(ret) } catch (const std::exception& ex) { set_last_error(ex.what()); fprintf(stderr, "q3tts llama shim: %s\n", ex.what()); return ret; }

extern "C" {

void ggml_backend_load_all(void) {}
void llama_backend_init(void) {}
void llama_backend_free(void) {}

llama_model_params llama_model_default_params(void) {
    llama_model_params p{};
    p.n_gpu_layers = 999; p.use_mmap = true; p.use_extra_bufts = true;
    return p;
}
struct llama_model* llama_model_load_from_file(const char* path, llama_model_params) {
    SHIM_TRY
    std::lock_guard<std::mutex> lk(g_mu);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) throw Error("no HIP device (this libllama replacement has no CPU path)");
    auto* m = new llama_model();
    m->path = path;
    // the talker is prefilled with up to 1024 prompt rows (engine.rs:456); chunks of 256 tokens per launch
    m->tf = std::make_shared<Transformer>(path, Q3_TALKER_NCTX, 256);
    m->vocab.n_tokens = m->tf->hp().n_vocab;
    return m;
    SHIM_CATCH(nullptr)
}
void llama_model_free(struct llama_model* m) { delete m; }
const struct llama_vocab* llama_model_get_vocab(const struct llama_model* m) { return &m->vocab; }
int32_t llama_model_n_embd(const struct llama_model* m) { return m->tf->hp().n_embd; }
int32_t llama_model_n_head(const struct llama_model* m) { return m->tf->hp().n_head; }
int32_t llama_model_n_layer(const struct llama_model* m) { return m->tf->hp().n_layer; }
uint32_t llama_n_ctx(const struct llama_context* c) { return c ? c->params.n_ctx : 0; }
int32_t llama_n_vocab(const struct llama_vocab* v) { return v ? v->n_tokens : 0; }
int32_t llama_vocab_n_tokens(const struct llama_vocab* v) { return v->n_tokens; }
llama_token llama_vocab_eos(const struct llama_vocab*) { return Q3_CODEC_EOS; }

llama_context_params llama_context_default_params(void) {
    llama_context_params p{};
    p.n_ctx = 512; p.n_batch = 2048; p.n_ubatch = 512; p.n_seq_max = 1; p.n_threads = 4; p.n_threads_batch = 4;
    p.rope_scaling_type = -1; p.pooling_type = -1; p.attention_type = -1; p.flash_attn_type = -1;
    p.yarn_ext_factor = -1.0f; p.yarn_attn_factor = 1.0f; p.yarn_beta_fast = 32.0f; p.yarn_beta_slow = 1.0f;
    p.defrag_thold = -1.0f; p.type_k = 1; p.type_v = 1; p.offload_kqv = true; p.op_offload = true; p.kv_unified = false;
    return p;
}
struct llama_context* llama_init_from_model(struct llama_model* m, llama_context_params p) {
    SHIM_TRY
    if (!m) throw Error("null model");
    const int n_ctx = p.n_ctx ? (int)p.n_ctx : Q3_TALKER_NCTX;
    if (n_ctx > m->tf->n_ctx()) throw Error("n_ctx exceeds the model's RoPE table");
    auto* c = new llama_context();
    c->model = m; c->params = p; c->mem.ctx = c;
    c->c.reset(new TfContext(m->tf, n_ctx));
    return c;
    SHIM_CATCH(nullptr)
}
void llama_free(struct llama_context* c) { delete c; }

llama_batch llama_batch_init(int32_t n_tokens, int32_t embd, int32_t n_seq_max) { // llama/mod.rs:536-537, 556-614
    llama_batch b{};
    if (embd) b.embd = (float*)calloc((size_t)n_tokens * embd, sizeof(float));
    else b.token = (int32_t*)calloc((size_t)n_tokens, sizeof(int32_t));
    b.pos = (int32_t*)calloc((size_t)n_tokens, sizeof(int32_t));
    b.n_seq_id = (int32_t*)calloc((size_t)n_tokens, sizeof(int32_t));
    b.seq_id = (int32_t**)calloc((size_t)n_tokens + 1, sizeof(int32_t*));
    for (int i = 0; i < n_tokens; i++) b.seq_id[i] = (int32_t*)calloc((size_t)(n_seq_max > 0 ? n_seq_max : 1), sizeof(int32_t));
    b.logits = (int8_t*)calloc((size_t)n_tokens, 1);
    { std::lock_guard<std::mutex> lk(g_mu); g_pos_cap[b.pos] = n_tokens; }
    return b;
}
void llama_batch_free(llama_batch b) {
    { std::lock_guard<std::mutex> lk(g_mu); g_pos_cap.erase(b.pos); }
    free(b.token); free(b.embd); free(b.pos); free(b.n_seq_id);
    if (b.seq_id) { for (int i = 0; b.seq_id[i]; i++) free(b.seq_id[i]); free(b.seq_id); }
    free(b.logits);
}

int32_t llama_decode(struct llama_context* c, llama_batch b) {
    SHIM_TRY
    if (!c || !b.embd || b.n_tokens <= 0) throw Error("llama_decode: embedding batch required");
    const auto& hp = c->model->tf->hp();
    const int n = b.n_tokens;
    const bool mrope = (hp.mrope_sec[0] + hp.mrope_sec[1] + hp.mrope_sec[2] + hp.mrope_sec[3]) > 0;
    int cap = 4 * n; // ints the caller may have written; unknown batches (not from llama_batch_init) are trusted
    { std::lock_guard<std::mutex> lk(g_mu); auto it = g_pos_cap.find(b.pos); if (it != g_pos_cap.end()) cap = it->second; }
    std::vector<int32_t> pos4((size_t)4 * n);
    for (int i = 0; i < n; i++)
        for (int s = 0; s < 4; s++) { // stream-major, engine.rs:306-314; entries past the allocation were never written (prompts > 1024
            const size_t idx = mrope ? (size_t)s * n + i : (size_t)i; // tokens, mod.rs:567-581): stream 0 stands in, never an out-of-bounds read
            pos4[(size_t)4 * i + s] = idx < (size_t)cap ? b.pos[idx] : (s == 3 ? 0 : ((size_t)i < (size_t)cap ? b.pos[i] : i));
        }
    // hidden rows for every token (one copy of the final-norm output); the output matrix only for rows the caller flagged -- plus, with
    // params.embeddings, nothing more: the reference reads logits and embeddings at the BATCH index of the flagged row (engine.rs:550-566)
    const bool all_rows = c->params.embeddings;
    std::vector<int8_t> want(n, 0);
    int n_flag = 0;
    for (int i = 0; i < n; i++) if (b.logits && b.logits[i]) { want[i] = 1; n_flag++; }
    if (n_flag == 0) want[n - 1] = 1;
    std::vector<float> hid((size_t)n * hp.n_embd), lg((size_t)n * hp.n_vocab, 0.0f);
    c->c->eval(b.embd, pos4.data(), n, hid.data(), lg.data(), 0, hp.n_vocab, want.data());
    c->logits.clear(); c->embd.clear(); c->n_out = 0;
    for (int i = 0; i < n; i++) {
        if (all_rows || want[i]) { // embeddings = true: rows addressable by batch index (unflagged logits rows are zero); else compacted
            c->logits.insert(c->logits.end(), lg.begin() + (size_t)i * hp.n_vocab, lg.begin() + (size_t)(i + 1) * hp.n_vocab);
            c->embd.insert(c->embd.end(), hid.begin() + (size_t)i * hp.n_embd, hid.begin() + (size_t)(i + 1) * hp.n_embd);
            c->n_out++;
        }
    }
    return 0;
    SHIM_CATCH(-1)
}
float* llama_get_embeddings(struct llama_context* c) { return c->embd.empty() ? nullptr : c->embd.data(); }
float* llama_get_logits(struct llama_context* c) { return c->logits.empty() ? nullptr : c->logits.data(); }
struct llama_memory_i* llama_get_memory(struct llama_context* c) { return &c->mem; }
void llama_memory_clear(struct llama_memory_i* mem, bool) { if (mem) mem->ctx->c->clear(); }
bool llama_memory_seq_rm(struct llama_memory_i* mem, int32_t seq, int32_t p0, int32_t p1) { // only (-1|0, 0, -1) is used (mod.rs:482)
    if (!mem) return false;
    if ((seq <= 0) && p0 <= 0 && p1 < 0) { mem->ctx->c->clear(); return true; }
    return false;
}
int32_t llama_memory_seq_pos_max(struct llama_memory_i* mem, int32_t) { return mem ? mem->ctx->c->n_past() - 1 : -1; }
struct llama_sampler* llama_sampler_init_temp(float t) { return new llama_sampler{t}; }
llama_token llama_sampler_sample(struct llama_sampler*, struct llama_context* c, int32_t idx) { // greedy over row idx
    if (!c || c->logits.empty()) return 0;
    const int nv = c->model->tf->hp().n_vocab;
    const int row = idx < 0 ? c->n_out - 1 : idx;
    const float* lg = c->logits.data() + (size_t)row * nv;
    int best = 0;
    for (int i = 1; i < nv; i++) if (lg[i] > lg[best]) best = i;
    return best;
}
void llama_sampler_free(struct llama_sampler* s) { delete s; }

void q3tts_llama_abi_sizes(int32_t* out) {
    out[0] = (int32_t)sizeof(llama_model_params); out[1] = (int32_t)offsetof(llama_model_params, n_gpu_layers);
    out[2] = (int32_t)offsetof(llama_model_params, vocab_only); out[3] = (int32_t)sizeof(llama_context_params);
    out[4] = (int32_t)offsetof(llama_context_params, embeddings); out[5] = (int32_t)offsetof(llama_context_params, n_samplers);
    out[6] = (int32_t)sizeof(llama_batch); out[7] = (int32_t)offsetof(llama_batch, logits);
}

} // extern "C"
