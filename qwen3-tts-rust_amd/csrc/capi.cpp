// capi.cpp -- extern "C" Boundary B (include/q3tts.h).  Every entry point catches, records last_error, returns a code.
#include "../../include/q3tts.h"
#include "engine.h"
#include "tfctx.h"
#include <algorithm>

using namespace q3;

struct q3tts_assets { std::unique_ptr<HostAssets> owned; const HostAssets* a = nullptr; };
struct q3tts_engine { std::unique_ptr<Engine> e; q3tts_assets assets_view; }; // the view lives exactly as long as the engine
struct q3tts_sampler { Sampler s; };
struct q3tts_chunker { std::unique_ptr<Chunker> c; };
struct q3tts_decoder { std::unique_ptr<CodecDecoder> d; hipStream_t st = nullptr; float* pinned = nullptr; size_t pinned_cap = 0; };
struct q3tts_tf { std::unique_ptr<TfContext> c; };

#define Q3_API_BEGIN try {
#define Q3_API_END(failval) } catch (const std::exception& ex) { set_last_error(ex.what()); return failval; } catch (...) { set_last_error("unknown error"); return failval; }

static void require_gpu() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) throw Error("no HIP device available: the HIP path is the only compute path (no CPU fallback)");
}

extern "C" {

const char* q3tts_last_error(void) { return last_error(); }
int q3tts_version(void) { return 100; }
int q3tts_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }

void q3tts_sampler_config_default(q3tts_sampler_config* c) { c->temperature = 0.7f; c->top_k = 40; c->top_p = 0.9f; c->has_seed = 0; c->seed = 0; }
void q3tts_engine_params_default(q3tts_engine_params* p) {
    p->model_dir = nullptr; p->quant = "q8_0"; p->device = 0; p->max_batch = 1; p->max_prompt = 1024;
    p->max_steps = Q3_DEFAULT_MAX_STEPS; p->load_codec = 1; p->use_graph = 1;
}

int q3tts_engine_create(const q3tts_engine_params* p, q3tts_engine** out) {
    Q3_API_BEGIN
    Q3_CHECK(p && out && p->model_dir, "null argument");
    require_gpu();
    Q3_HIP(hipSetDevice(p->device));
    EngineParams ep;
    ep.model_dir = p->model_dir; ep.quant = p->quant ? p->quant : "q8_0"; ep.max_batch = p->max_batch; ep.max_prompt = p->max_prompt;
    ep.max_steps = p->max_steps; ep.load_codec = p->load_codec != 0; ep.use_graph = p->use_graph != 0;
    auto* h = new q3tts_engine();
    try { h->e.reset(new Engine(ep)); } catch (...) { delete h; throw; }
    h->assets_view.a = &h->e->assets();
    *out = h;
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}
void q3tts_engine_destroy(q3tts_engine* e) { delete e; }
int32_t q3tts_engine_device(q3tts_engine* e) { return e ? e->e->device() : -1; }

static GenRequest to_gen(const q3tts_request& q);
int q3tts_generate_batch(q3tts_engine* e, q3tts_request* reqs, int32_t n, int32_t want_pcm) {
    Q3_API_BEGIN
    Q3_CHECK(e && reqs && n >= 1, "bad arguments");
    std::vector<GenRequest> rq(n);
    for (int i = 0; i < n; i++) rq[i] = to_gen(reqs[i]);
    std::vector<GenResult> res;
    e->e->generate_batch(rq, res, want_pcm != 0);
    for (int i = 0; i < n; i++) {
        reqs[i].n_frames = res[i].n_frames;
        if (reqs[i].codes_out) std::copy(res[i].codes.begin(), res[i].codes.end(), reqs[i].codes_out);
        const int64_t np = std::min<int64_t>((int64_t)res[i].pcm.size(), reqs[i].pcm_out ? reqs[i].pcm_capacity : 0);
        if (np > 0) std::copy(res[i].pcm.begin(), res[i].pcm.begin() + np, reqs[i].pcm_out);
        reqs[i].n_pcm = np;
        reqs[i].prefill_ms = res[i].prefill_ms; reqs[i].first_chunk_ms = res[i].first_chunk_ms; reqs[i].total_ms = res[i].total_ms;
    }
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}

static GenRequest to_gen(const q3tts_request& q) {
    GenRequest g;
    g.prompt = q.prompt; g.n_prompt = q.n_prompt; g.max_steps = q.max_steps; g.mask_eos = q.mask_eos != 0;
    g.sampler.temperature = q.sampler.temperature; g.sampler.top_k = q.sampler.top_k; g.sampler.top_p = q.sampler.top_p;
    g.sampler.has_seed = q.sampler.has_seed != 0; g.sampler.seed = q.sampler.seed;
    return g;
}
int q3tts_submit(q3tts_engine* e, const q3tts_request* r, int32_t want_pcm, int64_t* req_id) {
    Q3_API_BEGIN
    Q3_CHECK(e && r && req_id, "null argument");
    *req_id = e->e->submit(to_gen(*r), want_pcm != 0, true);
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}
int q3tts_poll(q3tts_engine* e, int64_t id, q3tts_req_status* o) {
    Q3_API_BEGIN
    Q3_CHECK(e && o, "null argument");
    const ReqStatus s = e->e->poll(id);
    o->state = s.state; o->n_frames = s.n_frames; o->n_pcm = s.n_pcm; o->queue_ms = s.queue_ms; o->prefill_ms = s.prefill_ms;
    o->first_chunk_ms = s.first_chunk_ms; o->total_ms = s.total_ms;
    if (s.state == REQ_FAILED) set_last_error(s.error.empty() ? "request failed" : s.error); // state says it failed; q3tts_last_error() says why
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}
int q3tts_fetch(q3tts_engine* e, int64_t id, int32_t* codes, int32_t frame_off, int32_t max_frames, float* pcm, int64_t pcm_off,
                int64_t pcm_cap, int32_t* got_frames, int64_t* got_pcm) {
    Q3_API_BEGIN
    Q3_CHECK(e, "null argument");
    int gf = 0; int64_t gp = 0;
    e->e->fetch(id, codes, frame_off, max_frames, pcm, pcm_off, pcm_cap, &gf, &gp);
    if (got_frames) *got_frames = gf;
    if (got_pcm) *got_pcm = gp;
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}
int q3tts_wait(q3tts_engine* e, int64_t id, double timeout_ms) {
    Q3_API_BEGIN
    Q3_CHECK(e, "null argument");
    return e->e->wait(id, timeout_ms) ? 0 : 1;
    Q3_API_END(Q3TTS_ERR)
}
int q3tts_release(q3tts_engine* e, int64_t id) { Q3_API_BEGIN Q3_CHECK(e, "null argument"); e->e->release(id); return Q3TTS_OK; Q3_API_END(Q3TTS_ERR) }
int q3tts_sched_start(q3tts_engine* e) { Q3_API_BEGIN Q3_CHECK(e, "null argument"); e->e->start_driver(); return Q3TTS_OK; Q3_API_END(Q3TTS_ERR) }
int q3tts_sched_stop(q3tts_engine* e) { Q3_API_BEGIN Q3_CHECK(e, "null argument"); e->e->stop_driver(); return Q3TTS_OK; Q3_API_END(Q3TTS_ERR) }
int q3tts_sched_step(q3tts_engine* e, int32_t* busy) {
    Q3_API_BEGIN
    Q3_CHECK(e, "null argument");
    const bool b = e->e->step();
    if (busy) *busy = b ? 1 : 0;
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}
int q3tts_voice_register(q3tts_engine* e, const float* spk, const int32_t* ref_codes, int32_t n_ref_codes, const int32_t* ref_text,
                         int32_t n_ref_text, int32_t* voice_id) {
    Q3_API_BEGIN
    Q3_CHECK(e && spk && voice_id, "null argument");
    Voice v;
    v.spk_emb.assign(spk, spk + 2048);
    if (ref_codes && n_ref_codes > 0) v.ref_codes.assign(ref_codes, ref_codes + n_ref_codes);
    if (ref_text && n_ref_text > 0) v.ref_text_ids.assign(ref_text, ref_text + n_ref_text);
    *voice_id = e->e->register_voice(v);
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}
int q3tts_submit_text(q3tts_engine* e, int32_t voice_id, const int32_t* text_ids, int32_t n_text, int32_t lang_id, const int32_t* instr_ids,
                      int32_t n_instr, const q3tts_sampler_config* sampler, int32_t max_steps, int32_t mask_eos, int32_t want_pcm,
                      int64_t* req_id) {
    Q3_API_BEGIN
    Q3_CHECK(e && text_ids && n_text >= 0 && req_id, "bad arguments");
    const Voice& v = e->e->voice(voice_id);
    std::vector<int32_t> t(text_ids, text_ids + n_text), ins;
    if (instr_ids) ins.assign(instr_ids, instr_ids + n_instr);
    int lang = lang_id;
    // engine.rs:398-428: clone voices (codes + ref text) take build_clone_prompt, presets build_core with marker + spk_emb
    PromptData pd = !v.ref_codes.empty()
        ? PromptBuilder::build_clone_prompt(e->e->assets(), t, v.ref_codes, v.ref_text_ids, v.spk_emb.data(), lang_id, instr_ids ? &ins : nullptr)
        : PromptBuilder::build_core(e->e->assets(), t, lang_id >= 0 ? &lang : nullptr, nullptr, v.spk_emb.data(), instr_ids ? &ins : nullptr, nullptr);
    GenRequest g;
    g.prompt = pd.embd.data(); g.n_prompt = pd.n_rows; g.max_steps = max_steps; g.mask_eos = mask_eos != 0;
    if (sampler) {
        g.sampler.temperature = sampler->temperature; g.sampler.top_k = sampler->top_k; g.sampler.top_p = sampler->top_p;
        g.sampler.has_seed = sampler->has_seed != 0; g.sampler.seed = sampler->seed;
    }
    *req_id = e->e->submit(g, want_pcm != 0, true);
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}

int q3tts_engine_stats(q3tts_engine* e, q3tts_stats* o) {
    Q3_API_BEGIN
    const EngineStats& s = e->e->stats;
    o->frame_loop_ms = s.frame_loop_ms; o->frames = s.frames; o->prefill_ms = s.prefill_ms; o->gemv_ms = s.gemv_ms;
    o->gemv_launches = s.gemv_launches; o->gemv_bytes = s.gemv_bytes; o->codec_ms = s.codec_ms; o->codec_calls = s.codec_calls;
    o->gu_ms = s.gu_ms; o->gu_launches = s.gu_launches; o->gu_bytes = s.gu_bytes;
    o->sched_steps = s.steps; o->slot_frames = s.slot_frames; o->graph_frames = s.graph_frames;
    o->talker_weight_bytes = (double)e->e->talker().weight_bytes(); o->predictor_weight_bytes = (double)e->e->predictor().weight_bytes();
    const auto& hp = e->e->talker().hp();
    o->kv_bytes_per_token = (double)hp.n_layer * 2 * hp.n_kv * 128 * 2;
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}
void q3tts_engine_reset_stats(q3tts_engine* e) { if (e) e->e->reset_stats(); }
void q3tts_engine_set_instrument(q3tts_engine* e, int32_t on) { if (e) e->e->set_instrument(on != 0); }
double q3tts_engine_bytes_per_step(q3tts_engine* e, int32_t batch, double mean_ctx) { return e ? (double)e->e->bytes_per_frame_step(batch, mean_ctx) : 0.0; }

// ---------------- assets / prompt ----------------
int q3tts_assets_open(const char* path, q3tts_assets** out) {
    Q3_API_BEGIN
    Q3_CHECK(path && out, "null argument");
    auto* h = new q3tts_assets();
    try { h->owned.reset(new HostAssets(path)); } catch (...) { delete h; throw; }
    h->a = h->owned.get();
    *out = h;
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}
void q3tts_assets_close(q3tts_assets* a) { delete a; }
const q3tts_assets* q3tts_engine_assets(q3tts_engine* e) { return e ? &e->assets_view : nullptr; }
int q3tts_assets_codec_embedding(const q3tts_assets* a, int32_t q, int32_t code, float* out) { Q3_API_BEGIN a->a->codec_embedding(q, code, out); return Q3TTS_OK; Q3_API_END(Q3TTS_ERR) }
int q3tts_assets_text_embedding(const q3tts_assets* a, int64_t tok, float* out) { Q3_API_BEGIN a->a->text_embedding(tok, out); return Q3TTS_OK; Q3_API_END(Q3TTS_ERR) }
int q3tts_assets_tts_pad(const q3tts_assets* a, float* out) { Q3_API_BEGIN std::copy(a->a->tts_pad(), a->a->tts_pad() + 2048, out); return Q3TTS_OK; Q3_API_END(Q3TTS_ERR) }

// host-side projection with the reference's exact loop (assets_manager.rs:383-399); used by the Boundary-A replay harness
extern "C" int q3tts_assets_proj_out(const q3tts_assets* a) { return (int)a->a->proj_out; }
extern "C" int q3tts_assets_project_host(const q3tts_assets* a, const float* x, float* out) {
    const HostAssets& h = *a->a;
    for (int64_t o = 0; o < h.proj_out; o++) {
        float sum = h.proj_b[o];
        const float* w = h.proj_w + (size_t)o * (size_t)h.proj_in;
        for (int64_t i = 0; i < h.proj_in; i++) { const float t = x[i] * w[i]; sum = sum + t; }
        out[o] = sum;
    }
    return Q3TTS_OK;
}
static int emit_rows(const PromptData& d, float* out, int32_t max_rows) {
    if (d.n_rows > max_rows) { set_last_error("prompt does not fit max_rows"); return -1; }
    std::copy(d.embd.begin(), d.embd.end(), out);
    return d.n_rows;
}
int q3tts_prompt_build_core(const q3tts_assets* a, const int32_t* text_ids, int32_t n_text, int32_t lang_id, int32_t spk_id,
                            const float* spk_emb, const int32_t* instr_ids, int32_t n_instr, const float* mid_rows, int32_t n_mid,
                            float* out, int32_t max_rows) {
    Q3_API_BEGIN
    std::vector<int32_t> t(text_ids, text_ids + n_text), ins;
    if (instr_ids) ins.assign(instr_ids, instr_ids + n_instr);
    std::vector<float> mid;
    if (mid_rows) mid.assign(mid_rows, mid_rows + (size_t)n_mid * 2048);
    int lang = lang_id, spk = spk_id;
    return emit_rows(PromptBuilder::build_core(*a->a, t, lang_id >= 0 ? &lang : nullptr, spk_id >= 0 ? &spk : nullptr, spk_emb,
                                               instr_ids ? &ins : nullptr, mid_rows ? &mid : nullptr), out, max_rows);
    Q3_API_END(-1)
}
int q3tts_prompt_build_clone(const q3tts_assets* a, const int32_t* text_ids, int32_t n_text, const int32_t* ref_codes, int32_t n_ref_codes,
                             const int32_t* ref_text_ids, int32_t n_ref_text, const float* spk_emb, int32_t lang_id,
                             const int32_t* instr_ids, int32_t n_instr, float* out, int32_t max_rows) {
    Q3_API_BEGIN
    std::vector<int32_t> t(text_ids, text_ids + n_text), rc(ref_codes, ref_codes + n_ref_codes), rt(ref_text_ids, ref_text_ids + n_ref_text), ins;
    if (instr_ids) ins.assign(instr_ids, instr_ids + n_instr);
    return emit_rows(PromptBuilder::build_clone_prompt(*a->a, t, rc, rt, spk_emb, lang_id, instr_ids ? &ins : nullptr), out, max_rows);
    Q3_API_END(-1)
}

// ---------------- sampler / chunker ----------------
q3tts_sampler* q3tts_sampler_new(float temperature, int32_t top_k, float top_p, uint64_t seed) { return new q3tts_sampler{Sampler(temperature, top_k, top_p, seed)}; }
void q3tts_sampler_free(q3tts_sampler* s) { delete s; }
int32_t q3tts_sampler_sample(q3tts_sampler* s, const float* logits, int32_t n_vocab, int32_t start, int32_t end) { return s->s.sample(logits, n_vocab, start, end); }
q3tts_chunker* q3tts_chunker_new(q3tts_decode_cb cb, void* user) {
    auto* h = new q3tts_chunker();
    h->c.reset(new Chunker([cb, user](const int64_t* codes, int n, bool fin) { if (cb) cb(user, codes, n, fin ? 1 : 0); }));
    return h;
}
void q3tts_chunker_free(q3tts_chunker* c) { delete c; }
int q3tts_chunker_push(q3tts_chunker* c, const int64_t* codes, int32_t n, int32_t is_final) { Q3_API_BEGIN c->c->push(codes, n, is_final != 0); return Q3TTS_OK; Q3_API_END(Q3TTS_ERR) }

// ---------------- codec decoder ----------------
int q3tts_decoder_create(const char* path, int32_t n_streams, q3tts_decoder** out) {
    Q3_API_BEGIN
    require_gpu();
    auto* h = new q3tts_decoder();
    try { h->d.reset(new CodecDecoder(path, n_streams, 64)); Q3_HIP(hipStreamCreate(&h->st)); } catch (...) { delete h; throw; }
    *out = h;
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}
int q3tts_decoder_create_ex(const char* path, int32_t n_streams, int32_t max_frames, int32_t max_group, q3tts_decoder** out) {
    Q3_API_BEGIN
    require_gpu();
    Q3_CHECK(path && out && n_streams >= 1 && max_frames >= 1 && max_group >= 1, "bad arguments");
    auto* h = new q3tts_decoder();
    try { h->d.reset(new CodecDecoder(path, n_streams, max_frames, 1, max_group)); Q3_HIP(hipStreamCreate(&h->st)); } catch (...) { delete h; throw; }
    *out = h;
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}
int q3tts_decoder_decode_group(q3tts_decoder* d, int32_t G, const int32_t* streams, const int64_t* codes, int32_t n_frames, float* wav) {
    Q3_API_BEGIN
    Q3_CHECK(d && streams && codes && wav && G >= 1 && n_frames >= 1, "bad arguments");
    const size_t per = (size_t)n_frames * d->d->samples_per_frame();
    if (d->pinned_cap < per * G) { // pinned staging for the async D2H copies of the pass
        if (d->pinned) (void)hipHostFree(d->pinned);
        Q3_HIP(hipHostMalloc((void**)&d->pinned, per * G * sizeof(float)));
        d->pinned_cap = per * G;
    }
    std::vector<float*> dst(G);
    for (int g = 0; g < G; g++) dst[g] = d->pinned + (size_t)g * per;
    std::vector<int> st(streams, streams + G);
    const int got = d->d->decode_group_async(d->st, G, st.data(), codes, n_frames, dst.data(), 0);
    Q3_HIP(hipStreamSynchronize(d->st));
    if (got < 0) throw Error("decode failed");
    std::copy(d->pinned, d->pinned + per * G, wav);
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}
void q3tts_decoder_destroy(q3tts_decoder* d) { if (d && d->st) (void)hipStreamDestroy(d->st); if (d && d->pinned) (void)hipHostFree(d->pinned); delete d; }
int q3tts_decoder_samples_per_frame(q3tts_decoder* d) { return d->d->samples_per_frame(); }
int q3tts_decoder_reset(q3tts_decoder* d, int32_t stream) { Q3_API_BEGIN d->d->reset(stream); return Q3TTS_OK; Q3_API_END(Q3TTS_ERR) }
int q3tts_decoder_decode(q3tts_decoder* d, int32_t stream, const int64_t* codes, int32_t n_frames, int32_t is_last, float* wav, int64_t* valid) {
    Q3_API_BEGIN
    const int got = d->d->decode(d->st, stream, codes, n_frames, is_last != 0, wav);
    if (got < 0) throw Error("decode failed");
    if (valid) *valid = got;
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}

// DecoderState export / import (onnx.rs:461-496)
int64_t q3tts_decoder_state_floats(q3tts_decoder* d) { return d ? (int64_t)d->d->state_floats() : -1; }
int q3tts_decoder_state_export(q3tts_decoder* d, int32_t stream, float* out, int64_t n_floats) { Q3_API_BEGIN Q3_CHECK(d && out && n_floats >= 0, "null argument"); d->d->state_export(stream, out, (size_t)n_floats); return Q3TTS_OK; Q3_API_END(Q3TTS_ERR) }
int q3tts_decoder_state_import(q3tts_decoder* d, int32_t stream, const float* in, int64_t n_floats) { Q3_API_BEGIN Q3_CHECK(d && in && n_floats >= 0, "null argument"); d->d->state_import(stream, in, (size_t)n_floats); return Q3TTS_OK; Q3_API_END(Q3TTS_ERR) }
/* entry i of the layout: name (valid until the decoder is destroyed), offset in floats, rows x cols; returns the number of entries */
int32_t q3tts_decoder_state_entry(q3tts_decoder* d, int32_t i, const char** name, int64_t* offset, int32_t* rows, int32_t* cols) {
    if (!d) return -1;
    static thread_local std::vector<CodecStateEntry> lay;
    lay = d->d->state_layout();
    if (i >= 0 && i < (int)lay.size()) { if (name) *name = lay[i].name.c_str(); if (offset) *offset = lay[i].offset; if (rows) *rows = lay[i].rows; if (cols) *cols = lay[i].cols; }
    return (int32_t)lay.size();
}

// ---------------- transformer contexts ----------------
int q3tts_tf_open(const char* path, int32_t n_ctx, int32_t max_tok, q3tts_tf** out) {
    Q3_API_BEGIN
    require_gpu();
    auto model = std::make_shared<Transformer>(path, n_ctx, max_tok);
    auto* h = new q3tts_tf();
    h->c.reset(new TfContext(model, n_ctx));
    *out = h;
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}
void q3tts_tf_close(q3tts_tf* t) { delete t; }
int q3tts_tf_dims(q3tts_tf* t, int32_t* n_embd, int32_t* n_layer, int32_t* n_head, int32_t* n_vocab) {
    const auto& hp = t->c->model().hp();
    if (n_embd) *n_embd = hp.n_embd; if (n_layer) *n_layer = hp.n_layer; if (n_head) *n_head = hp.n_head; if (n_vocab) *n_vocab = hp.n_vocab;
    return Q3TTS_OK;
}
void q3tts_tf_clear(q3tts_tf* t) { t->c->clear(); }
int q3tts_tf_eval(q3tts_tf* t, const float* x, const int32_t* pos4, int32_t ntok, float* hidden_out, float* logits_out, int32_t row0, int32_t row1) {
    Q3_API_BEGIN
    t->c->eval(x, pos4, ntok, hidden_out, logits_out, row0, row1);
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}

} // extern "C"
