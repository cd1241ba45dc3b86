// engine.cpp -- see engine.h.  Loop structure cites /root/reference/src/tts/engine.rs line numbers.
#include "engine.h"
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include "kdev.h"
#include <functional>

namespace q3 {

static std::string quant_dir(const std::string& q) { // engine.rs:91-95 (+ this engine's extra dirs)
    if (q == "q5_k_m") return "gguf_q5_k_m";
    if (q == "q8_0") return "gguf_q8_0";
    if (q == "bf16") return "gguf_bf16";
    if (q == "f16") return "gguf_f16";
    return "gguf";
}
static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

Engine::Engine(const EngineParams& p) : p_(p) {
    Q3_CHECK(p.max_batch >= 1 && p.max_batch <= 512, "max_batch out of range");
    Q3_HIP(hipGetDevice(&dev_));
    Q3_HIP(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
    const std::string dir = p.model_dir + "/" + quant_dir(p.quant);
    assets_.reset(new HostAssets(dir + "/qwen3_assets.gguf"));
    const int B = p.max_batch;
    B_ = B;
    talker_.reset(new Transformer(dir + "/qwen3_tts_talker.gguf", Q3_TALKER_NCTX, std::max(256, B)));
    predictor_.reset(new Transformer(dir + "/qwen3_tts_predictor.gguf", Q3_PRED_NCTX, 2 * B));
    if (const char* e = std::getenv("Q3_PRED_FUSED_MAX")) predictor_->set_fused_max_tokens(atoi(e)); // experiment knob
    // The predictor never holds more than 17 positions per sequence: wide steps use the single-wave attention kernel (one wave per
    // token and kv head, no workgroup barrier).  At one sequence it loses to k_attention_fused (3.24 vs 3.06 ms/frame: the serial PV
    // of both heads costs more than the barriers it removes), hence the token threshold.
    {
        int min_tok = 8;
        if (const char* e = std::getenv("Q3_SHORT_ATTN_MIN")) min_tok = atoi(e); // 0 = never
        // (the engine drives the predictor through positions 0..16 only -- 2 prompt rows + 15 code passes -- i.e. one KV page)
        predictor_->set_short_attention(min_tok);
    }
    Q3_CHECK(talker_->hp().n_embd == Q3_EMBD, "talker n_embd must be 2048 (reference hard-codes 2048-wide rows)");
    dP_ = predictor_->hp().n_embd;
    Q3_CHECK(assets_->proj_out == dP_ && assets_->proj_in == Q3_EMBD, "proj shape does not match predictor n_embd");
    Q3_CHECK(assets_->n_codec == 16, "need 16 codec embedding tables");
    Q3_CHECK(predictor_->hp().n_vocab >= 15 * Q3_CODEBOOK_SIZE, "predictor vocab < 15*2048");
    // ---- device assets ----
    std::vector<float> wt((size_t)Q3_EMBD * dP_);
    for (int o = 0; o < dP_; o++)
        for (int i = 0; i < Q3_EMBD; i++) wt[(size_t)i * dP_ + o] = assets_->proj_w[(size_t)o * Q3_EMBD + i];
    d_proj_wt_.alloc(wt.size()); d_proj_wt_.upload(wt.data(), wt.size());
    Q3_CHECK(dP_ % 16 == 0, "projection width must be a multiple of 16");
    {   // blocked copy [dP/16][2048][16] for the per-frame projection of m_hidden
        std::vector<float> wb((size_t)Q3_EMBD * dP_);
        for (int o = 0; o < dP_; o++)
            for (int i = 0; i < Q3_EMBD; i++) wb[((size_t)(o / 16) * Q3_EMBD + i) * 16 + (o % 16)] = assets_->proj_w[(size_t)o * Q3_EMBD + i];
        d_proj_wblk_.alloc(wb.size()); d_proj_wblk_.upload(wb.data(), wb.size());
    }
    d_proj_b_.alloc(dP_); d_proj_b_.upload(assets_->proj_b, dP_);
    d_tts_pad_.alloc(Q3_EMBD); d_tts_pad_.upload(assets_->tts_pad(), Q3_EMBD);
    std::vector<const float*> ptrs(16);
    std::vector<int64_t> rows(16);
    for (int q = 0; q < 16; q++) {
        const size_t n = (size_t)assets_->codec_rows[q] * Q3_EMBD;
        d_codec_tab_[q].alloc(n); d_codec_tab_[q].upload(assets_->codec[q], n);
        // pre-projected table P_q[c] = project(E_q[c]) : same arithmetic as assets_manager.rs:401-417,439-442, done once
        d_proj_tab_[q].alloc((size_t)assets_->codec_rows[q] * dP_);
        launch_project_table(st_, d_codec_tab_[q].p, assets_->codec_rows[q], d_proj_wt_.p, d_proj_b_.p, Q3_EMBD, dP_, d_proj_tab_[q].p);
        ptrs[q] = d_codec_tab_[q].p; rows[q] = assets_->codec_rows[q];
    }
    d_tab_ptrs_.alloc(16); d_tab_ptrs_.upload(ptrs.data(), 16);
    d_tab_rows_.alloc(16); d_tab_rows_.upload(rows.data(), 16);
    // ---- KV pools ----
    const int pages_per_seq = (p.max_prompt + p.max_steps + 1 + 63) / 64;
    // talker page-table row B is the scratch sequence: idle slots of the frame graph read and write its single page, so the pages of a
    // slot that is being prefilled (or was just recycled) are never touched by the graph
    kv_t_.reset(new KvPool(talker_->hp().n_layer, talker_->hp().n_kv, B * pages_per_seq + 1, B + 1, pages_per_seq));
    kv_p_.reset(new KvPool(predictor_->hp().n_layer, predictor_->hp().n_kv, B, B, 1));
    for (int b = 0; b < B; b++) kv_p_->ensure(b, 16);
    pred_identity_pages_ = true; // sequence b owns physical page b: the predictor's attention can skip the page-table lookup
    for (int b = 0; b < B; b++) if (kv_p_->page(b, 0) != b) pred_identity_pages_ = false;
    if (const char* e = std::getenv("Q3_UNIFORM_META")) if (e[0] == '0') pred_identity_pages_ = false;
    kv_t_->ensure(B, 1);
    {
        const char* e = std::getenv("Q3_ASYNC_PREFILL");
        async_pf_ = B > 1 && !(e && e[0] == '0');
    }
    if (async_pf_) {
        talker_pf_.reset(new Transformer(*talker_, 256)); // shares the talker's device weights; own activation workspace
        Q3_HIP(hipStreamCreateWithFlags(&st_pf_, hipStreamNonBlocking));
        Q3_HIP(hipEventCreate(&pf_done_)); Q3_HIP(hipEventCreate(&pf_e0_));
        arena_pf_cap_ = (size_t)4096 * (Q3_EMBD * 4 + 24) + ((size_t)1 << 16); // prompts of one admission: at most 4096 rows + routing tables
        Q3_HIP(hipHostMalloc((void**)&arena_pf_, arena_pf_cap_));
        d_prompt_pf_.alloc((size_t)256 * Q3_EMBD); d_pfa_seq_.alloc(256); d_pfa_slot_.alloc(256); d_pfa_pos_.alloc(4 * 256);
    }
    // ---- per-slot state ----
    hist_stride_ = p.max_steps * 16;
    tl_stride_ = (Q3_SAMPLE_END + 31) & ~31;
    d_tseq_.alloc(B); d_tslot_.alloc(B); d_tpos_.alloc(4 * B); d_keys_.alloc(16 * B); d_next_key0_.alloc(B); d_nframes_.alloc(B); d_finished_.alloc(B);
    d_maxframes_.alloc(B); d_maskeos_.alloc(B); d_hist_.alloc((size_t)B * hist_stride_);
    d_temp_.alloc(B); d_topp_.alloc(B); d_topk_.alloc(B); d_rngkey_.alloc((size_t)8 * B); d_draws_.alloc(B);
    d_temp_.zero(); d_topp_.zero(); d_topk_.zero(); d_rngkey_.zero(); d_draws_.zero();
    d_tlogits_.alloc((size_t)B * tl_stride_); d_thidden_.alloc((size_t)B * Q3_EMBD); d_pin_.alloc((size_t)2 * B * dP_);
    d_fb_.alloc((size_t)B * Q3_EMBD);
    d_prompt_.alloc((size_t)talker_->max_tok() * Q3_EMBD);
    d_pf_seq_.alloc(talker_->max_tok()); d_pf_slot_.alloc(talker_->max_tok()); d_pf_pos_.alloc(4 * (size_t)talker_->max_tok());
    d_hist_.zero(); d_tlogits_.zero(); d_thidden_.zero();
    // predictor routing: pass i (i = 2..15) handles position i of every slot; tables are [16][B] so any graph width can use them
    std::vector<int32_t> seq(B), slot((size_t)16 * B), pos((size_t)16 * B * 4);
    for (int b = 0; b < B; b++) {
        seq[b] = b;
        for (int i = 0; i < 16; i++) { slot[(size_t)i * B + b] = i; for (int s = 0; s < 4; s++) pos[((size_t)i * B + b) * 4 + s] = i; } // engine.rs:316-318
    }
    d_pseq_.alloc(B); d_pseq_.upload(seq.data(), B);
    d_pslot_.alloc(slot.size()); d_pslot_.upload(slot.data(), slot.size());
    d_ppos_.alloc(pos.size()); d_ppos_.upload(pos.data(), pos.size());
    h_maxf_.assign(B, 0); h_fin_.assign(B, 1); h_nfr_.assign(B, 0); h_mask_.assign(B, -1); h_nprompt_.assign(B, 0); h_topk_.assign(B, 0);
    h_temp_.assign(B, 0.0f); h_topp_.assign(B, 1.0f);
    slot_req_.assign(B, nullptr); slot_live_.assign(B, 0);
    if (async_pf_) { d_pf_logits_.alloc((size_t)B * tl_stride_); d_pf_hidden_.alloc((size_t)B * Q3_EMBD); }
    {
        std::vector<q3_u64> k0((size_t)16 * B, pack_key(-INFINITY, 0)), n0(B, pack_key(-INFINITY, 0));
        d_keys_.upload(k0.data(), k0.size()); d_next_key0_.upload(n0.data(), n0.size());
    }
    arena_cap_ = (size_t)talker_->max_tok() * Q3_EMBD * 4 + (size_t)B * 4096 + ((size_t)1 << 20);
    Q3_HIP(hipHostMalloc((void**)&arena_, arena_cap_));
    Q3_HIP(hipDeviceSynchronize()); // memsets / table kernels above ran on other streams than st_
    upload_slot_state();
    if (p.load_codec) {
        // codec streams outnumber the slots: a retired sequence's last chunks still drain while its slot is already reused
        // up to 16 streams share one decode pass; two lanes (HIP streams + scratch) let consecutive groups overlap
        // streams per decode pass, measured on MI355X: 64 slots 617 / 607 / 600 audio-s/s at 16 / 32 / 64, 256 slots 821 / 860 / 836
        static const int group_env = [] { const char* e = std::getenv("Q3_CODEC_GROUP"); return e ? atoi(e) : 0; }();
        const int group_cap = group_env > 0 ? group_env : (B >= 128 ? 32 : 16);
        const int n_cs = B + std::min(B, 16), gmax = std::max(1, std::min(B, group_cap));
        static const int lanes_env = [] { const char* e = std::getenv("Q3_CODEC_LANES"); return e ? atoi(e) : 0; }();
        const int n_lanes = B > 1 ? (lanes_env > 0 ? std::min(lanes_env, 8) : 2) : 1;
        codec_.reset(new CodecDecoder(p.model_dir + "/onnx/q3tts_codec.gguf", n_cs, 4, n_lanes, gmax));
        st2_.resize(n_lanes);
        for (auto& s2 : st2_) Q3_HIP(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        slot_cap_ = (size_t)p.max_steps * codec_->samples_per_frame();
        Q3_HIP(hipHostMalloc((void**)&pcm_pinned_, (size_t)n_cs * slot_cap_ * sizeof(float)));
        for (int c = n_cs - 1; c >= 0; c--) cs_free_.push_back(c);
    }
    Q3_HIP(hipStreamSynchronize(st_));
    Q3_HIP(hipDeviceSynchronize());
    arena_used_ = 0;
}

void* Engine::stage_alloc(size_t bytes) {
    bytes = (bytes + 63) & ~(size_t)63;
    Q3_CHECK(bytes <= arena_cap_, "staging request larger than the arena");
    if (arena_used_ + bytes > arena_cap_) { Q3_HIP(hipStreamSynchronize(st_)); arena_used_ = 0; } // everything staged so far has been consumed
    void* p = arena_ + arena_used_;
    arena_used_ += bytes;
    return p;
}
void Engine::h2d(void* dst, const void* src, size_t bytes) {
    void* p = stage_alloc(bytes);
    std::memcpy(p, src, bytes);
    Q3_HIP(hipMemcpyAsync(dst, p, bytes, hipMemcpyHostToDevice, st_));
}
void* Engine::d2h_begin(const void* src, size_t bytes) {
    void* p = stage_alloc(bytes);
    Q3_HIP(hipMemcpyAsync(p, src, bytes, hipMemcpyDeviceToHost, st_));
    return p;
}

// ---------------------------------------------------------------------------------------------------------------
struct Engine::Req {
    int64_t id = 0;
    GenRequest r; std::vector<float> prompt_own; bool want_pcm = false;
    int state = REQ_QUEUED; int slot = -1, cs = -1; int fed = 0;
    std::vector<int32_t> codes; std::vector<float> pcm;
    size_t pcm_enq = 0;   // samples enqueued by the decoder thread (its private cursor)
    size_t pcm_ready = 0; // samples whose decode has completed (harvested events)
    std::unique_ptr<Chunker> chunker;
    uint64_t seed = 0;
    hipEvent_t ev_admit = nullptr; bool first_seen = false;
    double t_submit = 0, t_admit = 0, t_done = 0, prefill_ms = 0, first_chunk_ms = 0;
    std::string error;
};

Engine::~Engine() {
    (void)hipSetDevice(dev_);
    try { stop_driver(); } catch (...) {}
    if (dec_started_) {
        { std::lock_guard<std::mutex> lk(dmu_); dec_stop_ = true; }
        dcv_.notify_all();
        dec_thread_.join();
    }
    for (auto& c : comp_) (void)hipEventDestroy(c.ev);
    for (auto& kv : reqs_) if (kv.second->ev_admit) (void)hipEventDestroy(kv.second->ev_admit);
    for (auto& g : graphs_) {
        if (g.second->exec) (void)hipGraphExecDestroy(g.second->exec);
        if (g.second->graph) (void)hipGraphDestroy(g.second->graph);
    }
    if (pcm_pinned_) (void)hipHostFree(pcm_pinned_);
    if (arena_) (void)hipHostFree(arena_);
    if (arena_pf_) (void)hipHostFree(arena_pf_);
    if (pf_done_) (void)hipEventDestroy(pf_done_);
    if (pf_e0_) (void)hipEventDestroy(pf_e0_);
    if (st_pf_) (void)hipStreamDestroy(st_pf_);
    for (auto s2 : st2_) (void)hipStreamDestroy(s2);
    if (st_) (void)hipStreamDestroy(st_);
}

size_t Engine::bytes_per_frame_step(int batch, double mean_ctx) const {
    // W_T + W_P (once) + B*ctx*KV bytes/token + B*(17 rows of 8 KB)
    const auto& t = talker_->hp();
    const size_t head_t = (size_t)tl_stride_ * talker_->head_bytes_per_row();
    const size_t head_p = (size_t)15 * Q3_CODEBOOK_SIZE * predictor_->head_bytes_per_row();
    const size_t kv_per_tok = (size_t)t.n_layer * 2 * t.n_kv * 128 * 2;
    return talker_->weight_bytes() + head_t + predictor_->weight_bytes() + head_p +
           (size_t)((double)batch * mean_ctx * (double)kv_per_tok) + (size_t)batch * 17 * 8192;
}

// One frame for fg.width slots (engine.rs:545-641).  code_0 of the frame is already in keys[b][0]: it was produced at the end
// of the previous frame (or of the prefill) by the talker head's argmax epilogue or by the device sampler.
void Engine::record_frame(FrameGraph& fg, bool sampled) {
    const int B = fg.width;
    const KvCache kvt = kv_t_->view(), kvp = kv_p_->view();
    // :565-573 predictor input = [project(m_hidden) ; project(E_0[code_0])]
    launch_project_blk(st_, d_thidden_.p, Q3_EMBD, d_proj_wblk_.p, d_proj_b_.p, Q3_EMBD, dP_, d_pin_.p, dP_, B);
    launch_gather_rows_keys(st_, d_proj_tab_[0].p, assets_->codec_rows[0], d_keys_.p, 16, dP_, d_pin_.p + (size_t)B * dP_, B);
    {   // :575-582 clear KV (= positions restart at 0) + 2-token prefill; :588-596 only slice q-1 of the 30720 logits is needed
        TokMeta tm{fg.seqA.p, fg.slotA.p, fg.posA.p};
        Transformer::Input in; in.x = d_pin_.p; in.x_stride = dP_;
        predictor_->set_same_seq_tokens(true);
        predictor_->forward(st_, in, 2 * B, tm, kvp, nullptr);
        ArgmaxEpi am{d_keys_.p + 1, 16, nullptr, 0};
        predictor_->head(st_, B, B, 0, Q3_CODEBOOK_SIZE, nullptr, 0, &am);
    }
    predictor_->set_same_seq_tokens(false);
    for (int q = 1; q < 15; q++) { // :602-610 decode project(E_q[code_q]) at pos q+1
        TokMeta tm{d_pseq_.p, d_pslot_.p + (size_t)(q + 1) * B_, d_ppos_.p + (size_t)(q + 1) * B_ * 4};
        KvCache kvq = kvp;
        if (pred_identity_pages_) { tm.uniform_pos = q + 1; kvq.page_table = nullptr; } // token t = sequence t = page t, position q+1 everywhere
        Transformer::Input in; in.x = d_proj_tab_[q].p; in.x_stride = dP_; in.idx_keys = d_keys_.p + q; in.idx_stride = 16;
        predictor_->forward(st_, in, B, tm, kvq, nullptr);
        ArgmaxEpi am{d_keys_.p + q + 1, 16, nullptr, 0};
        predictor_->head(st_, 0, B, q * Q3_CODEBOOK_SIZE, Q3_CODEBOOK_SIZE, nullptr, 0, &am);
    }
    // :622-631 feedback ; :633-639 talker step at pos = cur_pos ; :550-555 next frame's code_0
    launch_feedback_keys(st_, d_tab_ptrs_.p, d_tab_rows_.p, d_keys_.p, 16, d_tts_pad_.p, d_fb_.p, B);
    {
        TokMeta tm{d_tseq_.p, d_tslot_.p, d_tpos_.p};
        Transformer::Input in; in.x = d_fb_.p; in.x_stride = Q3_EMBD;
        talker_->set_same_seq_tokens(false);
        talker_->forward(st_, in, B, tm, kvt, nullptr);
        if (sampled) { // llama/mod.rs:666-776 on device; greedy slots of a mixed batch take the kernel's T<=0 branch
            talker_->head(st_, 0, B, 0, tl_stride_, d_tlogits_.p, tl_stride_, nullptr, -1, d_thidden_.p);
            SampleArgs sa{d_tlogits_.p, tl_stride_, Q3_SAMPLE_END, d_temp_.p, d_topk_.p, d_topp_.p, d_maskeos_.p, d_rngkey_.p, d_draws_.p, d_next_key0_.p, 1};
            launch_sample(st_, sa, B);
        } else {
            ArgmaxEpi am{d_next_key0_.p, 1, d_maskeos_.p, 0};
            talker_->head(st_, 0, B, 0, tl_stride_, nullptr, 0, &am, Q3_SAMPLE_END, d_thidden_.p);
        }
    }
    AdvanceKeysArgs a{B, d_finished_.p, d_nframes_.p, d_maxframes_.p, d_keys_.p, d_next_key0_.p, d_hist_.p, hist_stride_, d_tslot_.p, d_tpos_.p};
    launch_advance_keys(st_, a);
    Q3_LAUNCH_CHECK();
}

std::mutex& capture_mutex() { static std::mutex mu; return mu; }

Engine::FrameGraph& Engine::frame_graph(int width, bool sampled, bool capture) {
    std::unique_ptr<FrameGraph>& slot = graphs_[width * 2 + (sampled ? 1 : 0)];
    // one graph construction at a time per process: engines of a multi-device group (or several engines on one device) build their frame
    // graphs from different threads; the synchronous allocations / uploads below and concurrent thread-local captures were seen to
    // invalidate another thread's capture on ROCm 7.2 ("operation failed due to a previous error during capture")
    std::unique_lock<std::mutex> build_lock(capture_mutex(), std::defer_lock);
    if (!slot || (capture && !slot->exec)) build_lock.lock();
    if (!slot) {
        slot.reset(new FrameGraph());
        FrameGraph& fg = *slot;
        fg.width = width;
        const int B = width; // pass A = positions 0 and 1 of every slot: 2B tokens, slot b's pair stays in its own sequence
        std::vector<int32_t> seqA(2 * B), slotA(2 * B), posA((size_t)2 * B * 4);
        for (int b = 0; b < B; b++) {
            seqA[b] = b; seqA[B + b] = b; slotA[b] = 0; slotA[B + b] = 1;
            for (int s = 0; s < 4; s++) { posA[(size_t)b * 4 + s] = 0; posA[((size_t)B + b) * 4 + s] = 1; }
        }
        fg.seqA.alloc(2 * B); fg.seqA.upload(seqA.data(), 2 * B);
        fg.slotA.alloc(2 * B); fg.slotA.upload(slotA.data(), 2 * B);
        fg.posA.alloc(posA.size()); fg.posA.upload(posA.data(), posA.size());
    }
    FrameGraph& fg = *slot;
    if (capture && !fg.exec) {
        (void)hipGetLastError();
        Q3_HIP(hipStreamBeginCapture(st_, hipStreamCaptureModeThreadLocal));
        try { record_frame(fg, sampled); }
        catch (...) { hipGraph_t dead = nullptr; (void)hipStreamEndCapture(st_, &dead); if (dead) (void)hipGraphDestroy(dead); throw; }
        Q3_HIP(hipStreamEndCapture(st_, &fg.graph));
        Q3_HIP(hipGraphInstantiate(&fg.exec, fg.graph, nullptr, nullptr, 0));
    }
    return fg;
}

void Engine::upload_slot_state() {
    const int W = B_;
    std::vector<int32_t> tslot(W, 0), tpos((size_t)4 * W, 0), tseq(W, W); // idle slots step on the scratch sequence (row W)
    for (int b = 0; b < W; b++) {
        if (!slot_live_.empty() && slot_live_[b]) { const int t = h_nprompt_[b] + h_nfr_[b]; tslot[b] = t; tpos[4 * b] = tpos[4 * b + 1] = tpos[4 * b + 2] = t; tseq[b] = b; }
    }
    h2d(d_tseq_.p, tseq.data(), (size_t)W * 4);
    h2d(d_maxframes_.p, h_maxf_.data(), (size_t)W * 4); h2d(d_finished_.p, h_fin_.data(), (size_t)W * 4); h2d(d_nframes_.p, h_nfr_.data(), (size_t)W * 4);
    h2d(d_tslot_.p, tslot.data(), (size_t)W * 4); h2d(d_tpos_.p, tpos.data(), (size_t)16 * W); h2d(d_maskeos_.p, h_mask_.data(), (size_t)W * 4);
    h2d(d_temp_.p, h_temp_.data(), (size_t)W * 4); h2d(d_topk_.p, h_topk_.data(), (size_t)W * 4); h2d(d_topp_.p, h_topp_.data(), (size_t)W * 4);
    slot_dirty_ = false;
}

// Prefill (engine.rs:455-462) of newly admitted sequences: all prompts as ONE token stream, chunked by the talker's launch
// width; tokens of different sequences share a launch (per-token seq/slot/pos routing); each sequence's last token gets the
// head: code_0 of its first frame (:550-555) + the hidden row the predictor starts from (:565-566).
void Engine::prefill(const std::vector<Req*>& batch, bool sampled) {
    size_t total = 0;
    for (Req* r : batch) total += (size_t)r->r.n_prompt;
    const int chunk = talker_->max_tok();
    std::vector<int32_t> seq(chunk), slot(chunk), pos((size_t)4 * chunk);
    size_t bi = 0; int t = 0; // cursor over (request, token)
    size_t done = 0;
    talker_->set_same_seq_tokens(true);
    while (done < total) {
        int n = 0;
        std::vector<std::pair<Req*, int>> lasts; // (request, index inside this chunk) of final prompt tokens
        const size_t left = total - done;
        float* stage = (float*)stage_alloc(std::min<size_t>(left, (size_t)chunk) * Q3_EMBD * 4);
        while (n < chunk && bi < batch.size()) {
            Req* r = batch[bi];
            std::copy(r->r.prompt + (size_t)t * Q3_EMBD, r->r.prompt + (size_t)(t + 1) * Q3_EMBD, stage + (size_t)n * Q3_EMBD);
            seq[n] = r->slot; slot[n] = t; pos[4 * n] = pos[4 * n + 1] = pos[4 * n + 2] = t; pos[4 * n + 3] = 0; // :306-314
            if (t == r->r.n_prompt - 1) { lasts.emplace_back(r, n); bi++; t = 0; } else t++;
            n++;
        }
        Q3_HIP(hipMemcpyAsync(d_prompt_.p, stage, (size_t)n * Q3_EMBD * 4, hipMemcpyHostToDevice, st_));
        h2d(d_pf_seq_.p, seq.data(), (size_t)n * 4); h2d(d_pf_slot_.p, slot.data(), (size_t)n * 4); h2d(d_pf_pos_.p, pos.data(), (size_t)n * 16);
        TokMeta tm{d_pf_seq_.p, d_pf_slot_.p, d_pf_pos_.p};
        Transformer::Input in; in.x = d_prompt_.p; in.x_stride = Q3_EMBD;
        talker_->forward(st_, in, n, tm, kv_t_->view(), nullptr);
        for (auto& lb : lasts) {
            const int sb = lb.first->slot, ti = lb.second;
            if (sampled) {
                talker_->head(st_, ti, 1, 0, tl_stride_, d_tlogits_.p + (size_t)sb * tl_stride_, tl_stride_, nullptr, -1, d_thidden_.p + (size_t)sb * Q3_EMBD);
                SampleArgs sa{d_tlogits_.p + (size_t)sb * tl_stride_, tl_stride_, Q3_SAMPLE_END, d_temp_.p + sb, d_topk_.p + sb, d_topp_.p + sb,
                              d_maskeos_.p + sb, d_rngkey_.p + (size_t)sb * 8, d_draws_.p + sb, d_keys_.p + (size_t)sb * 16, 16};
                launch_sample(st_, sa, 1);
            } else {
                ArgmaxEpi am{d_keys_.p + (size_t)sb * 16, 16, d_maskeos_.p + sb, 0};
                talker_->head(st_, ti, 1, 0, tl_stride_, nullptr, 0, &am, Q3_SAMPLE_END, d_thidden_.p + (size_t)sb * Q3_EMBD);
            }
        }
        done += (size_t)n;
    }
}

// ---------------- decoder thread (the reference decodes on a second thread too, engine.rs:495-543): it owns every codec
// launch, so the ~200 kernel launches of a chunk never delay the AR stream's launches ----------------
void Engine::decoder_main() {
    try {
        Q3_HIP(hipSetDevice(dev_)); // HIP's current device is per thread
        const int spf = codec_->samples_per_frame(), gmax = codec_->max_group(), n_lanes = (int)st2_.size();
        std::map<Req*, std::pair<int, hipEvent_t>> last;            // request -> (lane, event after its latest chunk)
        int next_lane = 0;
        for (;;) {
            // take the oldest task; if it is a chunk, add the first pending chunk of other requests with the same frame count:
            // one decode pass then serves the whole group (weights streamed once, one set of launches)
            std::vector<DecTask> grp;
            {
                std::unique_lock<std::mutex> lk(dmu_);
                dcv_.wait(lk, [&] { return dec_stop_ || !dq_.empty(); });
                if (dq_.empty()) return;
                grp.push_back(std::move(dq_.front()));
                dq_.pop_front();
                if (!grp[0].fence && !grp[0].reset && gmax > 1) {
                    const size_t nc = grp[0].codes.size();
                    std::vector<Req*> seen{grp[0].r};
                    for (auto it = dq_.begin(); it != dq_.end() && (int)grp.size() < gmax;) {
                        const bool dup = std::find(seen.begin(), seen.end(), it->r) != seen.end();
                        if (!dup) seen.push_back(it->r);
                        if (!dup && !it->fence && !it->reset && it->codes.size() == nc) { grp.push_back(std::move(*it)); it = dq_.erase(it); }
                        else ++it;
                    }
                }
            }
            const int G = (int)grp.size();
            int lane = next_lane;
            { auto it = last.find(grp[0].r); if (G == 1 && it != last.end()) lane = it->second.first; else next_lane = (next_lane + 1) % n_lanes; }
            for (auto& t : grp) { // a request's chunks must run in order even when consecutive groups use different lanes
                auto it = last.find(t.r);
                if (it != last.end() && it->second.first != lane) Q3_HIP(hipStreamWaitEvent(st2_[lane], it->second.second, 0));
            }
            if (grp[0].reset) codec_->reset_async(st2_[lane], grp[0].r->cs); // AudioDecoder::create_state for a new request
            else if (!grp[0].fence) {
                const int nf = (int)grp[0].codes.size() / 16;
                std::vector<int> streams(G);
                std::vector<float*> dst(G);
                std::vector<int64_t> codes((size_t)G * nf * 16);
                for (int g = 0; g < G; g++) {
                    Req* r = grp[g].r;
                    Q3_CHECK(r->pcm_enq + (size_t)nf * spf <= slot_cap_, "pcm staging overflow");
                    streams[g] = r->cs; dst[g] = pcm_pinned_ + (size_t)r->cs * slot_cap_ + r->pcm_enq;
                    std::copy(grp[g].codes.begin(), grp[g].codes.end(), codes.begin() + (size_t)g * nf * 16);
                }
                // engine.rs:520: decode the chunk(s) -- enqueued on a codec stream, overlapping the next AR frames
                const int got = codec_->decode_group_async(st2_[lane], G, streams.data(), codes.data(), nf, dst.data(), lane);
                for (int g = 0; g < G; g++) grp[g].r->pcm_enq += (size_t)std::max(got, 0);
            }
            for (auto& t : grp) {
                hipEvent_t ev = nullptr;
                if (!t.reset) {
                    Q3_HIP(hipEventCreate(&ev));
                    Q3_HIP(hipEventRecord(ev, st2_[lane])); // first one = the reference's first stream_tx.send (:522-523)
                }
                if (t.fence) { auto it = last.find(t.r); if (it != last.end()) { (void)hipEventDestroy(it->second.second); last.erase(it); } }
                else {
                    hipEvent_t ord;
                    auto it = last.find(t.r);
                    if (it == last.end()) { Q3_HIP(hipEventCreateWithFlags(&ord, hipEventDisableTiming)); last[t.r] = {lane, ord}; }
                    else { ord = it->second.second; it->second.first = lane; }
                    Q3_HIP(hipEventRecord(ord, st2_[lane]));
                }
                if (t.reset) continue;
                std::lock_guard<std::mutex> lk(dmu_);
                comp_.push_back(Completion{t.r, ev, t.r->pcm_enq, t.fence});
                if (!t.fence) stats.codec_calls++;
            }
            dcv_.notify_all();
        }
    } catch (const std::exception& ex) {
        std::lock_guard<std::mutex> lk(dmu_);
        derr_ = ex.what();
        dcv_.notify_all();
    }
}

// collects finished codec work: PCM watermark per request, first-chunk latency, request completion
void Engine::harvest(bool block) {
    for (;;) {
        Completion c;
        {
            std::unique_lock<std::mutex> lk(dmu_);
            if (!derr_.empty()) throw Error("decoder thread: " + derr_);
            if (comp_.empty()) {
                if (!block) return;
                dcv_.wait_for(lk, std::chrono::milliseconds(2));
                if (comp_.empty()) return;
            }
            c = comp_.front();
            if (!block && hipEventQuery(c.ev) != hipSuccess) return; // completions of a lane finish in order; keep it simple: FIFO
            comp_.pop_front();
        }
        Q3_HIP(hipEventSynchronize(c.ev));
        Req* r = c.r;
        std::lock_guard<std::mutex> lk(mu_);
        r->pcm_ready = c.pcm_after;
        if (!r->first_seen && c.pcm_after > 0) {
            float ms = 0;
            Q3_HIP(hipEventElapsedTime(&ms, r->ev_admit, c.ev));
            r->first_chunk_ms = (r->t_admit - r->t_submit) + ms;
            r->first_seen = true;
        }
        (void)hipEventDestroy(c.ev);
        if (c.fence) {
            r->pcm.assign(pcm_pinned_ + (size_t)r->cs * slot_cap_, pcm_pinned_ + (size_t)r->cs * slot_cap_ + c.pcm_after);
            cs_free_.push_back(r->cs);
            r->cs = -1;
            r->state = REQ_DONE; r->t_done = now_ms();
            n_draining_--;
            cv_.notify_all();
        }
        block = false; // after one blocking wait, drain whatever else is ready
    }
}

int64_t Engine::submit(const GenRequest& g, bool want_pcm, bool copy_prompt) {
    Q3_CHECK(g.prompt && g.n_prompt >= 1 && g.n_prompt <= p_.max_prompt, "prompt length out of range");
    Q3_CHECK(g.max_steps >= 0 && g.max_steps <= p_.max_steps, "max_steps out of range");
    std::unique_ptr<Req> r(new Req());
    r->r = g;
    if (copy_prompt) { r->prompt_own.assign(g.prompt, g.prompt + (size_t)g.n_prompt * Q3_EMBD); r->r.prompt = r->prompt_own.data(); }
    r->want_pcm = want_pcm && codec_;
    r->t_submit = now_ms();
    r->seed = g.sampler.has_seed ? g.sampler.seed
                                 : (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::system_clock::now().time_since_epoch()).count(); // :473-478
    std::lock_guard<std::mutex> lk(mu_);
    r->id = next_id_++;
    Req* raw = r.get();
    reqs_[raw->id] = std::move(r);
    pending_.push_back(raw);
    cv_.notify_all();
    return raw->id;
}

// moves queued requests into free slots (lowest slot first, so the live slots stay packed and a narrow graph suffices)
void Engine::admit() {
    if (async_pf_ && !pf_batch_.empty()) return; // one admission wave in flight at a time
    std::vector<Req*> batch;
    {
        std::lock_guard<std::mutex> lk(mu_);
        // admissions are grouped under load: wait until an eighth of the slots is free -- unless the engine is idle or the free slots
        // already cover the whole queue (a synchronous prefill stalls every running sequence; an asynchronous one still shares the GPU)
        int reserved = 0;
        for (int b = 0; b < B_; b++) if (slot_req_[b]) reserved++;
        static const int quantum_div = [] { const char* e = std::getenv("Q3_ADMIT_DIV"); return e ? std::max(1, atoi(e)) : 8; }();
        const int free_slots = B_ - reserved, quantum = std::max(1, B_ / quantum_div);
        if (free_slots < quantum && n_active_ > 0 && free_slots < (int)pending_.size()) return;
        size_t rows = 0;
        while (!pending_.empty() && reserved < B_) {
            Req* r = pending_.front();
            if (r->want_pcm && cs_free_.empty()) break; // every codec stream still drains: admit after the next harvest
            if (async_pf_ && n_active_ > 0 && !batch.empty() && rows + (size_t)r->r.n_prompt > 4096) break; // staging arena of one asynchronous wave
            int slot = -1;
            for (int b = 0; b < B_; b++) if (!slot_req_[b]) { slot = b; break; }
            if (slot < 0) break;
            pending_.pop_front();
            r->slot = slot;
            if (r->want_pcm) { r->cs = cs_free_.back(); cs_free_.pop_back(); }
            r->state = REQ_RUNNING; r->t_admit = now_ms();
            slot_req_[slot] = r; reserved++;
            rows += (size_t)r->r.n_prompt;
            batch.push_back(r);
        }
    }
    if (batch.empty()) return;
    for (Req* r : batch) {
        if (r->want_pcm) { // the decoder thread clears the codec stream's state right before the request's first chunk
            { std::lock_guard<std::mutex> lk(dmu_); dq_.push_back(DecTask{r, {}, false, false, true}); }
            dcv_.notify_all();
        }
        r->chunker.reset(new Chunker([this, r](const int64_t* codes, int n_codes, bool is_final) {
            if (!r->want_pcm) return;
            { std::lock_guard<std::mutex> lk(dmu_); dq_.push_back(DecTask{r, std::vector<int64_t>(codes, codes + n_codes), is_final, false, false}); }
            dcv_.notify_all();
        }));
        Q3_HIP(hipEventCreate(&r->ev_admit));
    }
    if (async_pf_ && n_active_ > 0) { prefill_async(batch); return; } // sequences are running: do not stall them
    // ---- in-line form (nothing is running, or a single-slot engine): one batched prefill of the whole wave on the AR stream ----
    arena_used_ = 0; // st_ is idle between scheduler operations
    for (Req* r : batch) kv_t_->release(r->slot), kv_t_->ensure(r->slot, r->r.n_prompt + r->r.max_steps + 1, st_);
    pf_batch_ = batch;
    bool sampled = false;
    for (Req* r : batch) if (r->r.sampler.temperature > 0.0f) sampled = true;
    for (int b = 0; b < B_; b++) if (slot_live_[b] && slot_req_[b]->r.sampler.temperature > 0.0f) sampled = true;
    hipEvent_t e0, e1;
    Q3_HIP(hipEventCreate(&e0)); Q3_HIP(hipEventCreate(&e1));
    activate(); // slot state first: the prefill heads write code_0 / hidden straight into the live arrays
    Q3_HIP(hipEventRecord(e0, st_));
    for (Req* r : batch) Q3_HIP(hipEventRecord(r->ev_admit, st_)); // the AR stream is idle here: marks "admitted"
    prefill(batch, sampled);
    Q3_HIP(hipEventRecord(e1, st_));
    Q3_HIP(hipStreamSynchronize(st_));
    { float ms = 0; Q3_HIP(hipEventElapsedTime(&ms, e0, e1)); stats.prefill_ms += ms; }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    const double t = now_ms();
    std::lock_guard<std::mutex> lk(mu_);
    for (Req* r : batch) r->prefill_ms = t - r->t_submit;
}

// Prefill of one admission wave on the second talker instance / stream: every chunk is enqueued without host waits (the prompts of
// the wave sit in their own pinned arena); each sequence's last token leaves its logits row and hidden row in staging buffers.
void Engine::prefill_async(const std::vector<Req*>& batch) {
    size_t total = 0;
    for (Req* r : batch) {
        total += (size_t)r->r.n_prompt;
        kv_t_->release(r->slot);
        kv_t_->ensure(r->slot, r->r.n_prompt + r->r.max_steps + 1, st_pf_);
    }
    const int chunk = talker_pf_->max_tok();
    unsigned char* cur = arena_pf_;
    size_t bi = 0; int t = 0, done_tok = 0;
    Q3_HIP(hipEventRecord(pf_e0_, st_pf_));
    for (Req* r : batch) Q3_HIP(hipEventRecord(r->ev_admit, st_pf_));
    talker_pf_->set_same_seq_tokens(true);
    while ((size_t)done_tok < total) {
        const int n_max = (int)std::min<size_t>(total - done_tok, (size_t)chunk);
        float* stage = (float*)cur; cur += (size_t)n_max * Q3_EMBD * 4;
        int32_t* seq = (int32_t*)cur; cur += (size_t)n_max * 4;
        int32_t* slot = (int32_t*)cur; cur += (size_t)n_max * 4;
        int32_t* pos = (int32_t*)cur; cur += (size_t)n_max * 16;
        Q3_CHECK((size_t)(cur - arena_pf_) <= arena_pf_cap_, "prefill staging arena overflow");
        int n = 0;
        std::vector<std::pair<Req*, int>> lasts;
        while (n < n_max && bi < batch.size()) {
            Req* r = batch[bi];
            std::copy(r->r.prompt + (size_t)t * Q3_EMBD, r->r.prompt + (size_t)(t + 1) * Q3_EMBD, stage + (size_t)n * Q3_EMBD);
            seq[n] = r->slot; slot[n] = t; pos[4 * n] = pos[4 * n + 1] = pos[4 * n + 2] = t; pos[4 * n + 3] = 0; // engine.rs:306-314
            if (t == r->r.n_prompt - 1) { lasts.emplace_back(r, n); bi++; t = 0; } else t++;
            n++;
        }
        Q3_HIP(hipMemcpyAsync(d_prompt_pf_.p, stage, (size_t)n * Q3_EMBD * 4, hipMemcpyHostToDevice, st_pf_));
        Q3_HIP(hipMemcpyAsync(d_pfa_seq_.p, seq, (size_t)n * 4, hipMemcpyHostToDevice, st_pf_));
        Q3_HIP(hipMemcpyAsync(d_pfa_slot_.p, slot, (size_t)n * 4, hipMemcpyHostToDevice, st_pf_));
        Q3_HIP(hipMemcpyAsync(d_pfa_pos_.p, pos, (size_t)n * 16, hipMemcpyHostToDevice, st_pf_));
        TokMeta tm{d_pfa_seq_.p, d_pfa_slot_.p, d_pfa_pos_.p};
        Transformer::Input in; in.x = d_prompt_pf_.p; in.x_stride = Q3_EMBD;
        talker_pf_->forward(st_pf_, in, n, tm, kv_t_->view(), nullptr);
        for (auto& lb : lasts) { // logits + hidden of the LAST prompt token (engine.rs:550-554,565-566); code_0 is drawn at activation
            const int sb = lb.first->slot;
            talker_pf_->head(st_pf_, lb.second, 1, 0, tl_stride_, d_pf_logits_.p + (size_t)sb * tl_stride_, tl_stride_, nullptr, -1, d_pf_hidden_.p + (size_t)sb * Q3_EMBD);
        }
        done_tok += n;
    }
    Q3_HIP(hipEventRecord(pf_done_, st_pf_));
    pf_batch_ = batch;
    n_prefilling_ = (int)batch.size();
}

// The admission wave joins the frame graph: slot state (positions, sampler parameters, RNG stream, frame counters) goes live, and --
// asynchronous form -- the staged hidden row is copied in and code_0 of the first frame is drawn from the staged logits with the
// slot's own sampler state (greedy slots take the sampler kernel's first-max branch).  Runs on st_ between two frame groups.
void Engine::activate() {
    if (pf_batch_.empty()) return;
    std::vector<Req*> batch;
    batch.swap(pf_batch_);
    arena_used_ = 0; // st_ is idle between scheduler operations
    const q3_u64 armed[17] = {pack_key(-INFINITY, 0), pack_key(-INFINITY, 0), pack_key(-INFINITY, 0), pack_key(-INFINITY, 0), pack_key(-INFINITY, 0),
                              pack_key(-INFINITY, 0), pack_key(-INFINITY, 0), pack_key(-INFINITY, 0), pack_key(-INFINITY, 0), pack_key(-INFINITY, 0),
                              pack_key(-INFINITY, 0), pack_key(-INFINITY, 0), pack_key(-INFINITY, 0), pack_key(-INFINITY, 0), pack_key(-INFINITY, 0),
                              pack_key(-INFINITY, 0), pack_key(-INFINITY, 0)};
    const uint32_t zero = 0;
    for (Req* r : batch) {
        const int b = r->slot;
        h_maxf_[b] = r->r.max_steps; h_fin_[b] = 0; h_nfr_[b] = 0; h_nprompt_[b] = r->r.n_prompt;
        h_mask_[b] = r->r.mask_eos ? Q3_CODEC_EOS : -1;
        h_temp_[b] = r->r.sampler.temperature; h_topk_[b] = r->r.sampler.top_k; h_topp_[b] = r->r.sampler.top_p;
        StdRng rng(r->seed); // llama/mod.rs:648: the device regenerates this ChaCha12 stream from the key and a draw counter
        h2d(d_rngkey_.p + (size_t)b * 8, rng.key(), 32);
        h2d(d_draws_.p + b, &zero, 4);
        h2d(d_keys_.p + (size_t)b * 16, armed, 16 * 8);
        h2d(d_next_key0_.p + b, armed, 8);
        slot_live_[b] = 1;
    }
    upload_slot_state();
    if (n_prefilling_ > 0) { // the wave was prefilled on the second talker instance
        float ms = 0;
        Q3_HIP(hipEventElapsedTime(&ms, pf_e0_, pf_done_));
        stats.prefill_ms += ms;
        for (Req* r : batch) {
            const int b = r->slot;
            Q3_HIP(hipMemcpyAsync(d_thidden_.p + (size_t)b * Q3_EMBD, d_pf_hidden_.p + (size_t)b * Q3_EMBD, (size_t)Q3_EMBD * 4, hipMemcpyDeviceToDevice, st_));
            SampleArgs sa{d_pf_logits_.p + (size_t)b * tl_stride_, tl_stride_, Q3_SAMPLE_END, d_temp_.p + b, d_topk_.p + b, d_topp_.p + b,
                          d_maskeos_.p + b, d_rngkey_.p + (size_t)b * 8, d_draws_.p + b, d_keys_.p + (size_t)b * 16, 16};
            launch_sample(st_, sa, 1);
        }
        Q3_HIP(hipStreamSynchronize(st_));
        n_prefilling_ = 0;
        const double t = now_ms();
        std::lock_guard<std::mutex> lk(mu_);
        for (Req* r : batch) r->prefill_ms = t - r->t_submit;
        n_active_ += (int)batch.size();
    } else {
        std::lock_guard<std::mutex> lk(mu_);
        n_active_ += (int)batch.size();
    }
}

void Engine::finish_ar(Req* r) { // engine.rs:644-649: final flush of the chunker, then the decoder drains
    r->chunker->push(nullptr, 0, true);
    const int b = r->slot;
    {
        std::lock_guard<std::mutex> lk(mu_);
        slot_req_[b] = nullptr; slot_live_[b] = 0; n_active_--;
        r->slot = -1;
        if (r->want_pcm) { r->state = REQ_DRAINING; n_draining_++; }
        else { r->state = REQ_DONE; r->t_done = now_ms(); cv_.notify_all(); }
    }
    if (r->want_pcm) {
        { std::lock_guard<std::mutex> lk(dmu_); dq_.push_back(DecTask{r, {}, false, true, false}); }
        dcv_.notify_all();
    }
    h_fin_[b] = 1; h_maxf_[b] = 0; h_nfr_[b] = 0; h_nprompt_[b] = 0; h_temp_[b] = 0.0f; h_mask_[b] = -1;
    kv_t_->release(b); // from the next group on the slot steps on the scratch sequence (upload_slot_state)
    slot_dirty_ = true;
}

static int pick_width(int hi, int W) { // graph widths: powers of two (and the full width)
    int w = 1;
    while (w < hi) w <<= 1;
    return std::min(w, W);
}

// one streaming step (4 frames, engine.rs:505-512) for every active slot
void Engine::run_group() {
    int hi = 0, remaining = 0;
    bool sampled = false;
    for (int b = 0; b < B_; b++)
        if (slot_live_[b]) {
            Req* r = slot_req_[b];
            hi = b + 1;
            remaining = std::max(remaining, r->r.max_steps - h_nfr_[b]);
            if (r->r.sampler.temperature > 0.0f) sampled = true;
        }
    arena_used_ = 0; // st_ is idle between scheduler operations: everything staged earlier has been consumed
    if (slot_dirty_) upload_slot_state();
    const int width = pick_width(hi, B_);
    const int group = std::max(1, std::min(4, remaining));
    const bool eager = instrument_ || !p_.use_graph;
    FrameGraph& fg = frame_graph(width, sampled, !eager);
    if (instrument_) { talker_->timer = &timer_; predictor_->timer = &timer_; talker_->timer_gu = &timer_gu_; }
    hipEvent_t e0, e1;
    Q3_HIP(hipEventCreate(&e0)); Q3_HIP(hipEventCreate(&e1));
    Q3_HIP(hipEventRecord(e0, st_));
    for (int g = 0; g < group; g++) {
        if (eager) record_frame(fg, sampled);
        else Q3_HIP(hipGraphLaunch(fg.exec, st_));
    }
    Q3_HIP(hipEventRecord(e1, st_));
    // results of the group: frame counters, finished flags and the (at most `group`) new code rows of every active slot
    const int32_t* p_nfr = (const int32_t*)d2h_begin(d_nframes_.p, (size_t)width * 4);
    const int32_t* p_fin = (const int32_t*)d2h_begin(d_finished_.p, (size_t)width * 4);
    std::vector<const int32_t*> p_hist(width, nullptr);
    for (int b = 0; b < width; b++)
        if (slot_live_[b]) {
            Req* r = slot_req_[b];
            const int rows = std::min(group, r->r.max_steps - r->fed);
            if (rows > 0) p_hist[b] = (const int32_t*)d2h_begin(d_hist_.p + (size_t)b * hist_stride_ + (size_t)r->fed * 16, (size_t)rows * 64);
        }
    Q3_HIP(hipStreamSynchronize(st_));
    { float ms = 0; Q3_HIP(hipEventElapsedTime(&ms, e0, e1)); stats.frame_loop_ms += ms; }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    stats.steps++; stats.slot_frames += (double)width * group; stats.graph_frames += group;
    talker_->timer = nullptr; predictor_->timer = nullptr; talker_->timer_gu = nullptr;
    if (instrument_) {
        stats.gemv_ms += timer_.collect_ms(); stats.gemv_bytes += timer_.bytes; stats.gemv_launches += timer_.launches; timer_.bytes = 0; timer_.launches = 0;
        stats.gu_ms += timer_gu_.collect_ms(); stats.gu_bytes += timer_gu_.bytes; stats.gu_launches += timer_gu_.launches; timer_gu_.bytes = 0; timer_gu_.launches = 0;
    }
    std::memcpy(h_nfr_.data(), p_nfr, (size_t)width * 4); std::memcpy(h_fin_.data(), p_fin, (size_t)width * 4);
    for (int b = 0; b < width; b++) {
        if (!slot_live_[b]) continue;
        Req* r = slot_req_[b];
        if (h_nfr_[b] > r->fed) { // hand new frames to the chunker (engine.rs:613-620)
            const int nnew = h_nfr_[b] - r->fed;
            const int32_t* hbuf = p_hist[b];
            { std::lock_guard<std::mutex> lk(mu_); r->codes.insert(r->codes.end(), hbuf, hbuf + (size_t)nnew * 16); r->fed = h_nfr_[b]; }
            for (int f = 0; f < nnew; f++) {
                int64_t fc[16];
                for (int q = 0; q < 16; q++) fc[q] = hbuf[(size_t)f * 16 + q];
                r->chunker->push(fc, 16, false);
            }
            stats.frames += nnew;
        }
        if (h_fin_[b] || h_nfr_[b] >= r->r.max_steps) finish_ar(r);
    }
}

static double g_t_harvest = 0, g_t_admit = 0, g_t_group = 0, g_t_idle = 0;
static const bool g_trace = std::getenv("Q3_SCHED_TRACE") != nullptr;
bool Engine::step() {
    std::lock_guard<std::mutex> step_lock(step_mu_);
    Q3_HIP(hipSetDevice(dev_)); // HIP's current device is per thread: any thread may drive any engine of a multi-device group
    if (codec_ && !dec_started_) { dec_started_ = true; dec_thread_ = std::thread([this] { decoder_main(); }); }
    const double t0 = now_ms();
    harvest(false);
    const double t1 = now_ms();
    if (async_pf_ && n_prefilling_ > 0) { // has the admission wave finished its prefill?  (wait for it when nothing else can run)
        if (n_active_ == 0) Q3_HIP(hipEventSynchronize(pf_done_));
        if (hipEventQuery(pf_done_) == hipSuccess) activate();
    }
    admit();
    const double t2 = now_ms();
    if (n_active_ > 0) run_group();
    else if (n_prefilling_ > 0) { Q3_HIP(hipEventSynchronize(pf_done_)); activate(); }
    else if (n_draining_ > 0) harvest(true);
    const double t3 = now_ms();
    if (g_trace) {
        g_t_harvest += t1 - t0; g_t_admit += t2 - t1; (n_active_ > 0 ? g_t_group : g_t_idle) += t3 - t2;
        fprintf(stderr, "[sched] harvest %.1f admit %.1f group %.1f drain-wait %.1f ms (frame_loop %.1f prefill %.1f)\n", g_t_harvest, g_t_admit, g_t_group, g_t_idle,
                stats.frame_loop_ms, stats.prefill_ms);
    }
    std::lock_guard<std::mutex> lk(mu_);
    return n_active_ > 0 || n_prefilling_ > 0 || n_draining_ > 0 || !pending_.empty();
}

ReqStatus Engine::poll(int64_t id) {
    std::lock_guard<std::mutex> lk(mu_);
    auto it = reqs_.find(id);
    Q3_CHECK(it != reqs_.end(), "unknown request id");
    const Req& r = *it->second;
    ReqStatus s;
    s.state = r.state; s.n_frames = r.fed; s.n_pcm = (int64_t)(r.state == REQ_DONE ? r.pcm.size() : r.pcm_ready);
    s.queue_ms = r.t_admit > 0 ? r.t_admit - r.t_submit : now_ms() - r.t_submit;
    s.prefill_ms = r.prefill_ms; s.first_chunk_ms = r.first_chunk_ms;
    s.total_ms = (r.state == REQ_DONE ? r.t_done : now_ms()) - r.t_submit;
    s.error = r.error;
    return s;
}

void Engine::fetch(int64_t id, int32_t* codes, int frame_off, int max_frames, float* pcm, int64_t pcm_off, int64_t pcm_cap, int* got_frames, int64_t* got_pcm) {
    std::lock_guard<std::mutex> lk(mu_);
    auto it = reqs_.find(id);
    Q3_CHECK(it != reqs_.end(), "unknown request id");
    const Req& r = *it->second;
    int nf = 0; int64_t np = 0;
    if (codes && frame_off >= 0 && frame_off < r.fed) {
        nf = std::min(max_frames, r.fed - frame_off);
        std::copy(r.codes.begin() + (size_t)frame_off * 16, r.codes.begin() + (size_t)(frame_off + nf) * 16, codes);
    }
    if (pcm && pcm_off >= 0) {
        const bool done = r.state == REQ_DONE;
        const int64_t avail = (int64_t)(done ? r.pcm.size() : r.pcm_ready);
        const float* src = done ? r.pcm.data() : (r.cs >= 0 ? pcm_pinned_ + (size_t)r.cs * slot_cap_ : nullptr);
        if (src && pcm_off < avail) { np = std::min(pcm_cap, avail - pcm_off); std::copy(src + pcm_off, src + pcm_off + np, pcm); }
    }
    if (got_frames) *got_frames = nf;
    if (got_pcm) *got_pcm = np;
}

bool Engine::wait(int64_t id, double timeout_ms) {
    const double t0 = now_ms();
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(mu_);
            auto it = reqs_.find(id);
            Q3_CHECK(it != reqs_.end(), "unknown request id");
            if (it->second->state == REQ_DONE || it->second->state == REQ_FAILED) return true;
            if (timeout_ms >= 0 && now_ms() - t0 > timeout_ms) return false;
            if (driver_on_) { cv_.wait_for(lk, std::chrono::milliseconds(5)); continue; }
        }
        step();
    }
}

void Engine::release(int64_t id) {
    std::lock_guard<std::mutex> lk(mu_);
    auto it = reqs_.find(id);
    Q3_CHECK(it != reqs_.end(), "unknown request id");
    Q3_CHECK(it->second->state == REQ_DONE || it->second->state == REQ_FAILED, "request still in flight");
    if (it->second->ev_admit) (void)hipEventDestroy(it->second->ev_admit);
    reqs_.erase(it);
}

void Engine::start_driver() {
    std::lock_guard<std::mutex> lk(mu_);
    if (driver_on_) return;
    driver_on_ = true; driver_stop_ = false;
    driver_ = std::thread([this] {
        try {
            Q3_HIP(hipSetDevice(dev_));
            for (;;) {
                const bool busy = step();
                std::unique_lock<std::mutex> lk(mu_);
                if (driver_stop_) return;
                if (!busy) cv_.wait_for(lk, std::chrono::milliseconds(1), [this] { return driver_stop_ || !pending_.empty(); });
            }
        } catch (const std::exception& ex) {
            std::lock_guard<std::mutex> lk(mu_);
            for (auto& kv : reqs_) if (kv.second->state != REQ_DONE) { kv.second->state = REQ_FAILED; kv.second->error = ex.what(); }
            cv_.notify_all();
        }
    });
}
void Engine::stop_driver() {
    {
        std::lock_guard<std::mutex> lk(mu_);
        if (!driver_on_) return;
        driver_stop_ = true;
        cv_.notify_all();
    }
    driver_.join();
    std::lock_guard<std::mutex> lk(mu_);
    driver_on_ = false;
}

int Engine::register_voice(const Voice& v) {
    Q3_CHECK(v.spk_emb.size() == (size_t)Q3_EMBD, "speaker embedding must have 2048 values");
    std::lock_guard<std::mutex> lk(mu_);
    voices_.push_back(v);
    return (int)voices_.size() - 1;
}
const Voice& Engine::voice(int id) const {
    Q3_CHECK(id >= 0 && id < (int)voices_.size(), "unknown voice id");
    return voices_[id];
}

void Engine::generate_batch(const std::vector<GenRequest>& reqs, std::vector<GenResult>& out, bool want_pcm) {
    const int n = (int)reqs.size();
    Q3_CHECK(n >= 1, "no requests");
    Q3_CHECK(!driver_on_, "generate_batch cannot be mixed with a running scheduler thread");
    out.assign(n, GenResult());
    for (const GenRequest& g : reqs) { // validate everything before anything is queued
        Q3_CHECK(g.prompt && g.n_prompt >= 1 && g.n_prompt <= p_.max_prompt, "prompt length out of range");
        Q3_CHECK(g.max_steps >= 0 && g.max_steps <= p_.max_steps, "max_steps out of range");
    }
    std::vector<int64_t> ids(n);
    for (int i = 0; i < n; i++) ids[i] = submit(reqs[i], want_pcm, false);
    double t_ar_end = 0;
    while (step()) {
        if (t_ar_end == 0 && n_active_ == 0) { std::lock_guard<std::mutex> lk(mu_); if (pending_.empty()) t_ar_end = now_ms(); }
    }
    if (t_ar_end > 0) stats.codec_ms += now_ms() - t_ar_end; // only the part of the codec work the AR loop did not hide
    for (int i = 0; i < n; i++) {
        {
            std::lock_guard<std::mutex> lk(mu_);
            Req& r = *reqs_.at(ids[i]);
            Q3_CHECK(r.state == REQ_DONE, "request did not finish");
            out[i].codes = std::move(r.codes); out[i].n_frames = r.fed; out[i].pcm = std::move(r.pcm);
            out[i].prefill_ms = r.prefill_ms; out[i].first_chunk_ms = r.first_chunk_ms; out[i].total_ms = r.t_done - r.t_submit;
        }
        release(ids[i]);
    }
}

} // namespace q3
