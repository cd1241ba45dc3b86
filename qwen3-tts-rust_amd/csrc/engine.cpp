// engine.cpp -- see engine.h.  Loop structure cites /root/reference/src/tts/engine.rs line numbers.
#include "engine.h"
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include "kdev.h"

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
    Q3_HIP(hipStreamCreate(&st_));
    const std::string dir = p.model_dir + "/" + quant_dir(p.quant);
    assets_.reset(new HostAssets(dir + "/qwen3_assets.gguf"));
    const int B = p.max_batch;
    B_ = B;
    talker_.reset(new Transformer(dir + "/qwen3_tts_talker.gguf", Q3_TALKER_NCTX, std::max(256, B)));
    predictor_.reset(new Transformer(dir + "/qwen3_tts_predictor.gguf", Q3_PRED_NCTX, 2 * B));
    // experiment switch: fold the predictor's attention into its o-proj launch (k_oproj_attn).  Measured on MI355X: 3.23 vs
    // 3.07 ms/frame -- the single-wave attention chain costs more than the launch it removes, so it stays off by default.
    if (const char* e = std::getenv("Q3_FOLD_ATTN")) predictor_->set_short_context(e[0] == '1');
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
    kv_t_.reset(new KvPool(talker_->hp().n_layer, talker_->hp().n_kv, B * pages_per_seq, B, pages_per_seq));
    kv_p_.reset(new KvPool(predictor_->hp().n_layer, predictor_->hp().n_kv, B, B, 1));
    for (int b = 0; b < B; b++) { kv_p_->ensure(b, 16); kv_t_->ensure(b, 1); }
    // ---- per-sequence state ----
    hist_stride_ = p.max_steps * 16;
    tl_stride_ = (Q3_SAMPLE_END + 31) & ~31;
    d_tseq_.alloc(B); d_tslot_.alloc(B); d_tpos_.alloc(4 * B); d_keys_.alloc(16 * B); d_next_key0_.alloc(B); d_nframes_.alloc(B); d_finished_.alloc(B);
    d_maxframes_.alloc(B); d_maskeos_.alloc(B); d_hist_.alloc((size_t)B * hist_stride_);
    d_tlogits_.alloc((size_t)B * tl_stride_); d_thidden_.alloc((size_t)B * Q3_EMBD); d_pin_.alloc((size_t)2 * B * dP_);
    d_plogits_.alloc((size_t)B * Q3_CODEBOOK_SIZE); d_fb_.alloc((size_t)B * Q3_EMBD);
    d_prompt_.alloc((size_t)p.max_prompt * Q3_EMBD); d_hid_all_.alloc((size_t)talker_->max_tok() * Q3_EMBD);
    d_pf_seq_.alloc(talker_->max_tok()); d_pf_slot_.alloc(talker_->max_tok()); d_pf_pos_.alloc(4 * (size_t)talker_->max_tok());
    d_hist_.zero(); d_tlogits_.zero(); d_thidden_.zero();
    // predictor routing: pass i (i = 1..15) handles position i for every sequence; pass A = positions 0 and 1 (2B tokens)
    std::vector<int32_t> seq(B), slot((size_t)16 * B), pos((size_t)16 * B * 4), seqA(2 * B), slotA(2 * B), posA((size_t)2 * B * 4);
    for (int b = 0; b < B; b++) {
        seq[b] = b;
        for (int i = 0; i < 16; i++) { slot[(size_t)i * B + b] = i; for (int s = 0; s < 4; s++) pos[((size_t)i * B + b) * 4 + s] = i; } // engine.rs:316-318
        seqA[b] = b; seqA[B + b] = b; slotA[b] = 0; slotA[B + b] = 1;
        for (int s = 0; s < 4; s++) { posA[(size_t)b * 4 + s] = 0; posA[((size_t)B + b) * 4 + s] = 1; }
    }
    d_pseq_.alloc(B); d_pseq_.upload(seq.data(), B);
    d_pslot_.alloc(slot.size()); d_pslot_.upload(slot.data(), slot.size());
    d_ppos_.alloc(pos.size()); d_ppos_.upload(pos.data(), pos.size());
    d_pseqA_.alloc(2 * B); d_pseqA_.upload(seqA.data(), 2 * B);
    d_pslotA_.alloc(2 * B); d_pslotA_.upload(slotA.data(), 2 * B);
    d_pposA_.alloc(posA.size()); d_pposA_.upload(posA.data(), posA.size());
    d_tseq_.upload(seq.data(), B);
    if (p.load_codec) {
        const int n_lanes = std::min(B, 8);
        codec_.reset(new CodecDecoder(p.model_dir + "/onnx/q3tts_codec.gguf", B, 4, n_lanes));
        st2_.resize(n_lanes);
        for (auto& s2 : st2_) Q3_HIP(hipStreamCreate(&s2));
        pcm_pinned_cap_ = (size_t)B * p.max_steps * codec_->samples_per_frame();
        Q3_HIP(hipHostMalloc((void**)&pcm_pinned_, pcm_pinned_cap_ * sizeof(float)));
    }
    Q3_HIP(hipStreamSynchronize(st_));
}

Engine::~Engine() {
    if (graph_exec_) (void)hipGraphExecDestroy(graph_exec_);
    if (graph_) (void)hipGraphDestroy(graph_);
    for (auto e : ev_pool_) (void)hipEventDestroy(e);
    if (pcm_pinned_) (void)hipHostFree(pcm_pinned_);
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

// One frame for B lock-stepped sequences (engine.rs:545-641).  code_0 of the frame is already in keys[b][0]: it was
// produced by the argmax epilogue of the talker head that ended the previous frame (or the prefill), or by the host sampler.
void Engine::record_frame(int B) {
    const KvCache kvt = kv_t_->view(), kvp = kv_p_->view();
    // :565-573 predictor input = [project(m_hidden) ; project(E_0[code_0])]
    launch_project_blk(st_, d_thidden_.p, Q3_EMBD, d_proj_wblk_.p, d_proj_b_.p, Q3_EMBD, dP_, d_pin_.p, dP_, B);
    launch_gather_rows_keys(st_, d_proj_tab_[0].p, assets_->codec_rows[0], d_keys_.p, 16, dP_, d_pin_.p + (size_t)B * dP_, B);
    {   // :575-582 clear KV (= positions restart at 0) + 2-token prefill; :588-596 only slice q-1 of the 30720 logits is needed
        TokMeta tm{d_pseqA_.p, d_pslotA_.p, d_pposA_.p};
        Transformer::Input in; in.x = d_pin_.p; in.x_stride = dP_;
        predictor_->set_same_seq_tokens(true);
        predictor_->forward(st_, in, 2 * B, tm, kvp, nullptr);
        ArgmaxEpi am{d_keys_.p + 1, 16, nullptr, 0};
        predictor_->head(st_, B, B, 0, Q3_CODEBOOK_SIZE, nullptr, 0, &am);
    }
    predictor_->set_same_seq_tokens(false);
    for (int q = 1; q < 15; q++) { // :602-610 decode project(E_q[code_q]) at pos q+1
        TokMeta tm{d_pseq_.p, d_pslot_.p + (size_t)(q + 1) * B, d_ppos_.p + (size_t)(q + 1) * B * 4};
        Transformer::Input in; in.x = d_proj_tab_[q].p; in.x_stride = dP_; in.idx_keys = d_keys_.p + q; in.idx_stride = 16;
        predictor_->forward(st_, in, B, tm, kvp, nullptr);
        ArgmaxEpi am{d_keys_.p + q + 1, 16, nullptr, 0};
        predictor_->head(st_, 0, B, q * Q3_CODEBOOK_SIZE, Q3_CODEBOOK_SIZE, nullptr, 0, &am);
    }
    // :622-631 feedback ; :633-639 talker step at pos = cur_pos ; :550-555 next frame's code_0 (greedy branch)
    launch_feedback_keys(st_, d_tab_ptrs_.p, d_tab_rows_.p, d_keys_.p, 16, d_tts_pad_.p, d_fb_.p, B);
    {
        TokMeta tm{d_tseq_.p, d_tslot_.p, d_tpos_.p};
        Transformer::Input in; in.x = d_fb_.p; in.x_stride = Q3_EMBD;
        talker_->set_same_seq_tokens(false);
        talker_->forward(st_, in, B, tm, kvt, nullptr);
        if (code0_given_) talker_->head(st_, 0, B, 0, tl_stride_, d_tlogits_.p, tl_stride_, nullptr, -1, d_thidden_.p);
        else {
            ArgmaxEpi am{d_next_key0_.p, 1, d_maskeos_.p, 0};
            talker_->head(st_, 0, B, 0, tl_stride_, nullptr, 0, &am, Q3_SAMPLE_END, d_thidden_.p);
        }
    }
    AdvanceKeysArgs a{B, d_finished_.p, d_nframes_.p, d_maxframes_.p, d_keys_.p, d_next_key0_.p, d_hist_.p, hist_stride_, d_tslot_.p, d_tpos_.p};
    launch_advance_keys(st_, a);
}

void Engine::build_graph(int B) {
    if (graph_exec_ && graph_B_ == B && graph_given_ == code0_given_) return;
    if (graph_exec_) { (void)hipGraphExecDestroy(graph_exec_); graph_exec_ = nullptr; }
    if (graph_) { (void)hipGraphDestroy(graph_); graph_ = nullptr; }
    Q3_HIP(hipStreamBeginCapture(st_, hipStreamCaptureModeThreadLocal));
    record_frame(B);
    Q3_HIP(hipStreamEndCapture(st_, &graph_));
    Q3_HIP(hipGraphInstantiate(&graph_exec_, graph_, nullptr, nullptr, 0));
    graph_B_ = B; graph_given_ = code0_given_;
}

void Engine::generate_batch(const std::vector<GenRequest>& reqs, std::vector<GenResult>& out, bool want_pcm) {
    const int B = (int)reqs.size();
    Q3_CHECK(B >= 1 && B <= B_, "batch size exceeds max_batch");
    out.assign(B, GenResult());
    const double t0 = now_ms();
    // the graph is captured for the engine's full batch width; unused slots idle as finished sequences
    const int W = B_;
    bool any_sampled = false;
    int max_steps_all = 0;
    std::vector<int32_t> maxf(W, 0), mask(W, -1), fin(W, 1), nfr(W, 0), tslot(W, 0), tpos((size_t)4 * W, 0);
    hipEvent_t ev0, ev1;
    Q3_HIP(hipEventCreate(&ev0)); Q3_HIP(hipEventCreate(&ev1));
    Q3_HIP(hipEventRecord(ev0, st_));
    bool any_sampled_req = false;
    for (int b = 0; b < B; b++) { mask[b] = reqs[b].mask_eos ? Q3_CODEC_EOS : -1; if (reqs[b].sampler.temperature > 0.0f) any_sampled_req = true; }
    d_maskeos_.upload(mask.data(), W);
    {
        std::vector<q3_u64> k0((size_t)16 * W, pack_key(-INFINITY, 0)), n0(W, pack_key(-INFINITY, 0));
        d_keys_.upload(k0.data(), k0.size()); d_next_key0_.upload(n0.data(), n0.size());
    }
    // ---------------- prefill (engine.rs:455-462): all prompts as ONE token stream, chunked by the talker's launch width ----------------
    // tokens of different sequences share a launch (per-token seq/slot/pos routing); each sequence's last token gets the head.
    {
        size_t total = 0;
        for (int b = 0; b < B; b++) {
            const GenRequest& r = reqs[b];
            Q3_CHECK(r.n_prompt >= 1 && r.n_prompt <= p_.max_prompt, "prompt length out of range");
            Q3_CHECK(r.max_steps >= 0 && r.max_steps <= p_.max_steps, "max_steps out of range");
            kv_t_->release(b);
            kv_t_->ensure(b, r.n_prompt + r.max_steps + 1);
            total += (size_t)r.n_prompt;
            maxf[b] = r.max_steps; fin[b] = 0;
            tslot[b] = r.n_prompt; tpos[4 * b] = tpos[4 * b + 1] = tpos[4 * b + 2] = r.n_prompt;
            max_steps_all = std::max(max_steps_all, r.max_steps);
            if (r.sampler.temperature > 0.0f) any_sampled = true;
        }
        const int chunk = talker_->max_tok();
        if (d_prompt_.n < (size_t)chunk * Q3_EMBD) d_prompt_.alloc((size_t)chunk * Q3_EMBD);
        std::vector<float> stage((size_t)chunk * Q3_EMBD);
        std::vector<int32_t> seq(chunk), slot(chunk), pos((size_t)4 * chunk);
        int b = 0, t = 0; // cursor over (sequence, token)
        size_t done = 0;
        talker_->set_same_seq_tokens(true);
        while (done < total) {
            int n = 0;
            std::vector<std::pair<int, int>> lasts; // (sequence, index inside this chunk) of final prompt tokens
            while (n < chunk && b < B) {
                const GenRequest& r = reqs[b];
                std::copy(r.prompt + (size_t)t * Q3_EMBD, r.prompt + (size_t)(t + 1) * Q3_EMBD, stage.begin() + (size_t)n * Q3_EMBD);
                seq[n] = b; slot[n] = t; pos[4 * n] = pos[4 * n + 1] = pos[4 * n + 2] = t; pos[4 * n + 3] = 0; // :306-314
                if (t == r.n_prompt - 1) { lasts.emplace_back(b, n); b++; t = 0; } else t++;
                n++;
            }
            Q3_HIP(hipMemcpyAsync(d_prompt_.p, stage.data(), (size_t)n * Q3_EMBD * 4, hipMemcpyHostToDevice, st_));
            Q3_HIP(hipMemcpyAsync(d_pf_seq_.p, seq.data(), (size_t)n * 4, hipMemcpyHostToDevice, st_));
            Q3_HIP(hipMemcpyAsync(d_pf_slot_.p, slot.data(), (size_t)n * 4, hipMemcpyHostToDevice, st_));
            Q3_HIP(hipMemcpyAsync(d_pf_pos_.p, pos.data(), (size_t)n * 16, hipMemcpyHostToDevice, st_));
            Q3_HIP(hipStreamSynchronize(st_)); // staging vectors are reused by the next chunk
            TokMeta tm{d_pf_seq_.p, d_pf_slot_.p, d_pf_pos_.p};
            Transformer::Input in; in.x = d_prompt_.p; in.x_stride = Q3_EMBD;
            talker_->forward(st_, in, n, tm, kv_t_->view(), nullptr);
            for (auto& lb : lasts) { // logits / code_0 + hidden of the LAST prompt token (engine.rs:550-554,565-566)
                const int sb = lb.first, ti = lb.second;
                if (any_sampled_req)
                    talker_->head(st_, ti, 1, 0, tl_stride_, d_tlogits_.p + (size_t)sb * tl_stride_, tl_stride_, nullptr, -1, d_thidden_.p + (size_t)sb * Q3_EMBD);
                else {
                    ArgmaxEpi am{d_keys_.p + (size_t)sb * 16, 16, d_maskeos_.p + sb, 0};
                    talker_->head(st_, ti, 1, 0, tl_stride_, nullptr, 0, &am, Q3_SAMPLE_END, d_thidden_.p + (size_t)sb * Q3_EMBD);
                }
            }
            done += (size_t)n;
        }
    }
    for (int b = B; b < W; b++) { tslot[b] = 0; } // idle slots write to their single reserved page
    d_maxframes_.upload(maxf.data(), W); d_finished_.upload(fin.data(), W);
    d_nframes_.upload(nfr.data(), W); d_tslot_.upload(tslot.data(), W); d_tpos_.upload(tpos.data(), (size_t)4 * W);
    Q3_HIP(hipEventRecord(ev1, st_));
    Q3_HIP(hipStreamSynchronize(st_));
    { float ms = 0; Q3_HIP(hipEventElapsedTime(&ms, ev0, ev1)); stats.prefill_ms += ms; }
    const double t_prefill = now_ms();
    for (int b = 0; b < B; b++) out[b].prefill_ms = t_prefill - t0;

    // ---------------- host-side per-sequence helpers ----------------
    std::vector<Sampler> samplers;
    for (int b = 0; b < B; b++) {
        const SamplerConfig& sc = reqs[b].sampler;
        const uint64_t seed = sc.has_seed ? sc.seed : (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(
                                                           std::chrono::system_clock::now().time_since_epoch()).count(); // :473-478
        samplers.emplace_back(sc.temperature, sc.top_k, sc.top_p, seed);
    }
    const int spf = codec_ ? codec_->samples_per_frame() : 0;
    std::vector<std::unique_ptr<Chunker>> chunkers;
    std::vector<size_t> pcm_len(B, 0);        // samples enqueued so far per sequence (pinned staging area, slot b)
    std::vector<hipEvent_t> ev_first(B, nullptr);
    const size_t slot_cap = codec_ ? (size_t)p_.max_steps * spf : 0;
    // decoder thread (the reference decodes on a second thread too, engine.rs:495-543): it owns every codec launch, so
    // the ~200 kernel launches of a chunk never delay the AR stream's launches on this thread
    struct DecTask { int b; std::vector<int64_t> codes; bool is_final; };
    std::deque<DecTask> dq;
    std::mutex dmu;
    std::condition_variable dcv;
    bool ddone = false;
    std::string derr;
    std::thread dec_thread;
    const bool use_codec = codec_ && want_pcm;
    if (use_codec) {
        for (int b = 0; b < B; b++) codec_->reset(b);
        dec_thread = std::thread([&]() {
            try {
                Q3_HIP(hipSetDevice(dev_)); // HIP's current device is per thread
                for (;;) {
                    DecTask t;
                    {
                        std::unique_lock<std::mutex> lk(dmu);
                        dcv.wait(lk, [&] { return ddone || !dq.empty(); });
                        if (dq.empty()) return;
                        t = std::move(dq.front());
                        dq.pop_front();
                    }
                    const int b = t.b, nf = (int)t.codes.size() / 16;
                    Q3_CHECK(pcm_len[b] + (size_t)nf * spf <= slot_cap, "pcm staging overflow");
                    const int lane = b % (int)st2_.size();
                    // engine.rs:520: decode the chunk -- enqueued on a codec stream, overlapping the next AR frames
                    const int got = codec_->decode_async(st2_[lane], b, t.codes.data(), nf, t.is_final, pcm_pinned_ + (size_t)b * slot_cap + pcm_len[b], lane);
                    pcm_len[b] += (size_t)std::max(got, 0);
                    stats.codec_calls++;
                    if (!ev_first[b]) { Q3_HIP(hipEventCreate(&ev_first[b])); Q3_HIP(hipEventRecord(ev_first[b], st2_[lane])); } // first stream_tx.send, :522-523
                }
            } catch (const std::exception& ex) { std::lock_guard<std::mutex> lk(dmu); derr = ex.what(); }
        });
    }
    for (int b = 0; b < B; b++) {
        chunkers.emplace_back(new Chunker([&, b](const int64_t* codes, int n_codes, bool is_final) {
            if (!use_codec) return;
            { std::lock_guard<std::mutex> lk(dmu); dq.push_back(DecTask{b, std::vector<int64_t>(codes, codes + n_codes), is_final}); }
            dcv.notify_one();
        }));
    }
    hipEvent_t ev_t0;
    Q3_HIP(hipEventCreate(&ev_t0));
    Q3_HIP(hipEventRecord(ev_t0, st_)); // the AR stream is idle here (prefill was synchronised): marks "prefill done"

    // ---------------- frame loop ----------------
    code0_given_ = any_sampled;
    LaunchTimer timer, timer_gu;
    const bool eager = instrument_ || !p_.use_graph;
    if (instrument_) { talker_->timer = &timer; predictor_->timer = &timer; talker_->timer_gu = &timer_gu; }
    if (!eager) build_graph(W);
    std::vector<int32_t> fed(B, 0), hbuf;
    std::vector<float> hlogits;
    int step = 0;
    bool all_done = (max_steps_all == 0);
    while (!all_done && step < max_steps_all) {
        const int group = any_sampled ? 1 : std::min(4, max_steps_all - step);
        Q3_HIP(hipEventRecord(ev0, st_));
        for (int g = 0; g < group; g++) {
            if (any_sampled) { // host sampler on logits [0,2160) (llama/mod.rs:666-775; greedy requests take its T<=0 branch)
                hlogits.resize((size_t)W * tl_stride_);
                d_tlogits_.download(hlogits.data(), hlogits.size());
                for (int b = 0; b < B; b++) {
                    float* lg = hlogits.data() + (size_t)b * tl_stride_;
                    if (reqs[b].mask_eos) lg[Q3_CODEC_EOS] = -INFINITY;
                    const q3_u64 key = pack_key(0.0f, samplers[b].sample(lg, tl_stride_, 0, Q3_SAMPLE_END));
                    Q3_HIP(hipMemcpy(d_keys_.p + (size_t)16 * b, &key, 8, hipMemcpyHostToDevice));
                }
            }
            if (eager) record_frame(W);
            else Q3_HIP(hipGraphLaunch(graph_exec_, st_));
        }
        Q3_HIP(hipEventRecord(ev1, st_));
        Q3_HIP(hipStreamSynchronize(st_));
        { float ms = 0; Q3_HIP(hipEventElapsedTime(&ms, ev0, ev1)); stats.frame_loop_ms += ms; }
        if (instrument_) {
            stats.gemv_ms += timer.collect_ms(); stats.gemv_bytes += timer.bytes; stats.gemv_launches += timer.launches; timer.bytes = 0; timer.launches = 0;
            stats.gu_ms += timer_gu.collect_ms(); stats.gu_bytes += timer_gu.bytes; stats.gu_launches += timer_gu.launches; timer_gu.bytes = 0; timer_gu.launches = 0;
        }
        step += group;
        d_nframes_.download(nfr.data(), W); d_finished_.download(fin.data(), W);
        all_done = true;
        for (int b = 0; b < B; b++) {
            if (nfr[b] > fed[b]) { // hand new frames to the chunker (engine.rs:613-620)
                const int nnew = nfr[b] - fed[b];
                hbuf.resize((size_t)nnew * 16);
                Q3_HIP(hipMemcpy(hbuf.data(), d_hist_.p + (size_t)b * hist_stride_ + (size_t)fed[b] * 16, (size_t)nnew * 64, hipMemcpyDeviceToHost));
                out[b].codes.insert(out[b].codes.end(), hbuf.begin(), hbuf.end());
                for (int f = 0; f < nnew; f++) {
                    int64_t fc[16];
                    for (int q = 0; q < 16; q++) fc[q] = hbuf[(size_t)f * 16 + q];
                    chunkers[b]->push(fc, 16, false);
                }
                stats.frames += nnew;
                fed[b] = nfr[b];
            }
            if (!fin[b] && nfr[b] < reqs[b].max_steps) all_done = false;
        }
    }
    talker_->timer = nullptr; predictor_->timer = nullptr; talker_->timer_gu = nullptr;
    for (int b = 0; b < B; b++) chunkers[b]->push(nullptr, 0, true); // :644
    {
        const double c0 = now_ms();
        if (use_codec) {
            { std::lock_guard<std::mutex> lk(dmu); ddone = true; }
            dcv.notify_one();
            dec_thread.join();                                   // engine.rs:647-649
            if (!derr.empty()) throw Error("decoder thread: " + derr);
        }
        for (auto s2 : st2_) Q3_HIP(hipStreamSynchronize(s2));
        stats.codec_ms += now_ms() - c0;     // only the part of the codec work the AR loop did not hide
    }
    for (int b = 0; b < B; b++) {
        out[b].n_frames = fed[b];
        if (want_pcm && codec_) out[b].pcm.assign(pcm_pinned_ + (size_t)b * slot_cap, pcm_pinned_ + (size_t)b * slot_cap + pcm_len[b]);
        if (ev_first[b]) {
            float ms = 0;
            Q3_HIP(hipEventElapsedTime(&ms, ev_t0, ev_first[b]));
            out[b].first_chunk_ms = out[b].prefill_ms + ms;
            (void)hipEventDestroy(ev_first[b]);
        }
        out[b].total_ms = now_ms() - t0;
    }
    (void)hipEventDestroy(ev_t0);
    (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1);
}

} // namespace q3
