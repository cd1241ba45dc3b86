// engine.h -- the autoregressive generation loop on device: what TtsEngine::run_inference_stream does
// (/root/reference/src/tts/engine.rs:445-656), for many utterances at once.
//
// Execution model: the engine owns `max_batch` sequence slots.  One frame step for all slots is a static
// hipGraph (everything a frame needs -- codes, positions, sampler state, finished flags -- lives on the device).
// A scheduler admits queued requests into free slots (batched multi-sequence prefill), replays the frame graph in
// groups of 4 frames (one streaming step, engine.rs:505-512), retires sequences on EOS / max_steps and hands
// finished 4-frame chunks to a decoder thread (engine.rs:495-543) that runs the codec on its own HIP streams.
// Sampling (greedy and temperature/top-k/top-p, llama/mod.rs:666-776) is on device.
#pragma once
#include "transformer.h"
#include "host_logic.h"
#include "codec.h"
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

namespace q3 {

struct SamplerConfig { // engine.rs:13-45
    float temperature = 0.7f; int top_k = 40; float top_p = 0.9f; bool has_seed = false; uint64_t seed = 0;
};

struct EngineParams {
    std::string model_dir;      // contains <quant dir>/ and onnx/
    std::string quant = "q8_0"; // engine.rs:91-95
    int max_batch = 1;
    int max_prompt = 1024;      // llama/mod.rs:567-581: effective prompt cap of the reference
    int max_steps = Q3_DEFAULT_MAX_STEPS;
    bool load_codec = true;
    bool use_graph = true;
};

struct GenRequest {
    const float* prompt = nullptr; int n_prompt = 0; // [n_prompt][2048] host
    SamplerConfig sampler;
    int max_steps = Q3_DEFAULT_MAX_STEPS;
    bool mask_eos = false;      // bench/test knob: exclude EOS so runs have a fixed length
};
struct GenResult {
    std::vector<int32_t> codes; int n_frames = 0;
    std::vector<float> pcm;
    double prefill_ms = 0, first_chunk_ms = 0, total_ms = 0;
};

struct EngineStats { // accumulated since reset
    double frame_loop_ms = 0; long frames = 0;   // device time of the AR frame loop (HIP events), all sequences
    double gemv_ms = 0; long gemv_launches = 0; double gemv_bytes = 0; // instrumented (eager) leg only: GEMV family except ...
    double gu_ms = 0; long gu_launches = 0; double gu_bytes = 0;       // ... the talker's gate/up kernel, timed on its own
    double codec_ms = 0; long codec_calls = 0;
    double prefill_ms = 0;
    long steps = 0; double slot_frames = 0;      // scheduler: frame-group launches, sum of graph widths x frames (occupancy)
    long graph_frames = 0;                       // frame-graph replays
};

enum ReqState { REQ_QUEUED = 0, REQ_RUNNING = 1, REQ_DRAINING = 2, REQ_DONE = 3, REQ_FAILED = -1 };
struct ReqStatus {
    int state = REQ_QUEUED; int n_frames = 0; int64_t n_pcm = 0; // frames emitted / PCM samples decoded so far
    double queue_ms = 0, prefill_ms = 0, first_chunk_ms = 0, total_ms = 0;
    std::string error; // REQ_FAILED: what the scheduler / decoder thread reported
};

struct Voice { // a registered voice: preset embedding or clone material (engine.rs:390-435, voice_file.rs:5-22)
    std::vector<float> spk_emb; std::vector<int32_t> ref_codes, ref_text_ids;
};


class Engine {
public:
    explicit Engine(const EngineParams& p);
    ~Engine();
    // blocking convenience: submit all, run the scheduler until they are done (requests beyond max_batch queue up)
    void generate_batch(const std::vector<GenRequest>& reqs, std::vector<GenResult>& out, bool want_pcm);

    // ---- continuous-batching scheduler ----
    int64_t submit(const GenRequest& r, bool want_pcm, bool copy_prompt = true); // thread-safe
    bool step();                                  // one scheduling iteration on the calling thread; false when idle
    ReqStatus poll(int64_t id);                   // thread-safe snapshot
    // copies frames [frame_off, ...) and PCM samples [pcm_off, ...) that are available now; returns counts
    void fetch(int64_t id, int32_t* codes, int frame_off, int max_frames, float* pcm, int64_t pcm_off, int64_t pcm_cap,
               int* got_frames, int64_t* got_pcm);
    bool wait(int64_t id, double timeout_ms);     // drives step() itself when no driver thread is running
    void release(int64_t id);
    void start_driver();                          // background thread looping step()
    void stop_driver();
    int register_voice(const Voice& v);
    const Voice& voice(int id) const;

    const HostAssets& assets() const { return *assets_; }
    Transformer& talker() { return *talker_; }
    Transformer& predictor() { return *predictor_; }
    CodecDecoder* codec() { return codec_.get(); }
    EngineStats stats; void reset_stats() { stats = EngineStats(); }
    void set_instrument(bool on) { instrument_ = on; }
    size_t bytes_per_frame_step(int batch, double mean_ctx) const; // algorithmic HBM bytes of one batched frame step
    hipStream_t stream() const { return st_; }
    int max_batch() const { return B_; }
    int device() const { return dev_; }

private:
    struct Req;
    struct FrameGraph { // one captured frame step for `width` slots (+ the pass-A routing tables of that width)
        int width = 0; hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
        DevBuf<int32_t> seqA, slotA, posA;
    };
    void record_frame(FrameGraph& fg, bool sampled);  // enqueue one frame step for fg.width slots on st_
    FrameGraph& frame_graph(int width, bool sampled, bool capture);
    void prefill(const std::vector<Req*>& batch, bool sampled);
    // Asynchronous admission (engines with more than one slot): the prompt of newly admitted requests is prefilled by a second talker
    // instance (own weights copy, own scratch, own stream) while the frame graph keeps stepping the running sequences; the slots
    // join the batch ("activate") between two frame groups once the prefill event has completed.
    void prefill_async(const std::vector<Req*>& batch);
    void activate();
    void admit();
    void run_group();
    void harvest(bool block);
    void finish_ar(Req* r);
    void decoder_main();
    void upload_slot_state();
    // All scheduler transfers go through a pinned arena on the AR stream: the engine's streams are non-blocking, so nothing the
    // scheduler does waits for the codec lanes (a synchronous hipMemcpy on the null stream would).
    void* stage_alloc(size_t bytes);
    void h2d(void* dst, const void* src, size_t bytes);            // async on st_, source copied into the arena first
    void* d2h_begin(const void* src, size_t bytes);                // async on st_ into the arena; valid after the next sync of st_
    unsigned char* arena_ = nullptr; size_t arena_cap_ = 0, arena_used_ = 0;

    EngineParams p_;
    hipStream_t st_ = nullptr; int dev_ = 0;
    std::vector<hipStream_t> st2_; // codec decoder lanes: overlapped with the next AR frames and with each other
    float* pcm_pinned_ = nullptr; size_t slot_cap_ = 0;
    std::unique_ptr<HostAssets> assets_;
    std::unique_ptr<Transformer> talker_, predictor_, talker_pf_;
    hipStream_t st_pf_ = nullptr; unsigned char* arena_pf_ = nullptr; size_t arena_pf_cap_ = 0;
    DevBuf<float> d_prompt_pf_, d_pf_logits_, d_pf_hidden_; DevBuf<int32_t> d_pfa_seq_, d_pfa_slot_, d_pfa_pos_;
    std::vector<Req*> pf_batch_; hipEvent_t pf_done_ = nullptr, pf_e0_ = nullptr; bool async_pf_ = false; int n_prefilling_ = 0;
    std::vector<char> slot_live_;      // slot_req_[b] set = reserved; live = stepping inside the frame graph
    std::unique_ptr<KvPool> kv_t_, kv_p_;
    std::unique_ptr<CodecDecoder> codec_;
    // device assets
    DevBuf<float> d_codec_tab_[16]; DevBuf<float> d_proj_tab_[16];
    DevBuf<float> d_proj_wt_, d_proj_wblk_, d_proj_b_, d_tts_pad_;
    DevBuf<const float*> d_tab_ptrs_; DevBuf<int64_t> d_tab_rows_;
    int dP_ = 0;
    // per-slot device state
    int B_ = 0;
    DevBuf<int32_t> d_tseq_, d_tslot_, d_tpos_, d_nframes_, d_finished_, d_maxframes_, d_hist_, d_maskeos_;
    DevBuf<q3_u64> d_keys_, d_next_key0_; // argmax keys: [B][16] codes of the current frame, [B] code_0 of the next frame
    DevBuf<float> d_temp_, d_topp_; DevBuf<int32_t> d_topk_; DevBuf<uint32_t> d_rngkey_, d_draws_; // device sampler state
    DevBuf<int32_t> d_pseq_, d_pslot_, d_ppos_;
    DevBuf<float> d_tlogits_, d_thidden_, d_pin_, d_fb_, d_prompt_;
    DevBuf<int32_t> d_pf_seq_, d_pf_slot_, d_pf_pos_;
    int hist_stride_ = 0, tl_stride_ = 0;
    std::map<int, std::unique_ptr<FrameGraph>> graphs_; // key = width*2 + sampled
    bool instrument_ = false; bool pred_identity_pages_ = false;
    // host mirrors of the slot state (refreshed from the device after every frame group)
    std::vector<int32_t> h_maxf_, h_fin_, h_nfr_, h_mask_, h_nprompt_, h_topk_;
    std::vector<float> h_temp_, h_topp_;
    bool slot_dirty_ = false;
    // scheduler state
    std::mutex mu_; std::condition_variable cv_;
    std::mutex step_mu_;               // step() is driven by one thread at a time (driver thread, wait() callers, generate_batch)
    std::map<int64_t, std::unique_ptr<Req>> reqs_;
    std::deque<Req*> pending_;
    std::vector<Req*> slot_req_;
    std::vector<int> cs_free_;          // free codec streams
    int64_t next_id_ = 1;
    int n_active_ = 0, n_draining_ = 0;
    std::vector<Voice> voices_;
    std::thread driver_; bool driver_on_ = false, driver_stop_ = false;
    // decoder thread
    struct DecTask { Req* r; std::vector<int64_t> codes; bool is_final; bool fence; bool reset; };
    struct Completion { Req* r; hipEvent_t ev; size_t pcm_after; bool fence; };
    std::thread dec_thread_; bool dec_started_ = false, dec_stop_ = false;
    std::mutex dmu_; std::condition_variable dcv_;
    std::deque<DecTask> dq_; std::deque<Completion> comp_;
    std::string derr_;
    LaunchTimer timer_, timer_gu_;
};

} // namespace q3
