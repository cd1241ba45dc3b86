// engine.h -- the autoregressive generation loop on device: what TtsEngine::run_inference_stream does
// (/root/reference/src/tts/engine.rs:445-656), batched over B lock-stepped sequences and captured as a hipGraph
// per frame.  Greedy (temperature <= 0) sampling stays on device; temperature > 0 samples on the host with
// the reference's sampler (llama/mod.rs:703-775) from logits copied back each frame.
#pragma once
#include "transformer.h"
#include "host_logic.h"
#include "codec.h"
#include <memory>

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

struct EngineStats { // accumulated over generate_batch calls since reset
    double frame_loop_ms = 0; long frames = 0;   // device time of the AR frame loop (HIP events), all sequences
    double gemv_ms = 0; long gemv_launches = 0; double gemv_bytes = 0; // instrumented (eager) leg only: GEMV family except ...
    double gu_ms = 0; long gu_launches = 0; double gu_bytes = 0;       // ... the talker's gate/up kernel, timed on its own
    double codec_ms = 0; long codec_calls = 0;
    double prefill_ms = 0;
};

class Engine {
public:
    explicit Engine(const EngineParams& p);
    ~Engine();
    void generate_batch(const std::vector<GenRequest>& reqs, std::vector<GenResult>& out, bool want_pcm);
    const HostAssets& assets() const { return *assets_; }
    Transformer& talker() { return *talker_; }
    Transformer& predictor() { return *predictor_; }
    CodecDecoder* codec() { return codec_.get(); }
    EngineStats stats; void reset_stats() { stats = EngineStats(); }
    void set_instrument(bool on) { instrument_ = on; }
    size_t bytes_per_frame_step(int batch, double mean_ctx) const; // algorithmic HBM bytes of one batched frame step
    hipStream_t stream() const { return st_; }

private:
    void record_frame(int B);           // enqueue one frame step for B sequences on st_
    void build_graph(int B);
    EngineParams p_;
    hipStream_t st_ = nullptr; int dev_ = 0;
    std::vector<hipStream_t> st2_; // codec decoder lanes: overlapped with the next AR frames and with each other
    float* pcm_pinned_ = nullptr; size_t pcm_pinned_cap_ = 0;
    std::unique_ptr<HostAssets> assets_;
    std::unique_ptr<Transformer> talker_, predictor_;
    std::unique_ptr<KvPool> kv_t_, kv_p_;
    std::unique_ptr<CodecDecoder> codec_;
    // device assets
    DevBuf<float> d_codec_tab_[16]; DevBuf<float> d_proj_tab_[16];
    DevBuf<float> d_proj_wt_, d_proj_wblk_, d_proj_b_, d_tts_pad_;
    DevBuf<const float*> d_tab_ptrs_; DevBuf<int64_t> d_tab_rows_;
    int dP_ = 0;
    // per-sequence device state
    int B_ = 0;
    DevBuf<int32_t> d_tseq_, d_tslot_, d_tpos_, d_nframes_, d_finished_, d_maxframes_, d_hist_, d_maskeos_;
    DevBuf<q3_u64> d_keys_, d_next_key0_; // argmax keys: [B][16] codes of the current frame, [B] code_0 of the next frame
    DevBuf<int32_t> d_pseq_, d_pslot_, d_ppos_, d_pseqA_, d_pslotA_, d_pposA_;
    DevBuf<float> d_tlogits_, d_thidden_, d_pin_, d_plogits_, d_fb_, d_prompt_, d_hid_all_;
    DevBuf<int32_t> d_pf_seq_, d_pf_slot_, d_pf_pos_;
    int hist_stride_ = 0, tl_stride_ = 0;
    hipGraph_t graph_ = nullptr; hipGraphExec_t graph_exec_ = nullptr; int graph_B_ = 0; bool graph_given_ = false;
    bool instrument_ = false;
    std::vector<hipEvent_t> ev_pool_; size_t ev_used_ = 0;
    bool code0_given_ = false; // frame variant: code0 already placed by the host sampler
    void gemv_timed(const Q8Mat& w, std::function<void()> fn);
    friend struct FrameRecorder;
};

} // namespace q3
