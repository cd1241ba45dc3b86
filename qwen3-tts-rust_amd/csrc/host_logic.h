// host_logic.h -- host-side (no GPU) parts of the hot path, mirroring the reference's Rust semantics:
//   Sampler       /root/reference/src/models/llama/mod.rs:627-776   (LlamaSampler)
//   Assets        /root/reference/src/assets_manager.rs:5-460       (host view: gathers, fallback, tts_pad)
//   PromptBuilder /root/reference/src/tts/prompt.rs:24-277
//   Chunker       /root/reference/src/tts/engine.rs:495-543
// Pure C++17; unit-tested on CPU through the C ABI.
#pragma once
#include "gguf.h"
#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <vector>

namespace q3 {

// rand 0.10 StdRng (ChaCha12) seeded through SeedableRng::seed_from_u64 (PCG32 expansion) [EXT, unpinned]
class StdRng {
public:
    explicit StdRng(uint64_t seed);
    uint32_t next_u32();
    const uint32_t* key() const { return key_; } // the device sampler regenerates the same stream from the key + a draw count
private:
    uint32_t key_[8]; uint64_t counter_ = 0; uint32_t buf_[16]; int idx_ = 16;
};

class Sampler { // llama/mod.rs:627-776
public:
    Sampler(float temperature, int top_k, float top_p, uint64_t seed)
        : temperature_(temperature), top_k_(top_k), top_p_(top_p), rng_(seed) {}
    static Sampler greedy() { return Sampler(0.0f, 0, 1.0f, 42); } // :653-664
    int32_t sample(const float* logits, int n_vocab, int start, int end);
    const StdRng& rng() const { return rng_; }
private:
    float temperature_; int top_k_; float top_p_; StdRng rng_;
};

class HostAssets { // assets_manager.rs (host view over the mmapped GGUF; tensors must be F32 :163-167)
public:
    explicit HostAssets(const std::string& gguf_path);
    void codec_embedding(int q, int32_t code, float* out2048) const; // :419-437
    void text_embedding(int64_t token, float* out2048) const;        // :444-460
    const float* tts_pad() const { return tts_pad_.data(); }         // :244-249
    const float* proj_w = nullptr; const float* proj_b = nullptr; int64_t proj_out = 0, proj_in = 0;
    const float* text_table = nullptr; int64_t text_rows = 0;
    const float* codec[16] = {}; int64_t codec_rows[16] = {}; int n_codec = 0;
private:
    std::unique_ptr<Gguf> g_;
    std::vector<float> tts_pad_;
};

struct PromptData { std::vector<float> embd; int n_rows = 0; }; // rows of 2048 f32 (prompt.rs:18-22)

class PromptBuilder { // prompt.rs:26-277 ; token ids are supplied by the caller (tokenizer is row f-3, out of scope)
public:
    static PromptData build_core(const HostAssets& a, const std::vector<int32_t>& text_ids, const int* lang_id,
                                 const int* spk_id, const float* spk_emb, const std::vector<int32_t>* instr_ids,
                                 const std::vector<float>* mid_rows);
    static PromptData build_clone_prompt(const HostAssets& a, const std::vector<int32_t>& text_ids,
                                         const std::vector<int32_t>& ref_codes, const std::vector<int32_t>& ref_text_ids,
                                         const float* spk_emb, int lang_id, const std::vector<int32_t>* instr_ids);
};

class Chunker { // engine.rs:505-541
public:
    using DecodeFn = std::function<void(const int64_t* codes, int n_codes, bool is_final)>;
    explicit Chunker(DecodeFn fn) : fn_(std::move(fn)) {}
    void push(const int64_t* codes, int n, bool is_final);
private:
    std::vector<int64_t> buf_; DecodeFn fn_;
};

} // namespace q3
