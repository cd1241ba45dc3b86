// host_logic.cpp -- see host_logic.h
#include "host_logic.h"
#include "q3_common.h"
#include <algorithm>
#include <cmath>

namespace q3 {

// ---------------- StdRng [EXT] ----------------
static inline uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
StdRng::StdRng(uint64_t state) {
    for (int i = 0; i < 8; i++) {
        state = state * 6364136223846793005ULL + 11634580027462260723ULL;
        const uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27);
        const uint32_t rot = (uint32_t)(state >> 59);
        key_[i] = (xs >> rot) | (xs << ((32 - rot) & 31));
    }
}
uint32_t StdRng::next_u32() {
    if (idx_ >= 16) {
        uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key_[0], key_[1], key_[2], key_[3],
                          key_[4], key_[5], key_[6], key_[7], (uint32_t)counter_, (uint32_t)(counter_ >> 32), 0, 0};
        uint32_t x[16];
        std::copy(s, s + 16, x);
        auto qr = [&](int a, int b, int c, int d) {
            x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 12);
            x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 8);  x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 7);
        };
        for (int r = 0; r < 6; r++) {
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
        }
        for (int i = 0; i < 16; i++) buf_[i] = x[i] + s[i];
        counter_++;
        idx_ = 0;
    }
    return buf_[idx_++];
}

// ---------------- Sampler ----------------
int32_t Sampler::sample(const float* logits, int n_vocab, int start, int end) {
    end = std::min(end, n_vocab);
    if (temperature_ <= 0.0f) { // :690-701
        float max_val = -INFINITY; int max_idx = start;
        for (int i = start; i < end; i++) if (logits[i] > max_val) { max_val = logits[i]; max_idx = i; }
        return max_idx;
    }
    std::vector<std::pair<int, float>> c;
    for (int i = start; i < end; i++) c.emplace_back(i, logits[i]);
    if (c.empty()) return start;
    std::stable_sort(c.begin(), c.end(), [](const auto& a, const auto& b) { return a.second > b.second; }); // :708
    if (top_k_ > 0 && (size_t)top_k_ < c.size()) c.resize((size_t)top_k_);                                  // :711-713
    const float max_logit = c[0].second;
    float sum = 0.0f;
    for (auto& e : c) { e.second = q3_expf((e.second - max_logit) / temperature_); sum += e.second; }      // :716-726
    if (sum > 0.0f) for (auto& e : c) e.second /= sum;
    if (top_p_ < 1.0f) {                                                                                     // :734-753
        float cum = 0.0f; size_t cut = c.size();
        for (size_t i = 0; i < c.size(); i++) { cum += c[i].second; if (cum >= top_p_) { cut = i + 1; break; } }
        c.resize(cut);
        float ns = 0.0f;
        for (auto& e : c) ns += e.second;
        if (ns > 0.0f) for (auto& e : c) e.second /= ns;
    }
    const float r = (float)rng_.next_u32() / 4294967296.0f;                                                  // :761
    float cum = 0.0f;
    for (auto& e : c) { cum += e.second; if (r < cum) return e.first; }
    return c[0].first;
}

// ---------------- HostAssets ----------------
HostAssets::HostAssets(const std::string& path) : g_(new Gguf(path)), tts_pad_(2048, 0.0f) {
    auto f32 = [&](const GgufTensor* t) -> const float* {
        if (t->type != Q3_T_F32) throw Error("Unsupported tensor type: " + std::to_string(t->type) + " (expected F32)");
        return reinterpret_cast<const float*>(t->data);
    };
    const GgufTensor* pw = g_->find("proj.weight");
    if (!pw) throw Error("proj.weight (tensor) missing");
    const GgufTensor* pb = g_->find("proj.bias");
    if (!pb) throw Error("proj.bias (tensor) missing");
    proj_w = f32(pw); proj_b = f32(pb);
    proj_out = pb->ne[0]; proj_in = pw->ne[0] * pw->rows() / proj_out;
    if (const GgufTensor* tt = g_->find("text_embd")) { text_table = f32(tt); text_rows = tt->ne[0] * tt->rows() / 2048; }
    for (int i = 0; i < 16; i++)
        if (const GgufTensor* t = g_->find("codec_embd." + std::to_string(i))) {
            codec[n_codec] = f32(t); codec_rows[n_codec] = t->ne[0] * t->rows() / 2048; n_codec++;
        }
    if (text_rows * 2048 >= (int64_t)(151671 + 1) * 2048) std::copy(text_table + (size_t)151671 * 2048, text_table + (size_t)151672 * 2048, tts_pad_.begin());
}
void HostAssets::codec_embedding(int q, int32_t code, float* out) const {
    if (q >= 0 && q < n_codec) {
        const int64_t c = code < 0 ? 0 : code;
        if ((c + 1) * 2048 <= codec_rows[q] * 2048) { std::copy(codec[q] + (size_t)c * 2048, codec[q] + (size_t)(c + 1) * 2048, out); return; }
    }
    std::fill(out, out + 2048, 0.0f);
}
void HostAssets::text_embedding(int64_t token, float* out) const {
    if (token >= 0 && (token + 1) * 2048 <= text_rows * 2048) { std::copy(text_table + (size_t)token * 2048, text_table + (size_t)(token + 1) * 2048, out); return; }
    for (int i = 0; i < 2048; i++) out[i] = std::fmod((float)((uint64_t)token * 17u + (uint64_t)i), 2.0f) - 1.0f; // :454-460
}

// ---------------- PromptBuilder ----------------
namespace {
struct Rows {
    PromptData d;
    float* add() { d.embd.resize(d.embd.size() + 2048); d.n_rows++; return d.embd.data() + d.embd.size() - 2048; }
};
void sum2(const float* a, const float* b, float* o) { for (int i = 0; i < 2048; i++) o[i] = a[i] + b[i]; }
}
PromptData PromptBuilder::build_core(const HostAssets& a, const std::vector<int32_t>& text_ids, const int* lang_id,
                                     const int* spk_id, const float* spk_emb, const std::vector<int32_t>* instr_ids,
                                     const std::vector<float>* mid_rows) {
    Rows R;
    std::vector<float> e(2048), marker(2048), pad0(2048), t(2048);
    auto text = [&](int64_t id) { a.text_embedding(id, R.add()); };
    auto mark_codec = [&](int32_t code) { a.codec_embedding(0, code, e.data()); sum2(marker.data(), e.data(), R.add()); };
    if (instr_ids) { // prompt.rs:154-169
        text(151644); text(872); text(198);
        for (int32_t id : *instr_ids) text(id);
        text(151645); text(198);
    }
    text(151644); text(77091); text(198); // :173-175
    a.text_embedding(Q3_TEXT_AUDIO_MARKER, marker.data());
    if (lang_id) { mark_codec(Q3_CODEC_THINK); mark_codec(Q3_CODEC_THINK_BOS); mark_codec(*lang_id); mark_codec(Q3_CODEC_THINK_EOS); } // :180-191
    else { mark_codec(Q3_CODEC_NOTHINK); mark_codec(Q3_CODEC_THINK_BOS); mark_codec(Q3_CODEC_THINK_EOS); }                         // :192-204
    if (spk_id) mark_codec(*spk_id);                                  // :207-214
    else if (spk_emb) sum2(marker.data(), spk_emb, R.add());          // :215-222
    if (mid_rows) { R.d.embd.insert(R.d.embd.end(), mid_rows->begin(), mid_rows->end()); R.d.n_rows += (int)(mid_rows->size() / 2048); } // :225-227
    a.codec_embedding(0, Q3_CODEC_PAD, pad0.data());                  // :232
    a.text_embedding(Q3_TEXT_BOS, t.data()); sum2(t.data(), pad0.data(), R.add());                    // :233-239
    for (int32_t id : text_ids) { a.text_embedding(id, t.data()); sum2(t.data(), pad0.data(), R.add()); } // :241-245
    a.text_embedding(Q3_TEXT_EOS, t.data()); sum2(t.data(), pad0.data(), R.add());                    // :248-254
    mark_codec(Q3_CODEC_BOS);                                         // :258-264
    return std::move(R.d);
}
PromptData PromptBuilder::build_clone_prompt(const HostAssets& a, const std::vector<int32_t>& text_ids,
                                             const std::vector<int32_t>& ref_codes, const std::vector<int32_t>& ref_text_ids,
                                             const float* spk_emb, int lang_id, const std::vector<int32_t>* instr_ids) {
    std::vector<float> mid;
    auto add = [&]() { mid.resize(mid.size() + 2048); return mid.data() + mid.size() - 2048; };
    std::vector<float> pad(2048), t(2048), marker(2048), e(2048), sum(2048);
    a.codec_embedding(0, Q3_CODEC_PAD, pad.data()); // :47
    std::vector<int64_t> ids;                       // :41-43
    ids.push_back(Q3_TEXT_BOS);
    for (int32_t id : ref_text_ids) ids.push_back(id);
    ids.push_back(Q3_TEXT_EOS);
    for (int64_t id : ids) { a.text_embedding(id, t.data()); sum2(t.data(), pad.data(), add()); } // :49-58
    a.text_embedding(Q3_TEXT_AUDIO_MARKER, marker.data());                                       // :67
    a.codec_embedding(0, Q3_CODEC_AUDIO_START, e.data()); sum2(marker.data(), e.data(), add());  // :68-74
    const size_t n_steps = ref_codes.size() / 16;                                                // :79
    for (size_t s = 0; s < n_steps; s++) {                                                       // :80-96
        std::fill(sum.begin(), sum.end(), 0.0f);
        for (int q = 0; q < 16; q++) { a.codec_embedding(q, ref_codes[s * 16 + q], e.data()); for (int i = 0; i < 2048; i++) sum[i] += e[i]; }
        sum2(marker.data(), sum.data(), add());
    }
    sum2(marker.data(), pad.data(), add());                                                      // :100-106
    return build_core(a, text_ids, &lang_id, nullptr, spk_emb, instr_ids, &mid);
}

// ---------------- Chunker ----------------
void Chunker::push(const int64_t* codes, int n, bool is_final) {
    buf_.insert(buf_.end(), codes, codes + n);
    if (buf_.size() >= (size_t)Q3_CHUNK_CODES || is_final) { // engine.rs:510
        const size_t valid = (buf_.size() / 16) * 16;      // :512
        if (valid > 0) {
            std::vector<int64_t> safe(buf_.begin(), buf_.begin() + valid);
            for (auto& v : safe) v = std::min<int64_t>(std::max<int64_t>(v, 0), 2047); // :515-519
            fn_(safe.data(), (int)valid, is_final);
            const size_t remaining = buf_.size() - valid;
            if (remaining > 0 && !is_final) buf_.erase(buf_.begin(), buf_.begin() + valid); // :528-533
            else buf_.clear();
        } else {
            buf_.clear(); // :535
        }
    }
}

} // namespace q3
