// tts_engine.cpp -- see tts_engine.hpp
#include "tts_engine.hpp"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <dirent.h>
#include <fstream>
#include <memory>
#include <sstream>
#include <stdexcept>

namespace q3tts {

// ---------------- minimal JSON (objects, arrays of numbers, strings) for VoiceFile ----------------
namespace {
struct JP {
    const std::string& s; size_t i = 0;
    void ws() { while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) i++; }
    [[noreturn]] void fail(const char* m) { throw std::runtime_error(std::string("json: ") + m + " at " + std::to_string(i)); }
    char peek() { ws(); if (i >= s.size()) fail("eof"); return s[i]; }
    void expect(char c) { if (peek() != c) fail("unexpected character"); i++; }
    uint32_t hex4() {
        if (i + 4 > s.size()) fail("truncated \\u escape");
        uint32_t v = 0;
        for (int k = 0; k < 4; k++) { const char c = s[i++]; v = v * 16 + (c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : (fail("bad hex digit"), 0)); }
        return v;
    }
    static void utf8(std::string& o, uint32_t cp) {
        if (cp < 0x80) o += (char)cp;
        else if (cp < 0x800) { o += (char)(0xC0 | (cp >> 6)); o += (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { o += (char)(0xE0 | (cp >> 12)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
        else { o += (char)(0xF0 | (cp >> 18)); o += (char)(0x80 | ((cp >> 12) & 0x3F)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
    }
    std::string str() {
        expect('"');
        std::string o;
        while (i < s.size() && s[i] != '"') {
            if (s[i] == '\\' && i + 1 < s.size()) {
                const char c = s[++i]; i++;
                switch (c) { // the full JSON escape set (serde_json and Python's json emit \uXXXX / \r / \b / \f for ref_text)
                    case 'n': o += '\n'; break; case 't': o += '\t'; break; case 'r': o += '\r'; break;
                    case 'b': o += '\b'; break; case 'f': o += '\f'; break;
                    case 'u': {
                        uint32_t cp = hex4();
                        if (cp >= 0xD800 && cp <= 0xDBFF) { // surrogate pair
                            if (i + 1 < s.size() && s[i] == '\\' && s[i + 1] == 'u') { i += 2; const uint32_t lo = hex4(); if (lo < 0xDC00 || lo > 0xDFFF) fail("bad low surrogate"); cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00); }
                            else fail("lone high surrogate");
                        } else if (cp >= 0xDC00 && cp <= 0xDFFF) fail("lone low surrogate");
                        utf8(o, cp);
                        break;
                    }
                    default: o += c; // \" \\ \/
                }
            } else o += s[i++];
        }
        if (i >= s.size()) fail("unterminated string");
        i++;
        return o;
    }
    double num() { ws(); size_t st = i; while (i < s.size() && (isdigit((unsigned char)s[i]) || strchr("+-.eE", s[i]))) i++; if (st == i) fail("number"); return std::stod(s.substr(st, i - st)); }
    void skip() { // any value
        char c = peek();
        if (c == '"') { str(); return; }
        if (c == '{') { i++; if (peek() == '}') { i++; return; } for (;;) { str(); expect(':'); skip(); if (peek() == ',') { i++; continue; } expect('}'); return; } }
        if (c == '[') { i++; if (peek() == ']') { i++; return; } for (;;) { skip(); if (peek() == ',') { i++; continue; } expect(']'); return; } }
        if (!strncmp(s.c_str() + i, "true", 4)) { i += 4; return; }
        if (!strncmp(s.c_str() + i, "false", 5)) { i += 5; return; }
        if (!strncmp(s.c_str() + i, "null", 4)) { i += 4; return; }
        num();
    }
    template <typename T> std::vector<T> arr() {
        std::vector<T> v;
        expect('[');
        if (peek() == ']') { i++; return v; }
        for (;;) { v.push_back((T)num()); if (peek() == ',') { i++; continue; } expect(']'); return v; }
    }
    std::optional<std::string> optstr() { if (peek() == 'n') { i += 4; return std::nullopt; } return str(); }
};
std::string esc(const std::string& s) { // control characters must be escaped (RFC 8259); UTF-8 bytes pass through
    std::string o;
    for (unsigned char c : s) {
        if (c == '"' || c == '\\') { o += '\\'; o += (char)c; }
        else if (c == '\n') o += "\\n"; else if (c == '\t') o += "\\t"; else if (c == '\r') o += "\\r"; else if (c == '\b') o += "\\b"; else if (c == '\f') o += "\\f";
        else if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; }
        else o += (char)c;
    }
    return o;
}
}

VoiceFile VoiceFile::load(const std::string& path) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("No such file or directory: " + path);
    std::stringstream ss; ss << f.rdbuf();
    const std::string txt = ss.str();
    JP p{txt};
    VoiceFile v;
    bool have_emb = false;
    p.expect('{');
    if (p.peek() != '}') for (;;) {
        const std::string k = p.str();
        p.expect(':');
        if (k == "ref_text") v.ref_text = p.str();
        else if (k == "audio_codes") v.audio_codes = p.arr<int64_t>();
        else if (k == "speaker_embedding" || k == "spk_emb") { v.speaker_embedding = p.arr<float>(); have_emb = true; } // serde alias, voice_file.rs:13
        else if (k == "name") v.name = p.optstr();
        else if (k == "gender") v.gender = p.optstr();
        else if (k == "age") v.age = p.optstr();
        else if (k == "description") v.description = p.optstr();
        else p.skip(); // unknown keys (e.g. spk_id) are dropped, like serde's default
        if (p.peek() == ',') { p.i++; continue; }
        p.expect('}');
        break;
    } else p.i++;
    if (!have_emb) throw std::runtime_error("missing field `speaker_embedding`");
    return v;
}
void VoiceFile::save(const std::string& path) const {
    std::ofstream f(path);
    if (!f) throw std::runtime_error("cannot create " + path);
    f << "{\n  \"ref_text\": \"" << esc(ref_text) << "\",\n  \"audio_codes\": [";
    for (size_t i = 0; i < audio_codes.size(); i++) f << (i ? ", " : "") << audio_codes[i];
    f << "],\n  \"speaker_embedding\": [";
    f.precision(9);
    for (size_t i = 0; i < speaker_embedding.size(); i++) f << (i ? ", " : "") << speaker_embedding[i];
    f << "]";
    auto opt = [&](const char* k, const std::optional<std::string>& v) { f << ",\n  \"" << k << "\": "; if (v) f << "\"" << esc(*v) << "\""; else f << "null"; };
    opt("name", name); opt("gender", gender); opt("age", age); opt("description", description);
    f << "\n}\n";
}

void AudioSample::save_wav(const std::string& path) const { // audio.rs:26-41
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot create " + path);
    const uint32_t data_bytes = (uint32_t)samples.size() * 2, riff = 36 + data_bytes, fmt_len = 16, byte_rate = sample_rate * channels * 2;
    const uint16_t pcm = 1, block = (uint16_t)(channels * 2), bits = 16;
    fwrite("RIFF", 1, 4, f); fwrite(&riff, 4, 1, f); fwrite("WAVEfmt ", 1, 8, f); fwrite(&fmt_len, 4, 1, f); fwrite(&pcm, 2, 1, f);
    fwrite(&channels, 2, 1, f); fwrite(&sample_rate, 4, 1, f); fwrite(&byte_rate, 4, 1, f); fwrite(&block, 2, 1, f); fwrite(&bits, 2, 1, f);
    fwrite("data", 1, 4, f); fwrite(&data_bytes, 4, 1, f);
    for (float s : samples) { float a = s * 32767.0f; a = a < -32768.0f ? -32768.0f : (a > 32767.0f ? 32767.0f : a); const int16_t v = (int16_t)a; fwrite(&v, 2, 1, f); }
    fclose(f);
}

// ---------------- utils/cache.rs ----------------
namespace cache {
void save_cache(const std::string& path, const std::vector<int64_t>& codes, const std::vector<float>& emb) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot create " + path);
    const uint32_t version = 1;
    const uint64_t nc = codes.size(), ne = emb.size();
    fwrite("TTSC", 1, 4, f); fwrite(&version, 4, 1, f);
    fwrite(&nc, 8, 1, f); if (nc) fwrite(codes.data(), 8, nc, f);
    fwrite(&ne, 8, 1, f); if (ne) fwrite(emb.data(), 4, ne, f);
    fclose(f);
}
void load_cache(const std::string& path, std::vector<int64_t>& codes, std::vector<float>& emb) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    auto fail = [&](const char* m) { fclose(f); throw std::runtime_error(m); };
    char magic[4]; uint32_t version = 0; uint64_t n = 0;
    if (fread(magic, 1, 4, f) != 4) fail("failed to fill whole buffer");
    if (memcmp(magic, "TTSC", 4) != 0) fail("Invalid magic bytes");
    if (fread(&version, 4, 1, f) != 1) fail("failed to fill whole buffer");
    if (version != 1) fail("Unsupported version");
    if (fread(&n, 8, 1, f) != 1 || n > (1ull << 32)) fail("failed to fill whole buffer");
    codes.resize(n);
    if (n && fread(codes.data(), 8, n, f) != n) fail("failed to fill whole buffer");
    if (fread(&n, 8, 1, f) != 1 || n > (1ull << 32)) fail("failed to fill whole buffer");
    emb.resize(n);
    if (n && fread(emb.data(), 4, n, f) != n) fail("failed to fill whole buffer");
    fclose(f);
}
} // namespace cache

AudioSample AudioSample::load_wav(const std::string& path) { // audio.rs:11-23 (RIFF/WAVE PCM chunks; samples interpreted as i16)
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    std::vector<unsigned char> d;
    unsigned char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) d.insert(d.end(), buf, buf + n);
    fclose(f);
    if (d.size() < 12 || memcmp(d.data(), "RIFF", 4) != 0 || memcmp(d.data() + 8, "WAVE", 4) != 0) throw std::runtime_error("no RIFF tag found");
    AudioSample a;
    bool have_fmt = false;
    for (size_t p = 12; p + 8 <= d.size();) {
        uint32_t len; memcpy(&len, d.data() + p + 4, 4);
        const unsigned char* body = d.data() + p + 8;
        const size_t avail = std::min<size_t>(len, d.size() - p - 8);
        if (memcmp(d.data() + p, "fmt ", 4) == 0 && avail >= 16) {
            uint16_t ch; uint32_t sr; memcpy(&ch, body + 2, 2); memcpy(&sr, body + 4, 4);
            a.channels = ch; a.sample_rate = sr; have_fmt = true;
        } else if (memcmp(d.data() + p, "data", 4) == 0) {
            if (!have_fmt) throw std::runtime_error("data chunk before fmt chunk");
            a.samples.resize(avail / 2);
            for (size_t i = 0; i < a.samples.size(); i++) { int16_t v; memcpy(&v, body + 2 * i, 2); a.samples[i] = (float)v / 32768.0f; }
            return a;
        }
        p += 8 + (size_t)len + (len & 1);
    }
    throw std::runtime_error("no data chunk found");
}

TtsEngine TtsEngine::new_(const std::string& model_dir, const std::string& quant, Tokenizer tok) {
    TtsEngine t;
    q3tts_engine_params p;
    q3tts_engine_params_default(&p);
    p.model_dir = model_dir.c_str(); p.quant = quant.c_str(); p.max_batch = 1;
    if (q3tts_engine_create(&p, &t.e_) != Q3TTS_OK) throw std::runtime_error(std::string("Failed to load TtsEngine: ") + q3tts_last_error());
    t.tok_ = std::move(tok);
    { // engine.rs:106-127: the voice-clone encoders are optional; a missing file leaves the Option empty
        const int32_t dev = q3tts_engine_device(t.e_);
        auto open_if = [&](const char* file, q3tts_onnx_session** out) {
            const std::string path = model_dir + "/onnx/" + file;
            if (FILE* f = fopen(path.c_str(), "rb")) {
                fclose(f);
                if (q3tts_onnx_session_open(path.c_str(), dev, out) != Q3TTS_OK) throw std::runtime_error(std::string("Failed to load ") + file + ": " + q3tts_last_error());
                char buf[1024];
                if (q3tts_onnx_session_unsupported(*out, buf, sizeof(buf)) > 0) throw std::runtime_error(std::string(file) + " uses operators this engine cannot execute: " + buf);
            }
        };
        open_if("qwen3_tts_codec_encoder.onnx", &t.enc_);
        open_if("qwen3_tts_speaker_encoder.onnx", &t.spk_);
    }
    if (!t.tok_) { // engine.rs:103-104 / utils/tokenizer.rs:9-15: <model_dir>/tokenizer/tokenizer.json through the engine's own BPE reader (row f-3)
        const std::string tj = model_dir + "/tokenizer/tokenizer.json";
        if (FILE* f = fopen(tj.c_str(), "rb")) {
            fclose(f);
            q3tts_tokenizer* th = nullptr;
            if (q3tts_tokenizer_open(tj.c_str(), &th) != Q3TTS_OK) { q3tts_engine_destroy(t.e_); t.e_ = nullptr; throw std::runtime_error(std::string("Failed to load tokenizer: ") + q3tts_last_error()); }
            std::shared_ptr<q3tts_tokenizer> sp(th, [](q3tts_tokenizer* p) { q3tts_tokenizer_close(p); });
            t.tok_ = [sp](const std::string& text) {
                std::vector<int32_t> ids(2 * text.size() + 16);
                const int32_t n = q3tts_tokenizer_encode(sp.get(), text.c_str(), ids.data(), (int32_t)ids.size());
                if (n < 0) throw std::runtime_error(std::string("Error encoding text: ") + q3tts_last_error());
                ids.resize((size_t)n);
                return ids;
            };
        }
    }
    // engine.rs:156-166: <model_dir>/preset_speakers, else ./speakers
    for (const std::string& d : {model_dir + "/preset_speakers", std::string("speakers")}) {
        if (DIR* dd = opendir(d.c_str())) { closedir(dd); t.load_speakers(d); break; }
    }
    return t;
}
TtsEngine::TtsEngine(TtsEngine&& o) noexcept { *this = std::move(o); }
TtsEngine& TtsEngine::operator=(TtsEngine&& o) noexcept {
    if (this != &o) {
        if (enc_) q3tts_onnx_session_close(enc_);
        if (spk_) q3tts_onnx_session_close(spk_);
        enc_ = o.enc_; spk_ = o.spk_; o.enc_ = o.spk_ = nullptr;
        if (e_) q3tts_engine_destroy(e_); e_ = o.e_; o.e_ = nullptr; tok_ = std::move(o.tok_); speakers_ = std::move(o.speakers_); max_steps_ = o.max_steps_; sampler_ = o.sampler_;
    }
    return *this;
}
TtsEngine::~TtsEngine() {
    if (enc_) q3tts_onnx_session_close(enc_);
    if (spk_) q3tts_onnx_session_close(spk_);
    if (e_) q3tts_engine_destroy(e_);
}

void TtsEngine::load_speakers(const std::string& dir) {
    DIR* d = opendir(dir.c_str());
    if (!d) throw std::runtime_error("cannot read " + dir);
    while (dirent* e = readdir(d)) {
        const std::string n = e->d_name;
        if (n.size() > 5 && n.substr(n.size() - 5) == ".json") {
            try { speakers_[n.substr(0, n.size() - 5)] = VoiceFile::load(dir + "/" + n); } catch (...) {} // `if let Ok(voice)`, :196
        }
    }
    closedir(d);
}
const VoiceFile& TtsEngine::get_speaker(const std::string& id) const {
    auto it = speakers_.find(id);
    if (it != speakers_.end()) return it->second;
    for (auto& kv : speakers_) if (kv.second.name && *kv.second.name == id) return kv.second;
    it = speakers_.find("vivian");
    if (it != speakers_.end()) return it->second;
    if (speakers_.empty()) throw std::runtime_error("No speakers loaded in engine!");
    return speakers_.begin()->second;
}

AudioSample TtsEngine::generate_with_voice(const std::string& text, const VoiceFile& voice, const std::optional<std::string>& instruct) {
    if (!tok_) throw std::runtime_error("no tokenizer attached (SURVEY row f-3): use generate_with_voice_ids");
    const auto ids = tok_(text);
    std::vector<int32_t> ins, rt;
    if (instruct) ins = tok_(*instruct);
    if (!voice.audio_codes.empty()) rt = tok_(voice.ref_text);
    return generate_with_voice_ids(ids, voice, instruct ? &ins : nullptr, voice.audio_codes.empty() ? nullptr : &rt);
}

int TtsEngine::build_prompt(const std::vector<int32_t>& text_ids, const VoiceFile& voice, const std::vector<int32_t>* ins,
                            const std::vector<int32_t>* ref_text_ids, std::vector<float>& prompt) const {
    const q3tts_assets* a = q3tts_engine_assets(e_);
    const int max_rows = 1024;
    prompt.assign((size_t)max_rows * 2048, 0.0f);
    int n;
    if (voice.audio_codes.empty()) // engine.rs:398-412: Chinese lang id 2055, speaker injected as marker + spk_emb
        n = q3tts_prompt_build_core(a, text_ids.data(), (int)text_ids.size(), 2055, -1, voice.speaker_embedding.data(), ins ? ins->data() : nullptr,
                                    ins ? (int)ins->size() : 0, nullptr, 0, prompt.data(), max_rows);
    else { // :413-427
        std::vector<int32_t> rc(voice.audio_codes.begin(), voice.audio_codes.end()), rt;
        if (ref_text_ids) rt = *ref_text_ids;
        n = q3tts_prompt_build_clone(a, text_ids.data(), (int)text_ids.size(), rc.data(), (int)rc.size(), rt.data(), (int)rt.size(),
                                     voice.speaker_embedding.data(), 2055, ins ? ins->data() : nullptr, ins ? (int)ins->size() : 0, prompt.data(), max_rows);
    }
    if (n < 0) throw std::runtime_error(q3tts_last_error());
    return n;
}
void TtsEngine::fill_request(q3tts_request& r, const std::vector<float>& prompt, int n) const {
    r.prompt = prompt.data(); r.n_prompt = n; r.max_steps = (int)max_steps_;
    r.sampler.temperature = sampler_.temperature; r.sampler.top_k = sampler_.top_k; r.sampler.top_p = sampler_.top_p;
    r.sampler.has_seed = sampler_.seed ? 1 : 0; r.sampler.seed = sampler_.seed.value_or(0);
}

AudioSample TtsEngine::generate_with_voice_ids(const std::vector<int32_t>& text_ids, const VoiceFile& voice, const std::vector<int32_t>* ins,
                                               const std::vector<int32_t>* ref_text_ids, std::vector<int32_t>* codes_out) {
    std::vector<float> prompt;
    const int n = build_prompt(text_ids, voice, ins, ref_text_ids, prompt);
    q3tts_request r{};
    fill_request(r, prompt, n);
    std::vector<int32_t> codes(max_steps_ * 16);
    AudioSample out;
    out.samples.resize(max_steps_ * 1920);
    r.codes_out = codes.data(); r.pcm_out = out.samples.data(); r.pcm_capacity = (int64_t)out.samples.size();
    if (q3tts_generate_batch(e_, &r, 1, 1) != Q3TTS_OK) throw std::runtime_error(q3tts_last_error());
    out.samples.resize((size_t)r.n_pcm);
    if (codes_out) codes_out->assign(codes.begin(), codes.begin() + (size_t)r.n_frames * 16);
    return out; // AudioSample{samples, 24000, 1}: engine.rs:651-655
}

AudioSample TtsEngine::generate_ids(const std::vector<int32_t>& text_ids, const std::string& ref_audio_path, const std::vector<int32_t>& ref_text_ids,
                                    const std::vector<int32_t>* ins, std::vector<int32_t>* codes_out) {
    // process_reference, engine.rs:275-301
    std::string cache_path = ref_audio_path;
    const size_t slash = cache_path.find_last_of('/'), dot = cache_path.find_last_of('.');
    if (dot != std::string::npos && (slash == std::string::npos || dot > slash)) cache_path.resize(dot);
    cache_path += ".cache"; // Path::with_extension("cache")
    VoiceFile v;
    bool hit = false;
    if (FILE* f = fopen(cache_path.c_str(), "rb")) {
        fclose(f);
        try { cache::load_cache(cache_path, v.audio_codes, v.speaker_embedding); hit = true; } catch (...) {} // `if let Ok(..)`: a bad cache falls through
    }
    if (!hit) {
        AudioSample a;
        try { a = AudioSample::load_wav(ref_audio_path); } catch (const std::exception& e) { throw std::runtime_error(std::string("Failed to load audio: ") + e.what()); }
        if (!enc_) throw std::runtime_error("AudioEncoder not loaded (required for processing raw audio)");     // :288-290
        try { v.audio_codes = encode_audio(a.samples); } catch (const std::exception& e) { throw std::runtime_error(std::string("Audio encode failed: ") + e.what()); }
        if (!spk_) throw std::runtime_error("SpeakerEncoder not loaded (required for processing raw audio)"); // :294-296
        try { v.speaker_embedding = encode_speaker(a.samples); } catch (const std::exception& e) { throw std::runtime_error(std::string("Speaker extraction failed: ") + e.what()); }
        try { cache::save_cache(cache_path, v.audio_codes, v.speaker_embedding); } catch (...) {} // `let _ = cache::save_cache(..)`
    }
    if (v.speaker_embedding.size() != 2048) throw std::runtime_error("cached speaker embedding must have 2048 values");
    return generate_with_voice_ids(text_ids, v, ins, &ref_text_ids, codes_out);
}
AudioSample TtsEngine::generate(const std::string& text, const std::string& ref_audio_path, const std::string& ref_text,
                                const std::optional<std::string>& instruct) {
    if (!tok_) throw std::runtime_error("no tokenizer attached (SURVEY row f-3): use generate_ids");
    std::vector<int32_t> ins;
    if (instruct) ins = tok_(*instruct);
    return generate_ids(tok_(text), ref_audio_path, tok_(ref_text), instruct ? &ins : nullptr);
}

AudioSample TtsEngine::generate_with_voice_ids_stream(const std::vector<int32_t>& text_ids, const VoiceFile& voice, const ChunkFn& on_chunk,
                                                      const std::vector<int32_t>* ins, const std::vector<int32_t>* ref_text_ids,
                                                      std::vector<int32_t>* codes_out) {
    std::vector<float> prompt;
    const int n = build_prompt(text_ids, voice, ins, ref_text_ids, prompt);
    q3tts_request r{};
    fill_request(r, prompt, n);
    int64_t id = 0;
    if (q3tts_submit(e_, &r, 1, &id) != Q3TTS_OK) throw std::runtime_error(q3tts_last_error());
    AudioSample out;
    std::vector<float> buf(4 * 1920 * 4);
    for (;;) {
        int32_t busy = 0;
        if (q3tts_sched_step(e_, &busy) != Q3TTS_OK) { q3tts_release(e_, id); throw std::runtime_error(q3tts_last_error()); }
        q3tts_req_status st{};
        if (q3tts_poll(e_, id, &st) != Q3TTS_OK) throw std::runtime_error(q3tts_last_error());
        while ((int64_t)out.samples.size() < st.n_pcm) { // hand over whatever the codec finished since the last look
            int64_t got = 0;
            if (q3tts_fetch(e_, id, nullptr, 0, 0, buf.data(), (int64_t)out.samples.size(), (int64_t)buf.size(), nullptr, &got) != Q3TTS_OK)
                throw std::runtime_error(q3tts_last_error());
            if (got <= 0) break;
            if (on_chunk) on_chunk(buf.data(), (size_t)got);
            out.samples.insert(out.samples.end(), buf.begin(), buf.begin() + got);
        }
        if (st.state == Q3TTS_REQ_DONE && (int64_t)out.samples.size() >= st.n_pcm) {
            if (codes_out) {
                codes_out->assign((size_t)st.n_frames * 16, 0);
                int32_t gf = 0;
                q3tts_fetch(e_, id, codes_out->data(), 0, st.n_frames, nullptr, 0, 0, &gf, nullptr);
            }
            break;
        }
        if (st.state == Q3TTS_REQ_FAILED) { q3tts_release(e_, id); throw std::runtime_error("generation failed"); }
    }
    q3tts_release(e_, id);
    return out;
}

std::vector<int64_t> TtsEngine::encode_audio(const std::vector<float>& audio) const { // onnx.rs:97-121
    if (!enc_) throw std::runtime_error("AudioEncoder not loaded. Please ensure models/onnx/qwen3_tts_codec_encoder.onnx exists.");
    const int64_t shape[2] = {1, (int64_t)audio.size()};
    if (q3tts_onnx_session_set_input(enc_, "input_values", 1, audio.data(), shape, 2) != Q3TTS_OK || q3tts_onnx_session_run(enc_) != Q3TTS_OK)
        throw std::runtime_error(q3tts_last_error());
    int32_t dtype = 0, rank = 0; int64_t sh[8];
    if (q3tts_onnx_session_output_info(enc_, "audio_codes", &dtype, &rank, sh) != Q3TTS_OK) throw std::runtime_error(q3tts_last_error());
    if (dtype != 7) throw std::runtime_error("audio_codes is not an int64 tensor");
    int64_t n = 1; for (int i = 0; i < rank; i++) n *= sh[i];
    std::vector<int64_t> codes((size_t)n);
    if (n && q3tts_onnx_session_output(enc_, "audio_codes", codes.data(), n * 8) != Q3TTS_OK) throw std::runtime_error(q3tts_last_error());
    return codes; // [1, frames, 16] flattened, as the reference keeps it
}
std::vector<float> TtsEngine::encode_speaker(const std::vector<float>& audio) const { // onnx.rs:140-163
    if (!spk_) throw std::runtime_error("SpeakerEncoder not loaded. Please ensure models/onnx/qwen3_tts_speaker_encoder.onnx exists.");
    const int32_t frames = q3tts_mel_frames((int32_t)audio.size());
    if (frames <= 0) throw std::runtime_error("audio too short for the mel front end");
    std::vector<float> mel((size_t)frames * 128);
    if (q3tts_mel(audio.data(), (int32_t)audio.size(), mel.data()) != Q3TTS_OK) throw std::runtime_error(q3tts_last_error());
    const int64_t shape[3] = {1, frames, 128};
    if (q3tts_onnx_session_set_input(spk_, "mels", 1, mel.data(), shape, 3) != Q3TTS_OK || q3tts_onnx_session_run(spk_) != Q3TTS_OK)
        throw std::runtime_error(q3tts_last_error());
    int32_t dtype = 0, rank = 0; int64_t sh[8];
    if (q3tts_onnx_session_output_info(spk_, "spk_emb", &dtype, &rank, sh) != Q3TTS_OK) throw std::runtime_error(q3tts_last_error());
    int64_t n = 1; for (int i = 0; i < rank; i++) n *= sh[i];
    std::vector<float> emb((size_t)n);
    if (dtype == 7) throw std::runtime_error("spk_emb is not a float tensor");
    if (n && q3tts_onnx_session_output(spk_, "spk_emb", emb.data(), n * 4) != Q3TTS_OK) throw std::runtime_error(q3tts_last_error());
    return emb;
}

// hound-style WAV payload for create_voice_file (engine.rs:336-370): f32, i16 or i32 samples; stereo keeps channel 0
static std::vector<float> read_wav_for_voice(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("WAV error: cannot open " + path);
    std::vector<unsigned char> d;
    unsigned char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) d.insert(d.end(), buf, buf + n);
    fclose(f);
    if (d.size() < 12 || memcmp(d.data(), "RIFF", 4) != 0 || memcmp(d.data() + 8, "WAVE", 4) != 0) throw std::runtime_error("WAV error: no RIFF tag found");
    uint16_t fmt = 0, ch = 0, bits = 0; uint32_t sr = 0; bool have_fmt = false;
    for (size_t p = 12; p + 8 <= d.size();) {
        uint32_t len; memcpy(&len, d.data() + p + 4, 4);
        const unsigned char* body = d.data() + p + 8;
        const size_t avail = std::min<size_t>(len, d.size() - p - 8);
        if (memcmp(d.data() + p, "fmt ", 4) == 0 && avail >= 16) {
            memcpy(&fmt, body, 2); memcpy(&ch, body + 2, 2); memcpy(&sr, body + 4, 4); memcpy(&bits, body + 14, 2);
            if (fmt == 0xFFFE && avail >= 26) memcpy(&fmt, body + 24, 2); // WAVE_FORMAT_EXTENSIBLE: the sub-format's first word
            have_fmt = true;
        } else if (memcmp(d.data() + p, "data", 4) == 0) {
            if (!have_fmt) throw std::runtime_error("WAV error: data chunk before fmt chunk");
            if (sr != 24000) throw std::runtime_error("Expected 24000Hz audio, found " + std::to_string(sr) + "Hz");
            std::vector<float> all;
            if (fmt == 3 && bits == 32) { all.resize(avail / 4); memcpy(all.data(), body, all.size() * 4); }
            else if (fmt == 1 && bits == 16) { all.resize(avail / 2); for (size_t i = 0; i < all.size(); i++) { int16_t v; memcpy(&v, body + 2 * i, 2); all[i] = (float)v / 32768.0f; } }
            else if (fmt == 1 && bits == 32) { all.resize(avail / 4); for (size_t i = 0; i < all.size(); i++) { int32_t v; memcpy(&v, body + 4 * i, 4); all[i] = (float)v / 2147483648.0f; } }
            else throw std::runtime_error(std::string("Unsupported WAV format: ") + (fmt == 3 ? "Float" : "Int") + " " + std::to_string(bits) + " bits");
            if (ch > 1) { std::vector<float> mono(all.size() / ch); for (size_t i = 0; i < mono.size(); i++) mono[i] = all[i * ch]; return mono; }
            return all;
        }
        p += 8 + (size_t)len + (len & 1);
    }
    throw std::runtime_error("WAV error: no data chunk found");
}

VoiceFile TtsEngine::create_voice_file(const std::string& audio_path, const std::string& ref_text) { // engine.rs:324-387
    if (!enc_) throw std::runtime_error("AudioEncoder not loaded. Please ensure models/onnx/qwen3_tts_codec_encoder.onnx exists.");
    if (!spk_) throw std::runtime_error("SpeakerEncoder not loaded. Please ensure models/onnx/qwen3_tts_speaker_encoder.onnx exists.");
    const std::vector<float> audio = read_wav_for_voice(audio_path);
    VoiceFile v;
    v.ref_text = ref_text;
    v.audio_codes = encode_audio(audio);
    v.speaker_embedding = encode_speaker(audio);
    return v;
}

} // namespace q3tts
