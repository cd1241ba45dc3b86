// CPU-only checks of the host mirror's file formats: ".cache" (utils/cache.rs), WAV in/out (utils/audio.rs), VoiceFile JSON.
#include "../../qwen3-tts-rust_amd/host/tts_engine.hpp"
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>

static std::string expect_error(const std::function<void()>& fn) {
    try { fn(); } catch (const std::exception& e) { return e.what(); }
    return "";
}

int main(int argc, char** argv) {
    using namespace q3tts;
    if (argc < 2) return 2;
    const std::string dir = argv[1];
    try {
        // written by Python with struct.pack in the reference's layout
        std::vector<int64_t> codes; std::vector<float> emb;
        cache::load_cache(dir + "/py.cache", codes, emb);
        if (codes.size() != 5 || codes[0] != 7 || codes[4] != -3 || emb.size() != 3 || emb[1] != 0.5f) throw std::runtime_error("load_cache content");
        cache::save_cache(dir + "/cpp.cache", codes, emb);                  // Python compares the bytes with its own
        if (expect_error([&] { cache::load_cache(dir + "/badmagic.cache", codes, emb); }) != "Invalid magic bytes") throw std::runtime_error("magic error text");
        if (expect_error([&] { cache::load_cache(dir + "/badver.cache", codes, emb); }) != "Unsupported version") throw std::runtime_error("version error text");
        if (expect_error([&] { cache::load_cache(dir + "/short.cache", codes, emb); }).empty()) throw std::runtime_error("truncated file must fail");
        AudioSample a = AudioSample::load_wav(dir + "/py.wav");            // 16-bit stereo, 22050 Hz, with an extra LIST chunk before data
        if (a.sample_rate != 22050 || a.channels != 2 || a.samples.size() != 6) throw std::runtime_error("load_wav header");
        if (a.samples[0] != -1.0f || a.samples[1] != 32767.0f / 32768.0f || a.samples[2] != 0.0f) throw std::runtime_error("load_wav scaling");
        AudioSample b; b.samples = {0.0f, 1.0f, -1.0f, 0.5f, 2.0f, -2.0f}; b.sample_rate = 24000; b.channels = 1;
        b.save_wav(dir + "/cpp.wav");
        AudioSample c = AudioSample::load_wav(dir + "/cpp.wav");
        const float want[6] = {0.0f, 32767.0f / 32768.0f, -32767.0f / 32768.0f, 16383.0f / 32768.0f, 32767.0f / 32768.0f, -1.0f}; // x32767, clamp, truncate
        for (int i = 0; i < 6; i++) if (c.samples[i] != want[i]) throw std::runtime_error("save_wav scaling at " + std::to_string(i));
        VoiceFile v = VoiceFile::load(dir + "/voice.json");                // alias spk_emb, unknown keys ignored
        if (v.speaker_embedding.size() != 4 || v.audio_codes.size() != 2 || v.ref_text != "hi" || !v.name || *v.name != "n") throw std::runtime_error("VoiceFile::load");
        v.save(dir + "/voice_out.json");
        VoiceFile w = VoiceFile::load(dir + "/voice_out.json");
        if (w.speaker_embedding != v.speaker_embedding || w.audio_codes != v.audio_codes || w.ref_text != v.ref_text) throw std::runtime_error("VoiceFile round trip");
        // written by Python's json with ensure_ascii (\uXXXX incl. a surrogate pair) and control characters: the full escape set
        VoiceFile u = VoiceFile::load(dir + "/voice_uni.json");
        if (u.ref_text != "\xe4\xbd\xa0\xe5\xa5\xbd \xf0\x9f\x8e\xa4\nline2\ttab\r\b\f\"q\"\\" || !u.name || *u.name != "\xc3\xa9") throw std::runtime_error("VoiceFile unicode escapes");
        u.save(dir + "/voice_uni_out.json");                                // Python parses this file and compares
        VoiceFile u2 = VoiceFile::load(dir + "/voice_uni_out.json");
        if (u2.ref_text != u.ref_text || *u2.name != *u.name) throw std::runtime_error("VoiceFile escape round trip");
        printf("OK\n");
    } catch (const std::exception& e) { fprintf(stderr, "FAILED: %s\n", e.what()); return 1; }
    return 0;
}
