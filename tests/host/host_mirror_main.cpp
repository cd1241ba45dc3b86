// Exercises the C++ host mirror of the reference API (qwen3-tts-rust_amd/host/tts_engine.hpp) on the tiny model:
// TtsEngine::new_, load_speakers / get_speaker (vivian fallback), set_sampler_config, set_max_steps, generate_with_voice_ids
// (preset and clone voices), the streaming form, create_voice_file's error, AudioSample::save_wav.  Prints the codes for the
// Python test to compare with the ctypes path.
#include "../../qwen3-tts-rust_amd/host/tts_engine.hpp"
#include <cstdio>
#include <stdexcept>

int main(int argc, char** argv) {
    using namespace q3tts;
    if (argc < 4) { fprintf(stderr, "usage: host_mirror_main <model_dir> <speakers_dir> <out.wav>\n"); return 2; }
    try {
        TtsEngine eng = TtsEngine::new_(argv[1], "q8_0");
        eng.load_speakers(argv[2]);
        const VoiceFile& v = eng.get_speaker("no-such-speaker");           // falls back to vivian (engine.rs:211-231)
        if (v.speaker_embedding.size() != 2048) throw std::runtime_error("speaker embedding size");
        SamplerConfig sc; sc.temperature = 0.0f; sc.seed = 42;
        eng.set_sampler_config(sc);
        eng.set_max_steps(10);
        std::vector<int32_t> ids;
        for (int i = 0; i < 8; i++) ids.push_back(100 + i);
        std::vector<int32_t> codes, codes2, codes3;
        AudioSample a = eng.generate_with_voice_ids(ids, v, nullptr, nullptr, &codes);
        a.save_wav(argv[3]);
        size_t n_chunks = 0, streamed = 0;
        AudioSample b = eng.generate_with_voice_ids_stream(ids, v, [&](const float*, size_t n) { n_chunks++; streamed += n; }, nullptr, nullptr, &codes2);
        if (codes != codes2 || a.samples != b.samples || streamed != b.samples.size() || n_chunks < 2) throw std::runtime_error("streaming result differs");
        VoiceFile clone = v;                                                // clone voice: reference codes + reference text ids
        for (int i = 0; i < 3 * 16; i++) clone.audio_codes.push_back((i * 37) % 2048);
        std::vector<int32_t> ref_text = {7, 8, 9};
        sc.temperature = 0.7f; eng.set_sampler_config(sc);                  // sampled, seeded
        AudioSample c = eng.generate_with_voice_ids(ids, clone, nullptr, &ref_text, &codes3);
        // generate (engine.rs:243-271) through process_reference's ".cache" short-cut (:276-281): same result as the clone VoiceFile path
        const std::string ref_wav = std::string(argv[3]) + ".ref.wav", ref_cache = std::string(argv[3]) + ".ref.cache";
        cache::save_cache(ref_cache, clone.audio_codes, clone.speaker_embedding);
        std::vector<int32_t> codes4;
        AudioSample d = eng.generate_ids(ids, ref_wav, ref_text, nullptr, &codes4);
        if (codes4 != codes3 || d.samples != c.samples) throw std::runtime_error("cache-backed generate differs from the clone VoiceFile path");
        try { eng.generate_ids(ids, std::string(argv[3]), ref_text); throw std::runtime_error("generate without cache or encoders should fail"); }
        catch (const std::runtime_error& e) { if (std::string(e.what()).find("AudioEncoder not loaded (required for processing raw audio)") == std::string::npos) throw; }
        try { eng.create_voice_file("x.wav", "hello"); throw std::runtime_error("create_voice_file should fail"); }
        catch (const std::runtime_error& e) { if (std::string(e.what()).find("AudioEncoder not loaded") == std::string::npos) throw; }
        if (eng.has_tokenizer()) { // text in -> audio out (engine.rs:390-435 with utils/tokenizer.rs): the text path must equal the ids path
            const std::string text = "Hello, it's 42 degrees!";
            const std::vector<int32_t> tids = eng.encode(text);
            sc.temperature = 0.0f; eng.set_sampler_config(sc);
            std::vector<int32_t> ca, cb;
            AudioSample ta = eng.generate_with_voice(text, v);
            AudioSample tb = eng.generate_with_voice_ids(tids, v, nullptr, nullptr, &cb);
            if (ta.samples != tb.samples || tids.empty()) throw std::runtime_error("text path differs from the ids path");
            printf("TEXTIDS");
            for (int32_t x : tids) printf(" %d", x);
            printf("\nTEXTCODES");
            for (int32_t x : cb) printf(" %d", x);
            printf("\n");
        }
        printf("PRESET");
        for (int32_t x : codes) printf(" %d", x);
        printf("\nCLONE");
        for (int32_t x : codes3) printf(" %d", x);
        printf("\nPCM %zu %zu chunks %zu duration %.3f\nOK\n", a.samples.size(), c.samples.size(), n_chunks, a.duration());
    } catch (const std::exception& e) { fprintf(stderr, "FAILED: %s\n", e.what()); return 1; }
    return 0;
}
