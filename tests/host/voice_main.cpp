// create_voice_file / generate through the voice-clone encoders: a model directory whose onnx/ holds
// qwen3_tts_codec_encoder.onnx and qwen3_tts_speaker_encoder.onnx (graphs written by the Python test) is opened by the C++ mirror of the
// reference API; the VoiceFile it produces from a 24 kHz WAV is saved for the test to compare with the same graphs run through ctypes.
#include "../../qwen3-tts-rust_amd/host/tts_engine.hpp"
#include <cstdio>
#include <stdexcept>

int main(int argc, char** argv) {
    using namespace q3tts;
    if (argc < 5) { fprintf(stderr, "usage: voice_main <model_dir> <ref.wav> <voice_out.json> <ref2.wav>\n"); return 2; }
    try {
        TtsEngine eng = TtsEngine::new_(argv[1], "q8_0");
        if (!eng.has_encoders()) throw std::runtime_error("encoders were not loaded");
        VoiceFile v = eng.create_voice_file(argv[2], "reference text");   // engine.rs:324-387
        if (v.ref_text != "reference text" || v.audio_codes.empty() || v.audio_codes.size() % 16 != 0 || v.speaker_embedding.size() != 2048)
            throw std::runtime_error("VoiceFile has the wrong shape");
        v.save(argv[3]);
        // generate (engine.rs:243-271) from raw reference audio: process_reference runs both encoders and writes "<audio>.cache" (:275-301)
        SamplerConfig sc; sc.temperature = 0.0f; sc.seed = 42;
        eng.set_sampler_config(sc);
        eng.set_max_steps(6);
        std::vector<int32_t> ids = {100, 101, 102, 103}, ref_text = {7, 8, 9}, c1, c2;
        AudioSample a = eng.generate_ids(ids, argv[4], ref_text, nullptr, &c1);
        std::string cache_path = argv[4];
        cache_path.resize(cache_path.find_last_of('.'));
        cache_path += ".cache";
        std::vector<int64_t> cc; std::vector<float> ce;
        cache::load_cache(cache_path, cc, ce);                            // must exist now
        if (ce.size() != 2048 || cc.empty()) throw std::runtime_error("cache written by process_reference has the wrong shape");
        AudioSample b = eng.generate_ids(ids, argv[4], ref_text, nullptr, &c2);   // second call: the cache short-cut
        if (c1 != c2 || a.samples != b.samples) throw std::runtime_error("cache-backed generate differs from the encoder-backed one");
        try { eng.create_voice_file(std::string(argv[2]) + ".16k.wav", "x"); throw std::runtime_error("16 kHz audio should be refused"); }
        catch (const std::runtime_error& e) { if (std::string(e.what()).find("Expected 24000Hz audio, found 16000Hz") == std::string::npos) throw; }
        printf("CODES %zu FRAMES %zu\nOK\n", v.audio_codes.size(), v.audio_codes.size() / 16);
        return 0;
    } catch (const std::exception& e) { fprintf(stderr, "voice_main: %s\n", e.what()); return 1; }
}
