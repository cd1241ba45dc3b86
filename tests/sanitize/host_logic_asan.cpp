// CPU-only AddressSanitizer/UBSan driver for the host-side logic (GGUF parser, assets, prompt builders, sampler, chunker).
// Built and run by tests/test_sanitize_cpu.py with g++ -fsanitize=address,undefined (GPU sanitizers are not available on the pool).
#include "../../qwen3-tts-rust_amd/csrc/host_logic.h"
#include "../../qwen3-tts-rust_amd/csrc/q3_common.h"
#include <cstdio>
#include <random>

int main(int argc, char** argv) {
    using namespace q3;
    if (argc < 2) { fprintf(stderr, "usage: host_logic_asan <qwen3_assets.gguf>\n"); return 2; }
    HostAssets a(argv[1]);
    std::mt19937 rng(7);
    std::vector<float> spk(2048);
    for (auto& v : spk) v = (float)(rng() % 2000) / 1000.0f - 1.0f;
    size_t rows = 0;
    for (int n_text : {0, 1, 7, 64}) {
        std::vector<int32_t> text(n_text), instr = {1, 2, 3};
        for (auto& t : text) t = (int32_t)(rng() % 200000) - 100; // includes out-of-range and negative ids (fallback pattern path)
        int lang = 2055, spk_id = 3065;
        rows += PromptBuilder::build_core(a, text, &lang, &spk_id, nullptr, nullptr, nullptr).n_rows;
        rows += PromptBuilder::build_core(a, text, nullptr, nullptr, spk.data(), &instr, nullptr).n_rows;
        std::vector<int32_t> ref_codes((size_t)(n_text % 5) * 16), ref_text(n_text % 3);
        for (auto& c : ref_codes) c = (int32_t)(rng() % 4096) - 10; // out-of-range codes -> zero rows
        for (auto& t : ref_text) t = (int32_t)(rng() % 1000);
        rows += PromptBuilder::build_clone_prompt(a, text, ref_codes, ref_text, spk.data(), lang, n_text % 2 ? &instr : nullptr).n_rows;
    }
    std::vector<float> out(2048);
    a.codec_embedding(0, -5, out.data()); a.codec_embedding(15, 1 << 30, out.data()); a.codec_embedding(99, 0, out.data());
    a.text_embedding(-1, out.data()); a.text_embedding((int64_t)1 << 40, out.data());
    // sampler: ties, -inf, tiny ranges, top_k / top_p edges
    long acc = 0;
    for (int trial = 0; trial < 50; trial++) {
        const int n = 1 + (int)(rng() % 3000);
        std::vector<float> lg(n);
        for (auto& v : lg) v = (float)(rng() % 1000) / 100.0f;
        if (n > 3) { lg[1] = lg[2] = 99.0f; lg[0] = -INFINITY; }
        Sampler s(trial % 3 ? 0.7f : 0.0f, trial % 4 ? 40 : 0, trial % 5 ? 0.9f : 1.0f, 42 + trial);
        for (int d = 0; d < 20; d++) acc += s.sample(lg.data(), n, trial % 2 ? 0 : n / 2, n + (trial % 7 == 0 ? 5 : 0));
    }
    // chunker: every frame count 0..9, with and without final
    for (int frames = 0; frames < 10; frames++) {
        long got = 0; int finals = 0;
        Chunker c([&](const int64_t* codes, int n, bool fin) { for (int i = 0; i < n; i++) got += codes[i]; finals += fin; });
        for (int f = 0; f < frames; f++) { int64_t fc[16]; for (int q = 0; q < 16; q++) fc[q] = f * 16 + q; c.push(fc, 16, false); }
        c.push(nullptr, 0, true);
        acc += got + finals;
    }
    printf("ok rows=%zu acc=%ld\n", rows, acc);
    return 0;
}
