// Ablation / timing harness for the batched int8 GEMM (k_gemm_q8_mfma2): times the production launchers and ablated instantiations of
// the kernel on the full model's shapes.  Build + run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Iqwen3-tts-rust_amd/csrc -o /tmp/ubench_gemm scripts/ubench_gemm.hip qwen3-tts-rust_amd/csrc/gguf.cpp qwen3-tts-rust_amd/csrc/transformer.cpp qwen3-tts-rust_amd/csrc/kernels_fused.hip && /tmp/ubench_gemm
#include "../qwen3-tts-rust_amd/csrc/kernels.hip"
#include "../qwen3-tts-rust_amd/csrc/transformer.h"
#include <functional>
#include <vector>
using namespace q3;

static float time_it(const std::function<void()>& f, int iters = 200) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; i++) f();
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; i++) f();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;
}

template <bool GU, int ABL>
static void run_abl(const Q8Mat& m, int nrows, int ff, const int8_t* xq, const uint16_t* xd, float* out, int8_t* aq, uint16_t* ad, int ntok, int z, const char* tag) {
    const int nseg = m.K >> 8, nsseg = (nseg + 7) / 8, nw = nseg < 8 ? nseg : 8, rgs = (GU ? ff : nrows) / 32;
    const float us = time_it([&] {
        hipLaunchKernelGGL((k_gemm_q8_mfma2<GU, ABL>), dim3(rgs, GU ? 1 : nsseg, z), dim3(64 * nw), 0, 0, m, 0, GU ? ff : nrows, xq, xd, out, nrows, ntok, GU ? ff : 0, aq, ad);
    });
    printf("    %-34s %7.2f us\n", tag, us);
}

template <bool GU>
static void run_wave(const Q8Mat& m, int nrows, int ff, const int8_t* xq, const uint16_t* xd, float* out, int8_t* aq, uint16_t* ad, int ntok) {
    const int nseg = m.K >> 8, nsseg = (nseg + 7) / 8, rgs = (GU ? ff : nrows) / 32;
    const float us = time_it([&] {
        hipLaunchKernelGGL((k_gemm_q8_wave<GU>), dim3((rgs + 3) / 4, GU ? 1 : nsseg, (ntok + 31) / 32), dim3(256), 0, 0, m, 0, GU ? ff : nrows, xq, xd, out, nrows, ntok, GU ? ff : 0, aq, ad);
    });
    printf("    %-34s %7.2f us  (%d waves)\n", "wave-per-tile form", us, rgs * (GU ? 1 : nsseg) * ((ntok + 31) / 32));
}

int main(int argc, char** argv) {
    const bool quick = argc > 1 && !strcmp(argv[1], "quick"); // production launcher only, talker shapes at 64 / 256 tokens (for PMC passes)
    struct Shape { const char* name; int n, k, gu; } shapes[] = {
        {"talker gate/up 12288x2048", 12288, 2048, 1}, {"talker down 2048x6144", 2048, 6144, 0}, {"talker qkv 4096x2048", 4096, 2048, 0},
        {"talker o 2048x2048", 2048, 2048, 0}, {"pred gate/up 6144x1024", 6144, 1024, 1}, {"pred qkv 4096x1024", 4096, 1024, 0},
        {"pred o 1024x2048", 1024, 2048, 0}, {"pred down 1024x3072", 1024, 3072, 0}};
    for (auto& sh : shapes) {
        if (quick && &sh - shapes > 1) break;
        const int n = sh.n, k = sh.k;
        std::vector<uint8_t> raw((size_t)n * (k / 32) * 34);
        for (size_t i = 0; i < raw.size(); i++) raw[i] = (uint8_t)(i * 2654435761u >> 13);
        for (size_t b = 0; b < (size_t)n * (k / 32); b++) { raw[b * 34] = 0x00; raw[b * 34 + 1] = 0x1C; } // d = 2^-8
        DevBuf<uint8_t> storage;
        Q8Mat m = q8mat_from_host(raw.data(), n, k, storage);
        for (int ntok : {64, 128, 256, 512}) {
            if (quick && ntok != 64 && ntok != 256) continue;
            DevBuf<int8_t> xq((size_t)ntok * k); DevBuf<uint16_t> xd((size_t)ntok * k / 32);
            std::vector<int8_t> hx(xq.n); for (size_t i = 0; i < hx.size(); i++) hx[i] = (int8_t)((i * 40503u >> 7) & 0xFF);
            std::vector<uint16_t> hd(xd.n, 0x2000);
            xq.upload(hx.data(), hx.size()); xd.upload(hd.data(), hd.size());
            const int nsseg = ((k >> 8) + 7) / 8, ff = n / 2;
            DevBuf<float> out((size_t)nsseg * ntok * n); DevBuf<int8_t> aq((size_t)ntok * n); DevBuf<uint16_t> ad((size_t)ntok * n / 32);
            const double mb = (double)n * k * 1.0625 / 1e6;
            printf("%s, %d tokens (%.1f MB weights; HBM floor %.2f us at 6.3 TB/s)\n", sh.name, ntok, mb, mb / 6.3);
            const float us = time_it([&] {
                if (sh.gu) launch_gateup_mfma(0, m, ff, xq.p, xd.p, aq.p, ad.p, ntok);
                else launch_gemv_q8(0, m, 0, n, xq.p, xd.p, out.p, n, ntok);
            });
            printf("    %-34s %7.2f us  (%.2f TB/s)\n", "production launcher", us, mb / us);
            if (quick) continue;
            if (sh.gu) run_wave<true>(m, n, ff, xq.p, xd.p, out.p, aq.p, ad.p, ntok); else run_wave<false>(m, n, ff, xq.p, xd.p, out.p, aq.p, ad.p, ntok);
            if (ntok != 64 && ntok != 256) continue;
            const int ntiles = (ntok + 31) / 32;
            for (int z : {1, 2, ntiles}) {
                if (z > ntiles) continue;
                char tag[64];
                snprintf(tag, sizeof tag, "z=%d full", z);
                if (sh.gu) run_abl<true, 0>(m, n, ff, xq.p, xd.p, out.p, aq.p, ad.p, ntok, z, tag); else run_abl<false, 0>(m, n, ff, xq.p, xd.p, out.p, aq.p, ad.p, ntok, z, tag);
            }
            const int z = ntiles < 2 ? ntiles : 2;
#define ABL_RUN(A, T) if (sh.gu) run_abl<true, A>(m, n, ff, xq.p, xd.p, out.p, aq.p, ad.p, ntok, z, T); else run_abl<false, A>(m, n, ff, xq.p, xd.p, out.p, aq.p, ad.p, ntok, z, T);
            ABL_RUN(1, "z=2 no mfma/valu");
            ABL_RUN(2, "z=2 no x loads");
            ABL_RUN(4, "z=2 no w loads");
            ABL_RUN(8, "z=2 no combine/epilogue");
            ABL_RUN(9, "z=2 loads only");
            ABL_RUN(6, "z=2 compute+epilogue only");
            ABL_RUN(15, "z=2 nothing (launch floor)");
        }
    }
    return 0;
}
