// In-graph timing + in-kernel phase stamps for the batched int8 GEMM on the full model's shapes.
//   * timing: NREP dependent launches captured into ONE hipGraph (the way the engine's frame graph runs them: device-side launch-to-launch,
//     not host-bound eager launches), cycling through enough weight copies that talker-sized matrices come from HBM as in production
//   * stamps (-DQ3_STAMPS=1|2): thread 0 of every workgroup stores the 100 MHz s_memrealtime counter at 7 points of k_gemm_q8_mfma
//     (2 = with an s_waitcnt vmcnt(0) in front of stamp 2, i.e. "operands arrived")
// A client of the library: scripts/build_ubench.sh links it against libq3tts_stamps.so = the product objects with kernels.hip rebuilt under -DQ3_STAMPS
// (one copy of every kernel in the process: a second copy in the executable would share its host stubs with the library's).
#include "../qwen3-tts-rust_amd/csrc/kernels.h"
#include "../qwen3-tts-rust_amd/csrc/transformer.h"
namespace q3 { void set_stamp_buffer(hipStream_t st, unsigned long long* p); void set_stamp_buffer_fused(hipStream_t st, unsigned long long* p); void init_kernel_attributes(); }
#include <algorithm>
#include <functional>
#include <vector>
using namespace q3;

static hipStream_t g_st;
// f(i) enqueues launch i on g_st; returns us per launch inside a graph of nrep launches
static float time_graph(const std::function<void(int)>& f, int nrep, int replays = 20) {
    hipGraph_t g; hipGraphExec_t ge;
    Q3_HIP(hipStreamBeginCapture(g_st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < nrep; i++) f(i);
    Q3_HIP(hipStreamEndCapture(g_st, &g));
    Q3_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) Q3_HIP(hipGraphLaunch(ge, g_st));
    Q3_HIP(hipEventRecord(e0, g_st));
    for (int i = 0; i < replays; i++) Q3_HIP(hipGraphLaunch(ge, g_st));
    Q3_HIP(hipEventRecord(e1, g_st));
    Q3_HIP(hipEventSynchronize(e1));
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipGraphExecDestroy(ge); hipGraphDestroy(g); hipEventDestroy(e0); hipEventDestroy(e1);
    return ms * 1e3f / (replays * nrep);
}

struct Shape { const char* name; int n, k, gu, copies; };

// Do independent dependency chains overlap?  NL lanes, each its own stream + graph of `reps` x (predictor q,k,v / o / gate-up / down at `ntok` tokens) on its
// own buffers; all lanes launched together, wall time from the first launch to the last completion.  One lane = the engine's frame today.
static void concurrency_test(int ntok) {
    struct Lane { hipStream_t st; hipGraph_t g; hipGraphExec_t ge; std::vector<DevBuf<uint8_t>> storage; std::vector<Q8Mat> mats; DevBuf<int8_t> xq, aq; DevBuf<uint16_t> xd, ad; DevBuf<float> out; };
    const int NL = 4, reps = 20;
    const int shapes[4][3] = {{4096, 1024, 0}, {1024, 2048, 0}, {6144, 1024, 1}, {1024, 3072, 0}};
    std::vector<Lane> lanes(NL);
    for (auto& L : lanes) {
        Q3_HIP(hipStreamCreateWithFlags(&L.st, hipStreamNonBlocking));
        L.storage.resize(4);
        for (int i = 0; i < 4; i++) {
            const int n = shapes[i][0], k = shapes[i][1];
            std::vector<uint8_t> raw((size_t)n * (k / 32) * 34);
            for (size_t j = 0; j < raw.size(); j++) raw[j] = (uint8_t)(j * 2654435761u >> 13);
            for (size_t b = 0; b < (size_t)n * (k / 32); b++) { raw[b * 34] = 0x00; raw[b * 34 + 1] = 0x1C; }
            L.mats.push_back(q8mat_from_host(raw.data(), n, k, L.storage[(size_t)i]));
        }
        L.xq.alloc((size_t)ntok * 3072); L.xd.alloc((size_t)ntok * 96); L.aq.alloc((size_t)ntok * 3072); L.ad.alloc((size_t)ntok * 96); L.out.alloc((size_t)2 * ntok * 6144);
        Q3_HIP(hipMemset(L.xq.p, 1, L.xq.n)); Q3_HIP(hipMemset(L.xd.p, 0x20, L.xd.n * 2));
        Q3_HIP(hipStreamBeginCapture(L.st, hipStreamCaptureModeThreadLocal));
        for (int r = 0; r < reps; r++)
            for (int i = 0; i < 4; i++) {
                if (shapes[i][2]) launch_gateup_mfma(L.st, L.mats[(size_t)i], shapes[i][0] / 2, L.xq.p, L.xd.p, L.aq.p, L.ad.p, ntok);
                else launch_gemv_q8(L.st, L.mats[(size_t)i], 0, shapes[i][0], L.xq.p, L.xd.p, L.out.p, shapes[i][0], ntok);
            }
        Q3_HIP(hipStreamEndCapture(L.st, &L.g));
        Q3_HIP(hipGraphInstantiate(&L.ge, L.g, nullptr, nullptr, 0));
    }
    hipEvent_t e0; std::vector<hipEvent_t> e1(NL);
    hipEventCreate(&e0); for (auto& e : e1) hipEventCreate(&e);
    for (int nl : {1, 2, 4}) {
        for (int warm = 0; warm < 2; warm++) for (int l = 0; l < nl; l++) Q3_HIP(hipGraphLaunch(lanes[(size_t)l].ge, lanes[(size_t)l].st));
        Q3_HIP(hipDeviceSynchronize());
        const int rounds = 10;
        Q3_HIP(hipEventRecord(e0, lanes[0].st));
        for (int l = 1; l < nl; l++) Q3_HIP(hipStreamWaitEvent(lanes[(size_t)l].st, e0, 0));
        for (int it = 0; it < rounds; it++) for (int l = 0; l < nl; l++) Q3_HIP(hipGraphLaunch(lanes[(size_t)l].ge, lanes[(size_t)l].st));
        float worst = 0;
        for (int l = 0; l < nl; l++) Q3_HIP(hipEventRecord(e1[(size_t)l], lanes[(size_t)l].st));
        for (int l = 0; l < nl; l++) { Q3_HIP(hipEventSynchronize(e1[(size_t)l])); float ms = 0; hipEventElapsedTime(&ms, e0, e1[(size_t)l]); worst = std::max(worst, ms); }
        const double launches = (double)rounds * reps * 4;
        printf("concurrency: %d lane(s) x %d tokens: %.1f us per 4-GEMM layer step per lane (%.2f us per launch per lane); aggregate %.2f us per launch\n", nl, ntok,
               worst * 1e3 / (rounds * reps), worst * 1e3 / launches, worst * 1e3 / (launches * nl));
    }
}

static void print_stamps(const std::vector<unsigned long long>& h, const char* const* names) {
    std::vector<size_t> wg;
    for (size_t b = 0; b < 4096; b++) if (h[b * 8]) wg.push_back(b);
    if (wg.empty()) { printf("    (no stamps)\n"); return; }
    unsigned long long t0 = ~0ull, t6 = 0;
    for (size_t b : wg) { t0 = std::min(t0, h[b * 8]); t6 = std::max(t6, h[b * 8 + 6]); }
    printf("    stamps: %zu workgroups, first entry -> last exit %.2f us\n", wg.size(), (double)(t6 - t0) / 100.0);
    for (int s = 0; s < 7; s++) {
        std::vector<double> v;
        for (size_t b : wg) if (h[b * 8 + s]) v.push_back((double)(h[b * 8 + s] - t0) / 100.0);
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        printf("      %-18s  min %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us after the first entry\n", names[s], v.front(), v[v.size() / 2], v[v.size() * 9 / 10], v.back());
    }
    std::vector<double> life;
    for (size_t b : wg) life.push_back((double)(h[b * 8 + 6] - h[b * 8]) / 100.0);
    std::sort(life.begin(), life.end());
    printf("      workgroup lifetime  min %6.2f  p50 %6.2f  max %6.2f us\n", life.front(), life[life.size() / 2], life.back());
}

// the code predictor's single-wave attention (k_attention_short) and the workgroup-per-token norm at 64 sequences: in-graph cost + stamps
static void small_kernels_test(int ntok) {
    const int n_head = 16, n_kv = 8, dq = 2048, dkv = 1024, d = 1024, stride = dq + 2 * dkv, n_layer = 5;
    DevBuf<float> qkv((size_t)ntok * stride), qn(128), kn(128), rc(512 * 64), rs(512 * 64), h((size_t)ntok * d), parts((size_t)2 * ntok * d), g(d), hout((size_t)ntok * d);
    std::vector<float> hv((size_t)ntok * stride); for (size_t i = 0; i < hv.size(); i++) hv[i] = 0.01f * (float)((i * 2654435761u >> 20) & 255) - 1.0f;
    qkv.upload(hv.data(), hv.size());
    std::vector<float> ones(2048, 1.0f); qn.upload(ones.data(), 128); kn.upload(ones.data(), 128); g.upload(ones.data(), d);
    std::vector<float> tab(512 * 64, 0.5f); rc.upload(tab.data(), tab.size()); rs.upload(tab.data(), tab.size());
    h.upload(hv.data(), h.n); parts.upload(hv.data(), parts.n);
    DevBuf<int32_t> sec(4); int32_t z4[4] = {0, 0, 0, 0}; sec.upload(z4, 4);
    KvPool pool(n_layer, n_kv, ntok, ntok, 1);
    KvCache kv = pool.view(); kv.page_table = nullptr;
    DevBuf<int8_t> aq((size_t)ntok * dq); DevBuf<uint16_t> ad((size_t)ntok * dq / 32);
    DevBuf<int8_t> xq((size_t)ntok * d); DevBuf<uint16_t> xd((size_t)ntok * d / 32);
    DevBuf<unsigned long long> stamps((size_t)4096 * 8);
    const char* an[7] = {"entry", "prologue done", "QK done", "softmax done", "", "", "exit"};
    for (int pos : {3, 16}) {
        TokMeta tm{nullptr, nullptr, nullptr}; tm.uniform_pos = pos;
        auto att = [&](int i) { launch_attention_short(g_st, qkv.p, stride, n_head, n_kv, qn.p, kn.p, 1e-6f, rc.p, rs.p, 512, sec.p, tm, kv, i % n_layer, aq.p, ad.p, ntok, nullptr); };
        printf("k_attention_short %d tokens, position %d: in-graph %.2f us/launch\n", ntok, pos, time_graph(att, 40));
        Q3_HIP(hipMemsetAsync(stamps.p, 0, stamps.n * 8, g_st));
        set_stamp_buffer_fused(g_st, stamps.p);
        for (int i = 0; i < 3; i++) att(i);
        Q3_HIP(hipStreamSynchronize(g_st));
        std::vector<unsigned long long> hs(stamps.n); stamps.download(hs.data(), hs.size());
        set_stamp_buffer_fused(g_st, nullptr);
        print_stamps(hs, an);
    }
    NormPro a{}; a.h_in = h.p; a.h_stride = d; a.parts = parts.p; a.nparts = 2; a.parts_stride = d; a.parts_slab = (size_t)ntok * d; a.h_out = hout.p; a.g = g.p; a.eps = 1e-6f;
    auto nrm = [&](int) { launch_rmsnorm_quant_wg(g_st, a, d, xq.p, xd.p, ntok); };
    printf("k_rmsnorm_quant_wg %d tokens d=%d 2 slabs: in-graph %.2f us/launch\n", ntok, d, time_graph(nrm, 40));
    // the same norm behind the GEMM that produces its slabs (the predictor's down-projection, K = 3072 -> 2 slabs), as in the frame graph: its inputs are then
    // lines other XCDs have just written
    {
        const int n = 1024, k = 3072;
        std::vector<uint8_t> raw((size_t)n * (k / 32) * 34);
        for (size_t i = 0; i < raw.size(); i++) raw[i] = (uint8_t)(i * 2654435761u >> 13);
        for (size_t b = 0; b < (size_t)n * (k / 32); b++) { raw[b * 34] = 0x00; raw[b * 34 + 1] = 0x1C; }
        DevBuf<uint8_t> storage; Q8Mat m = q8mat_from_host(raw.data(), n, k, storage);
        DevBuf<int8_t> fq((size_t)ntok * k); DevBuf<uint16_t> fd((size_t)ntok * k / 32);
        Q3_HIP(hipMemset(fq.p, 1, fq.n)); Q3_HIP(hipMemset(fd.p, 0x20, fd.n * 2));
        auto gemm_only = [&](int) { launch_gemv_q8(g_st, m, 0, n, fq.p, fd.p, parts.p, d, ntok); };
        auto pair = [&](int) { launch_gemv_q8(g_st, m, 0, n, fq.p, fd.p, parts.p, d, ntok); launch_rmsnorm_quant_wg(g_st, a, d, xq.p, xd.p, ntok); };
        const float tg = time_graph(gemm_only, 40), tp = time_graph(pair, 40);
        printf("down GEMM alone %.2f us; GEMM + norm pair %.2f us -> norm behind its producer costs %.2f us in-graph\n", tg, tp, tp - tg);
        Q3_HIP(hipMemsetAsync(stamps.p, 0, stamps.n * 8, g_st));
        set_stamp_buffer_fused(g_st, stamps.p);
        for (int i = 0; i < 3; i++) pair(i);
        Q3_HIP(hipStreamSynchronize(g_st));
        std::vector<unsigned long long> hs(stamps.n); stamps.download(hs.data(), hs.size());
        set_stamp_buffer_fused(g_st, nullptr);
        const char* nn[7] = {"entry", "norm+quant done", "", "", "", "", "exit"};
        print_stamps(hs, nn);
    }
}

int main(int argc, char** argv) {
    Q3_HIP(hipStreamCreateWithFlags(&g_st, hipStreamNonBlocking));
    init_kernel_attributes();
    if (argc > 1 && !strcmp(argv[1], "small")) { small_kernels_test(64); return 0; }
    if (argc > 1 && !strcmp(argv[1], "conc")) { concurrency_test(64); concurrency_test(32); concurrency_test(16); return 0; }
    const Shape shapes[] = {
        {"talker gate/up 12288x2048", 12288, 2048, 1, 28}, {"talker down 2048x6144", 2048, 6144, 0, 28}, {"talker qkv 4096x2048", 4096, 2048, 0, 28},
        {"talker o 2048x2048", 2048, 2048, 0, 28}, {"pred gate/up 6144x1024", 6144, 1024, 1, 5}, {"pred qkv 4096x1024", 4096, 1024, 0, 5},
        {"pred o 1024x2048", 1024, 2048, 0, 5}, {"pred down 1024x3072", 1024, 3072, 0, 5}, {"pred head 2048x1024", 2048, 1024, 0, 5}};
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    DevBuf<unsigned long long> stamps((size_t)4096 * 8);
    for (const Shape& sh : shapes) {
        if (only >= 0 && &sh - shapes != only) continue;
        const int n = sh.n, k = sh.k;
        std::vector<uint8_t> raw((size_t)n * (k / 32) * 34);
        for (size_t i = 0; i < raw.size(); i++) raw[i] = (uint8_t)(i * 2654435761u >> 13);
        for (size_t b = 0; b < (size_t)n * (k / 32); b++) { raw[b * 34] = 0x00; raw[b * 34 + 1] = 0x1C; } // d = 2^-8
        std::vector<DevBuf<uint8_t>> storage((size_t)sh.copies);
        std::vector<Q8Mat> mats;
        for (int c = 0; c < sh.copies; c++) mats.push_back(q8mat_from_host(raw.data(), n, k, storage[(size_t)c]));
        for (int ntok : {64, 32, 128}) {
            DevBuf<int8_t> xq((size_t)ntok * k); DevBuf<uint16_t> xd((size_t)ntok * k / 32);
            std::vector<int8_t> hx(xq.n); for (size_t i = 0; i < hx.size(); i++) hx[i] = (int8_t)((i * 40503u >> 7) & 0xFF);
            std::vector<uint16_t> hd(xd.n, 0x2000);
            xq.upload(hx.data(), hx.size()); xd.upload(hd.data(), hd.size());
            const int nsseg = ((k >> 8) + 7) / 8, ff = n / 2;
            DevBuf<float> out((size_t)nsseg * ntok * n); DevBuf<int8_t> aq((size_t)ntok * n); DevBuf<uint16_t> ad((size_t)ntok * n / 32);
            const double mb = (double)n * k * 1.0625 / 1e6;
            auto launch = [&](int i) {
                const Q8Mat& m = mats[(size_t)(i % sh.copies)];
                if (sh.gu) launch_gateup_mfma(g_st, m, ff, xq.p, xd.p, aq.p, ad.p, ntok);
                else launch_gemv_q8(g_st, m, 0, n, xq.p, xd.p, out.p, n, ntok);
            };
            const float us = time_graph(launch, 2 * sh.copies);
            printf("%-28s %3d tok  %6.2f MB  in-graph %6.2f us/launch  (%.2f TB/s; HBM floor %.2f us)\n", sh.name, ntok, mb, us, mb / us, mb / 6.3);
#ifdef Q3_STAMPS
            if (ntok != 64) continue;
            // one stamped launch (after the timing runs: caches as in a steady stream of launches)
            Q3_HIP(hipMemsetAsync(stamps.p, 0, stamps.n * 8, g_st));
            set_stamp_buffer(g_st, stamps.p);
            for (int i = 0; i < 3; i++) launch(i + 1);
            Q3_HIP(hipStreamSynchronize(g_st));
            std::vector<unsigned long long> h(stamps.n);
            stamps.download(h.data(), h.size());
            set_stamp_buffer(g_st, nullptr);
            // workgroups that ran: stamp 0 != 0 (the last of the 3 launches overwrote the earlier ones)
            std::vector<size_t> wg;
            for (size_t b = 0; b < 4096; b++) if (h[b * 8]) wg.push_back(b);
            if (wg.empty()) continue;
            unsigned long long t0 = ~0ull, t6 = 0;
            for (size_t b : wg) { t0 = std::min(t0, h[b * 8]); t6 = std::max(t6, h[b * 8 + 6]); }
            printf("    stamps: %zu workgroups, first entry -> last exit %.2f us\n", wg.size(), (double)(t6 - t0) / 100.0);
            const char* names[7] = {"entry", "chain 0 done", "barrier A (0)", "barrier B (0)", "up chain done", "up barrier A", "exit"};
            for (int s = 0; s < 7; s++) {
                std::vector<double> v;
                for (size_t b : wg) if (h[b * 8 + s]) v.push_back((double)(h[b * 8 + s] - t0) / 100.0);
                if (v.empty()) continue;
                std::sort(v.begin(), v.end());
                printf("      %-18s  min %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us after the first entry\n", names[s], v.front(), v[v.size() / 2], v[v.size() * 9 / 10], v.back());
            }
            std::vector<double> life;
            for (size_t b : wg) life.push_back((double)(h[b * 8 + 6] - h[b * 8]) / 100.0);
            std::sort(life.begin(), life.end());
            printf("      workgroup lifetime  min %6.2f  p50 %6.2f  max %6.2f us\n", life.front(), life[life.size() / 2], life.back());
            std::vector<double> clk;
            for (size_t b : wg) if (h[b * 8 + 6] > h[b * 8]) clk.push_back((double)h[b * 8 + 7] / ((double)(h[b * 8 + 6] - h[b * 8]) * 10.0)); // cycles per ns = GHz
            std::sort(clk.begin(), clk.end());
            if (!clk.empty()) printf("      shader clock (s_memtime / s_memrealtime over the workgroup's life)  p50 %.2f GHz  min %.2f  max %.2f\n", clk[clk.size() / 2], clk.front(), clk.back());
#endif
        }
    }
    return 0;
}
