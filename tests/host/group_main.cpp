// C harness for the multi-GPU entry points of include/q3tts.h (q3tts_group_*, q3tts_comm_*): what a Rust host would bind.
//   group_main <model_dir> <n_engines>   (n_engines = 0: one engine per visible device; with a single GPU and n_engines = 2 the device is
//   listed twice, which exercises two engines + the peer-copy broadcast path on one card)
// Registers a CLONE voice through the group (payload uploaded to engine 0's device, broadcast, registered everywhere), submits text
// requests round-robin, and checks that the same request gives the same codes whichever engine / device it was sharded to -- in particular
// that a voice registered on device 0 generates on device 1.  Prints the codes for the Python test to compare with a plain single engine.
#include "../../include/q3tts.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { if ((x) != Q3TTS_OK) { fprintf(stderr, "FAILED %s: %s\n", #x, q3tts_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: group_main <model_dir> <n_engines>\n"); return 2; }
    const int n_vis = q3tts_device_count();
    if (n_vis < 1) { fprintf(stderr, "no HIP device\n"); return 3; }
    int n = atoi(argv[2]);
    if (n <= 0) n = n_vis;
    std::vector<int32_t> devs(n);
    for (int i = 0; i < n; i++) devs[i] = i % n_vis;
    q3tts_engine_params p;
    q3tts_engine_params_default(&p);
    p.model_dir = argv[1]; p.quant = "q8_0"; p.max_batch = 2; p.max_steps = 16; p.load_codec = 1;
    q3tts_group* g = nullptr;
    CHECK(q3tts_group_create(&p, devs.data(), n, &g));
    if (q3tts_group_size(g) != n) { fprintf(stderr, "group size\n"); return 1; }
    printf("ENGINES %d DEVICES_VISIBLE %d RCCL %d COMM_AVAILABLE %d\n", n, n_vis, q3tts_group_uses_rccl(g), q3tts_comm_available());
    std::vector<float> spk(2048);
    for (int i = 0; i < 2048; i++) spk[i] = 0.01f * (float)((i * 37) % 101 - 50);
    std::vector<int32_t> ref_codes(3 * 16), ref_text = {7, 8, 9}, text = {100, 101, 102, 103, 104, 105, 106, 107};
    for (int i = 0; i < 48; i++) ref_codes[i] = (i * 37) % 2048;
    int32_t preset = -1, clone = -1;
    CHECK(q3tts_group_voice_register(g, spk.data(), nullptr, 0, nullptr, 0, &preset));
    CHECK(q3tts_group_voice_register(g, spk.data(), ref_codes.data(), (int32_t)ref_codes.size(), ref_text.data(), (int32_t)ref_text.size(), &clone));
    if (preset != 0 || clone != 1) { fprintf(stderr, "voice ids %d %d\n", preset, clone); return 1; }
    {   // the exchange step must be able to say what it ran on: distinct devices + librccl => "RCCL, n ranks", and both registrations counted there;
        // one device / a device listed twice => peer copies.  A multi-GPU box that silently fell back to copies fails HERE.
        q3tts_group_info_t gi;
        CHECK(q3tts_group_info(g, &gi));
        const bool distinct = n <= n_vis;
        const bool expect_rccl = n > 1 && distinct && gi.rccl_loaded;
        printf("GROUP_INFO devices %d rccl_loaded %d rccl_ranks %d distinct %d reg_rccl %lld reg_peer %lld\n", gi.n_devices, gi.rccl_loaded, gi.rccl_ranks,
               gi.distinct_devices, (long long)gi.registrations_rccl, (long long)gi.registrations_peer_copy);
        if (gi.n_devices != n || gi.distinct_devices != (distinct ? 1 : 0) || gi.rccl_ranks != (expect_rccl ? n : 0) ||
            gi.registrations_rccl != (expect_rccl ? 2 : 0) || gi.registrations_peer_copy != ((!expect_rccl && n > 1) ? 2 : 0)) {
            fprintf(stderr, "group info does not match the path the registrations must have taken\n");
            return 1;
        }
    }
    CHECK(q3tts_group_start(g));
    q3tts_sampler_config sc;
    q3tts_sampler_config_default(&sc);
    sc.temperature = 0.0f; sc.has_seed = 1; sc.seed = 42;
    const int n_req = 2 * n + 1;
    std::vector<int64_t> ids(n_req);
    for (int i = 0; i < n_req; i++) CHECK(q3tts_group_submit_text(g, clone, text.data(), (int32_t)text.size(), 2055, nullptr, 0, &sc, 6, 1, 1, &ids[i]));
    std::vector<std::vector<int32_t>> codes(n_req);
    std::vector<int64_t> npcm(n_req);
    for (int i = 0; i < n_req; i++) {
        if (q3tts_group_wait(g, ids[i], 120000.0) != 0) { fprintf(stderr, "wait: %s\n", q3tts_last_error()); return 1; }
        q3tts_req_status st;
        CHECK(q3tts_group_poll(g, ids[i], &st));
        if (st.state != Q3TTS_REQ_DONE || st.n_frames != 6) { fprintf(stderr, "request %d state %d frames %d: %s\n", i, st.state, st.n_frames, q3tts_last_error()); return 1; }
        codes[i].resize(6 * 16);
        std::vector<float> pcm((size_t)st.n_pcm);
        int32_t gf = 0; int64_t gp = 0;
        CHECK(q3tts_group_fetch(g, ids[i], codes[i].data(), 0, 6, pcm.data(), 0, st.n_pcm, &gf, &gp));
        npcm[i] = gp;
        if (gf != 6 || gp != st.n_pcm || q3tts_group_device_of(g, ids[i]) != devs[i % n]) { fprintf(stderr, "fetch / sharding of request %d\n", i); return 1; }
        CHECK(q3tts_group_release(g, ids[i]));
    }
    for (int i = 1; i < n_req; i++)
        if (codes[i] != codes[0] || npcm[i] != npcm[0]) { fprintf(stderr, "request %d (engine %d) differs from request 0\n", i, i % n); return 1; }
    CHECK(q3tts_group_stop(g));
    printf("CLONE");
    for (int v : codes[0]) printf(" %d", v);
    printf("\nPCM %lld\n", (long long)npcm[0]);
    // one process per GPU form with a single rank: id -> communicator -> collective registration on engine 0
    if (q3tts_comm_available()) {
        uint8_t id[128];
        q3tts_comm* c = nullptr;
        CHECK(q3tts_comm_unique_id(id));
        CHECK(q3tts_comm_create(id, 0, 1, devs[0], &c));
        int32_t vid = -1;
        CHECK(q3tts_comm_voice_register(c, q3tts_group_engine(g, 0), 0, spk.data(), ref_codes.data(), 48, ref_text.data(), 3, &vid));
        if (vid != 2) { fprintf(stderr, "comm voice id %d\n", vid); return 1; }
        q3tts_comm_destroy(c);
        printf("COMM ok\n");
    }
    q3tts_group_destroy(g);
    printf("OK\n");
    return 0;
}
