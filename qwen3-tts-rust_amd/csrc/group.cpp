// group.cpp -- multi-GPU serving behind the C ABI (include/q3tts.h "multi-GPU").
//
// Utterances are independent (own prompt, KV pages, sampler state, decoder state; /root/reference/src/tts/engine.rs:445-656 touches no
// cross-request state), so GPUs are REQUEST-SHARDED: every device holds the full weights, requests go round-robin, and the data path has
// no collective.  The one exchange step is a voice registration: the VoiceFile payload (speaker embedding 2048 f32 = 8 KB, plus reference
// codes / reference-text ids for clone voices; utils/voice_file.rs:5-22) is sent from the device that owns it to all others --
//   * q3tts_group_*: ONE process drives N devices (one engine, one scheduler thread, one decoder thread per device).  Registration uploads
//     the payload to the first device and broadcasts it device-to-device with ncclBroadcast on communicators from ncclCommInitAll (RCCL,
//     xGMI between the GPUs of a node); when librccl cannot be loaded, or the group has one device, or lists a device twice, the copies
//     are hipMemcpyPeerAsync.  Every engine registers the voice from ITS OWN device's copy of the bytes.
//   * q3tts_comm_*: ONE process per GPU (the layout bench.py and the driver's --gpus N runs use): rank 0 creates a 128-byte id, every rank
//     builds its communicator from it (ncclCommInitRank), and q3tts_comm_voice_register is the collective call.
// RCCL is bound at run time (dlopen "librccl.so"), so libq3tts.so has no link-time dependency on it and single-GPU hosts need not have it.
#include "../../include/q3tts.h"
#include "engine.h"
#include <dlfcn.h>
#include <atomic>

using namespace q3;

#define Q3_API_BEGIN try {
#define Q3_API_END(failval) } catch (const std::exception& ex) { set_last_error(ex.what()); return failval; } catch (...) { set_last_error("unknown error"); return failval; }

namespace {

// ---- the few RCCL entry points this file uses (public NCCL API; signatures as in rccl.h) ----
struct NcclId { char internal[128]; };
typedef void* ncclComm_t;
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(NcclId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, NcclId, int) = nullptr;
    int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok() const { return h != nullptr; }
};
Rccl& rccl() {
    static Rccl r = [] {
        Rccl x;
        if (const char* e = std::getenv("Q3_NO_RCCL")) if (e[0] == '1') return x; // A/B switch: force the peer-copy path
        void* h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return x;
        auto sym = [&](const char* n) { return dlsym(h, n); };
        x.GetUniqueId = (int (*)(NcclId*))sym("ncclGetUniqueId");
        x.CommInitRank = (int (*)(ncclComm_t*, int, NcclId, int))sym("ncclCommInitRank");
        x.CommInitAll = (int (*)(ncclComm_t*, int, const int*))sym("ncclCommInitAll");
        x.CommDestroy = (int (*)(ncclComm_t))sym("ncclCommDestroy");
        x.Broadcast = (int (*)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t))sym("ncclBroadcast");
        x.GroupStart = (int (*)())sym("ncclGroupStart");
        x.GroupEnd = (int (*)())sym("ncclGroupEnd");
        x.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
        if (x.GetUniqueId && x.CommInitRank && x.CommInitAll && x.CommDestroy && x.Broadcast && x.GroupStart && x.GroupEnd) x.h = h;
        return x;
    }();
    return r;
}
void nccl_check(int rc, const char* what) {
    if (rc != 0) throw Error(std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(rc) : "RCCL error ") + " (" + std::to_string(rc) + ")");
}
constexpr int kNcclChar = 0; // ncclInt8 / ncclChar

// ---- voice payload: header {n_emb, n_codes, n_text, magic} i64 x 4, then f32 embedding, i32 codes, i32 text ids ----
constexpr int64_t kVoiceMagic = 0x5133564F49434531ll; // "Q3VOICE1"
constexpr size_t kHdrBytes = 32;
std::vector<uint8_t> pack_voice(const float* spk, const int32_t* codes, int32_t n_codes, const int32_t* text, int32_t n_text) {
    Q3_CHECK(spk, "speaker embedding missing");
    Q3_CHECK(n_codes >= 0 && n_text >= 0 && (n_codes == 0 || codes) && (n_text == 0 || text), "bad voice arrays");
    const int64_t hdr[4] = {Q3_EMBD, n_codes, n_text, kVoiceMagic};
    std::vector<uint8_t> b(kHdrBytes + (size_t)Q3_EMBD * 4 + (size_t)n_codes * 4 + (size_t)n_text * 4);
    std::memcpy(b.data(), hdr, kHdrBytes);
    std::memcpy(b.data() + kHdrBytes, spk, (size_t)Q3_EMBD * 4);
    if (n_codes) std::memcpy(b.data() + kHdrBytes + (size_t)Q3_EMBD * 4, codes, (size_t)n_codes * 4);
    if (n_text) std::memcpy(b.data() + kHdrBytes + (size_t)Q3_EMBD * 4 + (size_t)n_codes * 4, text, (size_t)n_text * 4);
    return b;
}
size_t payload_bytes(const int64_t hdr[4]) {
    Q3_CHECK(hdr[3] == kVoiceMagic && hdr[0] == Q3_EMBD && hdr[1] >= 0 && hdr[1] <= (1 << 24) && hdr[2] >= 0 && hdr[2] <= (1 << 20), "corrupt voice header");
    return (size_t)hdr[0] * 4 + (size_t)hdr[1] * 4 + (size_t)hdr[2] * 4;
}
Voice unpack_voice(const uint8_t* b) {
    int64_t hdr[4];
    std::memcpy(hdr, b, kHdrBytes);
    (void)payload_bytes(hdr);
    Voice v;
    const float* e = reinterpret_cast<const float*>(b + kHdrBytes);
    v.spk_emb.assign(e, e + hdr[0]);
    const int32_t* c = reinterpret_cast<const int32_t*>(b + kHdrBytes + (size_t)hdr[0] * 4);
    v.ref_codes.assign(c, c + hdr[1]);
    v.ref_text_ids.assign(c + hdr[1], c + hdr[1] + hdr[2]);
    return v;
}

struct DevSlot { // per-device staging for the broadcast
    int dev = 0; hipStream_t st = nullptr; uint8_t* d_buf = nullptr; uint8_t* h_buf = nullptr; size_t cap = 0;
    void ensure(size_t n) {
        if (n <= cap) return;
        Q3_HIP(hipSetDevice(dev));
        if (d_buf) (void)hipFree(d_buf);
        if (h_buf) (void)hipHostFree(h_buf);
        cap = (n + 4095) & ~(size_t)4095;
        Q3_HIP(hipMalloc((void**)&d_buf, cap));
        Q3_HIP(hipHostMalloc((void**)&h_buf, cap));
    }
    void release() {
        (void)hipSetDevice(dev);
        if (d_buf) (void)hipFree(d_buf);
        if (h_buf) (void)hipHostFree(h_buf);
        if (st) (void)hipStreamDestroy(st);
        d_buf = h_buf = nullptr; st = nullptr; cap = 0;
    }
};

} // namespace

struct q3tts_engine; // defined in capi.cpp
extern "C" {
int q3tts_engine_create(const q3tts_engine_params* p, q3tts_engine** out);
void q3tts_engine_destroy(q3tts_engine* e);
}

struct q3tts_group {
    std::vector<q3tts_engine*> eng;
    std::vector<DevSlot> slot;
    std::vector<ncclComm_t> comm; // empty => peer copies
    std::atomic<uint64_t> rr{0};
    std::mutex mu; // serialises registrations
    bool started = false;
    bool distinct = true;                 // every listed device is a different one
    int64_t n_reg_rccl = 0, n_reg_peer = 0; // registrations by the path their bytes took (q3tts_group_info)
    bool path_logged = false;
    ~q3tts_group() {
        for (auto c : comm) if (c) (void)rccl().CommDestroy(c);
        for (auto& s : slot) s.release();
        for (auto* e : eng) if (e) q3tts_engine_destroy(e);
    }
};
struct q3tts_comm {
    ncclComm_t comm = nullptr; int rank = 0, world = 1; DevSlot slot;
    ~q3tts_comm() { if (comm) (void)rccl().CommDestroy(comm); slot.release(); }
};

static const int kEngineBits = 8; // group request id = (engine request id << 8) | engine index
static inline int eng_of(int64_t id) { return (int)(id & ((1 << kEngineBits) - 1)); }
static inline int64_t rid_of(int64_t id) { return id >> kEngineBits; }

extern "C" {

int q3tts_group_create(const q3tts_engine_params* p, const int32_t* device_ids, int32_t n_dev, q3tts_group** out) {
    Q3_API_BEGIN
    Q3_CHECK(p && out && device_ids && n_dev >= 1 && n_dev <= (1 << kEngineBits), "bad arguments");
    int n_vis = 0;
    if (hipGetDeviceCount(&n_vis) != hipSuccess || n_vis <= 0) throw Error("no HIP device available: the HIP path is the only compute path (no CPU fallback)");
    bool distinct = true;
    for (int i = 0; i < n_dev; i++) {
        Q3_CHECK(device_ids[i] >= 0 && device_ids[i] < n_vis, "device id out of range");
        for (int j = 0; j < i; j++) if (device_ids[j] == device_ids[i]) distinct = false;
    }
    std::unique_ptr<q3tts_group> g(new q3tts_group());
    g->distinct = distinct;
    g->eng.assign(n_dev, nullptr);
    g->slot.resize(n_dev);
    for (int i = 0; i < n_dev; i++) {
        q3tts_engine_params pi = *p;
        pi.device = device_ids[i];
        if (q3tts_engine_create(&pi, &g->eng[i]) != Q3TTS_OK) throw Error(std::string("engine on device ") + std::to_string(device_ids[i]) + ": " + last_error());
        g->slot[i].dev = device_ids[i];
        Q3_HIP(hipSetDevice(device_ids[i]));
        Q3_HIP(hipStreamCreateWithFlags(&g->slot[i].st, hipStreamNonBlocking));
        for (int j = 0; j < i; j++) { // peer access for the copy path (and for RCCL's direct xGMI transport)
            int can = 0;
            if (device_ids[j] != device_ids[i] && hipDeviceCanAccessPeer(&can, device_ids[i], device_ids[j]) == hipSuccess && can) {
                (void)hipDeviceEnablePeerAccess(device_ids[j], 0); (void)hipGetLastError();
                (void)hipSetDevice(device_ids[j]); (void)hipDeviceEnablePeerAccess(device_ids[i], 0); (void)hipGetLastError();
                (void)hipSetDevice(device_ids[i]);
            }
        }
    }
    if (n_dev > 1 && distinct && rccl().ok()) {
        g->comm.assign(n_dev, nullptr);
        std::vector<int> devs(device_ids, device_ids + n_dev);
        nccl_check(rccl().CommInitAll(g->comm.data(), n_dev, devs.data()), "ncclCommInitAll");
    }
    *out = g.release();
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}
void q3tts_group_destroy(q3tts_group* g) { delete g; }
int32_t q3tts_group_size(q3tts_group* g) { return g ? (int32_t)g->eng.size() : 0; }
q3tts_engine* q3tts_group_engine(q3tts_group* g, int32_t i) { return (g && i >= 0 && i < (int)g->eng.size()) ? g->eng[i] : nullptr; }
int32_t q3tts_group_uses_rccl(q3tts_group* g) { return g && !g->comm.empty() ? 1 : 0; }
int q3tts_group_info(q3tts_group* g, q3tts_group_info_t* out) {
    Q3_API_BEGIN
    Q3_CHECK(g && out, "null argument");
    std::lock_guard<std::mutex> lk(g->mu);
    out->n_devices = (int32_t)g->eng.size();
    out->rccl_loaded = rccl().ok() ? 1 : 0;
    out->rccl_ranks = (int32_t)g->comm.size();
    out->distinct_devices = g->distinct ? 1 : 0;
    out->registrations_rccl = g->n_reg_rccl;
    out->registrations_peer_copy = g->n_reg_peer;
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}

int q3tts_group_voice_register(q3tts_group* g, const float* spk, const int32_t* ref_codes, int32_t n_ref_codes, const int32_t* ref_text,
                               int32_t n_ref_text, int32_t* voice_id) {
    Q3_API_BEGIN
    Q3_CHECK(g && spk && voice_id, "null argument");
    std::lock_guard<std::mutex> lk(g->mu);
    const std::vector<uint8_t> pay = pack_voice(spk, ref_codes, n_ref_codes, ref_text, n_ref_text);
    const int n = (int)g->eng.size();
    { std::lock_guard<std::mutex> cap(capture_mutex()); for (auto& s : g->slot) s.ensure(pay.size()); } // allocations never overlap a running engine's capture
    // the owner (first device) uploads once ...
    DevSlot& s0 = g->slot[0];
    Q3_HIP(hipSetDevice(s0.dev));
    std::memcpy(s0.h_buf, pay.data(), pay.size());
    Q3_HIP(hipMemcpyAsync(s0.d_buf, s0.h_buf, pay.size(), hipMemcpyHostToDevice, s0.st));
    Q3_HIP(hipStreamSynchronize(s0.st));
    // ... and the bytes travel device to device
    if (!g->path_logged) { // once per group: which path registrations take (a multi-GPU run must be able to show "RCCL, N ranks", not assume it)
        g->path_logged = true;
        std::fprintf(stderr, "[q3tts group] voice registrations: %s over %d device(s)%s\n", !g->comm.empty() ? "ncclBroadcast (RCCL communicators from ncclCommInitAll)" :
                     n > 1 ? "hipMemcpyPeerAsync (no RCCL communicators)" : "single device, no exchange", n,
                     g->comm.empty() && n > 1 ? (!g->distinct ? " -- a device is listed twice" : !rccl().ok() ? " -- librccl not loaded" : "") : "");
    }
    if (!g->comm.empty()) g->n_reg_rccl++; else if (n > 1) g->n_reg_peer++;
    if (!g->comm.empty()) {
        nccl_check(rccl().GroupStart(), "ncclGroupStart");
        for (int i = 0; i < n; i++) {
            Q3_HIP(hipSetDevice(g->slot[i].dev));
            nccl_check(rccl().Broadcast(g->slot[i].d_buf, g->slot[i].d_buf, pay.size(), kNcclChar, 0, g->comm[i], g->slot[i].st), "ncclBroadcast");
        }
        nccl_check(rccl().GroupEnd(), "ncclGroupEnd");
    } else {
        for (int i = 1; i < n; i++) {
            Q3_HIP(hipSetDevice(g->slot[i].dev));
            Q3_HIP(hipMemcpyPeerAsync(g->slot[i].d_buf, g->slot[i].dev, s0.d_buf, s0.dev, pay.size(), g->slot[i].st));
        }
    }
    int vid = -1;
    for (int i = 0; i < n; i++) { // every engine registers from its own device's copy
        DevSlot& s = g->slot[i];
        Q3_HIP(hipSetDevice(s.dev));
        if (i > 0) std::memset(s.h_buf, 0, pay.size());
        Q3_HIP(hipMemcpyAsync(s.h_buf, s.d_buf, pay.size(), hipMemcpyDeviceToHost, s.st));
        Q3_HIP(hipStreamSynchronize(s.st));
        const Voice v = unpack_voice(s.h_buf);
        int32_t id = -1;
        if (q3tts_voice_register(g->eng[i], v.spk_emb.data(), v.ref_codes.empty() ? nullptr : v.ref_codes.data(), (int32_t)v.ref_codes.size(),
                                 v.ref_text_ids.empty() ? nullptr : v.ref_text_ids.data(), (int32_t)v.ref_text_ids.size(), &id) != Q3TTS_OK)
            throw Error(std::string("voice registration on engine ") + std::to_string(i) + ": " + last_error());
        if (i == 0) vid = id;
        Q3_CHECK(id == vid, "engines of a group must hand out the same voice ids (register voices through the group only)");
    }
    *voice_id = vid;
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}

static int pick_engine(q3tts_group* g) { return (int)(g->rr.fetch_add(1) % g->eng.size()); } // request i -> device i % n

int q3tts_group_submit(q3tts_group* g, const q3tts_request* r, int32_t want_pcm, int64_t* req_id) {
    Q3_API_BEGIN
    Q3_CHECK(g && r && req_id, "null argument");
    const int i = pick_engine(g);
    int64_t id = 0;
    if (q3tts_submit(g->eng[i], r, want_pcm, &id) != Q3TTS_OK) return Q3TTS_ERR;
    *req_id = (id << kEngineBits) | i;
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}
int q3tts_group_submit_text(q3tts_group* g, int32_t voice_id, const int32_t* text_ids, int32_t n_text, int32_t lang_id, const int32_t* instr_ids,
                            int32_t n_instr, const q3tts_sampler_config* sampler, int32_t max_steps, int32_t mask_eos, int32_t want_pcm, int64_t* req_id) {
    Q3_API_BEGIN
    Q3_CHECK(g && req_id, "null argument");
    const int i = pick_engine(g);
    int64_t id = 0;
    if (q3tts_submit_text(g->eng[i], voice_id, text_ids, n_text, lang_id, instr_ids, n_instr, sampler, max_steps, mask_eos, want_pcm, &id) != Q3TTS_OK) return Q3TTS_ERR;
    *req_id = (id << kEngineBits) | i;
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}
#define Q3_GROUP_ENGINE(g, id) ((g) && eng_of(id) < (int)(g)->eng.size() ? (g)->eng[eng_of(id)] : nullptr)
int32_t q3tts_group_device_of(q3tts_group* g, int64_t req_id) { return (g && eng_of(req_id) < (int)g->slot.size()) ? g->slot[eng_of(req_id)].dev : -1; }
int q3tts_group_poll(q3tts_group* g, int64_t req_id, q3tts_req_status* out) {
    q3tts_engine* e = Q3_GROUP_ENGINE(g, req_id);
    if (!e) { set_last_error("unknown group request id"); return Q3TTS_ERR; }
    return q3tts_poll(e, rid_of(req_id), out);
}
int q3tts_group_fetch(q3tts_group* g, int64_t req_id, int32_t* codes_out, int32_t frame_off, int32_t max_frames, float* pcm_out, int64_t pcm_off,
                      int64_t pcm_cap, int32_t* got_frames, int64_t* got_pcm) {
    q3tts_engine* e = Q3_GROUP_ENGINE(g, req_id);
    if (!e) { set_last_error("unknown group request id"); return Q3TTS_ERR; }
    return q3tts_fetch(e, rid_of(req_id), codes_out, frame_off, max_frames, pcm_out, pcm_off, pcm_cap, got_frames, got_pcm);
}
int q3tts_group_wait(q3tts_group* g, int64_t req_id, double timeout_ms) {
    q3tts_engine* e = Q3_GROUP_ENGINE(g, req_id);
    if (!e) { set_last_error("unknown group request id"); return Q3TTS_ERR; }
    return q3tts_wait(e, rid_of(req_id), timeout_ms);
}
int q3tts_group_release(q3tts_group* g, int64_t req_id) {
    q3tts_engine* e = Q3_GROUP_ENGINE(g, req_id);
    if (!e) { set_last_error("unknown group request id"); return Q3TTS_ERR; }
    return q3tts_release(e, rid_of(req_id));
}
int q3tts_group_start(q3tts_group* g) { // one scheduler thread per device
    if (!g) return Q3TTS_ERR;
    for (auto* e : g->eng) if (q3tts_sched_start(e) != Q3TTS_OK) return Q3TTS_ERR;
    g->started = true;
    return Q3TTS_OK;
}
int q3tts_group_stop(q3tts_group* g) {
    if (!g) return Q3TTS_ERR;
    int rc = Q3TTS_OK;
    for (auto* e : g->eng) if (q3tts_sched_stop(e) != Q3TTS_OK) rc = Q3TTS_ERR;
    g->started = false;
    return rc;
}

// ---------------- one process per GPU ----------------
int q3tts_comm_available(void) { return rccl().ok() ? 1 : 0; }
int q3tts_comm_unique_id(uint8_t out[128]) {
    Q3_API_BEGIN
    Q3_CHECK(out, "null argument");
    if (!rccl().ok()) throw Error("librccl.so could not be loaded");
    NcclId id;
    nccl_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
    std::memcpy(out, id.internal, 128);
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}
int q3tts_comm_create(const uint8_t id[128], int32_t rank, int32_t world, int32_t device, q3tts_comm** out) {
    Q3_API_BEGIN
    Q3_CHECK(id && out && world >= 1 && rank >= 0 && rank < world, "bad arguments");
    if (!rccl().ok()) throw Error("librccl.so could not be loaded");
    std::unique_ptr<q3tts_comm> c(new q3tts_comm());
    c->rank = rank; c->world = world; c->slot.dev = device;
    Q3_HIP(hipSetDevice(device));
    Q3_HIP(hipStreamCreateWithFlags(&c->slot.st, hipStreamNonBlocking));
    NcclId nid;
    std::memcpy(nid.internal, id, 128);
    nccl_check(rccl().CommInitRank(&c->comm, world, nid, rank), "ncclCommInitRank");
    *out = c.release();
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}
void q3tts_comm_destroy(q3tts_comm* c) { delete c; }
int q3tts_comm_voice_register(q3tts_comm* c, q3tts_engine* e, int32_t root, const float* spk, const int32_t* ref_codes, int32_t n_ref_codes,
                              const int32_t* ref_text, int32_t n_ref_text, int32_t* voice_id) {
    Q3_API_BEGIN
    Q3_CHECK(c && e && voice_id && root >= 0 && root < c->world, "bad arguments");
    DevSlot& s = c->slot;
    Q3_HIP(hipSetDevice(s.dev));
    std::vector<uint8_t> pay;
    if (c->rank == root) pay = pack_voice(spk, ref_codes, n_ref_codes, ref_text, n_ref_text);
    // 1) the 32-byte header (sizes are known at the root only), 2) the payload
    { std::lock_guard<std::mutex> cap(capture_mutex()); s.ensure(kHdrBytes); }
    if (c->rank == root) { std::memcpy(s.h_buf, pay.data(), kHdrBytes); Q3_HIP(hipMemcpyAsync(s.d_buf, s.h_buf, kHdrBytes, hipMemcpyHostToDevice, s.st)); }
    nccl_check(rccl().Broadcast(s.d_buf, s.d_buf, kHdrBytes, kNcclChar, root, c->comm, s.st), "ncclBroadcast(header)");
    Q3_HIP(hipMemcpyAsync(s.h_buf, s.d_buf, kHdrBytes, hipMemcpyDeviceToHost, s.st));
    Q3_HIP(hipStreamSynchronize(s.st));
    int64_t hdr[4];
    std::memcpy(hdr, s.h_buf, kHdrBytes);
    const size_t total = kHdrBytes + payload_bytes(hdr);
    { std::lock_guard<std::mutex> cap(capture_mutex()); s.ensure(total); }
    if (c->rank == root) { std::memcpy(s.h_buf, pay.data(), total); Q3_HIP(hipMemcpyAsync(s.d_buf, s.h_buf, total, hipMemcpyHostToDevice, s.st)); }
    nccl_check(rccl().Broadcast(s.d_buf, s.d_buf, total, kNcclChar, root, c->comm, s.st), "ncclBroadcast(voice)");
    if (c->rank != root) std::memset(s.h_buf, 0, total);
    Q3_HIP(hipMemcpyAsync(s.h_buf, s.d_buf, total, hipMemcpyDeviceToHost, s.st));
    Q3_HIP(hipStreamSynchronize(s.st));
    const Voice v = unpack_voice(s.h_buf);
    return q3tts_voice_register(e, v.spk_emb.data(), v.ref_codes.empty() ? nullptr : v.ref_codes.data(), (int32_t)v.ref_codes.size(),
                                v.ref_text_ids.empty() ? nullptr : v.ref_text_ids.data(), (int32_t)v.ref_text_ids.size(), voice_id);
    Q3_API_END(Q3TTS_ERR)
}

} // extern "C"
