// capi_ops.cpp -- kernel-level C-ABI entry points for the parity tests (host buffers in, host buffers out).
#include "../../include/q3tts.h"
#include "transformer.h"
#include "host_logic.h"

using namespace q3;
#define Q3_API_BEGIN try {
#define Q3_API_END(failval) } catch (const std::exception& ex) { set_last_error(ex.what()); return failval; } catch (...) { set_last_error("unknown error"); return failval; }
static void require_gpu() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) throw Error("no HIP device available: the HIP path is the only compute path (no CPU fallback)");
}

extern "C" {

int q3tts_op_gemv_q8(const void* w, int32_t n, int32_t k, const int8_t* xq, const uint16_t* xd, int32_t ntok, float* y, int32_t lpr) {
    Q3_API_BEGIN
    require_gpu();
    DevBuf<uint8_t> storage;
    Q8Mat m = q8mat_from_host(w, n, k, storage);
    DevBuf<int8_t> dxq((size_t)ntok * k); dxq.upload(xq, dxq.n);
    DevBuf<uint16_t> dxd((size_t)ntok * (k / 32)); dxd.upload(xd, dxd.n);
    const int nsseg = ((k >> 8) + 7) / 8;
    DevBuf<float> parts((size_t)nsseg * ntok * n);
    launch_gemv_q8(0, m, 0, n, dxq.p, dxd.p, parts.p, n, ntok, lpr);
    Q3_HIP(hipDeviceSynchronize());
    std::vector<float> hp(parts.n);
    parts.download(hp.data(), hp.size());
    for (int t = 0; t < ntok; t++)
        for (int r = 0; r < n; r++) { // spec S3: super-segment sums added in order
            float v = hp[((size_t)0 * ntok + t) * n + r];
            for (int s = 1; s < nsseg; s++) v = v + hp[((size_t)s * ntok + t) * n + r];
            y[(size_t)t * n + r] = v;
        }
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}

/* fused gate/up GEMM + SwiGLU + int8 quantisation of the batched (>= 16 token) layer path; w = [2*ff][k] Q8_0 rows, gate rows first */
int q3tts_op_gateup_q8(const void* w, int32_t ff, int32_t k, const int8_t* xq, const uint16_t* xd, int32_t ntok, int8_t* aq, uint16_t* ad) {
    Q3_API_BEGIN
    require_gpu();
    Q3_CHECK(w && xq && xd && aq && ad && ff > 0 && ff % 32 == 0 && k % 256 == 0 && ntok >= 1, "bad gate/up shape");
    DevBuf<uint8_t> storage;
    Q8Mat m = q8mat_from_host(w, 2 * ff, k, storage);
    DevBuf<int8_t> dxq((size_t)ntok * k); dxq.upload(xq, dxq.n);
    DevBuf<uint16_t> dxd((size_t)ntok * (k / 32)); dxd.upload(xd, dxd.n);
    DevBuf<int8_t> daq((size_t)ntok * ff); DevBuf<uint16_t> dad((size_t)ntok * (ff / 32));
    if (!launch_gateup_mfma(0, m, ff, dxq.p, dxd.p, daq.p, dad.p, ntok)) throw Error("shape not served by the fused gate/up matrix-core kernel");
    Q3_LAUNCH_CHECK();
    Q3_HIP(hipDeviceSynchronize());
    daq.download(aq, daq.n); dad.download(ad, dad.n);
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}

int q3tts_op_gemv_kq(const void* const* raws, const int32_t* types, const int32_t* rows, int32_t nparts, int32_t k, const int8_t* xq, const uint16_t* xd,
                     int32_t ntok, float* y, int32_t lpr) {
    Q3_API_BEGIN
    require_gpu();
    Q3_CHECK(raws && types && rows && nparts >= 1 && nparts <= 3 && xq && xd && y && ntok >= 1 && k % 256 == 0, "bad K-quant gemv arguments");
    KqPart parts[3];
    int n = 0;
    for (int i = 0; i < nparts; i++) { parts[i] = KqPart{raws[i], types[i], rows[i]}; n += rows[i]; }
    DevBuf<uint8_t> storage;
    Q8Mat m = kqmat_from_host(parts, nparts, k, storage);
    DevBuf<int8_t> dxq((size_t)ntok * k); dxq.upload(xq, dxq.n);
    DevBuf<uint16_t> dxd((size_t)ntok * (k / 32)); dxd.upload(xd, dxd.n);
    const int nsseg = ((k >> 8) + 7) / 8;
    DevBuf<float> parts_out((size_t)nsseg * ntok * n);
    launch_gemv_q8(0, m, 0, n, dxq.p, dxd.p, parts_out.p, n, ntok, lpr);
    Q3_LAUNCH_CHECK();
    Q3_HIP(hipDeviceSynchronize());
    std::vector<float> hp(parts_out.n);
    parts_out.download(hp.data(), hp.size());
    for (int t = 0; t < ntok; t++)
        for (int r = 0; r < n; r++) { // spec S3: super-segment sums added in order
            float v = hp[((size_t)0 * ntok + t) * n + r];
            for (int s2 = 1; s2 < nsseg; s2++) v = v + hp[((size_t)s2 * ntok + t) * n + r];
            y[(size_t)t * n + r] = v;
        }
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}
int q3tts_op_gateup_kq(const void* gate_raw, const void* up_raw, int32_t type, int32_t ff, int32_t k, const int8_t* xq, const uint16_t* xd, int32_t ntok,
                       int8_t* aq, uint16_t* ad) {
    Q3_API_BEGIN
    require_gpu();
    Q3_CHECK(gate_raw && up_raw && xq && xd && aq && ad && ff > 0 && ff % 32 == 0 && k % 256 == 0 && ntok >= 1, "bad gate/up shape");
    KqPart parts[2] = {KqPart{gate_raw, type, ff}, KqPart{up_raw, type, ff}};
    DevBuf<uint8_t> storage;
    Q8Mat m = kqmat_from_host(parts, 2, k, storage);
    DevBuf<int8_t> dxq((size_t)ntok * k); dxq.upload(xq, dxq.n);
    DevBuf<uint16_t> dxd((size_t)ntok * (k / 32)); dxd.upload(xd, dxd.n);
    DevBuf<int8_t> daq((size_t)ntok * ff); DevBuf<uint16_t> dad((size_t)ntok * (ff / 32));
    if (!launch_gateup_mfma(0, m, ff, dxq.p, dxd.p, daq.p, dad.p, ntok)) throw Error("shape not served by the fused gate/up matrix-core kernel");
    Q3_LAUNCH_CHECK();
    Q3_HIP(hipDeviceSynchronize());
    daq.download(aq, daq.n); dad.download(ad, dad.n);
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}

int q3tts_op_matmul_float(const void* w, int32_t type, int32_t n, int32_t k, int32_t row0, int32_t nrows, const float* x, int32_t ntok, float* y) {
    Q3_API_BEGIN
    require_gpu();
    Q3_CHECK(type == Q3_T_F32 || type == Q3_T_F16 || type == Q3_T_BF16, "float type expected");
    Q3_CHECK(n > 0 && k > 0 && k % 256 == 0 && ntok > 0 && row0 >= 0 && nrows > 0 && row0 + nrows <= n, "bad matmul shape");
    const size_t esz = type == Q3_T_F32 ? 4 : 2;
    DevBuf<uint8_t> dw((size_t)n * k * esz); dw.upload((const uint8_t*)w, dw.n);
    DevBuf<uint8_t> dwt((size_t)((n + 63) / 64) * 64 * k * esz);
    launch_tile_float(0, dw.p, dwt.p, type, n, k);
    FMat m; m.w = dw.p; m.wt = dwt.p; m.type = type; m.N = n; m.K = k;
    DevBuf<float> dx((size_t)ntok * k); dx.upload(x, dx.n);
    DevBuf<float> dy((size_t)ntok * nrows);
    launch_gemv_float(0, m, row0, nrows, dx.p, k, dy.p, nrows, ntok);
    Q3_HIP(hipDeviceSynchronize());
    dy.download(y, dy.n);
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}

int q3tts_op_rmsnorm_quant(const float* x, const float* g, int32_t d, int32_t ntok, float eps, int8_t* xq, uint16_t* xd, float* xn) {
    Q3_API_BEGIN
    require_gpu();
    Q3_CHECK(d % 256 == 0 && d <= 2048, "d must be a multiple of 256 and <= 2048");
    DevBuf<float> dx((size_t)ntok * d), dg(d), dxn((size_t)ntok * d);
    dx.upload(x, dx.n); dg.upload(g, d);
    DevBuf<int8_t> dq((size_t)ntok * d); DevBuf<uint16_t> dd((size_t)ntok * d / 32);
    NormArgs a{};
    a.h_in = dx.p; a.h_stride = d; a.g = dg.p; a.eps = eps; a.d = d; a.xq = dq.p; a.xd = dd.p; a.xn_out = dxn.p;
    launch_rmsnorm_quant(0, a, ntok);
    Q3_HIP(hipDeviceSynchronize());
    if (xq) dq.download(xq, dq.n);
    if (xd) dd.download(xd, dd.n);
    if (xn) dxn.download(xn, dxn.n);
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}

int q3tts_op_swiglu_quant(const float* gu, int32_t ff, int32_t ntok, int8_t* aq, uint16_t* ad) {
    Q3_API_BEGIN
    require_gpu();
    DevBuf<float> dgu((size_t)ntok * 2 * ff); dgu.upload(gu, dgu.n);
    DevBuf<int8_t> dq((size_t)ntok * ff); DevBuf<uint16_t> dd((size_t)ntok * ff / 32);
    launch_swiglu_quant(0, dgu.p, ff, dq.p, dd.p, ntok);
    Q3_HIP(hipDeviceSynchronize());
    dq.download(aq, dq.n); dd.download(ad, dd.n);
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}

int q3tts_op_argmax(const float* logits, int32_t n, int32_t start, int32_t end, int32_t mask_idx, int32_t* out) {
    Q3_API_BEGIN
    require_gpu();
    DevBuf<float> dl(n); dl.upload(logits, n);
    DevBuf<int32_t> dm(1), dout(1);
    dm.upload(&mask_idx, 1);
    launch_argmax(0, dl.p, n, start, end, dm.p, dout.p, 1, 0, 1);
    Q3_HIP(hipDeviceSynchronize());
    dout.download(out, 1);
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}

/* n_draws consecutive samples of one sequence from the SAME logits (the RNG stream advances between draws) */
int q3tts_op_sample(const float* logits, int32_t n, float temperature, int32_t top_k, float top_p, uint64_t seed, int32_t mask_idx,
                    int32_t n_draws, int32_t* out) {
    Q3_API_BEGIN
    require_gpu();
    Q3_CHECK(logits && out && n >= 1 && n_draws >= 1, "bad arguments");
    StdRng rng(seed);
    DevBuf<float> dl(n), dt(1), dp(1); DevBuf<int32_t> dk(1), dm(1); DevBuf<uint32_t> dkey(8), ddraw(1); DevBuf<q3_u64> dout(n_draws);
    const uint32_t zero = 0;
    dl.upload(logits, n); dt.upload(&temperature, 1); dp.upload(&top_p, 1); dk.upload(&top_k, 1); dm.upload(&mask_idx, 1);
    dkey.upload(rng.key(), 8); ddraw.upload(&zero, 1);
    for (int i = 0; i < n_draws; i++) {
        SampleArgs a{dl.p, n, n, dt.p, dk.p, dp.p, dm.p, dkey.p, ddraw.p, dout.p + i, 1};
        launch_sample(0, a, 1);
    }
    Q3_HIP(hipDeviceSynchronize());
    std::vector<q3_u64> keys(n_draws);
    dout.download(keys.data(), n_draws);
    for (int i = 0; i < n_draws; i++) out[i] = (int32_t)(~(uint32_t)(keys[i] & 0xFFFFFFFFull));
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}

int q3tts_op_project(const float* x, const float* w, const float* b, int32_t n_in, int32_t n_out, float* y) {
    Q3_API_BEGIN
    require_gpu();
    std::vector<float> wt((size_t)n_in * n_out);
    for (int o = 0; o < n_out; o++) for (int i = 0; i < n_in; i++) wt[(size_t)i * n_out + o] = w[(size_t)o * n_in + i];
    DevBuf<float> dx(n_in), dw(wt.size()), db(n_out), dy(n_out);
    dx.upload(x, n_in); dw.upload(wt.data(), wt.size()); db.upload(b, n_out);
    launch_project(0, dx.p, n_in, dw.p, db.p, n_in, n_out, dy.p, n_out, 1);
    Q3_HIP(hipDeviceSynchronize());
    dy.download(y, n_out);
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}

int q3tts_mel_frames(int32_t n) { const int plen = n + 768; return (plen > 1024 ? plen - 1024 : 0) / 256 + 1; }

} // extern "C"

// ---------------- ONNX graph ingestion (host only, no GPU needed) ----------------
#include "onnx_reader.h"
#include "onnx_exec.h"
struct q3tts_onnx { std::unique_ptr<OnnxModel> m; std::string text; };
extern "C" {
int q3tts_onnx_open(const char* path, q3tts_onnx** out) {
    Q3_API_BEGIN
    Q3_CHECK(path && out, "null argument");
    auto* h = new q3tts_onnx();
    try { h->m.reset(new OnnxModel(path)); } catch (...) { delete h; throw; }
    *out = h;
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}
void q3tts_onnx_close(q3tts_onnx* m) { delete m; }
int q3tts_onnx_counts(q3tts_onnx* m, int32_t* n_nodes, int32_t* n_init, int32_t* n_in, int32_t* n_out) {
    if (!m) return Q3TTS_ERR;
    if (n_nodes) *n_nodes = (int32_t)m->m->nodes.size();
    if (n_init) *n_init = (int32_t)m->m->initializers.size();
    if (n_in) *n_in = (int32_t)m->m->inputs.size();
    if (n_out) *n_out = (int32_t)m->m->outputs.size();
    return Q3TTS_OK;
}
int64_t q3tts_onnx_summary(q3tts_onnx* m, char* buf, int64_t cap) {
    if (!m) return -1;
    m->text = m->m->summary();
    if (buf && cap > 0) { const size_t n = std::min((size_t)cap - 1, m->text.size()); std::memcpy(buf, m->text.data(), n); buf[n] = 0; }
    return (int64_t)m->text.size() + 1;
}
int q3tts_onnx_node(q3tts_onnx* m, int32_t i, const char** op_type, const char** name, int32_t* n_in, int32_t* n_out, int32_t* n_attr) {
    Q3_API_BEGIN
    Q3_CHECK(m && i >= 0 && i < (int)m->m->nodes.size(), "node index out of range");
    const OnnxNode& n = m->m->nodes[i];
    if (op_type) *op_type = n.op_type.c_str();
    if (name) *name = n.name.c_str();
    if (n_in) *n_in = (int32_t)n.inputs.size();
    if (n_out) *n_out = (int32_t)n.outputs.size();
    if (n_attr) *n_attr = (int32_t)n.attrs.size();
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}
const char* q3tts_onnx_node_input(q3tts_onnx* m, int32_t i, int32_t j) {
    return (m && i >= 0 && i < (int)m->m->nodes.size() && j >= 0 && j < (int)m->m->nodes[i].inputs.size()) ? m->m->nodes[i].inputs[j].c_str() : nullptr;
}
const char* q3tts_onnx_node_output(q3tts_onnx* m, int32_t i, int32_t j) {
    return (m && i >= 0 && i < (int)m->m->nodes.size() && j >= 0 && j < (int)m->m->nodes[i].outputs.size()) ? m->m->nodes[i].outputs[j].c_str() : nullptr;
}
/* INT / INTS attribute -> values (returns the count, -1 when the node has no such attribute) */
int32_t q3tts_onnx_node_attr_ints(q3tts_onnx* m, int32_t i, const char* attr, int64_t* out, int32_t cap) {
    if (!m || i < 0 || i >= (int)m->m->nodes.size() || !attr) return -1;
    const OnnxAttr* a = m->m->nodes[i].attr(attr);
    if (!a) return -1;
    if (a->type == 2 || (a->ints.empty() && a->type == 0)) { if (out && cap > 0) out[0] = a->i; return 1; }
    for (int k = 0; k < (int)a->ints.size() && k < cap; k++) out[k] = a->ints[k];
    return (int32_t)a->ints.size();
}
int32_t q3tts_onnx_node_attr_float(q3tts_onnx* m, int32_t i, const char* attr, float* out) {
    if (!m || i < 0 || i >= (int)m->m->nodes.size() || !attr || !out) return -1;
    const OnnxAttr* a = m->m->nodes[i].attr(attr);
    if (!a) return -1;
    *out = a->f;
    return 1;
}
int q3tts_onnx_initializer(q3tts_onnx* m, int32_t i, const char** name, int32_t* dtype, int64_t* dims8, int32_t* ndims, const void** data, int64_t* nbytes) {
    Q3_API_BEGIN
    Q3_CHECK(m && i >= 0 && i < (int)m->m->initializers.size(), "initializer index out of range");
    const OnnxTensor& t = m->m->initializers[i];
    Q3_CHECK(t.dims.size() <= 8, "more than 8 dimensions");
    if (name) *name = t.name.c_str();
    if (dtype) *dtype = t.data_type;
    if (ndims) *ndims = (int32_t)t.dims.size();
    if (dims8) for (size_t d = 0; d < t.dims.size(); d++) dims8[d] = t.dims[d];
    const void* p = t.raw; int64_t nb = (int64_t)t.raw_bytes;
    if (!p && !t.float_data.empty()) { p = t.float_data.data(); nb = (int64_t)t.float_data.size() * 4; }
    else if (!p && !t.int64_data.empty()) { p = t.int64_data.data(); nb = (int64_t)t.int64_data.size() * 8; }
    else if (!p && !t.int32_data.empty()) { p = t.int32_data.data(); nb = (int64_t)t.int32_data.size() * 4; }
    if (data) *data = p;
    if (nbytes) *nbytes = nb;
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}
const char* q3tts_onnx_op_kernel(const char* op_type) { return op_type ? onnx_op_kernel(op_type) : nullptr; }
/* 0: the graph carries the streaming-decoder inputs / outputs of onnx.rs:355-455; 1: something is missing (listed in buf) */
int q3tts_onnx_decoder_contract(q3tts_onnx* m, char* buf, int64_t cap) {
    if (!m) return Q3TTS_ERR;
    const std::string miss = m->m->check_decoder_contract();
    if (buf && cap > 0) { const size_t n = std::min((size_t)cap - 1, miss.size()); std::memcpy(buf, miss.data(), n); buf[n] = 0; }
    return miss.empty() ? 0 : 1;
}
} // extern "C"

// ---------------- tokenizer (host only) ----------------
#include "tokenizer.h"
struct q3tts_tokenizer { std::unique_ptr<Tokenizer> t; std::string text; };
extern "C" {
int q3tts_tokenizer_open(const char* tokenizer_json, q3tts_tokenizer** out) {
    Q3_API_BEGIN
    Q3_CHECK(tokenizer_json && out, "null argument");
    auto* h = new q3tts_tokenizer();
    try { h->t.reset(new Tokenizer(tokenizer_json)); } catch (...) { delete h; throw; }
    *out = h;
    return Q3TTS_OK;
    Q3_API_END(Q3TTS_ERR)
}
void q3tts_tokenizer_close(q3tts_tokenizer* t) { delete t; }
/* returns the number of ids (which may exceed cap: call again with a larger buffer), < 0 on error */
int32_t q3tts_tokenizer_encode(q3tts_tokenizer* t, const char* text_utf8, int32_t* ids, int32_t cap) {
    Q3_API_BEGIN
    Q3_CHECK(t && text_utf8, "null argument");
    const std::vector<int32_t> v = t->t->encode(text_utf8);
    for (size_t i = 0; i < v.size() && (int32_t)i < cap; i++) ids[i] = v[i];
    return (int32_t)v.size();
    Q3_API_END(-1)
}
/* returns the byte length of the decoded UTF-8 text (without NUL; may exceed cap - 1), < 0 on error */
int64_t q3tts_tokenizer_decode(q3tts_tokenizer* t, const int32_t* ids, int32_t n, char* buf, int64_t cap) {
    Q3_API_BEGIN
    Q3_CHECK(t && (ids || n == 0) && n >= 0, "bad arguments");
    t->text = t->t->decode(std::vector<int32_t>(ids, ids + n));
    if (buf && cap > 0) { const size_t k = std::min((size_t)cap - 1, t->text.size()); std::memcpy(buf, t->text.data(), k); buf[k] = 0; }
    return (int64_t)t->text.size();
    Q3_API_END(-1)
}
int64_t q3tts_text_nfc(const char* utf8, char* out, int64_t cap) {
    try {
        if (!utf8) return 0;
        const std::string r = q3::nfc_utf8(utf8);
        if (out && cap > 0) { const size_t n = std::min<size_t>((size_t)cap - 1, r.size()); memcpy(out, r.data(), n); out[n] = 0; }
        return (int64_t)r.size() + 1;
    } catch (const std::exception& e) { q3::set_last_error(e.what()); return -1; }
}
int32_t q3tts_tokenizer_vocab_size(q3tts_tokenizer* t) { return t ? t->t->vocab_size() : 0; }
} // extern "C"

// ---- ONNX graph execution (onnx_exec.h) ----
struct q3tts_onnx_session { std::unique_ptr<q3::OnnxSession> s; };
extern "C" {
int q3tts_onnx_session_open(const char* path, int32_t device, q3tts_onnx_session** out) {
    try {
        if (!path || !out) throw q3::Error("q3tts_onnx_session_open: null argument");
        auto* h = new q3tts_onnx_session();
        try { h->s.reset(new q3::OnnxSession(path, device)); } catch (...) { delete h; throw; }
        *out = h;
        return 0;
    } catch (const std::exception& e) { q3::set_last_error(e.what()); return 1; }
}
void q3tts_onnx_session_close(q3tts_onnx_session* s) { delete s; }
int32_t q3tts_onnx_session_unsupported(q3tts_onnx_session* s, char* buf, int64_t cap) {
    if (!s) return -1;
    const auto v = s->s->unsupported_ops();
    std::string t;
    for (size_t i = 0; i < v.size(); i++) t += (i ? "," : "") + v[i];
    if (buf && cap > 0) { const size_t n = std::min<size_t>((size_t)cap - 1, t.size()); memcpy(buf, t.data(), n); buf[n] = 0; }
    return (int32_t)v.size();
}
int q3tts_onnx_session_set_input(q3tts_onnx_session* s, const char* name, int32_t dtype, const void* data, const int64_t* shape, int32_t rank) {
    try {
        if (!s || !name || (rank > 0 && !shape)) throw q3::Error("q3tts_onnx_session_set_input: null argument");
        std::vector<int64_t> sh(shape, shape + rank);
        int64_t n = 1; for (auto d : sh) { if (d < 0) throw q3::Error("negative dimension"); n *= d; }
        if (n > 0 && !data) throw q3::Error("q3tts_onnx_session_set_input: null data");
        s->s->set_input(name, dtype, data, sh);
        return 0;
    } catch (const std::exception& e) { q3::set_last_error(e.what()); return 1; }
}
int q3tts_onnx_session_run(q3tts_onnx_session* s) {
    try { if (!s) throw q3::Error("null session"); s->s->run(); return 0; } catch (const std::exception& e) { q3::set_last_error(e.what()); return 1; }
}
int q3tts_onnx_session_output_info(q3tts_onnx_session* s, const char* name, int32_t* dtype, int32_t* rank, int64_t* shape8) {
    try {
        if (!s || !name) throw q3::Error("null argument");
        const q3::XTensor& t = s->s->value(name);
        if (t.shape.size() > 8) throw q3::Error("rank above 8");
        if (dtype) *dtype = t.dtype;
        if (rank) *rank = (int32_t)t.shape.size();
        if (shape8) for (size_t i = 0; i < t.shape.size(); i++) shape8[i] = t.shape[i];
        return 0;
    } catch (const std::exception& e) { q3::set_last_error(e.what()); return 1; }
}
int q3tts_onnx_session_output(q3tts_onnx_session* s, const char* name, void* dst, int64_t cap_bytes) {
    try {
        if (!s || !name || !dst || cap_bytes < 0) throw q3::Error("null argument");
        s->s->fetch(s->s->value(name), dst, (size_t)cap_bytes);
        return 0;
    } catch (const std::exception& e) { q3::set_last_error(e.what()); return 1; }
}
int64_t q3tts_onnx_session_launches(q3tts_onnx_session* s) { return s ? (int64_t)s->s->launches() : 0; }
int q3tts_onnx_op_executable(const char* op_type) { return op_type && q3::onnx_exec_supports(op_type) ? 1 : 0; }
}
struct q3tts_onnx_decoder { std::unique_ptr<q3::OnnxStreamDecoder> d; std::vector<float> last; }; // last = the most recent chunk's PCM (re-fetchable)
extern "C" {
int q3tts_onnx_decoder_open(const char* path, int32_t device, q3tts_onnx_decoder** out) {
    try {
        if (!path || !out) throw q3::Error("q3tts_onnx_decoder_open: null argument");
        auto* h = new q3tts_onnx_decoder();
        try { h->d.reset(new q3::OnnxStreamDecoder(path, device)); } catch (...) { delete h; throw; }
        *out = h;
        return 0;
    } catch (const std::exception& e) { q3::set_last_error(e.what()); return 1; }
}
void q3tts_onnx_decoder_close(q3tts_onnx_decoder* d) { delete d; }
int q3tts_onnx_decoder_reset(q3tts_onnx_decoder* d) {
    try { if (!d) throw q3::Error("null decoder"); d->d->reset(); d->last.clear(); return 0; } catch (const std::exception& e) { q3::set_last_error(e.what()); return 1; }
}
static int onnx_decoder_copy_out(q3tts_onnx_decoder* d, float* pcm, int64_t cap, int64_t* n_out) {
    *n_out = (int64_t)d->last.size();
    if ((int64_t)d->last.size() > cap || (!pcm && !d->last.empty())) {
        q3::set_last_error("pcm buffer too small for the chunk (" + std::to_string(d->last.size()) + " samples); the chunk is kept: call q3tts_onnx_decoder_fetch with a larger buffer");
        return 2;
    }
    if (!d->last.empty()) memcpy(pcm, d->last.data(), d->last.size() * 4);
    return 0;
}
int q3tts_onnx_decoder_decode(q3tts_onnx_decoder* d, const int64_t* codes, int32_t n_frames, int32_t is_final, float* pcm, int64_t cap, int64_t* n_out) {
    try {
        if (!d || !n_out) throw q3::Error("q3tts_onnx_decoder_decode: null argument");
        d->last = d->d->decode(codes, n_frames, is_final != 0); // the streaming state has advanced: the PCM stays in the handle until the next decode / reset
        return onnx_decoder_copy_out(d, pcm, cap, n_out);
    } catch (const std::exception& e) { q3::set_last_error(e.what()); return 1; }
}
int q3tts_onnx_decoder_fetch(q3tts_onnx_decoder* d, float* pcm, int64_t cap, int64_t* n_out) {
    try {
        if (!d || !n_out) throw q3::Error("q3tts_onnx_decoder_fetch: null argument");
        return onnx_decoder_copy_out(d, pcm, cap, n_out);
    } catch (const std::exception& e) { q3::set_last_error(e.what()); return 1; }
}
}
