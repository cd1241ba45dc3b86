// transformer.cpp -- see transformer.h.  (compiled by hipcc as HIP: contains the load-time repack kernel)
#include "transformer.h"
#include <cstdlib>
#include <cmath>

namespace q3 {

// ---- load-time repack: GGUF Q8_0 rows ([n][K/32]{f16 d; i8 qs[32]}) -> Q8Mat tiles (kernels.h) ----
__global__ void k_repack_q8(const uint8_t* __restrict__ raw, int n, int K, int row_off, uint8_t* __restrict__ qs,
                            uint16_t* __restrict__ sc, int sc_row_extra) { // quants go to row row_off + r of qs, scales to row row_off + sc_row_extra + r
    const int nb = K >> 5, nseg = K >> 8;
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (size_t)n * nb) return;
    const int r = (int)(id / nb), b = (int)(id % nb);
    const uint8_t* src = raw + id * 34;
    const int row = row_off + r, rg = row >> 5, r32 = row & 31;
    uint16_t d;
    memcpy(&d, src, 2);
    const int srow = row + sc_row_extra;
    sc[(((size_t)(srow >> 5) * nseg + (b >> 3)) * 32 + (srow & 31)) * 8 + (b & 7)] = d;
    uint8_t* dst = qs + ((size_t)rg * nb + b) * 1024 + r32 * 16;
    for (int i = 0; i < 16; i++) { dst[i] = src[2 + i]; dst[512 + i] = src[18 + i]; }
}

// Q5_K super-block {f16 d, dmin; u8 scales[12]; u8 qh[32]; u8 qs[128]} -> packed planes (kernels.h) + {d,dmin} + {sc[8], m[8]}
__global__ void k_repack_q5k(const uint8_t* __restrict__ raw, int n, int K, int row_off, uint8_t* __restrict__ qs, const uint32_t* __restrict__ rg_off,
                             uint16_t* __restrict__ sc, uint8_t* __restrict__ meta) {
    const int nseg = K >> 8;
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (size_t)n * nseg) return;
    const int r = (int)(id / nseg), s = (int)(id % nseg);
    const uint8_t* blk = raw + id * 176;
    const int row = row_off + r, rg = row >> 5, r32 = row & 31;
    const size_t vidx = ((size_t)rg * nseg + s) * 32 + r32;
    uint16_t d, dm;
    memcpy(&d, blk, 2); memcpy(&dm, blk + 2, 2);
    for (int i = 0; i < 8; i++) sc[vidx * 8 + i] = i == 0 ? d : i == 1 ? dm : 0;
    const uint8_t *scales = blk + 4, *qh = blk + 16, *ql = blk + 48;
    uint8_t* base = qs + (size_t)rg_off[rg] * 16 + (size_t)s * 5120;
    for (int j = 0; j < 8; j++) {
        int scv, mv;
        if (j < 4) { scv = scales[j] & 63; mv = scales[j + 4] & 63; }
        else { scv = (scales[j + 4] & 0xF) | ((scales[j - 4] >> 6) << 4); mv = (scales[j + 4] >> 4) | ((scales[j] >> 6) << 4); }
        meta[vidx * 16 + j] = (uint8_t)scv; meta[vidx * 16 + 8 + j] = (uint8_t)mv;
        const int jj = j >> 1, hi = j & 1;
        uint8_t q[32];
        uint32_t H = 0;
        for (int l = 0; l < 32; l++) {
            q[l] = (uint8_t)(hi ? (ql[32 * jj + l] >> 4) : (ql[32 * jj + l] & 0xF));
            H |= (uint32_t)((qh[l] >> (2 * jj + hi)) & 1) << (8 * (l & 3) + (l >> 2));
        }
        for (int h = 0; h < 2; h++)
            for (int w = 0; w < 2; w++)
                for (int bb = 0; bb < 4; bb++) base[j * 512 + h * 256 + r32 * 8 + w * 4 + bb] = (uint8_t)(q[16 * h + 8 * w + bb] | (q[16 * h + 8 * w + 4 + bb] << 4));
        memcpy(base + 4096 + r32 * 32 + j * 4, &H, 4);
    }
}
// Q6_K super-block {u8 ql[128]; u8 qh[64]; i8 scales[16]; f16 d} -> packed planes (values kept +32, unsigned 6 bit) + d + 16 sub-block scales
__global__ void k_repack_q6k(const uint8_t* __restrict__ raw, int n, int K, int row_off, uint8_t* __restrict__ qs, const uint32_t* __restrict__ rg_off,
                             uint16_t* __restrict__ sc, uint8_t* __restrict__ meta) {
    const int nseg = K >> 8;
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (size_t)n * nseg) return;
    const int r = (int)(id / nseg), s = (int)(id % nseg);
    const uint8_t* blk = raw + id * 210;
    const int row = row_off + r, rg = row >> 5, r32 = row & 31;
    const size_t vidx = ((size_t)rg * nseg + s) * 32 + r32;
    uint16_t d;
    memcpy(&d, blk + 208, 2);
    for (int i = 0; i < 8; i++) sc[vidx * 8 + i] = i == 0 ? d : 0;
    for (int i = 0; i < 16; i++) meta[vidx * 16 + i] = blk[192 + i];
    uint8_t* base = qs + (size_t)rg_off[rg] * 16 + (size_t)s * 6144;
    for (int j = 0; j < 8; j++) {
        const int half = j >> 2, grp = j & 3;
        const uint8_t* L = blk + 64 * half;
        const uint8_t* Hq = blk + 128 + 32 * half;
        uint8_t q[32];
        uint32_t H0 = 0, H1 = 0;
        for (int l = 0; l < 32; l++) {
            int lo, hb;
            switch (grp) {
                case 0: lo = L[l] & 0xF; hb = (Hq[l] >> 0) & 3; break;
                case 1: lo = L[l + 32] & 0xF; hb = (Hq[l] >> 2) & 3; break;
                case 2: lo = L[l] >> 4; hb = (Hq[l] >> 4) & 3; break;
                default: lo = L[l + 32] >> 4; hb = (Hq[l] >> 6) & 3; break;
            }
            q[l] = (uint8_t)lo;
            H0 |= (uint32_t)(hb & 1) << (8 * (l & 3) + (l >> 2));
            H1 |= (uint32_t)(hb >> 1) << (8 * (l & 3) + (l >> 2));
        }
        for (int h = 0; h < 2; h++)
            for (int w = 0; w < 2; w++)
                for (int bb = 0; bb < 4; bb++) base[j * 512 + h * 256 + r32 * 8 + w * 4 + bb] = (uint8_t)(q[16 * h + 8 * w + bb] | (q[16 * h + 8 * w + 4 + bb] << 4));
        memcpy(base + 4096 + r32 * 64 + j * 8, &H0, 4);
        memcpy(base + 4096 + r32 * 64 + j * 8 + 4, &H1, 4);
    }
}

Q8Mat q8mat_from_host(const void* raw_q8_0, int n, int k, DevBuf<uint8_t>& storage) {
    Q3_CHECK(k % 256 == 0 && n > 0, "bad Q8_0 matrix shape");
    Q8Mat m;
    m.N = n; m.Npad = (n + 31) & ~31; m.K = k;
    const size_t qs_bytes = (size_t)m.Npad * k, sc_bytes = (size_t)m.Npad * (k / 32) * 2;
    storage.alloc(qs_bytes + sc_bytes);
    storage.zero();
    m.qs = storage.p;
    m.sc = reinterpret_cast<const uint16_t*>(storage.p + qs_bytes);
    const size_t raw_bytes = (size_t)n * (k / 32) * 34;
    DevBuf<uint8_t> raw(raw_bytes);
    raw.upload((const uint8_t*)raw_q8_0, raw_bytes);
    const size_t nblk = (size_t)n * (k / 32);
    hipLaunchKernelGGL(k_repack_q8, dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, 0, raw.p, n, k, 0, storage.p,
                       reinterpret_cast<uint16_t*>(storage.p + qs_bytes), 0);
    Q3_HIP(hipDeviceSynchronize());
    return m;
}

Q8Mat kqmat_from_host(const KqPart* parts, int nparts, int k, DevBuf<uint8_t>& storage) {
    Q3_CHECK(parts && nparts >= 1 && nparts <= 3 && k % 256 == 0, "bad K-quant matrix description");
    Q8Mat m;
    int n = 0;
    for (int i = 0; i < nparts; i++) { Q3_CHECK(parts[i].rows > 0 && parts[i].rows % 32 == 0, "tensor rows must be a multiple of 32"); n += parts[i].rows; }
    m.N = n; m.Npad = n; m.K = k;
    const int nrg = n / 32;
    std::vector<uint8_t> types((size_t)nrg);
    std::vector<uint32_t> off((size_t)nrg);
    size_t qs_bytes = 0;
    { int rg = 0; for (int i = 0; i < nparts; i++) for (int j = 0; j < parts[i].rows / 32; j++, rg++) { types[(size_t)rg] = (uint8_t)parts[i].type; off[(size_t)rg] = (uint32_t)(qs_bytes / 16); qs_bytes += q3_rowgroup_bytes(parts[i].type, k); } }
    const size_t sc_bytes = (size_t)n * (k / 32) * 2, meta_bytes = (size_t)n * (k / 256) * 16, ty_bytes = (size_t)((nrg + 15) & ~15), off_bytes = (size_t)((nrg * 4 + 15) & ~15);
    storage.alloc(qs_bytes + sc_bytes + meta_bytes + ty_bytes + off_bytes);
    storage.zero();
    uint8_t* base = storage.p;
    uint16_t* sc = reinterpret_cast<uint16_t*>(base + qs_bytes);
    uint8_t* meta = base + qs_bytes + sc_bytes;
    uint8_t* ty = meta + meta_bytes;
    uint32_t* offd = reinterpret_cast<uint32_t*>(ty + ty_bytes);
    Q3_HIP(hipMemcpy(ty, types.data(), types.size(), hipMemcpyHostToDevice));
    Q3_HIP(hipMemcpy(offd, off.data(), off.size() * 4, hipMemcpyHostToDevice));
    m.qs = base; m.sc = sc; m.meta = meta; m.rg_type = ty; m.rg_off = offd; m.qbytes = qs_bytes;
    int last = -1, row_off = 0;
    for (int i = 0; i < nparts; i++) {
        const int t = parts[i].type, rows = parts[i].rows;
        Q3_CHECK(t == Q3_T_Q8_0 || t == Q3_T_Q5_K || t == Q3_T_Q6_K, "unsupported weight type");
        if (t != last) {
            const int rg0 = row_off / 32;
            if (m.nparts == 0) { m.p0_type = t; m.p0_off = off[(size_t)rg0]; }
            else if (m.nparts == 1) { m.p1_rg0 = rg0; m.p1_type = t; m.p1_off = off[(size_t)rg0]; }
            else { m.p2_rg0 = rg0; m.p2_type = t; m.p2_off = off[(size_t)rg0]; }
            m.nparts++; last = t;
        }
        const size_t raw_bytes = t == Q3_T_Q8_0 ? (size_t)rows * (k / 32) * 34 : (size_t)rows * (k / 256) * (t == Q3_T_Q5_K ? 176 : 210);
        DevBuf<uint8_t> raw(raw_bytes);
        raw.upload((const uint8_t*)parts[i].raw, raw_bytes);
        if (t == Q3_T_Q8_0) {
            const size_t nblk = (size_t)rows * (k / 32);
            hipLaunchKernelGGL(k_repack_q8, dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, 0, raw.p, rows, k, 0, base + (size_t)off[(size_t)(row_off / 32)] * 16, sc, row_off);
        } else {
            const size_t nsb = (size_t)rows * (k / 256);
            if (t == Q3_T_Q5_K) hipLaunchKernelGGL(k_repack_q5k, dim3((unsigned)((nsb + 255) / 256)), dim3(256), 0, 0, raw.p, rows, k, row_off, base, offd, sc, meta);
            else hipLaunchKernelGGL(k_repack_q6k, dim3((unsigned)((nsb + 255) / 256)), dim3(256), 0, 0, raw.p, rows, k, row_off, base, offd, sc, meta);
        }
        Q3_HIP(hipDeviceSynchronize());
        row_off += rows;
    }
    return m;
}

Q8Mat Transformer::make_mat(const Gguf& g, const std::vector<std::pair<std::string, int>>& rows, int N, int K) {
    Q3_CHECK(K % 256 == 0, "K must be a multiple of 256");
    Q8Mat m;
    m.N = N; m.Npad = (N + 31) & ~31; m.K = K;
    const int nrg = m.Npad / 32;
    // the tensors of the fused matrix decide the storage of every 32-row group (Q8_0 tiles or packed K-quant planes)
    std::vector<uint8_t> types((size_t)nrg, (uint8_t)Q3_T_Q8_0);
    bool kq = false;
    for (auto& r : rows) {
        const GgufTensor& t = g.need(r.first);
        if (t.type != Q3_T_Q8_0 && t.type != Q3_T_Q5_K && t.type != Q3_T_Q6_K)
            throw Error("tensor " + r.first + ": ggml type " + std::to_string(t.type) + " is not supported by the HIP path (Q8_0, Q5_K, Q6_K)");
        Q3_CHECK(r.second % 32 == 0 && t.ne[1] % 32 == 0 && r.second + t.ne[1] <= m.Npad, "bad row range for " + r.first);
        for (int64_t i = 0; i < t.ne[1] / 32; i++) types[(size_t)(r.second / 32 + i)] = (uint8_t)t.type;
        kq = kq || t.type != Q3_T_Q8_0;
    }
    std::vector<uint32_t> off((size_t)nrg);
    size_t qs_bytes = 0;
    for (int i = 0; i < nrg; i++) { off[(size_t)i] = (uint32_t)(qs_bytes / 16); qs_bytes += q3_rowgroup_bytes(kq ? types[(size_t)i] : Q3_T_Q8_0, K); }
    Q3_CHECK(qs_bytes / 16 < (1ull << 32), "matrix too large for 32-bit row-group offsets");
    const size_t sc_bytes = (size_t)m.Npad * (K / 32) * 2;
    const size_t meta_bytes = kq ? (size_t)m.Npad * (K / 256) * 16 : 0, ty_bytes = (size_t)((nrg + 15) & ~15), off_bytes = kq ? (size_t)((nrg * 4 + 15) & ~15) : 0;
    blobs_.emplace_back(qs_bytes + sc_bytes + meta_bytes + ty_bytes + off_bytes);
    blobs_.back().zero();
    uint8_t* base = blobs_.back().p;
    m.qs = base;
    m.sc = reinterpret_cast<const uint16_t*>(base + qs_bytes);
    mat_meta_[m.qs] = base + qs_bytes + sc_bytes;
    mat_types_[m.qs] = base + qs_bytes + sc_bytes + meta_bytes;
    mat_off_[m.qs] = reinterpret_cast<uint32_t*>(base + qs_bytes + sc_bytes + meta_bytes + ty_bytes);
    if (kq) {
        Q3_HIP(hipMemcpy(mat_types_[m.qs], types.data(), types.size(), hipMemcpyHostToDevice));
        Q3_HIP(hipMemcpy(mat_off_[m.qs], off.data(), off.size() * 4, hipMemcpyHostToDevice));
        m.meta = mat_meta_[m.qs]; m.rg_type = mat_types_[m.qs]; m.rg_off = mat_off_[m.qs]; m.qbytes = qs_bytes;
        all_q8_ = false;
        // tensor map for the kernel arguments: runs of equal type (rows not covered by a tensor stay Q8_0-typed zero tiles)
        int last = -1;
        for (int i = 0; i < nrg; i++) {
            if (last == types[(size_t)i]) continue;
            last = types[(size_t)i];
            Q3_CHECK(m.nparts < 3, "more than 3 runs of weight types in one fused matrix");
            if (m.nparts == 0) { m.p0_type = last; m.p0_off = off[(size_t)i]; }
            else if (m.nparts == 1) { m.p1_rg0 = i; m.p1_type = last; m.p1_off = off[(size_t)i]; }
            else { m.p2_rg0 = i; m.p2_type = last; m.p2_off = off[(size_t)i]; }
            m.nparts++;
        }
    }
    for (auto& r : rows) load_into(g, r.first, m, r.second, K);
    return m;
}

void Transformer::load_into(const Gguf& g, const std::string& name, Q8Mat& dst, int row_off, int K_expect) {
    const GgufTensor& t = g.need(name);
    Q3_CHECK(t.ne[0] == K_expect, "unexpected K for " + name);
    DevBuf<uint8_t> raw(t.nbytes);
    raw.upload(t.data, t.nbytes);
    uint8_t* qs = const_cast<uint8_t*>(dst.qs);
    uint16_t* sc = const_cast<uint16_t*>(dst.sc);
    if (t.type == Q3_T_Q8_0) {
        const size_t nblk = (size_t)t.ne[1] * (t.ne[0] / 32);
        // in a K-quant matrix the Q8_0 row groups start at their own offsets; the tile layout inside a row group is the same
        uint8_t* q0 = qs;
        int ro = row_off;
        if (dst.rg_off) { // row groups of one tensor are contiguous: rebase on the first one
            uint32_t o = 0;
            Q3_HIP(hipMemcpy(&o, dst.rg_off + row_off / 32, 4, hipMemcpyDeviceToHost));
            q0 = qs + (size_t)o * 16; ro = 0;
        }
        hipLaunchKernelGGL(k_repack_q8, dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, 0, raw.p, (int)t.ne[1], (int)t.ne[0], ro, q0, sc, row_off - ro);
    } else {
        const size_t nsb = (size_t)t.ne[1] * (t.ne[0] / 256);
        uint8_t* meta = mat_meta_.at(dst.qs);
        if (t.type == Q3_T_Q5_K) hipLaunchKernelGGL(k_repack_q5k, dim3((unsigned)((nsb + 255) / 256)), dim3(256), 0, 0, raw.p, (int)t.ne[1], (int)t.ne[0], row_off, qs, dst.rg_off, sc, meta);
        else hipLaunchKernelGGL(k_repack_q6k, dim3((unsigned)((nsb + 255) / 256)), dim3(256), 0, 0, raw.p, (int)t.ne[1], (int)t.ne[0], row_off, qs, dst.rg_off, sc, meta);
    }
    Q3_HIP(hipDeviceSynchronize());
}

FMat Transformer::make_fmat(const Gguf& g, const std::vector<std::string>& names, int K_expect) {
    FMat m;
    size_t total = 0;
    for (auto& n : names) {
        const GgufTensor& t = g.need(n);
        Q3_CHECK(t.ne[0] == K_expect && (t.type == Q3_T_F32 || t.type == Q3_T_F16 || t.type == Q3_T_BF16), "bad float tensor " + n);
        if (m.N == 0) m.type = t.type;
        Q3_CHECK(t.type == m.type, "mixed float types in one fused matrix: " + n);
        m.N += (int)t.ne[1]; total += t.nbytes;
    }
    m.K = K_expect;
    blobs_.emplace_back(total);
    size_t off = 0;
    for (auto& n : names) { const GgufTensor& t = g.need(n); Q3_HIP(hipMemcpy(blobs_.back().p + off, t.data, t.nbytes, hipMemcpyHostToDevice)); off += t.nbytes; }
    m.w = blobs_.back().p;
    if (K_expect % 256 == 0) { // tiled copy for the many-token kernel (k_gemm_float_mfma)
        const size_t esz = m.type == Q3_T_F32 ? 4 : 2;
        blobs_.emplace_back((size_t)((m.N + 63) / 64) * 64 * K_expect * esz);
        launch_tile_float(nullptr, m.w, blobs_.back().p, m.type, m.N, K_expect);
        Q3_HIP(hipDeviceSynchronize());
        m.wt = blobs_.back().p;
    }
    return m;
}

float* Transformer::load_f32(const Gguf& g, const std::string& name, int64_t n_expect) {
    const GgufTensor& t = g.need(name);
    Q3_CHECK(t.type == Q3_T_F32 && t.ne[0] * t.rows() == n_expect, "bad f32 tensor " + name);
    blobs_.emplace_back((size_t)n_expect * 4);
    blobs_.back().upload(t.data, (size_t)n_expect * 4);
    return reinterpret_cast<float*>(blobs_.back().p);
}

Transformer::Transformer(const std::string& path, int n_ctx, int max_tok) : n_ctx_(n_ctx), max_tok_(max_tok) {
    init_kernel_attributes();
    if (const char* e = std::getenv("Q3_UNFUSED")) fused = !(e[0] == '1'); // A/B switch for the parity tests (9 launches/layer)
    Gguf g(path);
    hp_.arch = g.kv_str("general.architecture", "qwen3");
    auto K = [&](const char* s) { return hp_.arch + "." + s; };
    hp_.n_embd = (int)g.kv_int(K("embedding_length"), 0);
    hp_.n_layer = (int)g.kv_int(K("block_count"), 0);
    hp_.n_head = (int)g.kv_int(K("attention.head_count"), 0);
    hp_.n_kv = (int)g.kv_int(K("attention.head_count_kv"), hp_.n_head);
    const int head_dim = (int)g.kv_int(K("attention.key_length"), hp_.n_head ? hp_.n_embd / hp_.n_head : 0);
    hp_.n_ff = (int)g.kv_int(K("feed_forward_length"), 0);
    hp_.eps = (float)g.kv_float(K("attention.layer_norm_rms_epsilon"), 1e-6);
    hp_.rope_base = (float)g.kv_float(K("rope.freq_base"), 1e6);
    if (auto* s = g.kv(K("rope.dimension_sections")))
        for (size_t i = 0; i < s->arr_i.size() && i < 4; i++) hp_.mrope_sec[i] = (int32_t)s->arr_i[i];
    Q3_CHECK(hp_.n_embd > 0 && hp_.n_layer > 0 && hp_.n_head > 0 && head_dim == Q3_HEAD_DIM, "unsupported hparams");
    Q3_CHECK(hp_.n_embd % 256 == 0 && hp_.n_embd <= 2048 && hp_.n_ff % 256 == 0 && hp_.n_head % hp_.n_kv == 0, "unsupported dims");
    Q3_CHECK(hp_.n_head * 128 <= 2048, "n_head*128 must be <= 2048 (single super-segment o-proj)");
    const int d = hp_.n_embd, dq = hp_.n_head * 128, dkv = hp_.n_kv * 128, ff = hp_.n_ff;
    layers_.resize(hp_.n_layer);
    {
        const int t0 = g.need("blk.0.attn_q.weight").type;
        float_mode_ = (t0 == Q3_T_F32 || t0 == Q3_T_F16 || t0 == Q3_T_BF16);
    }
    for (int l = 0; l < hp_.n_layer; l++) {
        Layer& L = layers_[l];
        const std::string p = "blk." + std::to_string(l) + ".";
        if (float_mode_) { // bf16 / f16 / f32 files: rows stay as stored, activations stay f32 (spec S3 float form)
            L.fqkv = make_fmat(g, {p + "attn_q.weight", p + "attn_k.weight", p + "attn_v.weight"}, d);
            L.fo = make_fmat(g, {p + "attn_output.weight"}, dq);
            L.fgu = make_fmat(g, {p + "ffn_gate.weight", p + "ffn_up.weight"}, d);
            L.fdown = make_fmat(g, {p + "ffn_down.weight"}, ff);
            L.attn_norm = load_f32(g, p + "attn_norm.weight", d);
            L.q_norm = load_f32(g, p + "attn_q_norm.weight", 128);
            L.k_norm = load_f32(g, p + "attn_k_norm.weight", 128);
            L.ffn_norm = load_f32(g, p + "ffn_norm.weight", d);
            const size_t lb = L.fqkv.bytes() + L.fo.bytes() + L.fgu.bytes() + L.fdown.bytes();
            if (l == 0) layer_weight_bytes_ = lb;
            weight_bytes_ += lb;
            continue;
        }
        L.wqkv = make_mat(g, {{p + "attn_q.weight", 0}, {p + "attn_k.weight", dq}, {p + "attn_v.weight", dq + dkv}}, dq + 2 * dkv, d);
        L.wo = make_mat(g, {{p + "attn_output.weight", 0}}, d, dq);
        L.wgu = make_mat(g, {{p + "ffn_gate.weight", 0}, {p + "ffn_up.weight", ff}}, 2 * ff, d);
        L.wdown = make_mat(g, {{p + "ffn_down.weight", 0}}, d, ff);
        if (L.wgu.nparts > 1) fused = false; // k_gateup_swiglu takes the type of the up rows from the gate rows; a file that mixes them runs the unfused sequence
        L.attn_norm = load_f32(g, p + "attn_norm.weight", d);
        L.q_norm = load_f32(g, p + "attn_q_norm.weight", 128);
        L.k_norm = load_f32(g, p + "attn_k_norm.weight", 128);
        L.ffn_norm = load_f32(g, p + "ffn_norm.weight", d);
        if (l == 0) layer_weight_bytes_ = L.wqkv.bytes() + L.wo.bytes() + L.wgu.bytes() + L.wdown.bytes();
        weight_bytes_ += L.wqkv.bytes() + L.wo.bytes() + L.wgu.bytes() + L.wdown.bytes();
    }
    if (const char* e = std::getenv("Q3_SPEC")) ggml_mode_ = std::string(e) == "ggml";
    if (ggml_mode_) {
        Q3_CHECK(!float_mode_, "Q3_SPEC=ggml serves Q8_0 / Q5_K_M files (float-weight files have no ggml-mode kernels)");
        gg_layers_.resize(hp_.n_layer);
        for (int l = 0; l < hp_.n_layer; l++) {
            const std::string p = "blk." + std::to_string(l) + ".";
            GgLayer& G = gg_layers_[l];
            G.wq = gg_load(g, p + "attn_q.weight", dq, d); G.wk = gg_load(g, p + "attn_k.weight", dkv, d); G.wv = gg_load(g, p + "attn_v.weight", dkv, d);
            G.wo = gg_load(g, p + "attn_output.weight", d, dq);
            G.gate = gg_load(g, p + "ffn_gate.weight", ff, d); G.up = gg_load(g, p + "ffn_up.weight", ff, d); G.down = gg_load(g, p + "ffn_down.weight", d, ff);
        }
    }
    output_norm_ = load_f32(g, "output_norm.weight", d);
    const GgufTensor& ot = g.need("output.weight");
    hp_.n_vocab = (int)ot.ne[1];
    if (float_mode_) { foutput_ = make_fmat(g, {"output.weight"}, d); fused = false; }
    else output_ = make_mat(g, {{"output.weight", 0}}, hp_.n_vocab, d);
    if (ggml_mode_) gg_output_ = gg_load(g, "output.weight", hp_.n_vocab, d);
    // RoPE tables: same double-precision expressions as the oracle (spec S5)
    std::vector<float> c((size_t)n_ctx * 64), s((size_t)n_ctx * 64);
    for (int p = 0; p < n_ctx; p++)
        for (int i = 0; i < 64; i++) {
            const double inv = std::pow((double)hp_.rope_base, -(double)i / 64.0);
            const double ang = (double)p * inv;
            c[(size_t)p * 64 + i] = (float)std::cos(ang);
            s[(size_t)p * 64 + i] = (float)std::sin(ang);
        }
    rope_cos_.alloc(c.size()); rope_cos_.upload(c.data(), c.size());
    rope_sin_.alloc(s.size()); rope_sin_.upload(s.data(), s.size());
    d_mrope_.alloc(4); d_mrope_.upload(hp_.mrope_sec, 4);
    alloc_workspace();
}

// second context over the same weights: see transformer.h
Transformer::Transformer(const Transformer& o, int max_tok)
    : fused(o.fused), float_mode_(o.float_mode_), foutput_(o.foutput_), hp_(o.hp_), n_ctx_(o.n_ctx_), max_tok_(max_tok), layers_(o.layers_), output_(o.output_),
      output_norm_(o.output_norm_), ws_(o.ws_), weight_bytes_(o.weight_bytes_), layer_weight_bytes_(o.layer_weight_bytes_), all_q8_(o.all_q8_),
      ggml_mode_(o.ggml_mode_), gg_layers_(o.gg_layers_), gg_output_(o.gg_output_) {
    alloc_workspace();
}

void Transformer::alloc_workspace() {
    const int d = hp_.n_embd, dq = hp_.n_head * 128, dkv = hp_.n_kv * 128, ff = hp_.n_ff;
    nparts_d_ = ((ff >> 8) + 7) / 8;
    const size_t T = (size_t)max_tok_;
    scratch_.alloc(T * 32); hid_.alloc(T * d); big_logits_.alloc(T * 2176);
    if (float_mode_) { xnf_.alloc(T * d); attf_.alloc(T * dq); actf_.alloc(T * ff); }
    h_.alloc(T * d); h2_.alloc(T * d); parts_o_.alloc(T * d); parts_d_.alloc((size_t)nparts_d_ * T * d);
    qkv_.alloc(T * (dq + 2 * dkv)); qrot_.alloc(T * dq); gu_.alloc(T * 2 * ff);
    const size_t mx = (size_t)(d > dq ? d : dq);
    xq_.alloc(T * mx); xd_.alloc(T * mx / 32); aq_.alloc(T * dq); ad_.alloc(T * dq / 32); fq_.alloc(T * ff); fd_.alloc(T * ff / 32);
    if (ggml_mode_) {
        const size_t kmax = (size_t)std::max(std::max(d, dq), ff);
        gg_q8_.alloc(T * kmax); gg_qk_.alloc(T * kmax); gg_d8_.alloc(T * kmax / 32); gg_dk_.alloc(T * kmax / 256); gg_bs_.alloc(T * kmax / 16);
        gg_xn_.alloc(T * kmax); gg_att_.alloc(T * dq); gg_act_f_.alloc(T * ff); gg_scores_.alloc(T * hp_.n_head * (size_t)n_ctx_);
        gg_act_ = GgAct{gg_q8_.p, gg_d8_.p, gg_qk_.p, gg_dk_.p, gg_bs_.p};
    }
}

// ---- ggml-arithmetic mode (ggml_mode.hip): raw GGUF matrices, llama.cpp's portable arithmetic, bit-exact with the test suite's CPU restatement of the mode ----
GgMat Transformer::gg_load(const Gguf& g, const std::string& name, int n_expect, int k_expect) {
    const GgufTensor& t = g.need(name);
    Q3_CHECK((int)t.ne[0] == k_expect && (int)t.ne[1] == n_expect, "ggml mode: shape of " + name);
    Q3_CHECK(t.type == Q3_T_Q8_0 || t.type == Q3_T_Q5_K || t.type == Q3_T_Q6_K, "ggml mode: type of " + name);
    GgMat m; m.type = t.type; m.n = n_expect; m.k = k_expect;
    m.row_bytes = t.type == Q3_T_Q8_0 ? (size_t)(k_expect / 32) * 34 : t.type == Q3_T_Q5_K ? (size_t)(k_expect / 256) * 176 : (size_t)(k_expect / 256) * 210;
    blobs_.emplace_back(m.row_bytes * (size_t)n_expect);
    blobs_.back().upload(reinterpret_cast<const uint8_t*>(t.data), m.row_bytes * (size_t)n_expect);
    m.p = blobs_.back().p;
    return m;
}
void Transformer::gg_linear(hipStream_t st, const GgMat& w, int row0, int nrows, const float* x, float* out, int out_stride, int ntok) {
    gg_quant(st, x, w.k, gg_act_, ntok);   // (ggml quantises the activations once per matmul call, to the weight type's vec_dot_type)
    gg_matvec(st, w, row0, nrows, gg_act_, out, out_stride, ntok);
}
void Transformer::forward_ggml(hipStream_t st, const Input& in, int ntok, const TokMeta& tm, const KvCache& kv, float* hidden_out) {
    const int d = hp_.n_embd, dq = hp_.n_head * 128, dkv = hp_.n_kv * 128, ff = hp_.n_ff, qs = dq + 2 * dkv;
    gg_load_rows(st, in.x, in.x_stride, in.idx, in.idx_stride, in.idx_keys, d, h_.p, ntok);
    for (int l = 0; l < hp_.n_layer; l++) {
        const Layer& L = layers_[l];
        const GgLayer& G = gg_layers_[l];
        gg_rmsnorm(st, h_.p, L.attn_norm, d, hp_.eps, gg_xn_.p, ntok);
        gg_quant(st, gg_xn_.p, d, gg_act_, ntok);
        gg_matvec(st, G.wq, 0, dq, gg_act_, qkv_.p, qs, ntok);
        gg_matvec(st, G.wk, 0, dkv, gg_act_, qkv_.p + dq, qs, ntok);
        gg_matvec(st, G.wv, 0, dkv, gg_act_, qkv_.p + dq + dkv, qs, ntok);
        gg_qk_rope_append(st, qkv_.p, qs, hp_.n_head, hp_.n_kv, L.q_norm, L.k_norm, hp_.eps, rope_cos_.p, rope_sin_.p, n_ctx_, d_mrope_.p, tm, kv, l, ntok);
        gg_attention(st, qkv_.p, qs, hp_.n_head, hp_.n_kv, tm, kv, l, gg_att_.p, gg_scores_.p, n_ctx_, ntok);
        gg_linear(st, G.wo, 0, d, gg_att_.p, parts_o_.p, d, ntok);
        gg_add(st, h_.p, parts_o_.p, (size_t)ntok * d);
        gg_rmsnorm(st, h_.p, L.ffn_norm, d, hp_.eps, gg_xn_.p, ntok);
        gg_quant(st, gg_xn_.p, d, gg_act_, ntok);
        gg_matvec(st, G.gate, 0, ff, gg_act_, gu_.p, ff, ntok);
        gg_matvec(st, G.up, 0, ff, gg_act_, gu_.p + (size_t)ntok * ff, ff, ntok);
        gg_swiglu(st, gu_.p, gu_.p + (size_t)ntok * ff, gg_act_f_.p, (size_t)ntok * ff);
        gg_linear(st, G.down, 0, d, gg_act_f_.p, parts_o_.p, d, ntok);
        gg_add(st, h_.p, parts_o_.p, (size_t)ntok * d);
    }
    gg_rmsnorm(st, h_.p, output_norm_, d, hp_.eps, hid_.p, ntok);
    if (hidden_out) launch_copy_f32(st, hid_.p, hidden_out, (size_t)ntok * d);
    Q3_LAUNCH_CHECK();
}


void Transformer::gemv(hipStream_t st, const Q8Mat& w, int row0, int nrows, const int8_t* xq, const uint16_t* xd, float* out,
                       int out_stride, int ntok) {
    if (timer) timer->begin(st);
    launch_gemv_q8(st, w, row0, nrows, xq, xd, out, out_stride, ntok);
    if (timer) timer->end(st, (double)nrows * ((double)w.K + (double)(w.K / 32) * 2.0));
}

void Transformer::forward(hipStream_t st, const Input& in, int ntok, const TokMeta& tm, const KvCache& kv, float* hidden_out) {
    Q3_CHECK(ntok >= 1 && ntok <= max_tok_, "ntok out of range");
    const int d = hp_.n_embd, dq = hp_.n_head * 128, dkv = hp_.n_kv * 128, ff = hp_.n_ff;
    if (ggml_mode_) { last_fused_ = false; last_ntok_ = ntok; forward_ggml(st, in, ntok, tm, kv, hidden_out); return; }
    if (float_mode_) { last_fused_ = false; last_ntok_ = ntok; forward_float(st, in, ntok, tm, kv, hidden_out); return; }
    const bool use_fused = fused && ntok <= fused_max_tok_; // batched steps (ntok > 8) take the weight-stationary token-sweep GEMM path
    last_fused_ = use_fused; last_ntok_ = ntok;
    if (use_fused) {
        // residual stream ping-pongs h_ <-> h2_: a fused prologue may not overwrite what other workgroups still read
        for (int l = 0; l < hp_.n_layer; l++) {
            const Layer& L = layers_[l];
            NormPro a{};
            if (l == 0) { a.h_in = in.x; a.h_stride = in.x_stride; a.idx_keys = in.idx_keys; a.idx_stride = in.idx_stride; a.nparts = 0; }
            else { a.h_in = h2_.p; a.h_stride = d; a.parts = parts_d_.p; a.nparts = nparts_d_; a.parts_stride = d; a.parts_slab = (size_t)ntok * d; }
            a.h_out = h_.p; a.g = L.attn_norm; a.eps = hp_.eps;
            if (timer) timer->begin(st);
            launch_gemv_q8_norm(st, L.wqkv, 0, dq + 2 * dkv, a, qkv_.p, dq + 2 * dkv, ntok, nullptr);
            if (timer) timer->end(st, (double)L.wqkv.bytes());
            if (same_seq_) {
                launch_qk_rope_append(st, qkv_.p, dq + 2 * dkv, nullptr, hp_.n_head, hp_.n_kv, L.q_norm, L.k_norm, hp_.eps, rope_cos_.p,
                                      rope_sin_.p, n_ctx_, d_mrope_.p, tm, kv, l, qrot_.p, ntok);
                launch_attention(st, qrot_.p, hp_.n_head, hp_.n_kv, tm, kv, l, nullptr, aq_.p, ad_.p, ntok);
            } else {
                if (short_attn_min_ > 0 && ntok >= short_attn_min_ && hp_.n_head == 2 * hp_.n_kv)
                    launch_attention_short(st, qkv_.p, dq + 2 * dkv, hp_.n_head, hp_.n_kv, L.q_norm, L.k_norm, hp_.eps, rope_cos_.p,
                                           rope_sin_.p, n_ctx_, d_mrope_.p, tm, kv, l, aq_.p, ad_.p, ntok);
                else
                launch_attention_fused(st, qkv_.p, dq + 2 * dkv, hp_.n_head, hp_.n_kv, L.q_norm, L.k_norm, hp_.eps, rope_cos_.p,
                                       rope_sin_.p, n_ctx_, d_mrope_.p, tm, kv, l, aq_.p, ad_.p, ntok);
            }
            gemv(st, L.wo, 0, d, aq_.p, ad_.p, parts_o_.p, d, ntok);
            NormPro b{};
            b.h_in = h_.p; b.h_stride = d; b.parts = parts_o_.p; b.nparts = 1; b.parts_stride = d; b.parts_slab = (size_t)ntok * d; b.h_out = h2_.p;
            b.g = L.ffn_norm; b.eps = hp_.eps;
            LaunchTimer* tg = timer_gu ? timer_gu : timer;
            if (tg) tg->begin(st);
            launch_gateup_swiglu(st, L.wgu, ff, b, fq_.p, fd_.p, ntok);
            if (tg) tg->end(st, (double)L.wgu.bytes());
            gemv(st, L.wdown, 0, d, fq_.p, fd_.p, parts_d_.p, d, ntok);
        }
        // final norm is the prologue of head(); hidden_out is produced there too
        if (hidden_out) head(st, 0, ntok, 0, 0, nullptr, 0, nullptr, -1, hidden_out);
        Q3_LAUNCH_CHECK();
        return;
    }
    const bool wgnorm = ntok > 8; // batched steps: workgroup-per-token norm kernel (same arithmetic, one global round trip)
    auto norm = [&](const NormArgs& a) {
        if (!wgnorm) { launch_rmsnorm_quant(st, a, ntok); return; }
        NormPro p{};
        p.h_in = a.h_in; p.h_stride = a.h_stride; p.idx_keys = a.idx_keys; p.idx_stride = a.idx_stride; p.parts = a.parts; p.nparts = a.nparts;
        p.parts_stride = a.parts_stride; p.parts_slab = (size_t)ntok * a.parts_stride; p.h_out = a.h_out; p.g = a.g; p.eps = a.eps; p.xn_out = a.xn_out;
        Q3_CHECK(a.idx == nullptr, "int32 row indices are not supported on the batched path");
        launch_rmsnorm_quant_wg(st, p, a.d, a.xq, a.xd, ntok);
    };
    // NB the wg kernel forbids h_out aliasing h_in only across workgroups; here each token is one workgroup, so in-place is safe
    for (int l = 0; l < hp_.n_layer; l++) {
        const Layer& L = layers_[l];
        NormArgs a{};
        if (l == 0) { a.h_in = in.x; a.h_stride = in.x_stride; a.idx = in.idx; a.idx_keys = in.idx_keys; a.idx_stride = in.idx_stride; a.nparts = 0; }
        else { a.h_in = h_.p; a.h_stride = d; a.parts = parts_d_.p; a.nparts = nparts_d_; a.parts_stride = d; }
        a.h_out = h_.p; a.g = L.attn_norm; a.eps = hp_.eps; a.d = d; a.xq = xq_.p; a.xd = xd_.p;
        norm(a);
        gemv(st, L.wqkv, 0, dq + 2 * dkv, xq_.p, xd_.p, qkv_.p, dq + 2 * dkv, ntok);
        if (short_attn_min_ > 0 && ntok >= short_attn_min_ && !same_seq_ && hp_.n_head == 2 * hp_.n_kv)
            launch_attention_short(st, qkv_.p, dq + 2 * dkv, hp_.n_head, hp_.n_kv, L.q_norm, L.k_norm, hp_.eps, rope_cos_.p, rope_sin_.p,
                                   n_ctx_, d_mrope_.p, tm, kv, l, aq_.p, ad_.p, ntok);
        else {
            launch_qk_rope_append(st, qkv_.p, dq + 2 * dkv, nullptr, hp_.n_head, hp_.n_kv, L.q_norm, L.k_norm, hp_.eps, rope_cos_.p,
                                  rope_sin_.p, n_ctx_, d_mrope_.p, tm, kv, l, qrot_.p, ntok);
            launch_attention(st, qrot_.p, hp_.n_head, hp_.n_kv, tm, kv, l, nullptr, aq_.p, ad_.p, ntok);
        }
        NormArgs b{};
        b.h_in = h_.p; b.h_stride = d; b.parts = parts_o_.p; b.nparts = 1; b.parts_stride = d; b.h_out = h_.p;
        b.g = L.ffn_norm; b.eps = hp_.eps; b.d = d; b.xq = xq_.p; b.xd = xd_.p;
        gemv(st, L.wo, 0, d, aq_.p, ad_.p, parts_o_.p, d, ntok);
        norm(b);
        bool gu_done = false;
        LaunchTimer* tg = timer_gu ? timer_gu : timer; // the talker's gate/up launch is timed on its own (bench roofline kernel)
        if (fused) {
            if (tg) tg->begin(st);
            gu_done = launch_gateup_mfma(st, L.wgu, ff, xq_.p, xd_.p, fq_.p, fd_.p, ntok);
            if (tg && gu_done) tg->end(st, (double)L.wgu.bytes()); // an unmatched begin() is simply re-recorded by the next one
        }
        if (!gu_done) {
            if (tg) { tg->begin(st); launch_gemv_q8(st, L.wgu, 0, 2 * ff, xq_.p, xd_.p, gu_.p, 2 * ff, ntok); tg->end(st, (double)L.wgu.bytes()); }
            else
            gemv(st, L.wgu, 0, 2 * ff, xq_.p, xd_.p, gu_.p, 2 * ff, ntok);
            launch_swiglu_quant(st, gu_.p, ff, fq_.p, fd_.p, ntok);
        }
        gemv(st, L.wdown, 0, d, fq_.p, fd_.p, parts_d_.p, d, ntok);
        if (l + 1 == hp_.n_layer) { // the final norm (the next layer's attention norm consumes the down-projection otherwise)
            NormArgs nx{};
            nx.h_in = h_.p; nx.h_stride = d; nx.parts = parts_d_.p; nx.nparts = nparts_d_; nx.parts_stride = d; nx.eps = hp_.eps; nx.d = d; nx.xq = xq_.p; nx.xd = xd_.p;
            nx.h_out = nullptr; nx.g = output_norm_; nx.xn_out = hidden_out ? hidden_out : hid_.p;
            norm(nx);
        }
    }
    Q3_LAUNCH_CHECK();
}

void Transformer::forward_float(hipStream_t st, const Input& in, int ntok, const TokMeta& tm, const KvCache& kv, float* hidden_out) {
    const int d = hp_.n_embd, dq = hp_.n_head * 128, dkv = hp_.n_kv * 128, ff = hp_.n_ff;
    auto norm = [&](const float* h_in, int h_stride, const unsigned long long* idx_keys, int idx_stride, const float* parts, const float* g, float* xn) {
        NormArgs a{};
        a.h_in = h_in; a.h_stride = h_stride; a.idx_keys = idx_keys; a.idx_stride = idx_stride; a.parts = parts; a.nparts = parts ? 1 : 0; a.parts_stride = d;
        a.h_out = h_.p; a.g = g; a.eps = hp_.eps; a.d = d; a.xq = xq_.p; a.xd = xd_.p; a.xn_out = xn; // int8 copy unused in float mode
        if (ntok <= 8) { launch_rmsnorm_quant(st, a, ntok); return; }
        NormPro p{}; // batched steps: workgroup-per-token norm kernel (same arithmetic)
        p.h_in = a.h_in; p.h_stride = a.h_stride; p.idx_keys = a.idx_keys; p.idx_stride = a.idx_stride; p.parts = a.parts; p.nparts = a.nparts;
        p.parts_stride = a.parts_stride; p.parts_slab = (size_t)ntok * a.parts_stride; p.h_out = a.h_out; p.g = a.g; p.eps = a.eps; p.xn_out = a.xn_out;
        launch_rmsnorm_quant_wg(st, p, a.d, a.xq, a.xd, ntok);
    };
    auto fgemv = [&](const FMat& w, const float* x, int xs, float* out, int os) {
        if (timer) timer->begin(st);
        launch_gemv_float(st, w, 0, w.N, x, xs, out, os, ntok);
        if (timer) timer->end(st, (double)w.bytes());
    };
    for (int l = 0; l < hp_.n_layer; l++) {
        const Layer& L = layers_[l];
        if (l == 0) norm(in.x, in.x_stride, in.idx_keys, in.idx_stride, nullptr, L.attn_norm, xnf_.p);
        else norm(h_.p, d, nullptr, 0, parts_d_.p, L.attn_norm, xnf_.p);
        fgemv(L.fqkv, xnf_.p, d, qkv_.p, dq + 2 * dkv);
        if (short_attn_min_ > 0 && ntok >= short_attn_min_ && !same_seq_ && hp_.n_head == 2 * hp_.n_kv)
            launch_attention_short(st, qkv_.p, dq + 2 * dkv, hp_.n_head, hp_.n_kv, L.q_norm, L.k_norm, hp_.eps, rope_cos_.p, rope_sin_.p,
                                   n_ctx_, d_mrope_.p, tm, kv, l, aq_.p, ad_.p, ntok, attf_.p);
        else {
            launch_qk_rope_append(st, qkv_.p, dq + 2 * dkv, nullptr, hp_.n_head, hp_.n_kv, L.q_norm, L.k_norm, hp_.eps, rope_cos_.p, rope_sin_.p,
                                  n_ctx_, d_mrope_.p, tm, kv, l, qrot_.p, ntok);
            launch_attention(st, qrot_.p, hp_.n_head, hp_.n_kv, tm, kv, l, attf_.p, aq_.p, ad_.p, ntok);
        }
        fgemv(L.fo, attf_.p, dq, parts_o_.p, d);
        norm(h_.p, d, nullptr, 0, parts_o_.p, L.ffn_norm, xnf_.p);
        LaunchTimer* tg = timer_gu ? timer_gu : timer;
        if (tg) tg->begin(st);
        const bool gu_fused = launch_gateup_float(st, L.fgu, ff, xnf_.p, d, actf_.p, ntok);
        if (tg && gu_fused) tg->end(st, (double)L.fgu.bytes()); // an unmatched begin() is simply re-recorded by the next one
        if (!gu_fused) {
            if (tg) { tg->begin(st); launch_gemv_float(st, L.fgu, 0, L.fgu.N, xnf_.p, d, gu_.p, 2 * ff, ntok); tg->end(st, (double)L.fgu.bytes()); }
            else
            fgemv(L.fgu, xnf_.p, d, gu_.p, 2 * ff);
            launch_swiglu_f32(st, gu_.p, ff, actf_.p, ntok);
        }
        fgemv(L.fdown, actf_.p, ff, parts_d_.p, d);
    }
    norm(h_.p, d, nullptr, 0, parts_d_.p, output_norm_, hidden_out ? hidden_out : hid_.p);
    if (hidden_out) launch_copy_f32(st, hidden_out, hid_.p, (size_t)ntok * d);
    Q3_LAUNCH_CHECK();
}

void Transformer::head(hipStream_t st, int tok0, int tok_count, int row0, int nrows, float* logits, int logits_stride,
                       const ArgmaxEpi* am, int nrows_valid, float* hidden_out) {
    head_impl(st, tok0, tok_count, row0, nrows, logits, logits_stride, am, nrows_valid, hidden_out);
    Q3_LAUNCH_CHECK();
}
void Transformer::head_impl(hipStream_t st, int tok0, int tok_count, int row0, int nrows, float* logits, int logits_stride,
                            const ArgmaxEpi* am, int nrows_valid, float* hidden_out) {
    const int d = hp_.n_embd;
    Q3_CHECK(row0 % 32 == 0 && (float_mode_ || row0 + nrows <= output_.Npad), "head row range");
    if (ggml_mode_) { // final-norm hidden states are in hid_ (f32): quantise them for the output matrix's type, plain dots, optional argmax keys
        if (hidden_out) launch_copy_f32(st, hid_.p + (size_t)tok0 * d, hidden_out, (size_t)tok_count * d);
        if (nrows <= 0) return;
        int nr = nrows;
        if (row0 + nr > gg_output_.n) nr = gg_output_.n - row0;   // (callers pass row counts padded to 32)
        float* dst = am ? big_logits_.p : logits;
        const int ls = am ? nrows : logits_stride;
        if (am && big_logits_.n < (size_t)tok_count * nrows) throw Error("head scratch too small");
        gg_linear(st, gg_output_, row0, nr, hid_.p + (size_t)tok0 * d, dst, ls, tok_count);
        if (am) launch_argmax_keys(st, big_logits_.p, nrows, std::min(nrows_valid > 0 ? nrows_valid : nrows, nr), am->mask_per_tok, am->keys, am->key_stride, tok_count);
        return;
    }
    if (last_fused_) {
        // final RMSNorm (+ last down-proj partials + residual) as the prologue of the output-matrix GEMV
        NormPro f{};
        f.h_in = h2_.p + (size_t)tok0 * d; f.h_stride = d;
        f.parts = parts_d_.p + (size_t)tok0 * d; f.nparts = nparts_d_; f.parts_stride = d;
        f.parts_slab = (size_t)last_ntok_ * d; // slabs are [p][ntok of the forward that wrote them][d]
        f.g = output_norm_; f.eps = hp_.eps; f.xn_out = hidden_out;
        if (nrows <= 0) { // hidden only: run the prologue through a 32-row GEMV whose output is discarded
            if (timer) timer->begin(st);
            launch_gemv_q8_norm(st, output_, 0, 32, f, scratch_logits(tok_count), 32, tok_count, nullptr);
            if (timer) timer->end(st, 32.0 * ((double)d * 1.0625));
            return;
        }
        if (timer) timer->begin(st);
        if (am) launch_gemv_q8_norm(st, output_, row0, nrows_valid > 0 ? nrows_valid : nrows, f, nullptr, 0, tok_count, am);
        else launch_gemv_q8_norm(st, output_, row0, nrows, f, logits, logits_stride, tok_count, nullptr);
        if (timer) timer->end(st, (double)nrows * ((double)d * 1.0625));
        return;
    }
    if (hidden_out) launch_copy_f32(st, hid_.p + (size_t)tok0 * d, hidden_out, (size_t)tok_count * d);
    if (nrows <= 0) return;
    if (float_mode_) {
        float* dst = am ? big_logits_.p : logits;
        const int ls = am ? nrows : logits_stride;
        if (timer) timer->begin(st);
        launch_gemv_float(st, foutput_, row0, nrows, hid_.p + (size_t)tok0 * d, d, dst, ls, tok_count);
        if (timer) timer->end(st, (double)nrows * d * (foutput_.type == Q3_T_F32 ? 4 : 2));
        if (am) launch_argmax_keys(st, big_logits_.p, nrows, nrows_valid > 0 ? nrows_valid : nrows, am->mask_per_tok, am->keys, am->key_stride, tok_count);
        return;
    }
    if (am) { // batched-step path: logits to scratch, then argmax keys
        const int nv = nrows_valid > 0 ? nrows_valid : nrows;
        if (big_logits_.n < (size_t)tok_count * nrows) throw Error("head scratch too small");
        gemv(st, output_, row0, nrows, xq_.p + (size_t)tok0 * d, xd_.p + (size_t)tok0 * (d / 32), big_logits_.p, nrows, tok_count);
        launch_argmax_keys(st, big_logits_.p, nrows, nv, am->mask_per_tok, am->keys, am->key_stride, tok_count);
        return;
    }
    gemv(st, output_, row0, nrows, xq_.p + (size_t)tok0 * d, xd_.p + (size_t)tok0 * (d / 32), logits, logits_stride, tok_count);
}

// ------------------------------------------------------------------------------------------------
KvPool::KvPool(int n_layer, int n_kv, int n_pages, int n_seq, int max_pages_per_seq)
    : n_layer_(n_layer), n_kv_(n_kv), n_pages_(n_pages), n_seq_(n_seq), max_pages_(max_pages_per_seq) {
    const size_t per_page = (size_t)n_layer * n_kv * 8192;
    k_.alloc(per_page * n_pages); v_.alloc(per_page * n_pages);
    k_.zero(); v_.zero();
    const size_t nt = (size_t)n_seq * max_pages_per_seq;
    Q3_HIP(hipHostMalloc((void**)&table_, nt * sizeof(int32_t)));
    std::fill(table_, table_ + nt, 0);
    d_table_.alloc(nt);
    d_table_.upload(table_, nt);
    for (int p = n_pages - 1; p >= 0; p--) free_.push_back(p);
    used_pages_.assign(n_seq, 0);
}
KvCache KvPool::view() const {
    KvCache c;
    c.k = k_.p; c.v = v_.p; c.page_table = d_table_.p; c.max_pages = max_pages_; c.n_layer = n_layer_; c.n_kv = n_kv_;
    return c;
}
int KvPool::alloc_page() { Q3_CHECK(!free_.empty(), "KV pool exhausted"); int p = free_.back(); free_.pop_back(); return p; }
void KvPool::free_page(int p) { free_.push_back(p); }
void KvPool::assign(int seq, int lp, int pp) {
    table_[(size_t)seq * max_pages_ + lp] = pp;
    Q3_HIP(hipMemcpy(d_table_.p + (size_t)seq * max_pages_ + lp, &pp, 4, hipMemcpyHostToDevice));
}
void KvPool::ensure(int seq, int n_positions) {
    const int need = (n_positions + 63) / 64;
    Q3_CHECK(seq >= 0 && seq < n_seq_, "KV sequence index out of range");
    Q3_CHECK(need <= max_pages_ && need <= n_pages_, "sequence exceeds max pages");
    while (used_pages_[seq] < need) { assign(seq, used_pages_[seq], alloc_page()); used_pages_[seq]++; }
}
void KvPool::ensure(int seq, int n_positions, hipStream_t st) {
    const int need = (n_positions + 63) / 64;
    Q3_CHECK(seq >= 0 && seq < n_seq_, "KV sequence index out of range");
    Q3_CHECK(need <= max_pages_ && need <= n_pages_, "sequence exceeds max pages");
    const int first = used_pages_[seq];
    while (used_pages_[seq] < need) { table_[(size_t)seq * max_pages_ + used_pages_[seq]] = alloc_page(); used_pages_[seq]++; }
    if (need > first) // rows of other sequences are untouched, and this row is not rewritten before `st` is synchronised by the caller
        Q3_HIP(hipMemcpyAsync(d_table_.p + (size_t)seq * max_pages_ + first, table_ + (size_t)seq * max_pages_ + first, (size_t)(need - first) * 4,
                              hipMemcpyHostToDevice, st));
}
KvPool::~KvPool() { if (table_) (void)hipHostFree(table_); }
void KvPool::release(int seq) {
    Q3_CHECK(seq >= 0 && seq < n_seq_, "KV sequence index out of range");
    for (int i = 0; i < used_pages_[seq]; i++) free_page(table_[(size_t)seq * max_pages_ + i]);
    used_pages_[seq] = 0;
}

} // namespace q3
