// onnx_reader.cpp -- see onnx_reader.h
#include "onnx_reader.h"
#include "q3_common.h"
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <sstream>

namespace q3 {
namespace {

// protobuf wire format: key = (field << 3) | wire_type; 0 varint, 1 fixed64, 2 length-delimited, 5 fixed32
struct Pb {
    const uint8_t* p; const uint8_t* end;
    bool done() const { return p >= end; }
    uint64_t varint() {
        uint64_t v = 0; int shift = 0;
        for (;;) {
            if (p >= end) throw Error("onnx: truncated varint");
            const uint8_t b = *p++;
            v |= (uint64_t)(b & 0x7F) << shift;
            if (!(b & 0x80)) return v;
            shift += 7;
            if (shift > 63) throw Error("onnx: varint too long");
        }
    }
    Pb sub() { // length-delimited payload
        const uint64_t n = varint();
        if (n > (uint64_t)(end - p)) throw Error("onnx: length-delimited field runs past its parent");
        Pb s{p, p + n};
        p += n;
        return s;
    }
    std::string str() { Pb s = sub(); return std::string((const char*)s.p, (size_t)(s.end - s.p)); }
    uint32_t fixed32() { if (4 > (size_t)(end - p)) throw Error("onnx: truncated fixed32"); uint32_t v; std::memcpy(&v, p, 4); p += 4; return v; }
    uint64_t fixed64() { if (8 > (size_t)(end - p)) throw Error("onnx: truncated fixed64"); uint64_t v; std::memcpy(&v, p, 8); p += 8; return v; }
    void skip(int wt) {
        switch (wt) {
            case 0: (void)varint(); break;
            case 1: (void)fixed64(); break;
            case 2: (void)sub(); break;
            case 5: (void)fixed32(); break;
            default: throw Error("onnx: unsupported wire type " + std::to_string(wt));
        }
    }
};
// repeated scalar fields arrive packed (wire type 2) or one by one
template <typename F> void rep_varint(Pb& m, int wt, F&& push) {
    if (wt == 2) { Pb s = m.sub(); while (!s.done()) push(s.varint()); } else push(m.varint());
}
void rep_float(Pb& m, int wt, std::vector<float>& out) {
    auto one = [&](uint32_t u) { float f; std::memcpy(&f, &u, 4); out.push_back(f); };
    if (wt == 2) { Pb s = m.sub(); while (!s.done()) one(s.fixed32()); } else one(m.fixed32());
}

void parse_tensor(Pb m, OnnxTensor& t) {
    while (!m.done()) {
        const uint64_t key = m.varint(); const int f = (int)(key >> 3), wt = (int)(key & 7);
        switch (f) {
            case 1: rep_varint(m, wt, [&](uint64_t v) { t.dims.push_back((int64_t)v); }); break;
            case 2: t.data_type = (int32_t)m.varint(); break;
            case 4: rep_float(m, wt, t.float_data); break;
            case 5: rep_varint(m, wt, [&](uint64_t v) { t.int32_data.push_back((int32_t)v); }); break;
            case 7: rep_varint(m, wt, [&](uint64_t v) { t.int64_data.push_back((int64_t)v); }); break;
            case 8: t.name = m.str(); break;
            case 9: { Pb s = m.sub(); t.raw = s.p; t.raw_bytes = (size_t)(s.end - s.p); break; }
            case 14: t.external = m.varint() == 1; break;
            default: m.skip(wt);
        }
    }
    for (auto d : t.dims) if (d < 0 || d > ((int64_t)1 << 40)) throw Error("onnx: implausible tensor dimension in " + t.name);
}
void parse_attr(Pb m, OnnxAttr& a) {
    while (!m.done()) {
        const uint64_t key = m.varint(); const int f = (int)(key >> 3), wt = (int)(key & 7);
        switch (f) {
            case 1: a.name = m.str(); break;
            case 2: { const uint32_t u = m.fixed32(); std::memcpy(&a.f, &u, 4); break; }
            case 3: a.i = (int64_t)m.varint(); break;
            case 4: a.s = m.str(); break;
            case 5: parse_tensor(m.sub(), a.t); break;
            case 7: rep_float(m, wt, a.floats); break;
            case 8: rep_varint(m, wt, [&](uint64_t v) { a.ints.push_back((int64_t)v); }); break;
            case 20: a.type = (int32_t)m.varint(); break;
            default: m.skip(wt);
        }
    }
}
void parse_node(Pb m, OnnxNode& n) {
    while (!m.done()) {
        const uint64_t key = m.varint(); const int f = (int)(key >> 3), wt = (int)(key & 7);
        switch (f) {
            case 1: n.inputs.push_back(m.str()); break;
            case 2: n.outputs.push_back(m.str()); break;
            case 3: n.name = m.str(); break;
            case 4: n.op_type = m.str(); break;
            case 5: n.attrs.emplace_back(); parse_attr(m.sub(), n.attrs.back()); break;
            case 7: n.domain = m.str(); break;
            default: m.skip(wt);
        }
    }
}
void parse_value_info(Pb m, OnnxValueInfo& v) {
    while (!m.done()) {
        const uint64_t key = m.varint(); const int f = (int)(key >> 3), wt = (int)(key & 7);
        if (f == 1) v.name = m.str();
        else if (f == 2) { // TypeProto
            Pb ty = m.sub();
            while (!ty.done()) {
                const uint64_t k2 = ty.varint();
                if ((k2 >> 3) == 1 && (k2 & 7) == 2) { // tensor_type
                    Pb tt = ty.sub();
                    while (!tt.done()) {
                        const uint64_t k3 = tt.varint();
                        if ((k3 >> 3) == 1) v.elem_type = (int32_t)tt.varint();
                        else if ((k3 >> 3) == 2) { // TensorShapeProto
                            Pb sh = tt.sub();
                            while (!sh.done()) {
                                const uint64_t k4 = sh.varint();
                                if ((k4 >> 3) == 1) { // Dimension
                                    Pb dm = sh.sub();
                                    int64_t val = -1; std::string par;
                                    while (!dm.done()) {
                                        const uint64_t k5 = dm.varint();
                                        if ((k5 >> 3) == 1) val = (int64_t)dm.varint();
                                        else if ((k5 >> 3) == 2) par = dm.str();
                                        else dm.skip((int)(k5 & 7));
                                    }
                                    v.shape.push_back(val); v.dim_params.push_back(par);
                                } else sh.skip((int)(k4 & 7));
                            }
                        } else tt.skip((int)(k3 & 7));
                    }
                } else ty.skip((int)(k2 & 7));
            }
        } else m.skip(wt);
    }
}
const char* dtype_name(int t) {
    switch (t) { case 1: return "f32"; case 2: return "u8"; case 3: return "i8"; case 6: return "i32"; case 7: return "i64"; case 9: return "bool";
                 case 10: return "f16"; case 11: return "f64"; case 16: return "bf16"; default: return "?"; }
}
} // namespace

OnnxModel::OnnxModel(const std::string& path) {
    fd_ = ::open(path.c_str(), O_RDONLY);
    if (fd_ < 0) throw Error("cannot open " + path);
    struct stat st;
    if (fstat(fd_, &st) != 0 || st.st_size <= 0) { ::close(fd_); fd_ = -1; throw Error("cannot stat " + path); }
    size_ = (size_t)st.st_size;
    void* mp = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
    if (mp == MAP_FAILED) { ::close(fd_); fd_ = -1; throw Error("mmap failed: " + path); }
    map_ = (uint8_t*)mp;
    try {
        Pb m{map_, map_ + size_};
        bool have_graph = false;
        while (!m.done()) {
            const uint64_t key = m.varint(); const int f = (int)(key >> 3), wt = (int)(key & 7);
            if (f == 1 && wt == 0) ir_version = (int64_t)m.varint();
            else if (f == 2 && wt == 2) producer = m.str();
            else if (f == 8 && wt == 2) { // OperatorSetIdProto {domain = 1, version = 2}
                Pb o = m.sub(); std::string dom; int64_t ver = 0;
                while (!o.done()) { const uint64_t k = o.varint(); if ((k >> 3) == 1) dom = o.str(); else if ((k >> 3) == 2) ver = (int64_t)o.varint(); else o.skip((int)(k & 7)); }
                opsets[dom] = ver;
            } else if (f == 7 && wt == 2) { // GraphProto
                have_graph = true;
                Pb g = m.sub();
                while (!g.done()) {
                    const uint64_t k = g.varint(); const int gf = (int)(k >> 3), gw = (int)(k & 7);
                    if (gf == 1 && gw == 2) { nodes.emplace_back(); parse_node(g.sub(), nodes.back()); }
                    else if (gf == 2 && gw == 2) graph_name = g.str();
                    else if (gf == 5 && gw == 2) { initializers.emplace_back(); parse_tensor(g.sub(), initializers.back()); }
                    else if (gf == 11 && gw == 2) { inputs.emplace_back(); parse_value_info(g.sub(), inputs.back()); }
                    else if (gf == 12 && gw == 2) { outputs.emplace_back(); parse_value_info(g.sub(), outputs.back()); }
                    else g.skip(gw);
                }
            } else m.skip(wt);
        }
        if (!have_graph) throw Error("Not an ONNX model (no GraphProto): " + path);
    } catch (...) {
        munmap(map_, size_); ::close(fd_); map_ = nullptr; fd_ = -1;
        throw;
    }
}
OnnxModel::~OnnxModel() { if (map_) munmap(map_, size_); if (fd_ >= 0) ::close(fd_); }

const OnnxTensor* OnnxModel::initializer(const std::string& name) const {
    for (auto& t : initializers) if (t.name == name) return &t;
    return nullptr;
}

const char* onnx_op_kernel(const std::string& op) {
    static const std::map<std::string, const char*> tab = {
        // convolution stack of the streaming decoder (csrc/codec.hip)
        {"Conv", "k_conv_gemm / k_conv_gemm_h (implicit GEMM over the extended buffer; depthwise k=7: k_dwconv7)"},
        {"ConvTranspose", "k_conv_gemm* with two taps and N = stride*cout"},
        {"MatMul", "k_conv_gemm* / k_skinny_gemm (M <= 16)"}, {"Gemm", "k_conv_gemm* / k_skinny_gemm with the bias epilogue"},
        {"Gather", "k_rvq_sum (codebook rows) / k_gather_rows_keys"}, {"Add", "GEMM epilogue EPI_RES / bias"}, {"Mul", "GEMM epilogue EPI_RES_SCALE (layer scale)"},
        {"Sin", "SnakeBeta: k_snake / GEMM epilogue EPI_SNAKE"}, {"Pow", "SnakeBeta (sin^2): k_snake"}, {"Exp", "SnakeBeta parameters (exp(alpha), 1/exp(beta)) folded at load"},
        {"Erf", "GEMM epilogue EPI_GELU"}, {"Gelu", "GEMM epilogue EPI_GELU"}, {"Sigmoid", "k_swiglu_rows"}, {"Softmax", "k_codec_attn"},
        {"LayerNormalization", "k_layernorm_rows"}, {"ReduceMean", "k_layernorm_rows / k_rmsnorm_rows"}, {"Sqrt", "k_rmsnorm_rows"}, {"Div", "k_rmsnorm_rows / k_codec_attn"},
        {"SimplifiedLayerNormalization", "k_rmsnorm_rows"}, {"RMSNormalization", "k_rmsnorm_rows"},
        {"Concat", "extended-buffer layout (history rows + new rows are one buffer): k_hist_all"}, {"Slice", "extended-buffer views / k_hist_all (state outputs)"},
        {"Transpose", "folded into GEMM addressing (time-major activations)"}, {"Reshape", "view"}, {"Unsqueeze", "view"}, {"Squeeze", "view"}, {"Cast", "load-time"},
        {"Clip", "k_conv_out_wave (final clamp)"}, {"Tanh", nullptr}, {"Where", "k_codec_attn (window mask)"}, {"Shape", "host"}, {"Constant", "host"},
        {"ConstantOfShape", "host"}, {"Expand", "view"}, {"Pad", "extended-buffer history rows"}, {"Sub", "k_layernorm_rows"}, {"Neg", "k_codec_rope"},
        {"Cos", "k_codec_rope_append (tables built at load)"}, {"Range", "host"}, {"Equal", "host"}, {"Less", "k_codec_attn (window mask)"}, {"Identity", "view"},
        // encoder-side ops that have no kernel in this engine yet (row a17)
        {"LSTM", nullptr}, {"GRU", nullptr}, {"Resize", nullptr}, {"InstanceNormalization", nullptr}, {"BatchNormalization", nullptr}, {"AveragePool", nullptr},
        {"GlobalAveragePool", nullptr}, {"Relu", nullptr}, {"LeakyRelu", nullptr}, {"Elu", nullptr}, {"ReduceSum", nullptr}, {"ArgMin", nullptr}, {"ArgMax", "k_argmax"},
    };
    auto it = tab.find(op);
    return it == tab.end() ? nullptr : it->second;
}

std::string OnnxModel::check_decoder_contract() const {
    // onnx.rs:355-455: inputs audio_codes, is_last, pre_conv_history, latent_buffer, conv_history, past_key_i / past_value_i (i = 0..7);
    // outputs final_wav, valid_samples, next_pre_conv_history, next_latent_buffer, next_conv_history, next_key_i / next_value_i
    auto has = [](const std::vector<OnnxValueInfo>& v, const std::string& n) { for (auto& x : v) if (x.name == n) return true; return false; };
    std::string miss;
    for (const char* n : {"audio_codes", "is_last", "pre_conv_history", "latent_buffer", "conv_history"}) if (!has(inputs, n)) miss += std::string(" input:") + n;
    for (int i = 0; i < 8; i++) for (const char* b : {"past_key_", "past_value_"}) if (!has(inputs, b + std::to_string(i))) miss += " input:" + std::string(b) + std::to_string(i);
    for (const char* n : {"final_wav", "valid_samples", "next_pre_conv_history", "next_latent_buffer", "next_conv_history"}) if (!has(outputs, n)) miss += std::string(" output:") + n;
    for (int i = 0; i < 8; i++) for (const char* b : {"next_key_", "next_value_"}) if (!has(outputs, b + std::to_string(i))) miss += " output:" + std::string(b) + std::to_string(i);
    return miss;
}

std::string OnnxModel::summary() const {
    std::ostringstream o;
    o << "ir_version " << ir_version << " producer '" << producer << "' graph '" << graph_name << "' opsets";
    for (auto& kv : opsets) o << " " << (kv.first.empty() ? "ai.onnx" : kv.first) << ":" << kv.second;
    o << "\n";
    auto vi = [&](const char* tag, const std::vector<OnnxValueInfo>& v) {
        for (auto& x : v) {
            bool init = initializer(x.name) != nullptr;
            if (init) continue; // (old exporters list initialisers among the graph inputs)
            o << tag << " " << x.name << " " << dtype_name(x.elem_type) << " [";
            for (size_t d = 0; d < x.shape.size(); d++) { if (d) o << ","; if (x.shape[d] >= 0) o << x.shape[d]; else o << (x.dim_params[d].empty() ? "?" : x.dim_params[d]); }
            o << "]\n";
        }
    };
    vi("input ", inputs); vi("output", outputs);
    size_t bytes = 0, n_ext = 0;
    for (auto& t : initializers) { bytes += t.raw_bytes + t.float_data.size() * 4 + t.int64_data.size() * 8 + t.int32_data.size() * 4; if (t.external) n_ext++; }
    o << "initializers " << initializers.size() << " (" << bytes << " bytes in file, " << n_ext << " external)\n";
    std::map<std::string, int> hist;
    for (auto& n : nodes) hist[n.op_type]++;
    int covered = 0;
    o << "nodes " << nodes.size() << "\n";
    for (auto& kv : hist) {
        const char* k = onnx_op_kernel(kv.first);
        if (k) covered += kv.second;
        o << "  " << kv.first << " x" << kv.second << " -> " << (k ? k : "NO KERNEL YET") << "\n";
    }
    o << "nodes served by existing kernels: " << covered << " / " << nodes.size() << "\n";
    const std::string miss = check_decoder_contract();
    o << "streaming-decoder I/O contract (onnx.rs:355-455): " << (miss.empty() ? "satisfied" : "missing" + miss) << "\n";
    return o.str();
}

} // namespace q3
