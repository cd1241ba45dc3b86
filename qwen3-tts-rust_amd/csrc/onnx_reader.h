// onnx_reader.h -- minimal ONNX ModelProto reader (protobuf wire format walked by hand; there is no protobuf / onnx dependency).
//
// the decoder the reference runs is `onnx/qwen3_tts_decoder.onnx` through ONNX Runtime
// (/root/reference/src/tts/engine.rs:488-502; I/O contract /root/reference/src/models/onnx.rs:355-455, 474-495), and the voice-clone
// encoders are `qwen3_tts_codec_encoder.onnx` / `qwen3_tts_speaker_encoder.onnx` (onnx.rs:97-163).  None of the three files is in this
// image, so this reader is exercised on graphs the tests write themselves; it is the ingestion half of replacing the Code2Wav analogue
// with the exported graph: initialisers (weights), the node list with attributes, graph inputs / outputs with shapes, and a table that
// says which HIP kernel of this engine serves each op type.
// Field numbers follow onnx.proto3 (ONNX IR, public) [EXT].
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace q3 {

struct OnnxTensor {       // TensorProto
    std::string name;
    int32_t data_type = 0; // 1 f32, 6 i32, 7 i64, 10 f16, 11 f64, 16 bf16, 9 bool, 2 u8, 3 i8
    std::vector<int64_t> dims;
    const uint8_t* raw = nullptr; size_t raw_bytes = 0;   // raw_data (points into the mapped file)
    std::vector<float> float_data; std::vector<int64_t> int64_data; std::vector<int32_t> int32_data; // typed repeated fields
    bool external = false;                                 // data_location = EXTERNAL: payload lives in a side file (not loaded)
    int64_t elements() const { int64_t n = 1; for (auto d : dims) if (d < 0 || __builtin_mul_overflow(n, d, &n)) return -1; return n; } // -1: negative dim / overflow
};
struct OnnxAttr {         // AttributeProto
    std::string name; int32_t type = 0; // 1 FLOAT 2 INT 3 STRING 4 TENSOR 6 FLOATS 7 INTS
    float f = 0; int64_t i = 0; std::string s; std::vector<float> floats; std::vector<int64_t> ints; OnnxTensor t;
};
struct OnnxNode {         // NodeProto
    std::string name, op_type, domain; std::vector<std::string> inputs, outputs; std::vector<OnnxAttr> attrs;
    const OnnxAttr* attr(const std::string& n) const { for (auto& a : attrs) if (a.name == n) return &a; return nullptr; }
};
struct OnnxValueInfo {    // ValueInfoProto (tensor types only)
    std::string name; int32_t elem_type = 0; std::vector<int64_t> shape; std::vector<std::string> dim_params; // shape[d] = -1 for symbolic dims
};
struct OnnxModel {
    int64_t ir_version = 0; std::string producer; std::map<std::string, int64_t> opsets; std::string graph_name;
    std::vector<OnnxNode> nodes; std::vector<OnnxTensor> initializers; std::vector<OnnxValueInfo> inputs, outputs;
    explicit OnnxModel(const std::string& path);   // throws q3::Error
    ~OnnxModel();
    OnnxModel(const OnnxModel&) = delete; OnnxModel& operator=(const OnnxModel&) = delete;
    const OnnxTensor* initializer(const std::string& name) const;
    std::string summary() const;                   // human-readable dump (tools/q3onnx_dump)
    // the streaming-decoder I/O contract of /root/reference/src/models/onnx.rs:355-455: empty string = satisfied, else what is missing
    std::string check_decoder_contract() const;
private:
    uint8_t* map_ = nullptr; size_t size_ = 0; int fd_ = -1;
};
// op type -> the kernel of this engine that computes it (nullptr: no kernel yet); used by the dump and by the graph lowering to come
const char* onnx_op_kernel(const std::string& op_type);

} // namespace q3
