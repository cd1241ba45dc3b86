// onnx_exec.h -- executes a parsed ONNX graph (onnx_reader.h) on the GPU: the second half of SURVEY.md 8f row f-2 and the engine behind
// row a17 (the voice-clone encoders).  The reference runs `qwen3_tts_codec_encoder.onnx` and `qwen3_tts_speaker_encoder.onnx` through
// ONNX Runtime sessions (/root/reference/src/models/onnx.rs:85-163: `input_values [1, T]` -> `audio_codes [1, F, 16]` i64 and
// `mels [1, n, 128]` -> `spk_emb [1, 2048]`); neither file is in this image, so this executor is a GENERAL interpreter of the operator set
// such exports use (convolutions, matmuls, normalisations, activations, reductions, shape arithmetic), checked operator by operator and on
// encoder-shaped graphs the tests write themselves (tests/test_gpu_onnx_exec.py, numpy as the reference).  It is off the hot path (it runs
// once per registered voice), so its kernels are plain and general -- one launch per node, f32 arithmetic, no fusion.
//
// Placement: float tensors live in HBM.  Small integer tensors (shapes, axes, indices of shape arithmetic: Shape -> Gather -> Concat ->
// Reshape chains) are evaluated on the host, where ONNX's integer semantics are exact and free; a value moves between the two on demand.
#pragma once
#include "onnx_reader.h"
#include "q3_common.h"
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace q3 {

struct XTensor {
    int dtype = 1;                       // ONNX element type: 1 f32, 7 i64, 9 bool (stored as f32 0 / 1 on the device, 0 / 1 on the host)
    std::vector<int64_t> shape;
    bool on_host = false;                // host-evaluated (values in hv); otherwise the payload is `dev`
    std::vector<double> hv;              // integers up to 2^53 and f32 values are exact in a double
    std::shared_ptr<DevBuf<uint8_t>> dev;
    int64_t numel() const { int64_t n = 1; for (auto d : shape) n *= d; return n; }
    size_t esize() const { return dtype == 7 ? 8 : 4; }
};

class OnnxSession {
public:
    OnnxSession(const std::string& path, int device);
    ~OnnxSession();
    const OnnxModel& model() const { return *model_; }
    // op types of the graph this executor cannot run (empty = the graph is executable)
    std::vector<std::string> unsupported_ops() const;
    void set_input(const std::string& name, int dtype, const void* data, const std::vector<int64_t>& shape);
    void bind_input(const std::string& name, const XTensor& t);     // a tensor already on the device (state fed back from a previous run)
    XTensor zeros(int dtype, const std::vector<int64_t>& shape);    // device tensor of zeros (any dimension may be 0)
    void run();                                                    // throws q3::Error naming the node on failure
    const XTensor& value(const std::string& name) const;           // any graph edge after run() (outputs stay alive)
    void fetch(const XTensor& t, void* dst, size_t cap_bytes) const; // f32 / bool -> float, i64 -> int64_t
    long launches() const { return launches_; }
private:
    struct Impl;
    std::unique_ptr<OnnxModel> model_;
    std::unique_ptr<Impl> impl_;
    int device_ = 0;
    long launches_ = 0;
};

// true when this executor has a kernel (or a host evaluation) for the op type
bool onnx_exec_supports(const std::string& op_type);

// AudioDecoder over a session (/root/reference/src/models/onnx.rs:322-458): the exported streaming decoder takes `audio_codes [1, N, 16]`, `is_last [1]`
// and the state tensors `pre_conv_history`, `latent_buffer`, `conv_history`, `past_key_i` / `past_value_i` (i = 0..7) and returns `final_wav`,
// `valid_samples` and the `next_*` state.  The state starts with zero-length time axes (DecoderState::new, onnx.rs:470-495) and stays on the device
// between chunks.  This is the general (one launch per node) route for the exported graph; the engine's own codec kernels are the fast route.
class OnnxStreamDecoder {
public:
    OnnxStreamDecoder(const std::string& path, int device);
    void reset();
    std::vector<float> decode(const int64_t* codes, int n_frames, bool is_final);   // AudioDecoder::decode, onnx.rs:341-458
    OnnxSession& session() { return s_; }
private:
    OnnxSession s_;
    std::vector<std::pair<std::string, std::string>> state_io_; // (input name, output name)
    std::map<std::string, XTensor> state_;
};

} // namespace q3
