// codec.h -- streaming codec decoder on device (codes -> 24 kHz PCM), the replacement for the ONNX AudioDecoder
// (/root/reference/src/models/onnx.rs:324-496).  Architecture "Q3TTS-codec-synth" (see DESIGN.md).
#pragma once
#include "q3_common.h"
#include <memory>
#include <string>
#include <vector>

namespace q3 {

struct CodecStateEntry { std::string name; int64_t offset; int rows, cols; }; // one named tensor of a stream's state blob ([rows][cols] f32 at `offset` floats)

class CodecDecoder {
public:
    // DecoderState (onnx.rs:461-496) out of / into the device: flat f32 blob per stream, layout = named history tensors + counters
    std::vector<CodecStateEntry> state_layout() const;
    size_t state_floats() const;
    void state_export(int stream, float* out, size_t n_floats) const; // n_floats must equal state_floats()
    void state_import(int stream, const float* in, size_t n_floats);  // validates the blob size and its trailer before committing
    // n_lanes: independent scratch sets so that decodes of different streams can run concurrently on different HIP streams
    // max_group: how many streams one pass may decode together (decode_group_async)
    CodecDecoder(const std::string& gguf_path, int n_streams, int max_frames_per_call, int n_lanes = 1, int max_group = 1);
    int max_group() const;
    int n_lanes() const;
    ~CodecDecoder();
    int samples_per_frame() const;
    void reset(int stream); // AudioDecoder::create_state (onnx.rs:338-340, 474-495)
    void reset_async(hipStream_t st, int stream); // same, ordered on `st` (call from the thread that issues the decodes)
    // codes: host [n_frames][16] (already clamped to [0,2047], engine.rs:515-519); pcm: host, n_frames*spf floats
    int decode(hipStream_t st, int stream, const int64_t* codes, int n_frames, bool is_last, float* pcm);
    // same, but returns right after enqueueing: pcm_pinned must be hipHostMalloc'd and stay valid until `st` is synchronised
    int decode_async(hipStream_t st, int stream, const int64_t* codes, int n_frames, bool is_last, float* pcm_pinned, int lane = 0);
    // G distinct streams, each with n_frames new frames: codes [G][n_frames][16], pcm[g] pinned destinations
    int decode_group_async(hipStream_t st, int G, const int* streams, const int64_t* codes, int n_frames, float* const* pcm_pinned, int lane = 0);
    double flops_per_frame() const;
private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

} // namespace q3
