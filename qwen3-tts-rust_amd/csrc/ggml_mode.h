// ggml_mode.h -- launchers of the opt-in "ggml-CPU" arithmetic mode (ggml_mode.hip; Q3_SPEC=ggml).
#pragma once
#include "q3_common.h"
#include "kernels.h"

namespace q3 {

struct GgMat { const uint8_t* p = nullptr; int type = 0, n = 0, k = 0; size_t row_bytes = 0; }; // a GGUF matrix exactly as stored (raw blocks, row-major)
struct GgAct { int8_t* q8; uint16_t* d8; int8_t* qk; float* dk; int16_t* bs; };                 // activations as Q8_0 (per 32) and Q8_K (per 256) blocks

void gg_load_rows(hipStream_t st, const float* x, int x_stride, const int32_t* idx, int idx_stride, const unsigned long long* idx_keys, int d, float* h, int ntok);
void gg_add(hipStream_t st, float* h, const float* o, size_t n);
void gg_rmsnorm(hipStream_t st, const float* x, const float* g, int d, float eps, float* y, int ntok);
void gg_quant(hipStream_t st, const float* x, int k, const GgAct& a, int ntok);
void gg_matvec(hipStream_t st, const GgMat& w, int row0, int nrows, const GgAct& a, float* out, int out_stride, int ntok);
void gg_qk_rope_append(hipStream_t st, float* qkv, int stride, int n_head, int n_kv, const float* q_norm_w, const float* k_norm_w, float eps, const float* rope_cos,
                       const float* rope_sin, int n_ctx, const int32_t* mrope_sec, const TokMeta& tm, const KvCache& kv, int layer, int ntok);
void gg_attention(hipStream_t st, const float* qkv, int stride, int n_head, int n_kv, const TokMeta& tm, const KvCache& kv, int layer, float* att, float* scores,
                  int n_ctx, int ntok);
void gg_swiglu(hipStream_t st, const float* g, const float* u, float* y, size_t n);

} // namespace q3
