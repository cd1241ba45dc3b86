/* q3o_spec_export.c -- ORACLE (test infrastructure): exports the inline helpers of include/q3tts_spec.h so tests can
 * pin them against numpy (f16/bf16 conversion, expf, SwiGLU, block quantisation, RoPE pair, M-RoPE sector map). */
#include "q3o.h"
float q3o_spec_f16_to_f32(uint16_t h) { return q3_f16_to_f32(h); }
uint16_t q3o_spec_f32_to_f16(float f) { return q3_f32_to_f16(f); }
uint16_t q3o_spec_f32_to_bf16(float f) { return q3_f32_to_bf16(f); }
float q3o_spec_expf(float x) { return q3_expf(x); }
float q3o_spec_swiglu(float g, float u) { return q3_swiglu(g, u); }
uint16_t q3o_spec_quant_block32(const float* x, int8_t* q) { return q3_quant_block32(x, q); }
int q3o_spec_mrope_stream(int i, const int32_t* sec) { return q3_mrope_stream(i, sec); }
void q3o_spec_f16_to_f32_n(const uint16_t* h, float* f, int64_t n) { for (int64_t i = 0; i < n; i++) f[i] = q3_f16_to_f32(h[i]); }
void q3o_spec_f32_to_f16_n(const float* f, uint16_t* h, int64_t n) { for (int64_t i = 0; i < n; i++) h[i] = q3_f32_to_f16(f[i]); }
void q3o_spec_expf_n(const float* x, float* y, int64_t n) { for (int64_t i = 0; i < n; i++) y[i] = q3_expf(x[i]); }
void q3o_spec_swiglu_vec(const float* g, const float* u, int64_t n, float* out) { for (int64_t i = 0; i < n; i++) out[i] = q3_swiglu(g[i], u[i]); }
