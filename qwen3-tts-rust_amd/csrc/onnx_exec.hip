// onnx_exec.hip -- see onnx_exec.h.  Kernels first (general N-d copies, element-wise maps with broadcasting, row reductions, matmul,
// convolutions), then the interpreter.  Operator semantics follow the public ONNX operator specification [EXT]; every operator is
// checked against numpy in tests/test_gpu_onnx_exec.py.
#include "onnx_exec.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <functional>
#include <limits>
#include <set>

namespace q3 {

// =============================================================== kernels ===============================================================
constexpr int XR = 8; // maximum rank
struct NdMap { int rank; int64_t oshape[XR]; int64_t istride[XR]; int64_t imod[XR]; int64_t ioff; };
// input offset of output element i: sum over dims of ((index_d % imod_d) * istride_d) + ioff  (stride 0 = broadcast, imod = tile period)
__device__ __forceinline__ int64_t nd_off(const NdMap& m, int64_t i) {
    int64_t off = m.ioff;
    for (int d = m.rank - 1; d >= 0; d--) { const int64_t q = i / m.oshape[d], r = i - q * m.oshape[d]; off += (r % m.imod[d]) * m.istride[d]; i = q; }
    return off;
}
template <typename T> __global__ void k_nd_copy(T* __restrict__ out, const T* __restrict__ in, NdMap m, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[nd_off(m, i)];
}

enum UOp { U_RELU, U_SIGMOID, U_TANH, U_EXP, U_LOG, U_SQRT, U_RECIP, U_NEG, U_ABS, U_SIN, U_COS, U_ERF, U_FLOOR, U_CEIL, U_ROUND, U_SIGN, U_ELU,
           U_LEAKY, U_HSIG, U_GELU, U_GELU_TANH, U_SOFTPLUS, U_SELU, U_NOT, U_CLIP, U_IDENT, U_SOFTSIGN, U_HSWISH, U_ISNAN, U_CELU, U_THRELU, U_MISH, U_TAN, U_ATAN, U_SINH, U_COSH, U_LOG1P_EXP_NEG };
__device__ __forceinline__ float apply_unary(int op, float x, float a, float b) {
    switch (op) {
        case U_RELU: return x > 0.f ? x : 0.f;
        case U_SIGMOID: return 1.f / (1.f + expf(-x));
        case U_TANH: return tanhf(x);
        case U_EXP: return expf(x);
        case U_LOG: return logf(x);
        case U_SQRT: return sqrtf(x);
        case U_RECIP: return 1.f / x;
        case U_NEG: return -x;
        case U_ABS: return fabsf(x);
        case U_SIN: return sinf(x);
        case U_COS: return cosf(x);
        case U_ERF: return erff(x);
        case U_FLOOR: return floorf(x);
        case U_CEIL: return ceilf(x);
        case U_ROUND: return rintf(x);
        case U_SIGN: return x > 0.f ? 1.f : x < 0.f ? -1.f : 0.f;
        case U_ELU: return x > 0.f ? x : a * (expf(x) - 1.f);
        case U_LEAKY: return x > 0.f ? x : a * x;
        case U_HSIG: return fminf(1.f, fmaxf(0.f, a * x + b));
        case U_GELU: return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
        case U_GELU_TANH: return 0.5f * x * (1.f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
        case U_SOFTPLUS: return x > 20.f ? x : log1pf(expf(x));
        case U_SELU: return b * (x > 0.f ? x : a * (expf(x) - 1.f));
        case U_NOT: return x != 0.f ? 0.f : 1.f;
        case U_CLIP: return fminf(b, fmaxf(a, x));
        case U_SOFTSIGN: return x / (1.f + fabsf(x));
        case U_HSWISH: return x * fminf(1.f, fmaxf(0.f, x / 6.f + 0.5f));
        case U_ISNAN: return x != x ? 1.f : 0.f;
        case U_CELU: return fmaxf(0.f, x) + fminf(0.f, a * (expf(x / a) - 1.f));
        case U_THRELU: return x > a ? x : 0.f;
        case U_MISH: return x * tanhf(x > 20.f ? x : log1pf(expf(x)));
        case U_TAN: return tanf(x);
        case U_ATAN: return atanf(x);
        case U_SINH: return sinhf(x);
        case U_COSH: return coshf(x);
        default: return x;
    }
}
__global__ void k_unary(float* __restrict__ out, const float* __restrict__ in, int op, float a, float b, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = apply_unary(op, in[i], a, b);
}

enum BOp { B_ADD, B_SUB, B_MUL, B_DIV, B_POW, B_MIN, B_MAX, B_EQ, B_LT, B_GT, B_LE, B_GE, B_AND, B_OR, B_XOR, B_MOD, B_FMOD, B_PRELU, B_IDIV };
__host__ __device__ __forceinline__ double apply_binary_d(int op, double x, double y) {
    switch (op) {
        case B_ADD: return x + y;
        case B_SUB: return x - y;
        case B_MUL: return x * y;
        case B_DIV: return x / y;
        case B_IDIV: return y != 0 ? (double)(long long)(x / y) : 0.0; // integer division truncates towards zero
        case B_POW: return pow(x, y);
        case B_MIN: return x < y ? x : y;
        case B_MAX: return x > y ? x : y;
        case B_EQ: return x == y ? 1.0 : 0.0;
        case B_LT: return x < y ? 1.0 : 0.0;
        case B_GT: return x > y ? 1.0 : 0.0;
        case B_LE: return x <= y ? 1.0 : 0.0;
        case B_GE: return x >= y ? 1.0 : 0.0;
        case B_AND: return (x != 0 && y != 0) ? 1.0 : 0.0;
        case B_OR: return (x != 0 || y != 0) ? 1.0 : 0.0;
        case B_XOR: return ((x != 0) != (y != 0)) ? 1.0 : 0.0;
        case B_MOD: { if (y == 0) return 0.0; double r = fmod(x, y); if (r != 0 && ((r < 0) != (y < 0))) r += y; return r; } // sign of the divisor (fmod = 0)
        case B_FMOD: return fmod(x, y);
        case B_PRELU: return x > 0 ? x : x * y;
        default: return x;
    }
}
__device__ __forceinline__ float apply_binary(int op, float x, float y) {
    switch (op) {
        case B_ADD: return x + y;
        case B_SUB: return x - y;
        case B_MUL: return x * y;
        case B_DIV: return x / y;
        case B_POW: return powf(x, y);
        case B_MIN: return fminf(x, y);
        case B_MAX: return fmaxf(x, y);
        case B_PRELU: return x > 0.f ? x : x * y;
        case B_FMOD: return fmodf(x, y);
        default: return (float)apply_binary_d(op, (double)x, (double)y);
    }
}
struct NdMap2 { int rank; int64_t oshape[XR]; int64_t as[XR]; int64_t bs[XR]; int64_t cs[XR]; };
__global__ void k_binary(float* __restrict__ out, const float* __restrict__ a, const float* __restrict__ b, int op, NdMap2 m, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t o = i;
    int64_t ao = 0, bo = 0;
    for (int d = m.rank - 1; d >= 0; d--) { const int64_t q = i / m.oshape[d], r = i - q * m.oshape[d]; ao += r * m.as[d]; bo += r * m.bs[d]; i = q; }
    out[o] = apply_binary(op, a[ao], b[bo]);
}
template <typename T>
__global__ void k_where(T* __restrict__ out, const float* __restrict__ c, const T* __restrict__ a, const T* __restrict__ b, NdMap2 m, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t o = i;
    int64_t ao = 0, bo = 0, co = 0;
    for (int d = m.rank - 1; d >= 0; d--) { const int64_t q = i / m.oshape[d], r = i - q * m.oshape[d]; ao += r * m.as[d]; bo += r * m.bs[d]; co += r * m.cs[d]; i = q; }
    out[o] = c[co] != 0.f ? a[ao] : b[bo];
}
__global__ void k_f32_to_i64(int64_t* __restrict__ out, const float* __restrict__ in, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (int64_t)in[i];
}
__global__ void k_i64_to_f32(float* __restrict__ out, const int64_t* __restrict__ in, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}
__global__ void k_fill(float* __restrict__ out, float v, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = v;
}
// out viewed as [outer][total][inner]; the input block [outer][part][inner] lands at axis offset `at`
template <typename T> __global__ void k_concat(T* __restrict__ out, const T* __restrict__ in, int64_t part, int64_t inner, int64_t total, int64_t at, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t in_ = i % inner, p = (i / inner) % part, o = i / (inner * part);
    out[(o * total + at + p) * inner + in_] = in[i];
}
// out[outer][j][inner] = data[outer][idx[j]][inner]
template <typename T> __global__ void k_gather(T* __restrict__ out, const T* __restrict__ data, const int64_t* __restrict__ idx, int64_t nidx, int64_t axis_dim, int64_t inner, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t in_ = i % inner, j = (i / inner) % nidx, o = i / (inner * nidx);
    int64_t k = idx[j];
    if (k < 0) k += axis_dim;
    k = k < 0 ? 0 : k >= axis_dim ? axis_dim - 1 : k;
    out[i] = data[(o * axis_dim + k) * inner + in_];
}

enum ROp { R_SUM, R_MEAN, R_MAX, R_MIN, R_PROD, R_L2, R_SUMSQ, R_L1, R_LOGSUMEXP, R_ARGMAX, R_ARGMIN };
// one workgroup per row; fixed-shape tree, so results do not depend on scheduling
__global__ void __launch_bounds__(256) k_reduce_rows(float* __restrict__ out, int64_t* __restrict__ out_idx, const float* __restrict__ in, int op, int64_t cols, int select_last) {
    __shared__ float sv[256];
    __shared__ int64_t si[256];
    const int64_t row = blockIdx.x;
    const float* x = in + row * cols;
    const bool arg = op == R_ARGMAX || op == R_ARGMIN;
    float acc = (op == R_MAX || op == R_ARGMAX || op == R_LOGSUMEXP) ? -INFINITY : (op == R_MIN || op == R_ARGMIN) ? INFINITY : op == R_PROD ? 1.f : 0.f;
    int64_t ai = 0;
    if (op == R_LOGSUMEXP) { // max first
        for (int64_t c = threadIdx.x; c < cols; c += 256) acc = fmaxf(acc, x[c]);
        sv[threadIdx.x] = acc; __syncthreads();
        for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) sv[threadIdx.x] = fmaxf(sv[threadIdx.x], sv[threadIdx.x + s]); __syncthreads(); }
        const float mx = sv[0]; __syncthreads();
        float t = 0.f;
        for (int64_t c = threadIdx.x; c < cols; c += 256) t += expf(x[c] - mx);
        sv[threadIdx.x] = t; __syncthreads();
        for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) sv[threadIdx.x] += sv[threadIdx.x + s]; __syncthreads(); }
        if (threadIdx.x == 0) out[row] = mx + logf(sv[0]);
        return;
    }
    for (int64_t c = threadIdx.x; c < cols; c += 256) {
        const float v = x[c];
        switch (op) {
            case R_SUM: case R_MEAN: acc += v; break;
            case R_MAX: acc = fmaxf(acc, v); break;
            case R_MIN: acc = fminf(acc, v); break;
            case R_PROD: acc *= v; break;
            case R_L2: case R_SUMSQ: acc += v * v; break;
            case R_L1: acc += fabsf(v); break;
            case R_ARGMAX: if (v > acc || (select_last && v == acc)) { acc = v; ai = c; } break;
            case R_ARGMIN: if (v < acc || (select_last && v == acc)) { acc = v; ai = c; } break;
        }
    }
    sv[threadIdx.x] = acc; si[threadIdx.x] = ai; __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            const float o = sv[threadIdx.x + s]; const int64_t oi = si[threadIdx.x + s];
            float m = sv[threadIdx.x]; int64_t mi = si[threadIdx.x];
            switch (op) {
                case R_MAX: m = fmaxf(m, o); break;
                case R_MIN: m = fminf(m, o); break;
                case R_PROD: m *= o; break;
                case R_ARGMAX: if (o > m || (o == m && (select_last ? oi > mi : oi < mi))) { m = o; mi = oi; } break;
                case R_ARGMIN: if (o < m || (o == m && (select_last ? oi > mi : oi < mi))) { m = o; mi = oi; } break;
                default: m += o; break;
            }
            sv[threadIdx.x] = m; si[threadIdx.x] = mi;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (arg) out_idx[row] = si[0];
        else out[row] = op == R_MEAN ? sv[0] / (float)cols : op == R_L2 ? sqrtf(sv[0]) : sv[0];
    }
}
__device__ __forceinline__ float block_sum(float v, float* sv) {
    sv[threadIdx.x] = v; __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) sv[threadIdx.x] += sv[threadIdx.x + s]; __syncthreads(); }
    const float r = sv[0]; __syncthreads();
    return r;
}
__device__ __forceinline__ float block_max(float v, float* sv) {
    sv[threadIdx.x] = v; __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) sv[threadIdx.x] = fmaxf(sv[threadIdx.x], sv[threadIdx.x + s]); __syncthreads(); }
    const float r = sv[0]; __syncthreads();
    return r;
}
__global__ void __launch_bounds__(256) k_softmax_rows(float* __restrict__ out, const float* __restrict__ in, int64_t cols, int logsm) {
    __shared__ float sv[256];
    const float* x = in + (int64_t)blockIdx.x * cols; float* y = out + (int64_t)blockIdx.x * cols;
    float m = -INFINITY;
    for (int64_t c = threadIdx.x; c < cols; c += 256) m = fmaxf(m, x[c]);
    m = block_max(m, sv);
    float s = 0.f;
    for (int64_t c = threadIdx.x; c < cols; c += 256) s += expf(x[c] - m);
    s = block_sum(s, sv);
    for (int64_t c = threadIdx.x; c < cols; c += 256) y[c] = logsm ? (x[c] - m) - logf(s) : expf(x[c] - m) / s;
}
// (x - mean) / sqrt(var + eps) per row; affine by column (LayerNormalization: g, b indexed by column) or by row group (InstanceNorm: channel = row % C)
__global__ void __launch_bounds__(256) k_norm_rows(float* __restrict__ out, const float* __restrict__ in, const float* __restrict__ g, const float* __restrict__ b,
                                                   int64_t cols, float eps, int by_row_channel, int64_t C) {
    __shared__ float sv[256];
    const int64_t row = blockIdx.x;
    const float* x = in + row * cols; float* y = out + row * cols;
    float s = 0.f;
    for (int64_t c = threadIdx.x; c < cols; c += 256) s += x[c];
    const float mean = block_sum(s, sv) / (float)cols;
    float v = 0.f;
    for (int64_t c = threadIdx.x; c < cols; c += 256) { const float d = x[c] - mean; v += d * d; }
    const float inv = 1.f / sqrtf(block_sum(v, sv) / (float)cols + eps);
    for (int64_t c = threadIdx.x; c < cols; c += 256) {
        const int64_t k = by_row_channel ? row % C : c;
        float t = (x[c] - mean) * inv;
        if (g) t *= g[k];
        if (b) t += b[k];
        y[c] = t;
    }
}
__global__ void k_batchnorm(float* __restrict__ out, const float* __restrict__ x, const float* __restrict__ sc, const float* __restrict__ bi,
                            const float* __restrict__ mean, const float* __restrict__ var, float eps, int64_t C, int64_t inner, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t c = (i / inner) % C;
    out[i] = (x[i] - mean[c]) / sqrtf(var[c] + eps) * sc[c] + bi[c];
}
__global__ void k_cumsum_rows(float* __restrict__ out, const float* __restrict__ in, int64_t rows, int64_t cols, int exclusive, int reverse) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    float acc = 0.f;
    for (int64_t k = 0; k < cols; k++) {
        const int64_t c = reverse ? cols - 1 - k : k;
        const float v = in[r * cols + c];
        if (exclusive) { out[r * cols + c] = acc; acc += v; } else { acc += v; out[r * cols + c] = acc; }
    }
}
// C[b][m][n] = alpha * sum_k A[b][m][k] B[b][k][n] (+ beta * bias), all operands through strides; 16 x 16 tiles staged in LDS
struct MmArgs { int64_t M, N, K; int64_t a_b, a_m, a_k, b_b, b_k, b_n; const float* bias; int64_t bias_m, bias_n; float alpha, beta; };
__global__ void __launch_bounds__(256) k_matmul(float* __restrict__ out, const float* __restrict__ A, const float* __restrict__ B, MmArgs g) {
    __shared__ float As[16][17], Bs[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t bz = blockIdx.z, m = (int64_t)blockIdx.y * 16 + ty, n = (int64_t)blockIdx.x * 16 + tx;
    const float* a = A + bz * g.a_b; const float* b = B + bz * g.b_b;
    float acc = 0.f;
    for (int64_t k0 = 0; k0 < g.K; k0 += 16) {
        As[ty][tx] = (m < g.M && k0 + tx < g.K) ? a[m * g.a_m + (k0 + tx) * g.a_k] : 0.f;
        Bs[ty][tx] = (k0 + ty < g.K && n < g.N) ? b[(k0 + ty) * g.b_k + n * g.b_n] : 0.f;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) acc = fmaf(As[ty][k], Bs[k][tx], acc);
        __syncthreads();
    }
    if (m < g.M && n < g.N) {
        float v = g.alpha * acc;
        if (g.bias) v += g.beta * g.bias[m * g.bias_m + n * g.bias_n];
        out[(bz * g.M + m) * g.N + n] = v;
    }
}
// The same product on the matrix cores (exact-f32 `v_mfma_f32_32x32x2_f32`, as the codec's k_conv_gemm): 64 x 64 output tile per workgroup (4 waves, 2 x 2),
// K tiles of 16 staged k-major in LDS with the next tile's loads in flight.  Operands through strides like k_matmul, plus an implicit-im2col B operand
// for 1-D convolutions: K index -> (channel, tap), N index -> output position, element = x[c][n*stride - pad + tap*dil] (0 outside the row), so
// Conv = W[M][C*kw] x im2col with the NCW output falling out row-major and nothing materialised.  ConvTranspose runs as W^T x X into per-tap
// columns followed by k_col2im1d.  Used for every MatMul / Gemm / ungrouped Conv1d / ConvTranspose1d with at least 16 rows and columns.
typedef float xf32x16 __attribute__((ext_vector_type(16)));
struct MmFast {
    int M, N, K;
    int64_t a_b, a_m, a_k, b_b, b_k, b_n;
    int conv, kw, cs, cp, cd, W;     // conv != 0: B is x[bz][c][pos] with row length W
    const float* bias; int64_t bias_m, bias_n; float alpha, beta;
};
__global__ void __launch_bounds__(256) k_mm_mfma(float* __restrict__ out, const float* __restrict__ A, const float* __restrict__ B, MmFast g) {
    __shared__ float As[16][65], Bs[16][65]; // 65: the k-fastest staging pattern writes 16 different k of one m -- distinct banks
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const float* a = A + (int64_t)blockIdx.z * g.a_b;
    const float* b = B + (int64_t)blockIdx.z * g.b_b;
    // which index runs fastest over the threads of a fetch: the one whose stride is 1 (coalescing); wave-uniform choices
    const bool a_mfast = g.a_m == 1 && g.a_k != 1;
    const bool b_nfast = g.conv ? true : (g.b_n == 1 || g.b_k != 1);
    xf32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;
    float ra[4], rb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int mm = a_mfast ? (tid & 63) : (tid >> 4) + 16 * i, kk = a_mfast ? (tid >> 6) + 4 * i : (tid & 15);
            const int m = m0 + mm, k = k0 + kk;
            ra[i] = (m < g.M && k < g.K) ? a[(int64_t)m * g.a_m + (int64_t)k * g.a_k] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int nn = b_nfast ? (tid & 63) : (tid >> 4) + 16 * i, kk = b_nfast ? (tid >> 6) + 4 * i : (tid & 15);
            const int n = n0 + nn, k = k0 + kk;
            float v = 0.0f;
            if (n < g.N && k < g.K) {
                if (g.conv) {
                    const int c = k / g.kw, tap = k - c * g.kw;
                    const int64_t pos = (int64_t)n * g.cs - g.cp + (int64_t)tap * g.cd;
                    if (pos >= 0 && pos < g.W) v = b[(int64_t)c * g.W + pos];
                } else v = b[(int64_t)k * g.b_k + (int64_t)n * g.b_n];
            }
            rb[i] = v;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int mm = a_mfast ? (tid & 63) : (tid >> 4) + 16 * i, kk = a_mfast ? (tid >> 6) + 4 * i : (tid & 15);
            As[kk][mm] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int nn = b_nfast ? (tid & 63) : (tid >> 4) + 16 * i, kk = b_nfast ? (tid >> 6) + 4 * i : (tid & 15);
            Bs[kk][nn] = rb[i];
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < g.K; k0 += 16) {
        stash();
        __syncthreads();
        if (k0 + 16 < g.K) fetch(k0 + 16);
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            const float av = As[kk + (lane >> 5)][wm * 32 + (lane & 31)];
            const float bv = Bs[kk + (lane >> 5)][wn * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int col = n0 + wn * 32 + (lane & 31);
    if (col < g.N) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < g.M) {
                float v = g.alpha * acc[r];
                if (g.bias) v += g.beta * g.bias[(int64_t)row * g.bias_m + (int64_t)col * g.bias_n];
                out[((int64_t)blockIdx.z * g.M + row) * g.N + col] = v;
            }
        }
    }
}
// overlap-add of a transposed 1-D convolution's per-tap columns Y[b][m*kw + k][i] (gather form, taps in increasing order): out[b][m][t]
__global__ void k_col2im1d(float* __restrict__ out, const float* __restrict__ Y, const float* __restrict__ bias, int64_t M, int64_t OW, int64_t W, int kw, int s, int p, int d, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t t = i % OW, m = (i / OW) % M, b = i / (OW * M);
    float acc = bias ? bias[m] : 0.f;
    const float* y = Y + (b * M + m) * kw * W;
    for (int k = 0; k < kw; k++) {
        const int64_t tt = t + p - (int64_t)k * d;
        if (tt < 0 || tt % s != 0) continue;
        const int64_t ix = tt / s;
        if (ix < W) acc += y[(int64_t)k * W + ix];
    }
    out[i] = acc;
}
static const bool g_onnx_naive = [] { const char* e = std::getenv("Q3_ONNX_NAIVE"); return e && e[0] == '1'; }(); // A/B: one-output-per-thread kernels everywhere
static bool mm_fast_ok(int64_t M, int64_t N, int64_t K, int64_t batch) {
    return !g_onnx_naive && M >= 16 && N >= 16 && K >= 1 && M < (1 << 30) && N < (1 << 30) && K < (1 << 30) && batch <= 65535 && (M + 63) / 64 <= 65535;
}
static void launch_mm_fast(float* out, const float* A, const float* B, const MmFast& g, int64_t batch) {
    hipLaunchKernelGGL(k_mm_mfma, dim3((unsigned)((g.N + 63) / 64), (unsigned)((g.M + 63) / 64), (unsigned)batch), dim3(256), 0, 0, out, A, B, g);
}
// direct 2-D convolution (1-D = H of 1), one output element per thread: general strides / pads / dilations / groups
struct ConvArgs { int64_t N, C, H, W, M, OH, OW, kh, kw, sh, sw, ph, pw, dh, dw, groups; };
__global__ void k_conv2d(float* __restrict__ out, const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, ConvArgs g, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t ow = i % g.OW, oh = (i / g.OW) % g.OH, m = (i / (g.OW * g.OH)) % g.M, b = i / (g.OW * g.OH * g.M);
    const int64_t cpg = g.C / g.groups, mpg = g.M / g.groups, grp = m / mpg;
    float acc = bias ? bias[m] : 0.f;
    for (int64_t c = 0; c < cpg; c++) {
        const float* xp = x + ((b * g.C + grp * cpg + c) * g.H) * g.W;
        const float* wp = w + ((m * cpg + c) * g.kh) * g.kw;
        for (int64_t ky = 0; ky < g.kh; ky++) {
            const int64_t iy = oh * g.sh - g.ph + ky * g.dh;
            if (iy < 0 || iy >= g.H) continue;
            for (int64_t kx = 0; kx < g.kw; kx++) {
                const int64_t ix = ow * g.sw - g.pw + kx * g.dw;
                if (ix < 0 || ix >= g.W) continue;
                acc = fmaf(xp[iy * g.W + ix], wp[ky * g.kw + kx], acc);
            }
        }
    }
    out[i] = acc;
}
// transposed convolution in gather form: weights [C][M / groups][kh][kw]
__global__ void k_convtr2d(float* __restrict__ out, const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, ConvArgs g, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t ow = i % g.OW, oh = (i / g.OW) % g.OH, m = (i / (g.OW * g.OH)) % g.M, b = i / (g.OW * g.OH * g.M);
    const int64_t cpg = g.C / g.groups, mpg = g.M / g.groups, grp = m / mpg, ml = m % mpg;
    float acc = bias ? bias[m] : 0.f;
    for (int64_t c = 0; c < cpg; c++) {
        const int64_t ci = grp * cpg + c;
        const float* xp = x + ((b * g.C + ci) * g.H) * g.W;
        const float* wp = w + ((ci * mpg + ml) * g.kh) * g.kw;
        for (int64_t ky = 0; ky < g.kh; ky++) {
            const int64_t ty = oh + g.ph - ky * g.dh;
            if (ty < 0 || ty % g.sh != 0) continue;
            const int64_t iy = ty / g.sh;
            if (iy >= g.H) continue;
            for (int64_t kx = 0; kx < g.kw; kx++) {
                const int64_t tx = ow + g.pw - kx * g.dw;
                if (tx < 0 || tx % g.sw != 0) continue;
                const int64_t ix = tx / g.sw;
                if (ix >= g.W) continue;
                acc = fmaf(xp[iy * g.W + ix], wp[ky * g.kw + kx], acc);
            }
        }
    }
    out[i] = acc;
}
// upper / lower triangle of the last two dims (k = diagonal offset)
__global__ void k_trilu(float* __restrict__ out, const float* __restrict__ in, int64_t rows, int64_t cols, int64_t k, int upper, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t c = i % cols, r = (i / cols) % rows;
    const bool keep = upper ? (c - r >= k) : (c - r <= k);
    out[i] = keep ? in[i] : 0.f;
}
// out[outer][j][inner] = data[outer][idx[outer][j][inner]][inner]   (GatherElements along one axis)
template <typename T> __global__ void k_gather_elems(T* __restrict__ out, const T* __restrict__ data, const int64_t* __restrict__ idx, int64_t nj, int64_t axis_dim, int64_t inner, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t in_ = i % inner, o = i / (inner * nj);
    int64_t k = idx[i];
    if (k < 0) k += axis_dim;
    k = k < 0 ? 0 : k >= axis_dim ? axis_dim - 1 : k;
    out[i] = data[(o * axis_dim + k) * inner + in_];
}
// ScatterND with whole-slice updates: out (a copy of data) [offsets[u] * slice + e] = updates[u * slice + e]
__global__ void k_scatter_rows(float* __restrict__ out, const float* __restrict__ upd, const int64_t* __restrict__ offsets, int64_t slice, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    out[offsets[i / slice] * slice + i % slice] = upd[i];
}
// nearest-neighbour Resize over the trailing spatial dims of an N-d tensor (asymmetric / floor: the exporters' default for integer factors)
struct ResizeArgs { int rank; int64_t oshape[XR]; int64_t ishape[XR]; float scale[XR]; };
__global__ void k_resize_nearest(float* __restrict__ out, const float* __restrict__ in, ResizeArgs a, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t o = i;
    int64_t off = 0, stride = 1;
    for (int d = a.rank - 1; d >= 0; d--) {
        const int64_t q = i / a.oshape[d], r = i - q * a.oshape[d];
        int64_t src = (int64_t)floorf((float)r / a.scale[d]);
        if (src > a.ishape[d] - 1) src = a.ishape[d] - 1;
        off += src * stride; stride *= a.ishape[d]; i = q;
    }
    out[o] = in[off];
}
struct PadArgs { int rank; int64_t oshape[XR]; int64_t ishape[XR]; int64_t begin[XR]; int mode; float value; }; // mode 0 constant, 1 reflect, 2 edge
__global__ void k_pad(float* __restrict__ out, const float* __restrict__ in, PadArgs p, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t o = i;
    int64_t off = 0, stride = 1;
    bool outside = false;
    for (int d = p.rank - 1; d >= 0; d--) {
        const int64_t q = i / p.oshape[d]; int64_t r = i - q * p.oshape[d] - p.begin[d];
        const int64_t L = p.ishape[d];
        if (r < 0 || r >= L) {
            if (p.mode == 0) outside = true;
            else if (p.mode == 2) r = r < 0 ? 0 : L - 1;
            else { if (L == 1) r = 0; else { const int64_t per = 2 * (L - 1); r = ((r % per) + per) % per; if (r >= L) r = per - r; } }
        }
        if (!outside) off += r * stride;
        stride *= L; i = q;
    }
    out[o] = outside ? p.value : in[off];
}

// ============================================================= interpreter =============================================================
static dim3 grid1(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }
static std::vector<int64_t> strides_of(const std::vector<int64_t>& s) { std::vector<int64_t> st(s.size(), 1); for (int d = (int)s.size() - 2; d >= 0; d--) st[d] = st[d + 1] * s[d + 1]; return st; }
static int64_t prod(const std::vector<int64_t>& s, size_t a = 0, size_t b = (size_t)-1) { int64_t n = 1; for (size_t i = a; i < std::min(b, s.size()); i++) n *= s[i]; return n; }
static std::vector<int64_t> bshape(const std::vector<int64_t>& a, const std::vector<int64_t>& b) {
    const size_t r = std::max(a.size(), b.size());
    std::vector<int64_t> o(r);
    for (size_t i = 0; i < r; i++) {
        const int64_t x = i + a.size() >= r ? a[i + a.size() - r] : 1, y = i + b.size() >= r ? b[i + b.size() - r] : 1;
        if (x != y && x != 1 && y != 1) throw Error("shapes do not broadcast");
        o[i] = x == 1 ? y : x;
    }
    return o;
}
static std::vector<int64_t> bstrides(const std::vector<int64_t>& in, const std::vector<int64_t>& out) { // strides of `in` seen through `out` (0 on broadcast dims)
    std::vector<int64_t> st = strides_of(in), o(out.size(), 0);
    for (size_t i = 0; i < in.size(); i++) { const size_t d = out.size() - in.size() + i; o[d] = in[i] == 1 ? 0 : st[i]; }
    return o;
}

static const std::map<std::string, int>& unary_table() {
    static const std::map<std::string, int> t = {{"Relu", U_RELU}, {"Sigmoid", U_SIGMOID}, {"Tanh", U_TANH}, {"Exp", U_EXP}, {"Log", U_LOG}, {"Sqrt", U_SQRT},
        {"Reciprocal", U_RECIP}, {"Neg", U_NEG}, {"Abs", U_ABS}, {"Sin", U_SIN}, {"Cos", U_COS}, {"Erf", U_ERF}, {"Floor", U_FLOOR}, {"Ceil", U_CEIL},
        {"Round", U_ROUND}, {"Sign", U_SIGN}, {"Elu", U_ELU}, {"LeakyRelu", U_LEAKY}, {"HardSigmoid", U_HSIG}, {"Gelu", U_GELU}, {"Softplus", U_SOFTPLUS},
        {"Selu", U_SELU}, {"Not", U_NOT}, {"Softsign", U_SOFTSIGN}, {"HardSwish", U_HSWISH}, {"IsNaN", U_ISNAN}, {"Celu", U_CELU}, {"ThresholdedRelu", U_THRELU},
        {"Mish", U_MISH}, {"Tan", U_TAN}, {"Atan", U_ATAN}, {"Sinh", U_SINH}, {"Cosh", U_COSH}};
    return t;
}
static const std::map<std::string, int>& binary_table() {
    static const std::map<std::string, int> t = {{"Add", B_ADD}, {"Sub", B_SUB}, {"Mul", B_MUL}, {"Div", B_DIV}, {"Pow", B_POW}, {"Min", B_MIN}, {"Max", B_MAX},
        {"Equal", B_EQ}, {"Less", B_LT}, {"Greater", B_GT}, {"LessOrEqual", B_LE}, {"GreaterOrEqual", B_GE}, {"And", B_AND}, {"Or", B_OR}, {"Xor", B_XOR},
        {"Mod", B_MOD}, {"PRelu", B_PRELU}, {"Sum", B_ADD}, {"Mean", B_ADD}};
    return t;
}
static const std::map<std::string, int>& reduce_table() {
    static const std::map<std::string, int> t = {{"ReduceSum", R_SUM}, {"ReduceMean", R_MEAN}, {"ReduceMax", R_MAX}, {"ReduceMin", R_MIN}, {"ReduceProd", R_PROD},
        {"ReduceL2", R_L2}, {"ReduceSumSquare", R_SUMSQ}, {"ReduceL1", R_L1}, {"ReduceLogSumExp", R_LOGSUMEXP}, {"ArgMax", R_ARGMAX}, {"ArgMin", R_ARGMIN}};
    return t;
}
static const std::set<std::string>& other_ops() {
    static const std::set<std::string> t = {"Identity", "Dropout", "Reshape", "Flatten", "Squeeze", "Unsqueeze", "Transpose", "Concat", "Slice", "Split", "Gather",
        "Shape", "Size", "Constant", "ConstantOfShape", "Range", "Cast", "Expand", "Tile", "Where", "Clip", "Softmax", "LogSoftmax", "LayerNormalization",
        "InstanceNormalization", "BatchNormalization", "MatMul", "Gemm", "Conv", "ConvTranspose", "Pad", "CumSum", "GlobalAveragePool", "GlobalMaxPool",
        "Trilu", "GatherElements", "ScatterND", "Resize", "GroupNormalization", "LpNormalization"};
    return t;
}
bool onnx_exec_supports(const std::string& op) { return unary_table().count(op) || binary_table().count(op) || reduce_table().count(op) || other_ops().count(op); }

struct OnnxSession::Impl {
    std::map<std::string, XTensor> vals;     // every live edge
    std::map<std::string, XTensor> consts;   // initialisers (uploaded once)
    std::map<std::string, XTensor> inputs;
    std::map<size_t, std::vector<std::shared_ptr<DevBuf<uint8_t>>>> pool; // buffers by byte size, reused between nodes / runs
    int64_t opset = 17;
    long* launches = nullptr;

    std::shared_ptr<DevBuf<uint8_t>> alloc(size_t bytes) {
        bytes = std::max<size_t>((bytes + 255) & ~(size_t)255, 256);
        auto& v = pool[bytes];
        for (auto& b : v) if (b.use_count() == 1) return b;
        v.push_back(std::make_shared<DevBuf<uint8_t>>(bytes));
        return v.back();
    }
    // every tensor is made here: a shape computed from graph data (Reshape / Expand / ConstantOfShape / Tile ... operands) with a negative
    // dimension or an element count that overflows becomes an Error before any buffer is sized from it
    static int64_t checked_numel(const std::vector<int64_t>& shape) {
        int64_t n = 1;
        for (auto d : shape) {
            Q3_CHECK(d >= 0, "tensor shape has a negative dimension");
            Q3_CHECK(!__builtin_mul_overflow(n, d, &n) && n <= ((int64_t)1 << 33), "tensor element count overflows / exceeds 2^33");
        }
        return n;
    }
    XTensor dev_tensor(int dtype, const std::vector<int64_t>& shape) {
        XTensor t; t.dtype = dtype; t.shape = shape; t.dev = alloc((size_t)std::max<int64_t>(checked_numel(shape), 1) * t.esize()); return t;
    }
    static XTensor host_tensor(int dtype, const std::vector<int64_t>& shape, std::vector<double> v) {
        Q3_CHECK((size_t)checked_numel(shape) == v.size(), "host tensor: " + std::to_string(v.size()) + " values for a shape of " + std::to_string(checked_numel(shape)) + " elements");
        XTensor t; t.dtype = dtype; t.shape = shape; t.on_host = true; t.hv = std::move(v); return t;
    }
    float* f(const XTensor& t) const { return reinterpret_cast<float*>(t.dev->p); }
    int64_t* i64(const XTensor& t) const { return reinterpret_cast<int64_t*>(t.dev->p); }
    void count() { if (launches) (*launches)++; }

    XTensor to_device(const XTensor& t) {
        if (!t.on_host) return t;
        XTensor d = dev_tensor(t.dtype, t.shape);
        const int64_t n = t.numel();
        if (n == 0) return d;
        if (t.dtype == 7) { std::vector<int64_t> h((size_t)n); for (int64_t i = 0; i < n; i++) h[(size_t)i] = (int64_t)t.hv[(size_t)i]; Q3_HIP(hipMemcpy(d.dev->p, h.data(), (size_t)n * 8, hipMemcpyHostToDevice)); }
        else { std::vector<float> h((size_t)n); for (int64_t i = 0; i < n; i++) h[(size_t)i] = (float)t.hv[(size_t)i]; Q3_HIP(hipMemcpy(d.dev->p, h.data(), (size_t)n * 4, hipMemcpyHostToDevice)); }
        return d;
    }
    XTensor to_host(const XTensor& t) {
        if (t.on_host) return t;
        const int64_t n = t.numel();
        Q3_CHECK(n <= (1 << 20), "tensor too large to evaluate on the host");
        XTensor h = host_tensor(t.dtype, t.shape, std::vector<double>((size_t)n));
        if (n == 0) return h;
        Q3_HIP(hipDeviceSynchronize());
        if (t.dtype == 7) { std::vector<int64_t> b((size_t)n); Q3_HIP(hipMemcpy(b.data(), t.dev->p, (size_t)n * 8, hipMemcpyDeviceToHost)); for (int64_t i = 0; i < n; i++) h.hv[(size_t)i] = (double)b[(size_t)i]; }
        else { std::vector<float> b((size_t)n); Q3_HIP(hipMemcpy(b.data(), t.dev->p, (size_t)n * 4, hipMemcpyDeviceToHost)); for (int64_t i = 0; i < n; i++) h.hv[(size_t)i] = (double)b[(size_t)i]; }
        return h;
    }
    // device tensor with f32 payload (i64 device tensors are converted; bool is f32 already)
    XTensor as_f32(const XTensor& t0) {
        XTensor t = to_device(t0);
        if (t.dtype != 7) return t;
        XTensor o = dev_tensor(1, t.shape);
        const int64_t n = t.numel();
        if (n) { hipLaunchKernelGGL(k_i64_to_f32, grid1(n), dim3(256), 0, 0, f(o), i64(t), n); count(); }
        return o;
    }
    XTensor as_i64(const XTensor& t0) {
        XTensor t = to_device(t0);
        if (t.dtype == 7) return t;
        XTensor o = dev_tensor(7, t.shape);
        const int64_t n = t.numel();
        if (n) { hipLaunchKernelGGL(k_f32_to_i64, grid1(n), dim3(256), 0, 0, i64(o), f(t), n); count(); }
        return o;
    }
    std::vector<int64_t> ints_of(const XTensor& t) { XTensor h = to_host(t); std::vector<int64_t> v(h.hv.size()); for (size_t i = 0; i < v.size(); i++) v[i] = (int64_t)h.hv[i]; return v; }

    // general strided copy (transpose / slice / expand / tile) on either side
    XTensor nd_copy(const XTensor& in, const std::vector<int64_t>& oshape, const std::vector<int64_t>& istride, const std::vector<int64_t>& imod, int64_t ioff) {
        Q3_CHECK(oshape.size() <= (size_t)XR, "rank above 8");
        const int64_t n = prod(oshape);
        if (in.on_host) {
            XTensor o = host_tensor(in.dtype, oshape, std::vector<double>((size_t)n));
            for (int64_t i = 0; i < n; i++) {
                int64_t r = i, off = ioff;
                for (int d = (int)oshape.size() - 1; d >= 0; d--) { const int64_t q = r / oshape[d], x = r - q * oshape[d]; off += (x % imod[d]) * istride[d]; r = q; }
                o.hv[(size_t)i] = in.hv[(size_t)off];
            }
            return o;
        }
        XTensor o = dev_tensor(in.dtype, oshape);
        if (n == 0) return o;
        NdMap m{}; m.rank = (int)oshape.size(); m.ioff = ioff;
        for (size_t d = 0; d < oshape.size(); d++) { m.oshape[d] = oshape[d]; m.istride[d] = istride[d]; m.imod[d] = imod[d]; }
        if (in.dtype == 7) hipLaunchKernelGGL(k_nd_copy<int64_t>, grid1(n), dim3(256), 0, 0, i64(o), i64(in), m, n);
        else hipLaunchKernelGGL(k_nd_copy<float>, grid1(n), dim3(256), 0, 0, f(o), f(in), m, n);
        count();
        return o;
    }
    static std::vector<int64_t> nomod(size_t r) { return std::vector<int64_t>(r, std::numeric_limits<int64_t>::max()); }
    XTensor transpose(const XTensor& in, const std::vector<int64_t>& perm) {
        Q3_CHECK(perm.size() == in.shape.size(), "Transpose: perm has " + std::to_string(perm.size()) + " entries for a rank-" + std::to_string(in.shape.size()) + " tensor");
        { std::vector<char> seen(perm.size(), 0); for (auto p : perm) { Q3_CHECK(p >= 0 && p < (int64_t)perm.size() && !seen[(size_t)p], "Transpose: perm is not a permutation"); seen[(size_t)p] = 1; } }
        const auto st = strides_of(in.shape);
        std::vector<int64_t> os(perm.size()), is(perm.size());
        for (size_t d = 0; d < perm.size(); d++) { os[d] = in.shape[(size_t)perm[d]]; is[d] = st[(size_t)perm[d]]; }
        return nd_copy(in, os, is, nomod(perm.size()), 0);
    }
    XTensor expand(const XTensor& in, const std::vector<int64_t>& oshape) {
        if (in.shape == oshape) return in;
        return nd_copy(in, oshape, bstrides(in.shape, oshape), nomod(oshape.size()), 0);
    }
    XTensor reshaped(const XTensor& in, const std::vector<int64_t>& shape) { XTensor o = in; o.shape = shape; Q3_CHECK(o.numel() == in.numel(), "reshape changes the element count"); return o; }

    XTensor binary(int op, const XTensor& a0, const XTensor& b0, int out_dtype) {
        const auto os = bshape(a0.shape, b0.shape);
        const int64_t n = prod(os);
        if (a0.on_host && b0.on_host) {
            XTensor o = host_tensor(out_dtype, os, std::vector<double>((size_t)n));
            const auto as = bstrides(a0.shape, os), bs = bstrides(b0.shape, os);
            const int hop = (op == B_DIV && a0.dtype == 7 && b0.dtype == 7) ? B_IDIV : op;
            for (int64_t i = 0; i < n; i++) {
                int64_t r = i, ao = 0, bo = 0;
                for (int d = (int)os.size() - 1; d >= 0; d--) { const int64_t q = r / os[d], x = r - q * os[d]; ao += x * as[d]; bo += x * bs[d]; r = q; }
                o.hv[(size_t)i] = apply_binary_d(hop, a0.hv[(size_t)ao], b0.hv[(size_t)bo]);
            }
            return o;
        }
        Q3_CHECK(!(a0.dtype == 7 && b0.dtype == 7 && !a0.on_host && !b0.on_host && op == B_DIV), "integer division of device tensors");
        XTensor a = as_f32(a0), b = as_f32(b0);
        XTensor o = dev_tensor(out_dtype == 7 ? 1 : out_dtype, os);
        if (n) {
            NdMap2 m{}; m.rank = (int)os.size();
            const auto as = bstrides(a.shape, os), bs = bstrides(b.shape, os);
            for (size_t d = 0; d < os.size(); d++) { m.oshape[d] = os[d]; m.as[d] = as[d]; m.bs[d] = bs[d]; }
            hipLaunchKernelGGL(k_binary, grid1(n), dim3(256), 0, 0, f(o), f(a), f(b), op, m, n);
            count();
        }
        if (out_dtype == 7) { XTensor r = as_i64(o); return r; } // integer arithmetic on large device tensors goes through f32 (exact below 2^24)
        return o;
    }
    XTensor unary(int op, const XTensor& x0, float a, float b) {
        if (x0.on_host && x0.dtype == 7 && (op == U_NEG || op == U_ABS || op == U_SIGN || op == U_IDENT || op == U_NOT)) {
            XTensor o = x0;
            for (auto& v : o.hv) v = op == U_NEG ? -v : op == U_ABS ? std::fabs(v) : op == U_SIGN ? (v > 0) - (v < 0) : op == U_NOT ? (v != 0 ? 0 : 1) : v;
            return o;
        }
        if (x0.on_host && (op == U_FLOOR || op == U_CEIL || op == U_SQRT || op == U_NEG || op == U_ABS || op == U_ROUND || op == U_NOT)) {
            XTensor o = x0;
            for (auto& v : o.hv) v = op == U_FLOOR ? std::floor(v) : op == U_CEIL ? std::ceil(v) : op == U_SQRT ? std::sqrt(v) : op == U_NEG ? -v : op == U_ABS ? std::fabs(v)
                                   : op == U_ROUND ? std::nearbyint(v) : (v != 0 ? 0 : 1);
            return o;
        }
        XTensor x = as_f32(x0);
        XTensor o = dev_tensor(op == U_NOT || op == U_ISNAN ? 9 : 1, x.shape);
        const int64_t n = x.numel();
        if (n) { hipLaunchKernelGGL(k_unary, grid1(n), dim3(256), 0, 0, f(o), f(x), op, a, b, n); count(); }
        return o;
    }
    // rows x cols view with the given axes moved last (in order); returns the permuted tensor and fills rows / cols
    XTensor axes_last(const XTensor& x, const std::vector<int64_t>& axes, int64_t& rows, int64_t& cols, std::vector<int64_t>* perm_out = nullptr) {
        const int r = (int)x.shape.size();
        std::vector<bool> red((size_t)r, false);
        for (auto a : axes) red[(size_t)a] = true;
        std::vector<int64_t> perm;
        for (int d = 0; d < r; d++) if (!red[(size_t)d]) perm.push_back(d);
        for (int d = 0; d < r; d++) if (red[(size_t)d]) perm.push_back(d);
        cols = 1; for (auto a : axes) cols *= x.shape[(size_t)a];
        rows = cols ? x.numel() / std::max<int64_t>(cols, 1) : 0;
        if (perm_out) *perm_out = perm;
        bool ident = true;
        for (int d = 0; d < r; d++) if (perm[(size_t)d] != d) ident = false;
        return ident ? x : transpose(x, perm);
    }
    std::vector<int64_t> norm_axes(std::vector<int64_t> axes, int rank) { for (auto& a : axes) { if (a < 0) a += rank; Q3_CHECK(a >= 0 && a < rank, "axis out of range"); } std::sort(axes.begin(), axes.end()); axes.erase(std::unique(axes.begin(), axes.end()), axes.end()); return axes; }
};

// Decodes a TensorProto payload into doubles after checking it against the declared dims: the element count is computed with overflow
// checks and capped, raw_data / the typed repeated field must hold at least that many elements (a truncated or crafted file must become
// an Error, not a read past the mapped file), and element types without a decoder are refused instead of becoming zeros.
static std::vector<double> tensor_values(const OnnxTensor& t, const std::string& what, int* dtype_out) {
    int64_t n = 1;
    for (auto d : t.dims) {
        Q3_CHECK(d >= 0, what + ": negative dimension");
        Q3_CHECK(!__builtin_mul_overflow(n, d, &n) && n <= ((int64_t)1 << 31), what + ": element count overflows / exceeds 2^31");
    }
    auto need_raw = [&](size_t esz) { Q3_CHECK(t.raw_bytes / esz >= (size_t)n, what + ": raw_data holds " + std::to_string(t.raw_bytes) + " bytes, dims need " + std::to_string((size_t)n * esz)); };
    auto need_typed = [&](size_t have) { Q3_CHECK(have >= (size_t)n, what + ": typed data holds " + std::to_string(have) + " elements, dims need " + std::to_string(n)); };
    std::vector<double> v((size_t)n);
    int dtype = 1;
    auto rd = [&](auto* typed) { for (int64_t i = 0; i < n; i++) v[(size_t)i] = (double)typed[i]; };
    switch (t.data_type) {
        case 1: if (t.raw) { need_raw(4); rd(reinterpret_cast<const float*>(t.raw)); } else { need_typed(t.float_data.size()); rd(t.float_data.data()); } break;
        case 7: dtype = 7; if (t.raw) { need_raw(8); rd(reinterpret_cast<const int64_t*>(t.raw)); } else { need_typed(t.int64_data.size()); rd(t.int64_data.data()); } break;
        case 6: dtype = 7; if (t.raw) { need_raw(4); rd(reinterpret_cast<const int32_t*>(t.raw)); } else { need_typed(t.int32_data.size()); rd(t.int32_data.data()); } break;
        case 9: dtype = 9; if (t.raw) { need_raw(1); rd(reinterpret_cast<const uint8_t*>(t.raw)); } else { need_typed(t.int32_data.size()); rd(t.int32_data.data()); } break;
        case 11: Q3_CHECK(t.raw || n == 0, what + ": double tensor without raw_data is not supported"); if (n) { need_raw(8); rd(reinterpret_cast<const double*>(t.raw)); } break;
        case 10: { Q3_CHECK(t.raw || n == 0, what + ": f16 tensor without raw_data"); if (n) { need_raw(2); const uint16_t* h = reinterpret_cast<const uint16_t*>(t.raw); for (int64_t i = 0; i < n; i++) v[(size_t)i] = (double)q3_f16_to_f32(h[i]); } break; }
        default: throw Error(what + ": element type " + std::to_string(t.data_type) + " is not supported");
    }
    if (dtype_out) *dtype_out = dtype;
    return v;
}

OnnxSession::OnnxSession(const std::string& path, int device) : model_(new OnnxModel(path)), impl_(new Impl()), device_(device) {
    Q3_HIP(hipSetDevice(device_));
    impl_->launches = &launches_;
    auto it = model_->opsets.find("");
    if (it != model_->opsets.end()) impl_->opset = it->second;
    else if (model_->opsets.count("ai.onnx")) impl_->opset = model_->opsets.at("ai.onnx");
    for (const auto& t : model_->initializers) {
        Q3_CHECK(!t.external, "initializer " + t.name + " keeps its data in an external file");
        int dtype = 1;
        std::vector<double> v = tensor_values(t, "initializer " + t.name, &dtype);
        const int64_t n = (int64_t)v.size();
        XTensor h = Impl::host_tensor(dtype, t.dims, std::move(v));
        // integer tensors and tiny float tensors (scalars, epsilons) stay on the host; weights go to HBM once
        impl_->consts[t.name] = (dtype == 7 || n <= 8) ? h : impl_->to_device(h);
    }
}
OnnxSession::~OnnxSession() = default;

std::vector<std::string> OnnxSession::unsupported_ops() const {
    std::set<std::string> s;
    for (const auto& n : model_->nodes) if (!onnx_exec_supports(n.op_type)) s.insert(n.op_type);
    return std::vector<std::string>(s.begin(), s.end());
}

// set_input / zeros / fetch / run allocate, copy and synchronise synchronously on the null stream: they take capture_mutex() so that they are safe
// beside a running engine whose scheduler thread may be capturing a frame graph (q3_common.h)
void OnnxSession::set_input(const std::string& name, int dtype, const void* data, const std::vector<int64_t>& shape) {
    std::lock_guard<std::mutex> cap(capture_mutex());
    Q3_HIP(hipSetDevice(device_));
    Q3_CHECK(dtype == 1 || dtype == 7, "inputs are f32 or i64");
    XTensor t = impl_->dev_tensor(dtype, shape);
    if (t.numel()) Q3_HIP(hipMemcpy(t.dev->p, data, (size_t)t.numel() * t.esize(), hipMemcpyHostToDevice));
    impl_->inputs[name] = t;
}
void OnnxSession::bind_input(const std::string& name, const XTensor& t) { impl_->inputs[name] = t; }
XTensor OnnxSession::zeros(int dtype, const std::vector<int64_t>& shape) {
    std::lock_guard<std::mutex> cap(capture_mutex());
    Q3_HIP(hipSetDevice(device_));
    XTensor t = impl_->dev_tensor(dtype, shape);
    if (t.numel()) Q3_HIP(hipMemset(t.dev->p, 0, (size_t)t.numel() * t.esize()));
    return t;
}
const XTensor& OnnxSession::value(const std::string& name) const {
    auto it = impl_->vals.find(name);
    if (it == impl_->vals.end()) throw Error("no value named " + name + " (run() first; only graph outputs and live edges are kept)");
    return it->second;
}
void OnnxSession::fetch(const XTensor& t, void* dst, size_t cap) const {
    const size_t n = (size_t)t.numel();
    Q3_CHECK(cap >= n * t.esize(), "fetch buffer too small");
    if (t.on_host) {
        if (t.dtype == 7) for (size_t i = 0; i < n; i++) reinterpret_cast<int64_t*>(dst)[i] = (int64_t)t.hv[i];
        else for (size_t i = 0; i < n; i++) reinterpret_cast<float*>(dst)[i] = (float)t.hv[i];
        return;
    }
    std::lock_guard<std::mutex> cap_lock(capture_mutex());
    Q3_HIP(hipDeviceSynchronize());
    if (n) Q3_HIP(hipMemcpy(dst, t.dev->p, n * t.esize(), hipMemcpyDeviceToHost));
}

void OnnxSession::run() {
    std::lock_guard<std::mutex> cap(capture_mutex());
    Q3_HIP(hipSetDevice(device_));
    Impl& I = *impl_;
    I.vals.clear();
    for (auto& kv : I.consts) I.vals[kv.first] = kv.second;
    for (const auto& vi : model_->inputs) {
        if (I.consts.count(vi.name)) continue; // initialisers listed as inputs (older exporters)
        auto it = I.inputs.find(vi.name);
        if (it == I.inputs.end()) throw Error("input " + vi.name + " was not set");
        I.vals[vi.name] = it->second;
    }
    // last use of every edge, so buffers return to the pool as soon as possible
    std::map<std::string, size_t> last;
    for (size_t k = 0; k < model_->nodes.size(); k++) for (auto& s : model_->nodes[k].inputs) last[s] = k;
    for (auto& o : model_->outputs) last[o.name] = (size_t)-1;

    for (size_t k = 0; k < model_->nodes.size(); k++) {
        const OnnxNode& nd = model_->nodes[k];
        try {
            auto has = [&](size_t i) { return i < nd.inputs.size() && !nd.inputs[i].empty(); };
            auto in = [&](size_t i) -> const XTensor& {
                Q3_CHECK(has(i), "missing input " + std::to_string(i));
                auto it = I.vals.find(nd.inputs[i]);
                if (it == I.vals.end()) throw Error("input " + nd.inputs[i] + " has no producer");
                return it->second;
            };
            auto ai = [&](const char* n, int64_t d) { auto* a = nd.attr(n); return a ? a->i : d; };
            auto af = [&](const char* n, float d) { auto* a = nd.attr(n); return a ? a->f : d; };
            auto as = [&](const char* n, const char* d) { auto* a = nd.attr(n); return a ? a->s : std::string(d); };
            auto aints = [&](const char* n) { auto* a = nd.attr(n); return a ? a->ints : std::vector<int64_t>(); };
            auto out = [&](size_t i, XTensor t) { if (i < nd.outputs.size() && !nd.outputs[i].empty()) I.vals[nd.outputs[i]] = std::move(t); };
            const std::string& op = nd.op_type;

            if (unary_table().count(op)) {
                int u = unary_table().at(op);
                float a = 0.f, b = 0.f;
                if (op == "Elu") a = af("alpha", 1.0f);
                else if (op == "LeakyRelu") a = af("alpha", 0.01f);
                else if (op == "HardSigmoid") { a = af("alpha", 0.2f); b = af("beta", 0.5f); }
                else if (op == "Selu") { a = af("alpha", 1.67326319217681884765625f); b = af("gamma", 1.05070102214813232421875f); }
                else if (op == "Gelu" && as("approximate", "none") == "tanh") u = U_GELU_TANH;
                else if (op == "Celu") a = af("alpha", 1.0f);
                else if (op == "ThresholdedRelu") a = af("alpha", 1.0f);
                out(0, I.unary(u, in(0), a, b));
            } else if (binary_table().count(op)) {
                int b = binary_table().at(op);
                if (op == "Mod" && ai("fmod", 0)) b = B_FMOD;
                const bool cmp = b == B_EQ || b == B_LT || b == B_GT || b == B_LE || b == B_GE || b == B_AND || b == B_OR || b == B_XOR;
                XTensor acc = in(0);
                for (size_t i = 1; i < nd.inputs.size(); i++) {
                    const XTensor& y = in(i);
                    const int od = cmp ? 9 : (acc.dtype == 7 && y.dtype == 7) ? 7 : 1;
                    acc = I.binary(b, acc, y, od);
                }
                if (op == "Mean" && nd.inputs.size() > 1) acc = I.binary(B_DIV, acc, Impl::host_tensor(1, {}, {(double)nd.inputs.size()}), 1);
                out(0, acc);
            } else if (op == "Identity" || op == "Dropout") {
                out(0, in(0));
            } else if (op == "Shape") {
                const auto& s = in(0).shape;
                int64_t b = ai("start", 0), e = ai("end", (int64_t)s.size());
                if (b < 0) b += (int64_t)s.size();
                if (e < 0) e += (int64_t)s.size();
                b = std::max<int64_t>(0, std::min<int64_t>(b, (int64_t)s.size())); e = std::max<int64_t>(b, std::min<int64_t>(e, (int64_t)s.size()));
                std::vector<double> v; for (int64_t i = b; i < e; i++) v.push_back((double)s[(size_t)i]);
                out(0, Impl::host_tensor(7, {(int64_t)v.size()}, v));
            } else if (op == "Size") {
                out(0, Impl::host_tensor(7, {}, {(double)in(0).numel()}));
            } else if (op == "Constant") {
                if (auto* a = nd.attr("value")) {
                    const OnnxTensor& t = a->t;
                    int dtype = 1;
                    std::vector<double> v = tensor_values(t, "Constant " + nd.name, &dtype);
                    const int64_t n = (int64_t)v.size();
                    XTensor h = Impl::host_tensor(dtype, t.dims, std::move(v));
                    out(0, (dtype == 7 || n <= 4096) ? h : I.to_device(h));
                } else if (auto* a2 = nd.attr("value_float")) out(0, Impl::host_tensor(1, {}, {(double)a2->f}));
                else if (auto* a3 = nd.attr("value_int")) out(0, Impl::host_tensor(7, {}, {(double)a3->i}));
                else if (auto* a4 = nd.attr("value_ints")) { std::vector<double> v(a4->ints.begin(), a4->ints.end()); out(0, Impl::host_tensor(7, {(int64_t)v.size()}, v)); }
                else if (auto* a5 = nd.attr("value_floats")) { std::vector<double> v(a5->floats.begin(), a5->floats.end()); out(0, Impl::host_tensor(1, {(int64_t)v.size()}, v)); }
                else throw Error("Constant without a supported value attribute");
            } else if (op == "ConstantOfShape") {
                const auto shape = I.ints_of(in(0));
                double v = 0; int dtype = 1;
                if (auto* a = nd.attr("value")) {
                    const OnnxTensor& t = a->t;
                    const std::vector<double> tv = tensor_values(t, "ConstantOfShape " + nd.name + " value", &dtype);
                    Q3_CHECK(tv.size() >= 1, "ConstantOfShape " + nd.name + ": empty value tensor");
                    v = tv[0];
                }
                const int64_t n = prod(shape);
                if (n <= 65536) out(0, Impl::host_tensor(dtype, shape, std::vector<double>((size_t)n, v)));
                else { XTensor o = I.dev_tensor(dtype == 7 ? 1 : dtype, shape); hipLaunchKernelGGL(k_fill, grid1(n), dim3(256), 0, 0, I.f(o), (float)v, n); I.count(); out(0, dtype == 7 ? I.as_i64(o) : o); }
            } else if (op == "Range") {
                const XTensor s = I.to_host(in(0)), l = I.to_host(in(1)), d = I.to_host(in(2));
                const double st = s.hv.at(0), li = l.hv.at(0), de = d.hv.at(0);
                const int64_t n = std::max<int64_t>(0, (int64_t)std::ceil((li - st) / de));
                std::vector<double> v((size_t)n);
                for (int64_t i = 0; i < n; i++) v[(size_t)i] = st + (double)i * de;
                out(0, Impl::host_tensor(s.dtype, {n}, v));
            } else if (op == "Cast") {
                const int64_t to = ai("to", 1);
                const XTensor& x = in(0);
                // f64 (11) is carried as f32 (documented precision loss); f16 / bf16 / string targets would need a rounding pass this executor does not have
                Q3_CHECK(to == 1 || to == 11 || to == 7 || to == 6 || to == 2 || to == 3 || to == 5 || to == 12 || to == 13 || to == 9,
                         "Cast to element type " + std::to_string(to) + " is not supported (tensors are f32 / i64 / bool on this executor)");
                const int dt = (to == 7 || to == 6 || to == 2 || to == 3 || to == 5 || to == 12 || to == 13) ? 7 : to == 9 ? 9 : 1;
                if (x.on_host) {
                    XTensor o = x; o.dtype = dt;
                    if (dt == 7) for (auto& v : o.hv) v = std::trunc(v);
                    else if (dt == 9) for (auto& v : o.hv) v = v != 0 ? 1 : 0;
                    else for (auto& v : o.hv) v = (double)(float)v;
                    out(0, o);
                } else if (dt == 7) out(0, I.as_i64(x));
                else if (dt == 9) { XTensor z = Impl::host_tensor(1, {}, {0.0}); XTensor e = I.binary(B_EQ, x, z, 9); out(0, I.unary(U_NOT, e, 0, 0)); }
                else { XTensor o = I.as_f32(x); o.dtype = 1; out(0, o); }
            } else if (op == "Reshape") {
                const XTensor& x = in(0);
                auto shape = I.ints_of(in(1));
                const bool allowzero = ai("allowzero", 0) != 0;
                int64_t known = 1; int neg = -1;
                for (size_t d = 0; d < shape.size(); d++) {
                    if (shape[d] == 0 && !allowzero) { Q3_CHECK(d < x.shape.size(), "Reshape: 0 beyond the input rank"); shape[d] = x.shape[d]; }
                    if (shape[d] == -1) neg = (int)d; else known *= shape[d];
                }
                if (neg >= 0) shape[(size_t)neg] = known ? x.numel() / known : 0;
                out(0, I.reshaped(x, shape));
            } else if (op == "Flatten") {
                const XTensor& x = in(0);
                int64_t a = ai("axis", 1); if (a < 0) a += (int64_t)x.shape.size();
                out(0, I.reshaped(x, {prod(x.shape, 0, (size_t)a), prod(x.shape, (size_t)a)}));
            } else if (op == "Squeeze" || op == "Unsqueeze") {
                const XTensor& x = in(0);
                std::vector<int64_t> axes = has(1) ? I.ints_of(in(1)) : aints("axes");
                std::vector<int64_t> s;
                if (op == "Squeeze") {
                    if (axes.empty()) { for (auto d : x.shape) if (d != 1) s.push_back(d); }
                    else { axes = I.norm_axes(axes, (int)x.shape.size()); for (size_t d = 0; d < x.shape.size(); d++) if (!std::binary_search(axes.begin(), axes.end(), (int64_t)d)) s.push_back(x.shape[d]); }
                } else {
                    const int r = (int)(x.shape.size() + axes.size());
                    axes = I.norm_axes(axes, r);
                    size_t src = 0;
                    for (int d = 0; d < r; d++) s.push_back(std::binary_search(axes.begin(), axes.end(), (int64_t)d) ? 1 : x.shape[src++]);
                }
                out(0, I.reshaped(x, s));
            } else if (op == "Transpose") {
                const XTensor& x = in(0);
                std::vector<int64_t> perm = aints("perm");
                if (perm.empty()) for (int d = (int)x.shape.size() - 1; d >= 0; d--) perm.push_back(d);
                out(0, I.transpose(x, perm));
            } else if (op == "Expand") {
                const XTensor& x = in(0);
                out(0, I.expand(x, bshape(x.shape, I.ints_of(in(1)))));
            } else if (op == "Tile") {
                const XTensor& x = in(0);
                const auto rep = I.ints_of(in(1));
                Q3_CHECK(rep.size() == x.shape.size(), "Tile: repeats has " + std::to_string(rep.size()) + " entries for a rank-" + std::to_string(x.shape.size()) + " tensor");
                std::vector<int64_t> os(x.shape.size());
                for (size_t d = 0; d < os.size(); d++) { Q3_CHECK(rep[d] >= 0 && rep[d] <= (1 << 24), "Tile: repeat count out of range"); os[d] = x.shape[d] * rep[d]; }
                std::vector<int64_t> mod = x.shape; for (auto& m : mod) m = std::max<int64_t>(m, 1);
                out(0, I.nd_copy(x, os, strides_of(x.shape), mod, 0));
            } else if (op == "Slice") {
                const XTensor& x = in(0);
                const int r = (int)x.shape.size();
                std::vector<int64_t> starts, ends, axes, steps;
                if (has(1)) { starts = I.ints_of(in(1)); ends = I.ints_of(in(2)); if (has(3)) axes = I.ints_of(in(3)); if (has(4)) steps = I.ints_of(in(4)); }
                else { starts = aints("starts"); ends = aints("ends"); axes = aints("axes"); }
                if (axes.empty()) for (size_t i = 0; i < starts.size(); i++) axes.push_back((int64_t)i);
                if (steps.empty()) steps.assign(starts.size(), 1);
                std::vector<int64_t> os = x.shape, st = strides_of(x.shape), is = st;
                int64_t off = 0;
                for (size_t i = 0; i < starts.size(); i++) {
                    int64_t a = axes[i]; if (a < 0) a += r;
                    const int64_t dim = x.shape[(size_t)a], step = steps[i];
                    Q3_CHECK(step != 0, "Slice step 0");
                    int64_t s = starts[i], e = ends[i];
                    if (s < 0) s += dim;
                    if (e < 0) e += dim;
                    if (step > 0) { s = std::max<int64_t>(0, std::min(s, dim)); e = std::max<int64_t>(0, std::min(e, dim)); }
                    else { s = std::max<int64_t>(0, std::min(s, dim - 1)); e = std::max<int64_t>(-1, std::min(e, dim - 1)); if (ends[i] < -dim) e = -1; }
                    const int64_t cnt = step > 0 ? std::max<int64_t>(0, (e - s + step - 1) / step) : std::max<int64_t>(0, (s - e + (-step) - 1) / (-step));
                    os[(size_t)a] = cnt; is[(size_t)a] = st[(size_t)a] * step; off += s * st[(size_t)a];
                }
                out(0, I.nd_copy(x, os, is, Impl::nomod(os.size()), off));
            } else if (op == "Split") {
                const XTensor& x = in(0);
                int64_t a = ai("axis", 0); if (a < 0) a += (int64_t)x.shape.size();
                Q3_CHECK(a >= 0 && a < (int64_t)x.shape.size(), "Split: axis out of range");
                std::vector<int64_t> sizes = has(1) ? I.ints_of(in(1)) : aints("split");
                { int64_t tot = 0; for (auto z : sizes) { Q3_CHECK(z >= 0, "Split: negative size"); tot += z; } Q3_CHECK(sizes.empty() || tot == x.shape[(size_t)a], "Split: sizes do not add up to the axis length"); }
                Q3_CHECK(sizes.empty() || sizes.size() <= nd.outputs.size(), "Split: more sizes than outputs");
                if (sizes.empty()) {
                    const int64_t parts = (int64_t)nd.outputs.size(), dim = x.shape[(size_t)a], each = (dim + parts - 1) / parts;
                    for (int64_t i = 0; i < parts; i++) sizes.push_back(std::min(each, dim - i * each));
                }
                const auto st = strides_of(x.shape);
                int64_t at = 0;
                for (size_t i = 0; i < sizes.size(); i++) {
                    std::vector<int64_t> os = x.shape; os[(size_t)a] = sizes[i];
                    out(i, I.nd_copy(x, os, st, Impl::nomod(os.size()), at * st[(size_t)a]));
                    at += sizes[i];
                }
            } else if (op == "Concat") {
                const int r = (int)in(0).shape.size();
                int64_t a = ai("axis", 0); if (a < 0) a += r;
                Q3_CHECK(a >= 0 && a < r, "Concat: axis out of range");
                for (size_t i = 1; i < nd.inputs.size(); i++) {
                    Q3_CHECK((int)in(i).shape.size() == r, "Concat: input " + std::to_string(i) + " (" + nd.inputs[i] + ") has rank " + std::to_string(in(i).shape.size()) + ", input 0 (" + nd.inputs[0] + ") rank " + std::to_string(r));
                    for (int d = 0; d < r; d++) Q3_CHECK(d == a || in(i).shape[(size_t)d] == in(0).shape[(size_t)d], "Concat: inputs differ off the concatenation axis");
                }
                bool all_host = true; int dtype = in(0).dtype; int64_t total = 0;
                for (size_t i = 0; i < nd.inputs.size(); i++) { all_host = all_host && in(i).on_host; total += in(i).shape.at((size_t)a); if (in(i).dtype != 7) dtype = in(i).dtype == 9 && dtype == 9 ? 9 : 1; }
                std::vector<int64_t> os = in(0).shape; os[(size_t)a] = total;
                const int64_t inner = prod(os, (size_t)a + 1), outer = prod(os, 0, (size_t)a);
                if (all_host) {
                    XTensor o = Impl::host_tensor(dtype, os, std::vector<double>((size_t)prod(os)));
                    int64_t at = 0;
                    for (size_t i = 0; i < nd.inputs.size(); i++) {
                        const XTensor& t = in(i); const int64_t part = t.shape[(size_t)a];
                        for (int64_t o_ = 0; o_ < outer; o_++) for (int64_t p = 0; p < part; p++) for (int64_t q = 0; q < inner; q++)
                            o.hv[(size_t)((o_ * total + at + p) * inner + q)] = t.hv[(size_t)((o_ * part + p) * inner + q)];
                        at += part;
                    }
                    out(0, o);
                } else {
                    XTensor o = I.dev_tensor(dtype, os);
                    int64_t at = 0;
                    for (size_t i = 0; i < nd.inputs.size(); i++) {
                        XTensor t = dtype == 7 ? I.as_i64(in(i)) : I.as_f32(in(i));
                        const int64_t part = t.shape[(size_t)a], n = t.numel();
                        if (n) {
                            if (dtype == 7) hipLaunchKernelGGL(k_concat<int64_t>, grid1(n), dim3(256), 0, 0, I.i64(o), I.i64(t), part, inner, total, at, n);
                            else hipLaunchKernelGGL(k_concat<float>, grid1(n), dim3(256), 0, 0, I.f(o), I.f(t), part, inner, total, at, n);
                            I.count();
                        }
                        at += part;
                    }
                    out(0, o);
                }
            } else if (op == "Gather") {
                const XTensor& x = in(0); const XTensor& idx = in(1);
                int64_t a = ai("axis", 0); if (a < 0) a += (int64_t)x.shape.size();
                Q3_CHECK(a >= 0 && a < (int64_t)x.shape.size(), "Gather: axis out of range");
                const int64_t dim = x.shape.at((size_t)a), inner = prod(x.shape, (size_t)a + 1), outer = prod(x.shape, 0, (size_t)a), nidx = idx.numel();
                std::vector<int64_t> os(x.shape.begin(), x.shape.begin() + a);
                os.insert(os.end(), idx.shape.begin(), idx.shape.end());
                os.insert(os.end(), x.shape.begin() + a + 1, x.shape.end());
                if (x.on_host && idx.on_host) {
                    XTensor o = Impl::host_tensor(x.dtype, os, std::vector<double>((size_t)prod(os)));
                    for (int64_t o_ = 0; o_ < outer; o_++) for (int64_t j = 0; j < nidx; j++) {
                        int64_t k2 = (int64_t)idx.hv[(size_t)j]; if (k2 < 0) k2 += dim;
                        Q3_CHECK(k2 >= 0 && k2 < dim, "Gather index out of range");
                        for (int64_t q = 0; q < inner; q++) o.hv[(size_t)((o_ * nidx + j) * inner + q)] = x.hv[(size_t)((o_ * dim + k2) * inner + q)];
                    }
                    out(0, o);
                } else {
                    XTensor xd = I.to_device(x), id = I.as_i64(idx);
                    XTensor o = I.dev_tensor(xd.dtype, os);
                    const int64_t n = o.numel();
                    if (n) {
                        if (xd.dtype == 7) hipLaunchKernelGGL(k_gather<int64_t>, grid1(n), dim3(256), 0, 0, I.i64(o), I.i64(xd), I.i64(id), nidx, dim, inner, n);
                        else hipLaunchKernelGGL(k_gather<float>, grid1(n), dim3(256), 0, 0, I.f(o), I.f(xd), I.i64(id), nidx, dim, inner, n);
                        I.count();
                    }
                    out(0, o);
                }
            } else if (op == "Where") {
                const XTensor& c = in(0); const XTensor& a = in(1); const XTensor& b = in(2);
                const auto os = bshape(bshape(c.shape, a.shape), b.shape);
                const int64_t n = prod(os);
                const int dtype = (a.dtype == 7 && b.dtype == 7) ? 7 : (a.dtype == 9 && b.dtype == 9) ? 9 : 1;
                if (c.on_host && a.on_host && b.on_host) {
                    XTensor o = Impl::host_tensor(dtype, os, std::vector<double>((size_t)n));
                    const auto cs = bstrides(c.shape, os), as_ = bstrides(a.shape, os), bs = bstrides(b.shape, os);
                    for (int64_t i = 0; i < n; i++) {
                        int64_t r = i, co = 0, ao = 0, bo = 0;
                        for (int d = (int)os.size() - 1; d >= 0; d--) { const int64_t q = r / os[d], x = r - q * os[d]; co += x * cs[d]; ao += x * as_[d]; bo += x * bs[d]; r = q; }
                        o.hv[(size_t)i] = c.hv[(size_t)co] != 0 ? a.hv[(size_t)ao] : b.hv[(size_t)bo];
                    }
                    out(0, o);
                } else {
                    XTensor cd = I.as_f32(c), ad = dtype == 7 ? I.as_i64(a) : I.as_f32(a), bd = dtype == 7 ? I.as_i64(b) : I.as_f32(b);
                    XTensor o = I.dev_tensor(dtype, os);
                    if (n) {
                        NdMap2 m{}; m.rank = (int)os.size();
                        const auto cs = bstrides(cd.shape, os), as_ = bstrides(ad.shape, os), bs = bstrides(bd.shape, os);
                        for (size_t d = 0; d < os.size(); d++) { m.oshape[d] = os[d]; m.as[d] = as_[d]; m.bs[d] = bs[d]; m.cs[d] = cs[d]; }
                        if (dtype == 7) hipLaunchKernelGGL(k_where<int64_t>, grid1(n), dim3(256), 0, 0, I.i64(o), I.f(cd), I.i64(ad), I.i64(bd), m, n);
                        else hipLaunchKernelGGL(k_where<float>, grid1(n), dim3(256), 0, 0, I.f(o), I.f(cd), I.f(ad), I.f(bd), m, n);
                        I.count();
                    }
                    out(0, o);
                }
            } else if (op == "Clip") {
                float lo = -std::numeric_limits<float>::infinity(), hi = std::numeric_limits<float>::infinity();
                if (nd.attr("min")) lo = af("min", lo);
                if (nd.attr("max")) hi = af("max", hi);
                if (has(1)) lo = (float)I.to_host(in(1)).hv.at(0);
                if (has(2)) hi = (float)I.to_host(in(2)).hv.at(0);
                out(0, I.unary(U_CLIP, in(0), lo, hi));
            } else if (reduce_table().count(op)) {
                const int r = reduce_table().at(op);
                const XTensor& x0 = in(0);
                const int rank = (int)x0.shape.size();
                const bool arg = r == R_ARGMAX || r == R_ARGMIN;
                std::vector<int64_t> axes;
                if (arg) axes = {ai("axis", 0)};
                else if (has(1)) axes = I.ints_of(in(1));
                else axes = aints("axes");
                const bool keep = ai("keepdims", 1) != 0;
                if (axes.empty() && !arg) {
                    if (ai("noop_with_empty_axes", 0)) { out(0, x0); continue; }
                    for (int d = 0; d < rank; d++) axes.push_back(d);
                }
                axes = I.norm_axes(axes, rank);
                std::vector<int64_t> os;
                for (int d = 0; d < rank; d++) { const bool red = std::binary_search(axes.begin(), axes.end(), (int64_t)d); if (!red) os.push_back(x0.shape[(size_t)d]); else if (keep) os.push_back(1); }
                if (x0.on_host && (r == R_SUM || r == R_PROD || r == R_MAX || r == R_MIN || r == R_MEAN)) { // shape arithmetic (ReduceProd of a shape, ...)
                    int64_t rows, cols;
                    XTensor p = I.axes_last(x0, axes, rows, cols);
                    XTensor o = Impl::host_tensor(x0.dtype, os, std::vector<double>((size_t)rows));
                    for (int64_t i = 0; i < rows; i++) {
                        double acc = r == R_PROD ? 1 : r == R_MAX ? -INFINITY : r == R_MIN ? INFINITY : 0;
                        for (int64_t c = 0; c < cols; c++) { const double v = p.hv[(size_t)(i * cols + c)]; acc = r == R_PROD ? acc * v : r == R_MAX ? std::max(acc, v) : r == R_MIN ? std::min(acc, v) : acc + v; }
                        o.hv[(size_t)i] = r == R_MEAN ? acc / (double)cols : acc;
                    }
                    out(0, o);
                } else {
                    int64_t rows, cols;
                    XTensor p = I.axes_last(I.as_f32(x0), axes, rows, cols);
                    XTensor o = I.dev_tensor(arg ? 7 : 1, os);
                    if (rows) {
                        Q3_CHECK(cols > 0, "reduction over an empty axis");
                        hipLaunchKernelGGL(k_reduce_rows, dim3((unsigned)rows), dim3(256), 0, 0, arg ? nullptr : I.f(o), arg ? I.i64(o) : nullptr, I.f(p), r, cols, (int)ai("select_last_index", 0));
                        I.count();
                    }
                    out(0, o);
                }
            } else if (op == "Softmax" || op == "LogSoftmax") {
                const XTensor x = I.as_f32(in(0));
                const int rank = (int)x.shape.size();
                int64_t a = ai("axis", I.opset >= 13 ? -1 : 1); if (a < 0) a += rank;
                XTensor o = I.dev_tensor(1, x.shape);
                if (I.opset < 13 || a == rank - 1) { // rows = everything before the axis, cols = the rest (opset < 13 flattens; for the last axis both readings agree)
                    const int64_t cols = prod(x.shape, (size_t)a), rows = prod(x.shape, 0, (size_t)a);
                    if (rows && cols) { hipLaunchKernelGGL(k_softmax_rows, dim3((unsigned)rows), dim3(256), 0, 0, I.f(o), I.f(x), cols, op == "LogSoftmax" ? 1 : 0); I.count(); }
                    out(0, o);
                } else {
                    int64_t rows, cols; std::vector<int64_t> perm;
                    XTensor p = I.axes_last(x, {a}, rows, cols, &perm);
                    XTensor q = I.dev_tensor(1, p.shape);
                    if (rows && cols) { hipLaunchKernelGGL(k_softmax_rows, dim3((unsigned)rows), dim3(256), 0, 0, I.f(q), I.f(p), cols, op == "LogSoftmax" ? 1 : 0); I.count(); }
                    std::vector<int64_t> inv(perm.size());
                    for (size_t d = 0; d < perm.size(); d++) inv[(size_t)perm[d]] = (int64_t)d;
                    out(0, I.transpose(q, inv));
                }
            } else if (op == "LayerNormalization") {
                const XTensor x = I.as_f32(in(0));
                int64_t a = ai("axis", -1); if (a < 0) a += (int64_t)x.shape.size();
                Q3_CHECK(a >= 0 && a < (int64_t)x.shape.size(), "LayerNormalization: axis out of range");
                const int64_t cols = prod(x.shape, (size_t)a), rows = prod(x.shape, 0, (size_t)a);
                XTensor g = has(1) ? I.as_f32(in(1)) : XTensor(), b = has(2) ? I.as_f32(in(2)) : XTensor();
                Q3_CHECK((!has(1) || g.numel() == cols) && (!has(2) || b.numel() == cols), "LayerNormalization: scale / bias length does not match the normalised extent");
                XTensor o = I.dev_tensor(1, x.shape);
                if (rows && cols) { hipLaunchKernelGGL(k_norm_rows, dim3((unsigned)rows), dim3(256), 0, 0, I.f(o), I.f(x), has(1) ? I.f(g) : nullptr, has(2) ? I.f(b) : nullptr, cols, af("epsilon", 1e-5f), 0, (int64_t)1); I.count(); }
                out(0, o);
            } else if (op == "InstanceNormalization") {
                const XTensor x = I.as_f32(in(0)), g = I.as_f32(in(1)), b = I.as_f32(in(2));
                const int64_t C = x.shape.at(1), rows = x.shape.at(0) * C, cols = prod(x.shape, 2);
                Q3_CHECK(g.numel() == C && b.numel() == C, "InstanceNormalization: scale / bias length does not match the channels");
                XTensor o = I.dev_tensor(1, x.shape);
                if (rows && cols) { hipLaunchKernelGGL(k_norm_rows, dim3((unsigned)rows), dim3(256), 0, 0, I.f(o), I.f(x), I.f(g), I.f(b), cols, af("epsilon", 1e-5f), 1, C); I.count(); }
                out(0, o);
            } else if (op == "BatchNormalization") {
                const XTensor x = I.as_f32(in(0)), sc = I.as_f32(in(1)), bi = I.as_f32(in(2)), mean = I.as_f32(in(3)), var = I.as_f32(in(4));
                const int64_t C = x.shape.at(1), inner = prod(x.shape, 2), n = x.numel();
                Q3_CHECK(sc.numel() == C && bi.numel() == C && mean.numel() == C && var.numel() == C, "BatchNormalization: parameter lengths do not match the channels");
                XTensor o = I.dev_tensor(1, x.shape);
                if (n) { hipLaunchKernelGGL(k_batchnorm, grid1(n), dim3(256), 0, 0, I.f(o), I.f(x), I.f(sc), I.f(bi), I.f(mean), I.f(var), af("epsilon", 1e-5f), C, inner, n); I.count(); }
                out(0, o);
            } else if (op == "GlobalAveragePool" || op == "GlobalMaxPool") {
                const XTensor x = I.as_f32(in(0));
                const int64_t rows = x.shape.at(0) * x.shape.at(1), cols = prod(x.shape, 2);
                std::vector<int64_t> os = x.shape; for (size_t d = 2; d < os.size(); d++) os[d] = 1;
                XTensor o = I.dev_tensor(1, os);
                if (rows) { hipLaunchKernelGGL(k_reduce_rows, dim3((unsigned)rows), dim3(256), 0, 0, I.f(o), (int64_t*)nullptr, I.f(x), op == "GlobalMaxPool" ? R_MAX : R_MEAN, cols, 0); I.count(); }
                out(0, o);
            } else if (op == "CumSum") {
                const XTensor x = I.as_f32(in(0));
                int64_t a = I.ints_of(in(1)).at(0); if (a < 0) a += (int64_t)x.shape.size();
                int64_t rows, cols; std::vector<int64_t> perm;
                XTensor p = I.axes_last(x, {a}, rows, cols, &perm);
                XTensor q = I.dev_tensor(1, p.shape);
                if (rows && cols) { hipLaunchKernelGGL(k_cumsum_rows, grid1(rows), dim3(256), 0, 0, I.f(q), I.f(p), rows, cols, (int)ai("exclusive", 0), (int)ai("reverse", 0)); I.count(); }
                std::vector<int64_t> inv(perm.size());
                for (size_t d = 0; d < perm.size(); d++) inv[(size_t)perm[d]] = (int64_t)d;
                out(0, I.transpose(q, inv));
            } else if (op == "MatMul" || op == "Gemm") {
                XTensor A = I.as_f32(in(0)), B = I.as_f32(in(1));
                MmArgs g{};
                g.alpha = 1.f; g.beta = 1.f;
                std::vector<int64_t> os;
                int64_t batch = 1;
                XTensor bias;
                if (op == "Gemm") {
                    Q3_CHECK(A.shape.size() == 2 && B.shape.size() == 2, "Gemm takes matrices");
                    const bool ta = ai("transA", 0) != 0, tb = ai("transB", 0) != 0;
                    g.M = ta ? A.shape[1] : A.shape[0]; g.K = ta ? A.shape[0] : A.shape[1]; g.N = tb ? B.shape[0] : B.shape[1];
                    Q3_CHECK((tb ? B.shape[1] : B.shape[0]) == g.K, "Gemm inner dimensions differ");
                    g.a_m = ta ? 1 : A.shape[1]; g.a_k = ta ? A.shape[1] : 1; g.b_k = tb ? 1 : B.shape[1]; g.b_n = tb ? B.shape[1] : 1;
                    g.alpha = af("alpha", 1.f); g.beta = af("beta", 1.f);
                    if (has(2)) {
                        bias = I.as_f32(in(2));
                        const auto bs = bstrides(bias.shape, {g.M, g.N});
                        g.bias = I.f(bias); g.bias_m = bs[0]; g.bias_n = bs[1];
                    }
                    os = {g.M, g.N};
                } else {
                    std::vector<int64_t> as_ = A.shape, bs_ = B.shape;
                    const bool va = as_.size() == 1, vb = bs_.size() == 1;
                    if (va) as_.insert(as_.begin(), 1);
                    if (vb) bs_.push_back(1);
                    g.M = as_[as_.size() - 2]; g.K = as_.back(); g.N = bs_.back();
                    Q3_CHECK(bs_[bs_.size() - 2] == g.K, "MatMul inner dimensions differ");
                    std::vector<int64_t> ab(as_.begin(), as_.end() - 2), bb(bs_.begin(), bs_.end() - 2);
                    const auto bd = bshape(ab, bb);
                    batch = prod(bd);
                    // operands whose batch dims are smaller than the broadcast batch are expanded first (rare: attention uses equal batches)
                    auto full = [&](XTensor& T, std::vector<int64_t>& s, const std::vector<int64_t>& own) {
                        if (own == bd) return;
                        if (prod(own) == 1) return; // pure broadcast: batch stride 0
                        std::vector<int64_t> tgt = bd; tgt.push_back(s[s.size() - 2]); tgt.push_back(s.back());
                        T = I.expand(I.reshaped(T, s), tgt); s = tgt;
                    };
                    full(A, as_, ab); full(B, bs_, bb);
                    g.a_b = (as_.size() > 2 && prod(as_, 0, as_.size() - 2) > 1) ? g.M * g.K : 0;
                    g.b_b = (bs_.size() > 2 && prod(bs_, 0, bs_.size() - 2) > 1) ? g.K * g.N : 0;
                    g.a_m = g.K; g.a_k = 1; g.b_k = g.N; g.b_n = 1;
                    os = bd;
                    if (!va) os.push_back(g.M);
                    if (!vb) os.push_back(g.N);
                }
                XTensor o = I.dev_tensor(1, os);
                if (o.numel()) {
                    if (mm_fast_ok(g.M, g.N, g.K, batch)) {
                        MmFast f{};
                        f.M = (int)g.M; f.N = (int)g.N; f.K = (int)g.K; f.a_b = g.a_b; f.a_m = g.a_m; f.a_k = g.a_k; f.b_b = g.b_b; f.b_k = g.b_k; f.b_n = g.b_n;
                        f.bias = g.bias; f.bias_m = g.bias_m; f.bias_n = g.bias_n; f.alpha = g.alpha; f.beta = g.beta;
                        launch_mm_fast(I.f(o), I.f(A), I.f(B), f, batch);
                    } else
                    hipLaunchKernelGGL(k_matmul, dim3((unsigned)((g.N + 15) / 16), (unsigned)((g.M + 15) / 16), (unsigned)batch), dim3(256), 0, 0, I.f(o), I.f(A), I.f(B), g);
                    I.count();
                }
                out(0, o);
            } else if (op == "Conv" || op == "ConvTranspose") {
                XTensor x = I.as_f32(in(0)), w = I.as_f32(in(1)), b = has(2) ? I.as_f32(in(2)) : XTensor();
                const int sp = (int)x.shape.size() - 2;
                Q3_CHECK(sp == 1 || sp == 2, "convolutions over 1 or 2 spatial dimensions");
                Q3_CHECK(w.shape.size() == x.shape.size(), op + ": weight rank does not match the input rank");
                auto two = [&](std::vector<int64_t> v, int64_t fill) { if (v.empty()) v.assign((size_t)sp, fill); if (sp == 1) v.insert(v.begin(), fill == 0 ? 0 : 1); return v; };
                std::vector<int64_t> strides = two(aints("strides"), 1), dil = two(aints("dilations"), 1), pads = aints("pads"), ks(w.shape.begin() + 2, w.shape.end());
                if (sp == 1) ks.insert(ks.begin(), 1);
                ConvArgs g{};
                g.N = x.shape[0]; g.C = x.shape[1]; g.H = sp == 1 ? 1 : x.shape[2]; g.W = x.shape.back();
                g.kh = ks[0]; g.kw = ks[1]; g.sh = strides[0]; g.sw = strides[1]; g.dh = dil[0]; g.dw = dil[1]; g.groups = ai("group", 1);
                // operand shapes against what k_conv2d / k_convtr2d index (they divide by groups and strides and trust the weight / bias extents)
                Q3_CHECK(strides.size() == 2 && dil.size() == 2 && (pads.empty() || pads.size() == (size_t)(2 * sp)), op + ": strides / dilations / pads do not match the spatial rank");
                Q3_CHECK(g.sh > 0 && g.sw > 0 && g.dh > 0 && g.dw > 0 && g.kh > 0 && g.kw > 0, op + ": strides, dilations and kernel extents must be positive");
                Q3_CHECK(g.groups > 0 && g.C % g.groups == 0, op + ": group does not divide the input channels");
                if (op == "Conv") Q3_CHECK(w.shape[1] * g.groups == g.C && w.shape[0] % g.groups == 0, "Conv: weight shape [M, C/group, ...] does not match the input channels");
                else Q3_CHECK(w.shape[0] == g.C, "ConvTranspose: weight shape [C, M/group, ...] does not match the input channels");
                Q3_CHECK(!has(2) || b.numel() == (op == "Conv" ? w.shape[0] : w.shape[1] * g.groups), op + ": bias length does not match the output channels");
                for (auto pv : pads) Q3_CHECK(pv >= 0, op + ": negative padding");
                std::vector<int64_t> pb(2, 0), pe(2, 0);
                if (!pads.empty()) { if (sp == 1) { pb[1] = pads[0]; pe[1] = pads[1]; } else { pb[0] = pads[0]; pb[1] = pads[1]; pe[0] = pads[2]; pe[1] = pads[3]; } }
                const std::string ap = as("auto_pad", "NOTSET");
                const int64_t inH[2] = {g.H, g.W};
                int64_t O[2];
                if (op == "Conv") {
                    g.M = w.shape[0];
                    for (int d = 0; d < 2; d++) {
                        const int64_t eff = (ks[(size_t)d] - 1) * dil[(size_t)d] + 1;
                        if (ap == "SAME_UPPER" || ap == "SAME_LOWER") {
                            O[d] = (inH[d] + strides[(size_t)d] - 1) / strides[(size_t)d];
                            const int64_t tot = std::max<int64_t>(0, (O[d] - 1) * strides[(size_t)d] + eff - inH[d]);
                            pb[(size_t)d] = ap == "SAME_UPPER" ? tot / 2 : tot - tot / 2; pe[(size_t)d] = tot - pb[(size_t)d];
                        } else {
                            if (ap == "VALID") { pb[(size_t)d] = pe[(size_t)d] = 0; }
                            O[d] = (inH[d] + pb[(size_t)d] + pe[(size_t)d] - eff) / strides[(size_t)d] + 1;
                        }
                    }
                } else {
                    g.M = w.shape[1] * g.groups;
                    std::vector<int64_t> opad = aints("output_padding"), oshape = aints("output_shape");
                    if (opad.empty()) opad.assign((size_t)sp, 0);
                    if (sp == 1) { opad.insert(opad.begin(), 0); if (!oshape.empty()) oshape.insert(oshape.begin(), 1); }
                    for (int d = 0; d < 2; d++) {
                        const int64_t eff = (ks[(size_t)d] - 1) * dil[(size_t)d] + 1;
                        if (!oshape.empty()) {
                            O[d] = oshape[(size_t)d];
                            const int64_t tot = std::max<int64_t>(0, (inH[d] - 1) * strides[(size_t)d] + opad[(size_t)d] + eff - O[d]);
                            pb[(size_t)d] = ap == "SAME_UPPER" ? tot / 2 : tot - tot / 2; pe[(size_t)d] = tot - pb[(size_t)d];
                        } else if (ap == "SAME_UPPER" || ap == "SAME_LOWER") {
                            O[d] = inH[d] * strides[(size_t)d];
                            const int64_t tot = std::max<int64_t>(0, (inH[d] - 1) * strides[(size_t)d] + opad[(size_t)d] + eff - O[d]);
                            pb[(size_t)d] = ap == "SAME_UPPER" ? tot / 2 : tot - tot / 2; pe[(size_t)d] = tot - pb[(size_t)d];
                        } else O[d] = (inH[d] - 1) * strides[(size_t)d] - pb[(size_t)d] - pe[(size_t)d] + eff + opad[(size_t)d];
                    }
                }
                g.ph = pb[0]; g.pw = pb[1]; g.OH = O[0]; g.OW = O[1];
                Q3_CHECK(g.OH >= 0 && g.OW >= 0, "negative convolution output size");
                std::vector<int64_t> os = {g.N, g.M};
                if (sp == 2) os.push_back(g.OH);
                os.push_back(g.OW);
                XTensor o = I.dev_tensor(1, os);
                const int64_t n = o.numel();
                if (n) {
                    const bool one_d = sp == 1 && g.groups == 1 && g.W < (1 << 30) && g.OW < (1 << 30) && g.sw < (1 << 20) && g.pw < (1 << 30) && g.dw < (1 << 20) && g.kw < (1 << 20);
                    if (op == "Conv" && one_d && mm_fast_ok(g.M, g.OW, g.C * g.kw, g.N)) {
                        // implicit GEMM: W[M][C * kw] x im2col(x)[C * kw][OW] per batch element; the NCW output is the product's row-major layout
                        MmFast f{};
                        f.M = (int)g.M; f.N = (int)g.OW; f.K = (int)(g.C * g.kw); f.a_b = 0; f.a_m = g.C * g.kw; f.a_k = 1; f.b_b = g.C * g.W;
                        f.conv = 1; f.kw = (int)g.kw; f.cs = (int)g.sw; f.cp = (int)g.pw; f.cd = (int)g.dw; f.W = (int)g.W;
                        f.bias = has(2) ? I.f(b) : nullptr; f.bias_m = 1; f.bias_n = 0; f.alpha = 1.f; f.beta = 1.f;
                        launch_mm_fast(I.f(o), I.f(w), I.f(x), f, g.N);
                    } else if (op == "ConvTranspose" && one_d && mm_fast_ok(g.M * g.kw, g.W, g.C, g.N)) {
                        // Y[b][m * kw + k][i] = sum_c w[c][m][k] x[b][c][i] (the weight tensor read as its own transpose), then the taps are overlap-added
                        XTensor y = I.dev_tensor(1, {g.N, g.M * g.kw, g.W});
                        MmFast f{};
                        f.M = (int)(g.M * g.kw); f.N = (int)g.W; f.K = (int)g.C; f.a_b = 0; f.a_m = 1; f.a_k = g.M * g.kw; f.b_b = g.C * g.W; f.b_k = g.W; f.b_n = 1;
                        f.alpha = 1.f; f.beta = 0.f;
                        launch_mm_fast(I.f(y), I.f(w), I.f(x), f, g.N);
                        I.count();
                        hipLaunchKernelGGL(k_col2im1d, grid1(n), dim3(256), 0, 0, I.f(o), I.f(y), has(2) ? I.f(b) : nullptr, g.M, g.OW, g.W, (int)g.kw, (int)g.sw, (int)g.pw, (int)g.dw, n);
                    } else if (op == "Conv") hipLaunchKernelGGL(k_conv2d, grid1(n), dim3(256), 0, 0, I.f(o), I.f(x), I.f(w), has(2) ? I.f(b) : nullptr, g, n);
                    else hipLaunchKernelGGL(k_convtr2d, grid1(n), dim3(256), 0, 0, I.f(o), I.f(x), I.f(w), has(2) ? I.f(b) : nullptr, g, n);
                    I.count();
                }
                out(0, o);
            } else if (op == "Pad") {
                const XTensor x = I.as_f32(in(0));
                const int rank = (int)x.shape.size();
                std::vector<int64_t> pads = has(1) ? I.ints_of(in(1)) : aints("pads");
                float value = af("value", 0.f);
                if (has(2)) value = (float)I.to_host(in(2)).hv.at(0);
                std::vector<int64_t> axes;
                if (has(3)) axes = I.ints_of(in(3)); else for (int d = 0; d < rank; d++) axes.push_back(d);
                Q3_CHECK(pads.size() == 2 * axes.size(), "Pad: pads do not match the axes");
                const std::string mode = as("mode", "constant");
                PadArgs p{}; p.rank = rank; p.mode = mode == "constant" ? 0 : mode == "reflect" ? 1 : mode == "edge" ? 2 : -1; p.value = value;
                Q3_CHECK(p.mode >= 0, "Pad mode " + mode);
                std::vector<int64_t> os = x.shape;
                for (int d = 0; d < rank; d++) { p.ishape[d] = x.shape[(size_t)d]; p.begin[d] = 0; }
                for (size_t i = 0; i < axes.size(); i++) { int64_t a = axes[i]; if (a < 0) a += rank; p.begin[a] = pads[i]; os[(size_t)a] += pads[i] + pads[i + axes.size()]; }
                for (int d = 0; d < rank; d++) p.oshape[d] = os[(size_t)d];
                XTensor o = I.dev_tensor(1, os);
                const int64_t n = o.numel();
                if (n) { hipLaunchKernelGGL(k_pad, grid1(n), dim3(256), 0, 0, I.f(o), I.f(x), p, n); I.count(); }
                out(0, o);
            } else if (op == "Trilu") {
                const XTensor x = I.as_f32(in(0));
                Q3_CHECK(x.shape.size() >= 2, "Trilu needs a matrix");
                const int64_t k2 = has(1) ? I.ints_of(in(1)).at(0) : 0, rows = x.shape[x.shape.size() - 2], cols = x.shape.back(), n = x.numel();
                XTensor o = I.dev_tensor(in(0).dtype == 9 ? 9 : 1, x.shape);
                if (n) { hipLaunchKernelGGL(k_trilu, grid1(n), dim3(256), 0, 0, I.f(o), I.f(x), rows, cols, k2, (int)ai("upper", 1), n); I.count(); }
                out(0, in(0).dtype == 7 ? I.as_i64(o) : o);
            } else if (op == "GatherElements") {
                const XTensor x = I.to_device(in(0)), idx = I.as_i64(in(1));
                int64_t a = ai("axis", 0); if (a < 0) a += (int64_t)x.shape.size();
                Q3_CHECK(idx.shape.size() == x.shape.size(), "GatherElements: rank mismatch");
                for (size_t d = 0; d < x.shape.size(); d++) Q3_CHECK((int64_t)d == a || idx.shape[d] == x.shape[d], "GatherElements: indices must match the data outside the axis");
                const int64_t inner = prod(x.shape, (size_t)a + 1), nj = idx.shape[(size_t)a], n = idx.numel();
                XTensor o = I.dev_tensor(x.dtype, idx.shape);
                if (n) {
                    if (x.dtype == 7) hipLaunchKernelGGL(k_gather_elems<int64_t>, grid1(n), dim3(256), 0, 0, I.i64(o), I.i64(x), I.i64(idx), nj, x.shape[(size_t)a], inner, n);
                    else hipLaunchKernelGGL(k_gather_elems<float>, grid1(n), dim3(256), 0, 0, I.f(o), I.f(x), I.i64(idx), nj, x.shape[(size_t)a], inner, n);
                    I.count();
                }
                out(0, o);
            } else if (op == "ScatterND") {
                const XTensor x = I.as_f32(in(0)), upd = I.as_f32(in(2));
                const XTensor ih = I.to_host(in(1)); // index tuples are small (cache positions): evaluated on the host into slice offsets
                Q3_CHECK(as("reduction", "none") == "none", "ScatterND with a reduction");
                const int64_t kdim = ih.shape.back(), nup = kdim ? ih.numel() / kdim : 0;
                Q3_CHECK(kdim >= 1 && kdim <= (int64_t)x.shape.size(), "ScatterND index depth");
                const int64_t slice = prod(x.shape, (size_t)kdim);
                const auto st = strides_of(x.shape);
                std::vector<double> offs((size_t)nup);
                for (int64_t u = 0; u < nup; u++) {
                    int64_t off = 0;
                    for (int64_t d = 0; d < kdim; d++) { int64_t v = (int64_t)ih.hv[(size_t)(u * kdim + d)]; if (v < 0) v += x.shape[(size_t)d]; Q3_CHECK(v >= 0 && v < x.shape[(size_t)d], "ScatterND index out of range"); off += v * st[(size_t)d]; }
                    offs[(size_t)u] = (double)(off / std::max<int64_t>(slice, 1));
                }
                XTensor o = I.dev_tensor(1, x.shape);
                if (x.numel()) Q3_HIP(hipMemcpyAsync(o.dev->p, x.dev->p, (size_t)x.numel() * 4, hipMemcpyDeviceToDevice, 0));
                const int64_t n = nup * slice;
                if (n) {
                    XTensor od = I.to_device(Impl::host_tensor(7, {nup}, offs));
                    hipLaunchKernelGGL(k_scatter_rows, grid1(n), dim3(256), 0, 0, I.f(o), I.f(upd), I.i64(od), slice, n);
                    I.count();
                }
                out(0, o);
            } else if (op == "Resize") {
                const XTensor x = I.as_f32(in(0));
                Q3_CHECK(as("mode", "nearest") == "nearest", "Resize: only nearest is implemented");
                const int rank = (int)x.shape.size();
                ResizeArgs a{}; a.rank = rank;
                std::vector<int64_t> os = x.shape;
                Q3_CHECK(rank >= 1 && rank <= 8, "Resize: rank out of range");
                const bool old_layout = I.opset < 11; // opset 10: inputs (X, scales); opset >= 11: (X, roi, scales, sizes)
                if (!old_layout && has(3)) { os = I.ints_of(in(3)); Q3_CHECK((int)os.size() == rank, "Resize sizes rank"); for (int d = 0; d < rank; d++) { Q3_CHECK(x.shape[(size_t)d] > 0 && os[(size_t)d] >= 0, "Resize: empty input extent"); a.scale[d] = (float)os[(size_t)d] / (float)x.shape[(size_t)d]; } }
                else { const int si = old_layout ? 1 : 2; Q3_CHECK(has(si), "Resize needs scales or sizes"); const XTensor sc = I.to_host(in(si)); Q3_CHECK((int)sc.hv.size() == rank, "Resize scales rank");
                       for (int d = 0; d < rank; d++) { Q3_CHECK(sc.hv[(size_t)d] > 0, "Resize: scales must be positive"); a.scale[d] = (float)sc.hv[(size_t)d]; os[(size_t)d] = (int64_t)std::floor((double)x.shape[(size_t)d] * sc.hv[(size_t)d]); } }
                { // the kernel samples x_in = floor(x_out / scale) (asymmetric + floor).  ONNX defaults to half_pixel + round_prefer_floor, which picks the same
                  // sample exactly when every scale is a whole number >= 1; any other combination is refused instead of silently resampling differently
                    const std::string ctm = as("coordinate_transformation_mode", old_layout ? "asymmetric" : "half_pixel"), nm = as("nearest_mode", old_layout ? "floor" : "round_prefer_floor");
                    bool whole = true;
                    for (int d = 0; d < rank; d++) whole = whole && a.scale[d] >= 1.0f && a.scale[d] == std::floor(a.scale[d]);
                    const bool native = ctm == "asymmetric" && nm == "floor";
                    const bool equivalent = whole && (ctm == "half_pixel" || ctm == "pytorch_half_pixel" || ctm == "asymmetric") && (nm == "round_prefer_floor" || nm == "floor");
                    Q3_CHECK(native || equivalent, "Resize: coordinate_transformation_mode=" + ctm + " nearest_mode=" + nm + " is only implemented for whole-number up-scaling");
                }
                for (int d = 0; d < rank; d++) { a.oshape[d] = os[(size_t)d]; a.ishape[d] = x.shape[(size_t)d]; }
                XTensor o = I.dev_tensor(1, os);
                const int64_t n = o.numel();
                if (n) { hipLaunchKernelGGL(k_resize_nearest, grid1(n), dim3(256), 0, 0, I.f(o), I.f(x), a, n); I.count(); }
                out(0, o);
            } else if (op == "GroupNormalization") {
                const XTensor x = I.as_f32(in(0)), g = I.as_f32(in(1)), b = I.as_f32(in(2));
                const int64_t G = ai("num_groups", 1), C = x.shape.at(1), sp = prod(x.shape, 2);
                Q3_CHECK(G > 0 && C % G == 0, "GroupNormalization groups");
                const int64_t rows = x.shape.at(0) * G, cols = (C / G) * sp;
                Q3_CHECK((g.numel() == C || g.numel() == G) && b.numel() == g.numel(), "GroupNormalization: scale / bias length is neither the channels nor the groups");
                XTensor nrm = I.dev_tensor(1, x.shape);
                if (rows && cols) { hipLaunchKernelGGL(k_norm_rows, dim3((unsigned)rows), dim3(256), 0, 0, I.f(nrm), I.f(x), (const float*)nullptr, (const float*)nullptr, cols, af("epsilon", 1e-5f), 0, (int64_t)1); I.count(); }
                // per-channel affine (opset 21 form; opset 18 files carry per-group scale / bias of length G)
                std::vector<int64_t> bs(x.shape.size(), 1); bs[1] = g.numel() == C ? C : G;
                XTensor y = nrm;
                if (g.numel() == C) { y = I.binary(B_MUL, nrm, I.reshaped(g, bs), 1); y = I.binary(B_ADD, y, I.reshaped(b, bs), 1); }
                else { std::vector<int64_t> gs = {x.shape[0], G, C / G}; gs.insert(gs.end(), x.shape.begin() + 2, x.shape.end()); std::vector<int64_t> gb(gs.size(), 1); gb[1] = G;
                       y = I.binary(B_ADD, I.binary(B_MUL, I.reshaped(nrm, gs), I.reshaped(g, gb), 1), I.reshaped(b, gb), 1); y = I.reshaped(y, x.shape); }
                out(0, y);
            } else if (op == "LpNormalization") {
                const XTensor x = I.as_f32(in(0));
                int64_t a = ai("axis", -1); if (a < 0) a += (int64_t)x.shape.size();
                const int64_t p2 = ai("p", 2);
                Q3_CHECK(p2 == 1 || p2 == 2, "LpNormalization p");
                int64_t rows, cols;
                XTensor perm = I.axes_last(x, {a}, rows, cols);
                std::vector<int64_t> ks = x.shape; ks[(size_t)a] = 1;
                XTensor nm = I.dev_tensor(1, ks);
                if (rows) { hipLaunchKernelGGL(k_reduce_rows, dim3((unsigned)rows), dim3(256), 0, 0, I.f(nm), (int64_t*)nullptr, I.f(perm), p2 == 2 ? R_L2 : R_L1, cols, 0); I.count(); }
                out(0, I.binary(B_DIV, x, nm, 1));
            } else {
                throw Error("operator is not supported by this executor");
            }
            Q3_LAUNCH_CHECK();
        } catch (const std::exception& e) {
            throw Error("node " + std::to_string(k) + " (" + nd.op_type + (nd.name.empty() ? "" : " '" + nd.name + "'") + "): " + e.what());
        }
        // release edges that nobody reads again
        for (auto& s : nd.inputs) { auto it = last.find(s); if (it != last.end() && it->second == k && !I.consts.count(s)) I.vals.erase(s); }
    }
    Q3_HIP(hipDeviceSynchronize());
    for (auto& o : model_->outputs) if (!I.vals.count(o.name)) throw Error("graph output " + o.name + " was not produced");
}

OnnxStreamDecoder::OnnxStreamDecoder(const std::string& path, int device) : s_(path, device) {
    const std::string miss = s_.model().check_decoder_contract();
    if (!miss.empty()) throw Error("not a streaming decoder graph (onnx.rs:355-455): missing" + miss);
    const auto bad = s_.unsupported_ops();
    if (!bad.empty()) { std::string t; for (auto& b : bad) t += " " + b; throw Error("the decoder graph uses operators without a kernel:" + t); }
    for (const char* n : {"pre_conv_history", "latent_buffer", "conv_history"}) state_io_.push_back({n, std::string("next_") + n});
    for (int i = 0; i < 8; i++) { state_io_.push_back({"past_key_" + std::to_string(i), "next_key_" + std::to_string(i)}); state_io_.push_back({"past_value_" + std::to_string(i), "next_value_" + std::to_string(i)}); }
    reset();
}
void OnnxStreamDecoder::reset() {
    // DecoderState::new (onnx.rs:470-495): every state tensor starts with a zero-length time axis.  The static dimensions come from the graph's declared
    // input shapes (symbolic / unknown dimensions -> 0); the reference's constants are the fallback when a graph declares no shape.
    state_.clear();
    for (auto& io : state_io_) {
        std::vector<int64_t> shape;
        for (const auto& vi : s_.model().inputs) if (vi.name == io.first) { shape = vi.shape; break; }
        if (shape.empty()) {
            if (io.first == "pre_conv_history") shape = {1, 512, -1};
            else if (io.first == "latent_buffer" || io.first == "conv_history") shape = {1, 1024, -1};
            else shape = {1, 16, -1, 64};
        }
        for (auto& d : shape) if (d < 0) d = 0;
        state_[io.first] = s_.zeros(1, shape);
    }
}
std::vector<float> OnnxStreamDecoder::decode(const int64_t* codes, int n_frames, bool is_final) {
    if (n_frames <= 0) return {};                                           // onnx.rs:350-353
    Q3_CHECK(codes != nullptr, "null codes");
    s_.set_input("audio_codes", 7, codes, {1, (int64_t)n_frames, 16});
    const float last = is_final ? 1.0f : 0.0f;
    s_.set_input("is_last", 1, &last, {1});
    for (auto& kv : state_) s_.bind_input(kv.first, kv.second);
    s_.run();
    const XTensor& wav = s_.value("final_wav");
    std::vector<float> pcm((size_t)wav.numel());
    s_.fetch(wav, pcm.data(), pcm.size() * 4);
    const XTensor& vs = s_.value("valid_samples");
    Q3_CHECK(vs.numel() >= 1, "valid_samples is empty");
    int64_t valid = 0;
    if (vs.dtype == 7) { std::vector<int64_t> v((size_t)vs.numel()); s_.fetch(vs, v.data(), v.size() * 8); valid = v[0]; }
    else { std::vector<float> v((size_t)vs.numel()); s_.fetch(vs, v.data(), v.size() * 4); valid = (int64_t)v[0]; }
    if (valid < 0) valid = 0;
    if ((size_t)valid < pcm.size()) pcm.resize((size_t)valid);              // `.take(valid_count)`
    for (auto& io : state_io_) state_[io.first] = s_.value(io.second);      // updated unconditionally, as the reference does
    return pcm;
}

} // namespace q3
