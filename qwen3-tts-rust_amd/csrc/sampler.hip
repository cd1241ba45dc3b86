// sampler.hip -- LlamaSampler::sample on device (/root/reference/src/models/llama/mod.rs:666-776), one workgroup per
// sequence, so that temperature > 0 decoding stays inside the per-frame hipGraph (no logits round trip to the host).
//
// Same steps, same f32 order as host_logic.cpp::Sampler::sample (the C ABI's q3tts_sampler_*):
//   T <= 0 : first-max argmax (strict >)                                                       :690-701
//   T  > 0 : stable sort descending -> top-k truncate -> e_i = exp((l_i - l_0)/T), sum in sorted order, divide
//            -> top-p: first prefix with cum >= p (inclusive), renormalise -> r = u32/2^32 from the sequence's
//            ChaCha12 stream -> first i with r < cum_i, else candidate 0                        :703-775
// The sort is a 4096-key bitonic network in LDS on (order-preserving float image, ~index) keys: descending key order
// == descending value with ties in ascending index order == the reference's stable sort.
#include "kernels.h"
#include "kdev.h"
#include "q3_common.h"

namespace q3 {

#define SAMPLE_NS 4096
#define SAMPLE_THREADS 1024

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }

// word `draw % 16` of ChaCha12 block `draw / 16` (rand_chacha layout: 64-bit block counter in words 12-13, stream id 0)
__device__ uint32_t chacha12_word(const uint32_t* __restrict__ key, uint32_t draw) {
    const uint64_t counter = draw >> 4;
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                      key[4], key[5], key[6], key[7], (uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
    uint32_t x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = s[i];
#define Q3_QR(a, b, c, d)                                                                                        \
    x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 12);                  \
    x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 8);  x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 7);
    for (int r = 0; r < 6; r++) {
        Q3_QR(0, 4, 8, 12) Q3_QR(1, 5, 9, 13) Q3_QR(2, 6, 10, 14) Q3_QR(3, 7, 11, 15)
        Q3_QR(0, 5, 10, 15) Q3_QR(1, 6, 11, 12) Q3_QR(2, 7, 8, 13) Q3_QR(3, 4, 9, 14)
    }
#undef Q3_QR
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) if ((int)(draw & 15u) == i) out = x[i] + s[i];
    return out;
}

__device__ __forceinline__ float key_value(u64 k) { // inverse of pack_key's float image
    const uint32_t ord = (uint32_t)(k >> 32);
    const uint32_t b = (ord & 0x80000000u) ? (ord & 0x7FFFFFFFu) : ~ord;
    return q3_bits_f32(b);
}

__global__ void __launch_bounds__(SAMPLE_THREADS) k_sample(SampleArgs a) {
    __shared__ u64 keys[SAMPLE_NS];
    __shared__ float prob[SAMPLE_NS];
    __shared__ float sh_f[2];
    __shared__ int sh_cut;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* lg = a.logits + (size_t)b * a.stride;
    const int mask = a.mask_idx ? a.mask_idx[b] : -1;
    const float T = a.temperature[b];
    const int n = a.n;
    for (int i = tid; i < SAMPLE_NS; i += SAMPLE_THREADS) {
        u64 k = 0; // pads sort below every real candidate (a real key always has a non-zero low word or high word)
        if (i < n) { const float v = (i == mask) ? -INFINITY : lg[i]; k = pack_key(v, i); }
        keys[i] = k;
    }
    __syncthreads();
    if (T <= 0.0f) { // greedy branch: max key == largest value, smallest index among equals
        u64 m = 0;
        for (int i = tid; i < SAMPLE_NS; i += SAMPLE_THREADS) m = keys[i] > m ? keys[i] : m;
        for (int s = 32; s >= 1; s >>= 1) { const u64 o = __shfl_xor(m, s); m = o > m ? o : m; }
        __syncthreads();
        if ((tid & 63) == 0) keys[tid >> 6] = m;
        __syncthreads();
        if (tid == 0) {
            u64 best = 0;
            for (int w = 0; w < SAMPLE_THREADS / 64; w++) best = keys[w] > best ? keys[w] : best;
            a.out_key[(size_t)b * a.out_stride] = pack_key(0.0f, key_code(best));
        }
        return;
    }
    // ---- bitonic sort, descending ----
    for (int k = 2; k <= SAMPLE_NS; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < SAMPLE_NS / 2; t += SAMPLE_THREADS) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)); // lower index of the pair
                const int p = i | j;
                const bool desc = (i & k) == 0;
                const u64 x = keys[i], y = keys[p];
                if ((x < y) == desc) { keys[i] = y; keys[p] = x; }
            }
            __syncthreads();
        }
    }
    const int top_k = a.top_k[b];
    int c = (top_k > 0 && top_k < n) ? top_k : n; // :711-713
    const float max_logit = key_value(keys[0]);
    for (int i = tid; i < c; i += SAMPLE_THREADS) { const float sc = (key_value(keys[i]) - max_logit) / T; prob[i] = q3_expf(sc); } // :716-722
    __syncthreads();
    if (tid == 0) { float sum = 0.0f; for (int i = 0; i < c; i++) sum = sum + prob[i]; sh_f[0] = sum; }
    __syncthreads();
    { const float sum = sh_f[0]; if (sum > 0.0f) for (int i = tid; i < c; i += SAMPLE_THREADS) prob[i] = prob[i] / sum; } // :724-730
    __syncthreads();
    const float top_p = a.top_p[b];
    if (top_p < 1.0f) { // :734-753
        if (tid == 0) {
            float cum = 0.0f; int cut = c;
            for (int i = 0; i < c; i++) { cum = cum + prob[i]; if (cum >= top_p) { cut = i + 1; break; } }
            float ns = 0.0f;
            for (int i = 0; i < cut; i++) ns = ns + prob[i];
            sh_cut = cut; sh_f[1] = ns;
        }
        __syncthreads();
        c = sh_cut;
        const float ns = sh_f[1];
        if (ns > 0.0f) for (int i = tid; i < c; i += SAMPLE_THREADS) prob[i] = prob[i] / ns;
        __syncthreads();
    }
    if (tid == 0) { // :761-775
        const uint32_t draw = a.draws[b];
        const float r = (float)chacha12_word(a.rng_key + (size_t)b * 8, draw) / 4294967296.0f;
        a.draws[b] = draw + 1;
        int pick = key_code(keys[0]);
        float cum = 0.0f;
        for (int i = 0; i < c; i++) { cum = cum + prob[i]; if (r < cum) { pick = key_code(keys[i]); break; } }
        a.out_key[(size_t)b * a.out_stride] = pack_key(0.0f, pick);
    }
}

void launch_sample(hipStream_t st, const SampleArgs& a, int B) {
    Q3_CHECK(a.n >= 1 && a.n <= SAMPLE_NS, "sampler range exceeds 4096 candidates");
    hipLaunchKernelGGL(k_sample, dim3(B), dim3(SAMPLE_THREADS), 0, st, a);
}

} // namespace q3
