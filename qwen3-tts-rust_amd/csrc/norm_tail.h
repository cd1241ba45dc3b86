// norm_tail.h -- the residual + RMSNorm + int8 quantisation of a batched step as the TAIL of the GEMM that produces its input.
// The batched (>= 16 token) layer path had two norm launches per layer (k_rmsnorm_quant_wg: 10.7 % of the 64-slot step's GPU time at ~4.7 us
// each, of which ~3 us is the launch floor).  A norm needs a token's complete row, i.e. the partial sums of EVERY workgroup of the producing
// GEMM, so it cannot be a plain epilogue; instead the last workgroups to finish do it: every workgroup takes a ticket when its stores are
// out (fence + atomic), the last W = min(#workgroups, #tokens) ticket holders wait until all tickets are taken and then normalise one token
// each (tokens ticket - first, + W, ...).  Which workgroup serves a token varies from run to run; what it computes does not -- the partial
// sums are complete and are added in the spec's fixed order -- so the result is bit-identical to the standalone kernel.
// No deadlock: tickets are taken at the END of a workgroup's GEMM work, so a waiting workgroup only ever waits for workgroups that are
// running or still to be scheduled, and every finished workgroup has released its slot for those.
// The last norm workgroup to finish re-arms the two counters, so a replayed hipGraph starts from zero again.
//
// MEASURED (MI355X, round 2): parity-green and a large loss -- C3 620 -> 301 audio-s/s.  The hand-off needs device-scope release / acquire
// fences, and on a part whose 8 XCDs have private L2s those are L2 write-backs and invalidations (buffer_wbl2 / buffer_inv), paid by every
// workgroup of every GEMM: far more than the ~3 us launch floor of the standalone norm kernel they replace.  The same physics is behind the
// 3.9-21 us grid barriers of round 1.  Kept opt-in (Q3_NORM_TAIL=1) as the record of the experiment; the default path launches the norm.
#pragma once
#include "kdev.h"
#include "kernels.h"

namespace q3 {

struct NormTail { NormPro a; int d; int8_t* xq; uint16_t* xd; unsigned* counters; }; // counters = nullptr: no tail

// every thread of every workgroup calls this after its last global store; lds: >= 10.5 KB, 16-B aligned, free for reuse; blockDim >= 64 * (d / 256)
__device__ __forceinline__ void norm_tail(const NormTail& t, int ntok, unsigned char* lds) {
    if (!t.counters) return;
    __shared__ unsigned ticket_s;
    __threadfence();                       // this thread's partial sums are visible device-wide before the ticket is taken
    __syncthreads();
    if (threadIdx.x == 0) ticket_s = atomicAdd(&t.counters[0], 1u);
    __syncthreads();
    const unsigned total = gridDim.x * gridDim.y * gridDim.z;
    const unsigned W = total < (unsigned)ntok ? total : (unsigned)ntok, first = total - W;
    const unsigned ticket = ticket_s;
    if (ticket < first) return;            // (uniform over the workgroup)
    if (threadIdx.x == 0) {
        while (__hip_atomic_load(&t.counters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < total) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    __threadfence();                       // acquire side: nothing stale between this CU and the other workgroups' partial sums
    int8_t* xq_s = reinterpret_cast<int8_t*>(lds);
    uint16_t* xd_s = reinterpret_cast<uint16_t*>(lds + 2048);
    float* vbuf = reinterpret_cast<float*>(lds + 2048 + 128);
    float* scal = reinterpret_cast<float*>(lds + 2048 + 128 + 8192);
    const NormPro& a = t.a;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nch = t.d >> 8, d = t.d;
    for (int tok = (int)(ticket - first); tok < ntok; tok += (int)W) {
        // the arithmetic of norm_quant_wg (kernels_fused.hip), wave w = 256-chunk w; waves beyond the row's chunks only keep the barriers
        const int c = wave;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f), g = v;
        if (c < nch) {
            const float* hin = a.h_in + (size_t)tok * a.h_stride;
            v = *reinterpret_cast<const float4*>(hin + 256 * c + 4 * lane);
            g = *reinterpret_cast<const float4*>(a.g + 256 * c + 4 * lane);
            if (a.nparts > 0) {
                const float* pp = a.parts + (size_t)tok * a.parts_stride + 256 * c + 4 * lane;
                float4 y = *reinterpret_cast<const float4*>(pp);
                for (int s = 1; s < a.nparts; s++) {
                    const float4 z = *reinterpret_cast<const float4*>(pp + (size_t)s * a.parts_slab);
                    y.x = y.x + z.x; y.y = y.y + z.y; y.z = y.z + z.z; y.w = y.w + z.w;
                }
                v.x = v.x + y.x; v.y = v.y + y.y; v.z = v.z + y.z; v.w = v.w + y.w;
            }
            if (a.h_out) *reinterpret_cast<float4*>(a.h_out + (size_t)tok * d + 256 * c + 4 * lane) = v;
            *reinterpret_cast<float4*>(vbuf + 256 * c + 4 * lane) = v;
        }
        __syncthreads();
        if (wave == 0) {
            float p = 0.0f;
            for (int cc = 0; cc < nch; cc++) {
                const float4 u = *reinterpret_cast<const float4*>(vbuf + 256 * cc + 4 * lane);
                p = q3_fmaf(u.x, u.x, p); p = q3_fmaf(u.y, u.y, p); p = q3_fmaf(u.z, u.z, p); p = q3_fmaf(u.w, u.w, p);
            }
            const float ss = wave_sum_bfly(p);
            const float mean = ss / (float)d;
            if (lane == 0) scal[0] = 1.0f / q3_sqrtf(mean + a.eps);
        }
        __syncthreads();
        if (c < nch) {
            const float scale = scal[0];
            float4 y;
            y.x = (v.x * scale) * g.x; y.y = (v.y * scale) * g.y; y.z = (v.z * scale) * g.z; y.w = (v.w * scale) * g.w;
            if (a.xn_out) *reinterpret_cast<float4*>(a.xn_out + (size_t)tok * d + 256 * c + 4 * lane) = y;
            float amax = fmaxf(fmaxf(q3_fabsf(y.x), q3_fabsf(y.y)), fmaxf(q3_fabsf(y.z), q3_fabsf(y.w)));
            amax = fmaxf(amax, xor_lane<1>(amax)); amax = fmaxf(amax, xor_lane<2>(amax)); amax = fmaxf(amax, xor_lane<4>(amax));
            const float dd = amax / 127.0f;
            const float id = (dd != 0.0f) ? (1.0f / dd) : 0.0f;
            const int q0 = (int)q3_rintf(y.x * id), q1 = (int)q3_rintf(y.y * id), q2 = (int)q3_rintf(y.z * id), q3v = (int)q3_rintf(y.w * id);
            const uint32_t pk = (uint32_t)(q0 & 0xFF) | ((uint32_t)(q1 & 0xFF) << 8) | ((uint32_t)(q2 & 0xFF) << 16) | ((uint32_t)(q3v & 0xFF) << 24);
            *reinterpret_cast<uint32_t*>(xq_s + 256 * c + 4 * lane) = pk;
            if ((lane & 7) == 0) xd_s[8 * c + (lane >> 3)] = f2h(dd);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < d / 16; i += blockDim.x)
            *reinterpret_cast<uint4*>(t.xq + (size_t)tok * d + 16 * i) = *reinterpret_cast<const uint4*>(xq_s + 16 * i);
        for (int i = threadIdx.x; i < d / 32; i += blockDim.x) t.xd[(size_t)tok * (d / 32) + i] = xd_s[i];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (atomicAdd(&t.counters[1], 1u) == W - 1) { // every norm workgroup is past its wait: re-arm for the next launch (or graph replay)
            __hip_atomic_store(&t.counters[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&t.counters[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

} // namespace q3
