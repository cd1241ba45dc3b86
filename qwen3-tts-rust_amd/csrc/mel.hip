// mel.hip -- log-mel front end of the speaker encoder on device, restating
// /root/reference/src/models/onnx.rs:167-320: 24 kHz, n_fft 1024, hop 256, reflect pad 384 (zeros when the clip is shorter
// than the pad, :255-271), periodic Hann, |X| = sqrt(re^2 + im^2 + 1e-9), 128 Slaney mel filters 0-12 kHz with Slaney
// normalisation, ln(max(., 1e-5)).  One workgroup per frame: windowed frame + twiddle table in LDS, direct 1024-point DFT
// for the 513 bins (0.5 MFLOP per frame: launch-bound, not worth an FFT), then the filterbank in ascending-k order.
#include "../../include/q3tts.h"
#include "q3_common.h"
#include <cmath>

namespace q3 {

constexpr int N_FFT = 1024, HOP = 256, N_MELS = 128, N_BINS = N_FFT / 2 + 1;

__global__ void __launch_bounds__(256) k_mel(const float* __restrict__ padded, int plen, const float* __restrict__ hann,
                                            const float* __restrict__ fb, float* __restrict__ mel) {
    __shared__ float xs[N_FFT];
    __shared__ float cs[N_FFT];
    __shared__ float sn[N_FFT];
    __shared__ float mag[N_BINS + 3];
    const int f = blockIdx.x, tid = threadIdx.x;
    const int start = f * HOP;
    for (int i = tid; i < N_FFT; i += 256) {
        xs[i] = padded[start + i] * hann[i];
        const float ang = 6.283185307179586f * (float)i / (float)N_FFT;
        cs[i] = cosf(ang); sn[i] = sinf(ang);
    }
    __syncthreads();
    for (int k = tid; k < N_BINS; k += 256) {
        float re = 0.0f, im = 0.0f;
        int idx = 0;
        for (int n = 0; n < N_FFT; n++) {
            re += xs[n] * cs[idx];
            im -= xs[n] * sn[idx];
            idx = (idx + k) & (N_FFT - 1);
        }
        mag[k] = sqrtf((re * re + im * im) + 1e-9f);
    }
    __syncthreads();
    if (tid < N_MELS) {
        float v = 0.0f;
        const float* w = fb + (size_t)tid * N_BINS;
        for (int k = 0; k < N_BINS; k++) v += w[k] * mag[k];
        mel[(size_t)f * N_MELS + tid] = logf(v > 1e-5f ? v : 1e-5f);
    }
}

static float hz_to_mel(float freq) { // onnx.rs:180-192
    const float f_sp = 200.0f / 3.0f, min_log_hz = 1000.0f, min_log_mel = min_log_hz / f_sp, logstep = logf(6.4f) / 27.0f;
    return freq >= min_log_hz ? min_log_mel + (logf(freq / min_log_hz) / logstep) : freq / f_sp;
}
static float mel_to_hz(float mel) { // onnx.rs:195-207
    const float f_sp = 200.0f / 3.0f, min_log_hz = 1000.0f, min_log_mel = min_log_hz / f_sp, logstep = logf(6.4f) / 27.0f;
    return mel >= min_log_mel ? min_log_hz * expf(logstep * (mel - min_log_mel)) : f_sp * mel;
}

int mel_device(const float* audio, int n, float* mel_out) {
    const int padding = (N_FFT - HOP) / 2, plen = padding + n + padding;
    int n_frames = (plen > N_FFT ? plen - N_FFT : 0) / HOP + 1; // :283
    while (n_frames > 0 && (n_frames - 1) * HOP + N_FFT > plen) n_frames--; // :288-290
    if (n_frames <= 0) return 0;
    std::vector<float> fb((size_t)N_MELS * N_BINS, 0.0f), edges(N_MELS + 2), hann(N_FFT), padded((size_t)plen);
    const float mmin = hz_to_mel(0.0f), mmax = hz_to_mel(12000.0f);
    for (int i = 0; i <= N_MELS + 1; i++) edges[i] = mel_to_hz(mmin + (mmax - mmin) * (float)i / (float)(N_MELS + 1));
    for (int m = 0; m < N_MELS; m++) {
        const float fl = edges[m], fc = edges[m + 1], fr = edges[m + 2], norm = 2.0f / (fr - fl);
        for (int k = 0; k < N_BINS; k++) {
            const float freq = (float)k * 24000.0f / (float)N_FFT;
            float w = 0.0f;
            if (freq >= fl && freq <= fc) w = (freq - fl) / (fc - fl);
            else if (freq > fc && freq <= fr) w = (fr - freq) / (fr - fc);
            fb[(size_t)m * N_BINS + k] = w * norm;
        }
    }
    int pi = 0;
    for (int i = padding; i >= 1; i--) padded[pi++] = (i < n) ? audio[i] : 0.0f;
    for (int i = 0; i < n; i++) padded[pi++] = audio[i];
    for (int i = 1; i <= padding; i++) { int idx = n - (1 + i); if (idx < 0) idx = 0; padded[pi++] = (idx < n) ? audio[idx] : 0.0f; }
    for (int i = 0; i < N_FFT; i++) hann[i] = 0.5f * (1.0f - cosf(2.0f * 3.14159265358979323846f * (float)i / (float)N_FFT));
    DevBuf<float> d_p(padded.size()), d_h(N_FFT), d_fb(fb.size()), d_mel((size_t)n_frames * N_MELS);
    d_p.upload(padded.data(), padded.size()); d_h.upload(hann.data(), N_FFT); d_fb.upload(fb.data(), fb.size());
    hipLaunchKernelGGL(k_mel, dim3(n_frames), dim3(256), 0, 0, d_p.p, plen, d_h.p, d_fb.p, d_mel.p);
    Q3_HIP(hipDeviceSynchronize());
    d_mel.download(mel_out, (size_t)n_frames * N_MELS);
    return n_frames;
}

} // namespace q3

extern "C" int q3tts_mel(const float* audio, int32_t n, float* mel_out) {
    try {
        int nd = 0;
        if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) throw q3::Error("no HIP device available: the HIP path is the only compute path (no CPU fallback)");
        if (!audio || !mel_out || n < 0) throw q3::Error("bad arguments");
        q3::mel_device(audio, n, mel_out);
        return Q3TTS_OK;
    } catch (const std::exception& ex) { q3::set_last_error(ex.what()); return Q3TTS_ERR; }
}
