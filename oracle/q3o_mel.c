/*
 * q3o_mel.c -- ORACLE (test infrastructure): log-mel front end of the speaker encoder, restating
 * /root/reference/src/models/onnx.rs:167-320 (24 kHz, n_fft 1024, hop 256, 128 Slaney mels 0-12 kHz,
 * reflect pad 384 with the zero-substitution quirk of :255-271, periodic Hann, |X| = sqrt(re^2+im^2+1e-9),
 * ln(max(mel,1e-5))).  The FFT itself (rustfft 6.4.1 in the reference) is evaluated in double here:
 * float-tolerance parity, not bit parity.
 */
#include "q3o.h"
#include <stdlib.h>

#define N_FFT 1024
#define HOP 256
#define N_MELS 128
#define N_BINS (N_FFT / 2 + 1)

static float hz_to_mel(float freq) { /* onnx.rs:180-192 */
    const float f_min = 0.0f, f_sp = 200.0f / 3.0f, min_log_hz = 1000.0f;
    const float min_log_mel = (min_log_hz - f_min) / f_sp;
    const float logstep = logf(6.4f) / 27.0f;
    if (freq >= min_log_hz) return min_log_mel + (logf(freq / min_log_hz) / logstep);
    return (freq - f_min) / f_sp;
}
static float mel_to_hz(float mel) { /* onnx.rs:195-207 */
    const float f_min = 0.0f, f_sp = 200.0f / 3.0f, min_log_hz = 1000.0f;
    const float min_log_mel = (min_log_hz - f_min) / f_sp;
    const float logstep = logf(6.4f) / 27.0f;
    if (mel >= min_log_mel) return min_log_hz * expf(logstep * (mel - min_log_mel));
    return f_min + f_sp * mel;
}

static void fft1024(double* re, double* im) { /* iterative radix-2, in place */
    const int n = N_FFT;
    for (int i = 1, j = 0; i < n; i++) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { double t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
    }
    for (int len = 2; len <= n; len <<= 1) {
        double ang = -2.0 * 3.14159265358979323846 / len;
        for (int i = 0; i < n; i += len)
            for (int k = 0; k < len / 2; k++) {
                double wr = cos(ang * k), wi = sin(ang * k);
                int a = i + k, b = i + k + len / 2;
                double xr = re[b] * wr - im[b] * wi, xi = re[b] * wi + im[b] * wr;
                re[b] = re[a] - xr; im[b] = im[a] - xi;
                re[a] += xr; im[a] += xi;
            }
    }
}

int q3o_mel(const float* audio, int n, float* mel_out) {
    const int padding = (N_FFT - HOP) / 2; /* 384, onnx.rs:251 */
    const int plen = padding + n + padding;
    int n_frames = (plen > N_FFT ? plen - N_FFT : 0) / HOP + 1; /* :283 */
    if (!mel_out) return n_frames;
    float* fb = (float*)calloc((size_t)N_MELS * N_BINS, 4);
    float edges[N_MELS + 2];
    float mel_min = hz_to_mel(0.0f), mel_max = hz_to_mel(12000.0f);
    for (int i = 0; i <= N_MELS + 1; i++) edges[i] = mel_to_hz(mel_min + (mel_max - mel_min) * (float)i / (float)(N_MELS + 1)); /* :216-219 */
    for (int m = 0; m < N_MELS; m++) { /* :228-246 */
        float fl = edges[m], fc = edges[m + 1], fr = edges[m + 2];
        float norm = 2.0f / (fr - fl);
        for (int k = 0; k < N_BINS; k++) {
            float freq = (float)k * 24000.0f / (float)N_FFT;
            float w = 0.0f;
            if (freq >= fl && freq <= fc) w = (freq - fl) / (fc - fl);
            else if (freq > fc && freq <= fr) w = (fr - freq) / (fr - fc);
            fb[m * N_BINS + k] = w * norm;
        }
    }
    float* padded = (float*)malloc((size_t)plen * 4);
    int pi = 0;
    for (int i = padding; i >= 1; i--) padded[pi++] = (i < n) ? audio[i] : 0.0f; /* :255-261 */
    for (int i = 0; i < n; i++) padded[pi++] = audio[i];
    for (int i = 1; i <= padding; i++) { /* :264-271 */
        int idx = n - (1 + i);
        if (idx < 0) idx = 0; /* saturating_sub */
        padded[pi++] = (idx < n) ? audio[idx] : 0.0f;
    }
    float hann[N_FFT];
    for (int i = 0; i < N_FFT; i++) hann[i] = 0.5f * (1.0f - cosf(2.0f * 3.14159265358979323846f * (float)i / (float)N_FFT)); /* :274-276 */
    int out_frames = 0;
    double re[N_FFT], im[N_FFT];
    float mag[N_BINS];
    for (int f = 0; f < n_frames; f++) {
        int start = f * HOP;
        if (start + N_FFT > plen) break; /* :288-290 */
        for (int i = 0; i < N_FFT; i++) { re[i] = (double)(padded[start + i] * hann[i]); im[i] = 0.0; }
        fft1024(re, im);
        for (int k = 0; k < N_BINS; k++) { float r = (float)re[k], q = (float)im[k]; mag[k] = sqrtf((r * r + q * q) + 1e-9f); } /* :301-304 */
        for (int m = 0; m < N_MELS; m++) { /* :307-315 */
            float v = 0.0f;
            for (int k = 0; k < N_BINS; k++) v += fb[m * N_BINS + k] * mag[k];
            mel_out[(size_t)out_frames * N_MELS + m] = logf(v > 1e-5f ? v : 1e-5f);
        }
        out_frames++;
    }
    free(fb); free(padded);
    return out_frames;
}
