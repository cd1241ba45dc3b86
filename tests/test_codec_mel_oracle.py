"""Codec decoder oracle: chunked streaming == one full pass, and the whole stack == an independent torch.nn.functional
re-implementation of the same architecture (conv1d / conv_transpose1d / layer_norm / gelu / softmax).
Mel front end: oracle vs a numpy restatement of /root/reference/src/models/onnx.rs:167-320."""
import os
import numpy as np
import pytest
import ggml_ref as G

torch = pytest.importorskip("torch")
F = torch.nn.functional


def _w(t, name):
    ty, ne, raw = t[name]
    return torch.from_numpy(np.array(raw).view(np.float32).reshape(list(reversed(ne))).copy())


def torch_codec(path, codes):
    kv, t = G.read_gguf(path)
    H, nh, hd, W = kv["codec.hidden"], kv["codec.n_heads"], kv["codec.head_dim"], kv["codec.window"]
    T = codes.shape[0]
    z = sum(_w(t, "codec.codebook.%d" % q)[codes[:, q]] for q in range(16)).T.unsqueeze(0)   # [1,512,T]
    causal = lambda x, w, b, dil=1, groups=1: F.conv1d(F.pad(x, ((w.shape[-1] - 1) * dil, 0)), w, b, dilation=dil, groups=groups)
    h = causal(z, _w(t, "codec.pre_conv.weight"), _w(t, "codec.pre_conv.bias"))[0].T                  # [T,H]
    rms = lambda x, g: x * torch.rsqrt((x * x).mean(-1, keepdim=True) + kv["codec.eps"]) * g
    pos = torch.arange(T, dtype=torch.float64)
    inv = torch.tensor(kv["codec.rope_base"], dtype=torch.float64) ** (-torch.arange(hd // 2, dtype=torch.float64) / (hd // 2))
    ang = pos[:, None] * inv[None]
    cs, sn = torch.cos(ang).float(), torch.sin(ang).float()

    def rope(x):  # [T, nh, hd]
        a, b = x[..., : hd // 2], x[..., hd // 2:]
        return torch.cat([a * cs[:, None] - b * sn[:, None], b * cs[:, None] + a * sn[:, None]], -1)
    idx = torch.arange(T)
    mask = (idx[None] <= idx[:, None]) & (idx[:, None] - idx[None] < W)
    for l in range(kv["codec.n_layers"]):
        p = "codec.tf.%d." % l
        xn = rms(h, _w(t, p + "attn_norm"))
        q = rope((xn @ _w(t, p + "wq").T).view(T, nh, hd)); k = rope((xn @ _w(t, p + "wk").T).view(T, nh, hd))
        v = (xn @ _w(t, p + "wv").T).view(T, nh, hd)
        s = torch.einsum("thd,shd->hts", q, k) / hd ** 0.5
        s = s.masked_fill(~mask[None], float("-inf"))
        a = torch.einsum("hts,shd->thd", torch.softmax(s, -1), v).reshape(T, nh * hd)
        h = h + _w(t, p + "ls_attn") * (a @ _w(t, p + "wo").T)
        xn = rms(h, _w(t, p + "ffn_norm"))
        h = h + _w(t, p + "ls_ffn") * ((F.silu(xn @ _w(t, p + "w_gate").T) * (xn @ _w(t, p + "w_up").T)) @ _w(t, p + "w_down").T)
    x = rms(h, _w(t, "codec.tf.norm")).T.unsqueeze(0)                                              # [1,H,T]

    def convt(x, w, b, s):  # keep the first T*s outputs (right trim k-s): the streamable causal form
        y = F.conv_transpose1d(x, w, b, stride=s)
        return y[..., : x.shape[-1] * s]
    for i in range(kv["codec.n_up"]):
        p = "codec.up.%d." % i
        x = convt(x, _w(t, p + "convt.weight"), _w(t, p + "convt.bias"), kv["codec.up_ratio.%d" % i])
        d = causal(x, _w(t, p + "dw.weight").unsqueeze(1), _w(t, p + "dw.bias"), groups=H)
        d = F.layer_norm(d.transpose(1, 2), (H,), _w(t, p + "ln.weight"), _w(t, p + "ln.bias"), 1e-6)
        d = F.gelu(d @ _w(t, p + "pw1.weight").T + _w(t, p + "pw1.bias")) @ _w(t, p + "pw2.weight").T + _w(t, p + "pw2.bias")
        x = x + (_w(t, p + "gamma") * d).transpose(1, 2)
    snake = lambda x, a, b: x + (1.0 / (torch.exp(b) + 1e-9))[None, :, None] * torch.sin(x * torch.exp(a)[None, :, None]) ** 2
    x = causal(x, _w(t, "codec.dec.conv_in.weight"), _w(t, "codec.dec.conv_in.bias"))
    for b in range(kv["codec.n_dec"]):
        p = "codec.dec.%d." % b
        r = kv["codec.dec_rate.%d" % b]
        x = convt(snake(x, _w(t, p + "snake.alpha"), _w(t, p + "snake.beta")), _w(t, p + "convt.weight"), _w(t, p + "convt.bias"), r)
        for u, dil in enumerate((1, 3, 9)):
            q = p + "ru.%d." % u
            y = causal(snake(x, _w(t, q + "snake1.alpha"), _w(t, q + "snake1.beta")), _w(t, q + "conv1.weight"), _w(t, q + "conv1.bias"), dil)
            y = causal(snake(y, _w(t, q + "snake2.alpha"), _w(t, q + "snake2.beta")), _w(t, q + "conv2.weight"), _w(t, q + "conv2.bias"))
            x = x + y
    x = causal(snake(x, _w(t, "codec.dec.snake_out.alpha"), _w(t, "codec.dec.snake_out.beta")), _w(t, "codec.dec.conv_out.weight").reshape(1, -1, 7),
               _w(t, "codec.dec.conv_out.bias").reshape(1))
    return x.clamp(-1, 1)[0, 0].numpy()


def test_codec_streaming_equals_full_and_matches_torch(tiny_model, oracle):
    path = os.path.join(tiny_model, "onnx", "q3tts_codec.gguf")
    c = oracle.Codec(path)
    assert c.spf == 1920
    rng = np.random.default_rng(7)
    codes = rng.integers(0, 2048, (21, 16))   # 21 frames > window(8)+chunking: exercises the sliding KV history
    c.reset(); full = c.decode(codes).copy()
    for chunks in ([4, 4, 4, 4, 4, 1], [1] * 21, [7, 14], [20, 1]):
        c.reset()
        parts, o = [], 0
        for i, n in enumerate(chunks):
            parts.append(c.decode(codes[o:o + n], i == len(chunks) - 1).copy()); o += n
        assert np.array_equal(np.concatenate(parts), full), chunks
    with torch.no_grad():
        ref = torch_codec(path, torch.from_numpy(codes))
    assert ref.shape == full.shape
    assert np.sqrt(np.mean((ref - full) ** 2)) < 2e-6 and np.abs(ref - full).max() < 3e-5
    assert 0.02 < full.std() < 0.6 and np.abs(full).max() <= 1.0
    c.close()


def numpy_mel(audio):
    """onnx.rs:167-320 restated with numpy (float64 FFT)."""
    sr, n_fft, hop, n_mels = 24000.0, 1024, 256, 128
    f32 = np.float32

    def hz_to_mel(f):
        f_sp = f32(200.0) / f32(3.0); min_log_hz = f32(1000.0); min_log_mel = min_log_hz / f_sp
        logstep = np.log(f32(6.4)) / f32(27.0)
        return min_log_mel + np.log(f / min_log_hz) / logstep if f >= min_log_hz else f / f_sp

    def mel_to_hz(m):
        f_sp = f32(200.0) / f32(3.0); min_log_hz = f32(1000.0); min_log_mel = min_log_hz / f_sp
        logstep = np.log(f32(6.4)) / f32(27.0)
        return min_log_hz * np.exp(logstep * (m - min_log_mel)) if m >= min_log_mel else f_sp * m
    mmin, mmax = hz_to_mel(f32(0)), hz_to_mel(f32(12000))
    edges = [mel_to_hz(f32(mmin + (mmax - mmin) * f32(i) / f32(n_mels + 1))) for i in range(n_mels + 2)]
    freqs = np.arange(n_fft // 2 + 1, dtype=np.float32) * f32(sr) / f32(n_fft)
    fb = np.zeros((n_mels, n_fft // 2 + 1), np.float32)
    for m in range(n_mels):
        fl, fc, fr = edges[m], edges[m + 1], edges[m + 2]
        up = (freqs - fl) / (fc - fl); dn = (fr - freqs) / (fr - fc)
        w = np.where((freqs >= fl) & (freqs <= fc), up, np.where((freqs > fc) & (freqs <= fr), dn, 0))
        fb[m] = w * (f32(2.0) / (fr - fl))
    pad = (n_fft - hop) // 2
    n = len(audio)
    head = [audio[i] if i < n else 0.0 for i in range(pad, 0, -1)]
    tail = [audio[max(n - 1 - i, 0)] if n > 0 else 0.0 for i in range(1, pad + 1)]
    x = np.concatenate([head, audio, tail]).astype(np.float32)
    hann = (0.5 * (1 - np.cos(2 * np.pi * np.arange(n_fft) / n_fft))).astype(np.float32)
    out = []
    for f in range((max(len(x) - n_fft, 0)) // hop + 1):
        if f * hop + n_fft > len(x):
            break
        spec = np.fft.rfft((x[f * hop:f * hop + n_fft] * hann).astype(np.float64))
        mag = np.sqrt(np.abs(spec) ** 2 + 1e-9)
        out.append(np.log(np.maximum(fb.astype(np.float64) @ mag, 1e-5)))
    return np.array(out, np.float32)


def test_mel_matches_transformers_audio_utils(oracle):
    """Independent pin of row a16: the installed `transformers.audio_utils` builds the same Slaney-scale, Slaney-normalised 128-filter
    bank (0-12 kHz, 513 bins) and a Hann/1024/256 power-1 spectrogram; the oracle's mel of a chirp must agree with it.  Framing differs
    only in the padding rule (reference: reflect pad 384 by hand, onnx.rs:255-271), so the clip is pre-padded the reference's way."""
    from transformers.audio_utils import mel_filter_bank, spectrogram, window_function
    rng = np.random.default_rng(5)
    t = np.arange(24000) / 24000.0
    audio = (0.4 * np.sin(2 * np.pi * (300 + 2500 * t) * t) + 0.02 * rng.standard_normal(t.size)).astype(np.float32)
    pad = 384
    padded = np.concatenate([audio[pad:0:-1], audio, audio[-2:-pad - 2:-1]]).astype(np.float64)
    fb = mel_filter_bank(num_frequency_bins=513, num_mel_filters=128, min_frequency=0.0, max_frequency=12000.0, sampling_rate=24000,
                         norm="slaney", mel_scale="slaney")
    spec = spectrogram(padded, window_function(1024, "hann", periodic=True), frame_length=1024, hop_length=256, fft_length=1024, power=1.0,
                       center=False, mel_filters=fb, mel_floor=1e-5, log_mel="log")
    ref = spec.T.astype(np.float32)                       # [frames][128]
    got = oracle.mel(audio)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 5e-3                 # |FFT| gets +1e-9 under the root in the reference; f32 vs f64 pipelines


def test_mel_matches_numpy_restatement(oracle):
    rng = np.random.default_rng(11)
    t = np.arange(24000 * 2) / 24000.0
    chirp = (0.4 * np.sin(2 * np.pi * (200 + 3000 * t) * t) + 0.05 * rng.standard_normal(t.size)).astype(np.float32)
    for audio in (chirp, chirp[:5000], chirp[:300], np.zeros(1500, np.float32)):   # incl. clip shorter than the 384 pad
        got = oracle.mel(audio)
        ref = numpy_mel(audio)
        assert got.shape == ref.shape and got.shape[1] == 128
        assert np.abs(got - ref).max() < 2e-3
    assert oracle.mel(chirp).shape[0] == (len(chirp) + 768 - 1024) // 256 + 1
