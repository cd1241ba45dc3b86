"""File formats of the C++ host mirror (qwen3-tts-rust_amd/host): ".cache" TTSC v1 (utils/cache.rs:5-67), WAV i16 in/out
(utils/audio.rs:11-41), VoiceFile JSON (utils/voice_file.rs:5-62).  Expected bytes are built here from the reference's layout."""
import json
import os
import shutil
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "qwen3-tts-rust_amd")


def cache_bytes(codes, emb, magic=b"TTSC", version=1):
    return magic + struct.pack("<I", version) + struct.pack("<Q", len(codes)) + b"".join(struct.pack("<q", c) for c in codes) + \
        struct.pack("<Q", len(emb)) + b"".join(struct.pack("<f", e) for e in emb)


def test_cache_wav_voicefile_formats(tmp_path):
    if shutil.which("g++") is None or not os.path.exists(os.path.join(PKG, "libq3tts_host.so")):
        pytest.skip("host library not built")
    d = str(tmp_path)
    codes, emb = [7, 1, 2047, 0, -3], [1.0, 0.5, -2.25]
    open(os.path.join(d, "py.cache"), "wb").write(cache_bytes(codes, emb))
    open(os.path.join(d, "badmagic.cache"), "wb").write(cache_bytes(codes, emb, magic=b"TTSX"))
    open(os.path.join(d, "badver.cache"), "wb").write(cache_bytes(codes, emb, version=2))
    open(os.path.join(d, "short.cache"), "wb").write(cache_bytes(codes, emb)[:30])
    pcm = struct.pack("<6h", -32768, 32767, 0, 1, -1, 100)
    fmt = struct.pack("<HHIIHH", 1, 2, 22050, 22050 * 4, 4, 16)
    extra = b"LIST" + struct.pack("<I", 3) + b"abc" + b"\x00"                      # odd-sized chunk + pad byte before the data chunk
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + extra + b"data" + struct.pack("<I", len(pcm)) + pcm
    open(os.path.join(d, "py.wav"), "wb").write(b"RIFF" + struct.pack("<I", len(body)) + body)
    json.dump({"ref_text": "hi", "audio_codes": [5, 6], "spk_emb": [0.25, -1.0, 2.0, 0.0], "name": "n", "extra_key": {"ignored": True}},
              open(os.path.join(d, "voice.json"), "w"))
    uni = "\u4f60\u597d \U0001F3A4\nline2\ttab\r\b\f\"q\"\\"
    json.dump({"ref_text": uni, "audio_codes": [], "speaker_embedding": [1.0], "name": "\u00e9"}, open(os.path.join(d, "voice_uni.json"), "w"))  # ensure_ascii
    exe = os.path.join(d, "host_formats_main")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "host", "host_formats_main.cpp"),
                           "-L" + PKG, "-lq3tts_host", "-lq3tts", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([exe, d], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "OK", r.stderr
    assert open(os.path.join(d, "cpp.cache"), "rb").read() == cache_bytes(codes, emb)     # byte-identical to the reference's writer
    raw = open(os.path.join(d, "cpp.wav"), "rb").read()
    assert raw[:4] == b"RIFF" and raw[8:16] == b"WAVEfmt " and struct.unpack("<HHI", raw[20:28]) == (1, 1, 24000)
    assert struct.unpack("<6h", raw[44:56]) == (0, 32767, -32767, 16383, 32767, -32768)    # audio.rs:36: (s*32767).clamp(-32768,32767) as i16
    out = json.load(open(os.path.join(d, "voice_out.json")))
    assert out["ref_text"] == "hi" and out["audio_codes"] == [5, 6] and out["speaker_embedding"] == [0.25, -1.0, 2.0, 0.0]
    out2 = json.load(open(os.path.join(d, "voice_uni_out.json"), encoding="utf-8"))
    assert out2["ref_text"] == uni and out2["name"] == "\u00e9"                        # control characters escaped, UTF-8 passed through
