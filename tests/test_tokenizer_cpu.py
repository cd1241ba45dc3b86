"""the engine's tokenizer (csrc/tokenizer.cpp behind q3tts_tokenizer_*) against the Python `tokenizers` package -- the
same library (version 0.22.2) the reference's `tokenizers` crate is, so this parity is PINNED by a real independent implementation.
No Qwen tokenizer.json is on disk (the reference downloads it at run time), so the test trains a byte-level BPE with the Qwen2
pre-tokenisation pattern and Qwen-style added tokens, saves tokenizer.json and compares ids / decoded text string by string."""
import json
import os
import numpy as np
import pytest

tokenizers = pytest.importorskip("tokenizers")

QWEN_PATTERN = r"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+"

CORPUS = [
    "The quick brown fox jumps over the lazy dog. It's 9:45am, isn't it? We've got 1,234 apples & 56 pears!",
    "你好，世界！今天天气怎么样？我们去公园散步吧。语音合成系统把文字变成声音。",
    "def f(x):\n    return x**2 + 3*x - 7  # comment\n\n\nclass A:\n\tpass\n",
    "Émilie naïve façade coöperate Ωmega straße ＡＢＣ１２３ ３.１４ ½ ²",
    "こんにちは、元気ですか？ 안녕하세요 Привет мир مرحبا بالعالم שלום",
    "emoji 🎤🎶👩‍💻 and tabs\t\tand   multiple   spaces    end  ",
    "I'M HERE, YOU'LL SEE; THEY'D'VE gone. 'twas the night... it's o'clock",
] * 3


def _build(tmp_path, vocab_size=700):
    from tokenizers import Tokenizer, Regex, models, pre_tokenizers, decoders, trainers, normalizers
    tok = Tokenizer(models.BPE())
    tok.normalizer = normalizers.NFC()
    tok.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.Split(Regex(QWEN_PATTERN), behavior="isolated", invert=False),
                                                 pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)])
    tok.decoder = decoders.ByteLevel()
    trainer = trainers.BpeTrainer(vocab_size=vocab_size, initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), show_progress=False,
                                  special_tokens=["<|endoftext|>", "<|im_start|>", "<|im_end|>", "<|audio_start|>"])
    tok.train_from_iterator(CORPUS, trainer)
    path = str(tmp_path / "tokenizer.json")
    tok.save(path)
    return tok, path


CASES = [
    "", " ", "  ", "a", " a", "a ", "Hello, world!", "hello  world", "hello   world  ", "it's", "IT'S", "I'll we've they'd you're I'm don't 'tis",
    "x'Sy'T", "line1\nline2", "line1\r\nline2\r\n\r\nline3", "trail spaces   \n  next", "tabs\t\tx", "\t x", " \n", "\n\n\n", "a\n \n b",
    "123", "a1b22c333", "3.14159", "½ ² ３", "price: $12.50!!", "wow!!!\nnext", " !!! ", "a...b", "x - y", "(a)[b]{c}", "~`@#$%^&*()_+-=",
    "你好，世界！", "今天天气 怎么样?", "中文mixed英文123", "こんにちは", "안녕하세요", "Привет мир", "مرحبا", "Émilie naïve façade", "straße ＡＢＣ",
    "emoji 🎤🎶", "👩‍💻", "a🎤b", "<|im_start|>user\nhi<|im_end|>\n<|im_start|>assistant\n", "<|endoftext|>", "x<|im_end|>y<|im_end|>", "<|im_start",
    "<|audio_start|><|im_start|>", "The quick brown fox jumps over the lazy dog.", "def f(x):\n    return x**2\n", "\u00a0nbsp\u2003emsp\u3000ideographic",
    "a\u0085b", "tab\x0bvt\x0cff", "UPPER lower MiXeD", "ſ long s 'ſ", "\u00e9 precomposed",
]


def test_encode_decode_match_tokenizers_package(q3, tmp_path):
    tok, path = _build(tmp_path)
    mine = q3.Tokenizer(path)
    rng = np.random.default_rng(0)
    alphabet = list("abc XYZ'019.,!?\n\t-你好世界🎤é") + ["<|im_end|>", "  ", "\r\n", "'ll", "'S"]
    cases = list(CASES) + ["".join(rng.choice(alphabet, size=int(rng.integers(1, 40)))) for _ in range(300)]
    for text in cases:
        ref = tok.encode(text, add_special_tokens=False).ids
        got = mine.encode(text)
        assert got == ref, (text, got[:20], ref[:20])
        assert mine.decode(ref) == tok.decode(ref, skip_special_tokens=False), text
    # the NFC normaliser is applied between added tokens, as the `tokenizers` crate does: decomposed input encodes like its composed form
    import unicodedata
    for dec in ("e\u0301 combining", "A\u030a\u0301ngstro\u0308m", "\u1100\u1161\u11a8 \u1112\u1161\u11ab jamo", "q\u0323\u0307 reorder", "\u212b \u2126 singletons",
                "x<|im_end|>e\u0301<|im_end|>\u0065\u0301", "\u0301 leading mark", "n\u0303o\u0303 \u0915\u093c"):
        ref = tok.encode(dec, add_special_tokens=False).ids
        assert mine.encode(dec) == ref, dec
        assert mine.encode(unicodedata.normalize("NFC", dec)) == ref, dec
    pool = [chr(c) for c in list(range(0x20, 0x7F)) + list(range(0xC0, 0x250)) + list(range(0x300, 0x370)) + list(range(0x1100, 0x1113)) + list(range(0x1161, 0x1176)) +
            list(range(0x11A8, 0x11C3)) + list(range(0xAC00, 0xAC40)) + [0x1E9B, 0x212B, 0x2126, 0x0344, 0x0F73, 0x0958, 0xFB1D, 0x2ADC, 0x4E2D, 0x1F3A4]]
    for _ in range(3000):  # q3tts_text_nfc against Python's unicodedata (the same Unicode version generated the tables)
        t = "".join(rng.choice(pool, size=int(rng.integers(1, 12))))
        assert q3.text_nfc(t) == unicodedata.normalize("NFC", t), [hex(ord(c)) for c in t]
    for _ in range(200):
        t = "".join(rng.choice(pool, size=int(rng.integers(1, 16))))
        assert mine.encode(t) == tok.encode(t, add_special_tokens=False).ids, [hex(ord(c)) for c in t]
    mine.close()


def test_tokenizer_json_variants_and_errors(q3, tmp_path):
    tok, path = _build(tmp_path, vocab_size=400)
    d = json.load(open(path, encoding="utf-8"))
    # legacy "a b" merge strings instead of [a, b] pairs (older tokenizer.json files)
    d2 = json.loads(json.dumps(d))
    d2["model"]["merges"] = [m if isinstance(m, str) else m[0] + " " + m[1] for m in d2["model"]["merges"]]
    p2 = str(tmp_path / "legacy.json")
    json.dump(d2, open(p2, "w", encoding="utf-8"), ensure_ascii=True)   # \\uXXXX escapes for the byte-level alphabet
    a, b = q3.Tokenizer(path), q3.Tokenizer(p2)
    for text in ("Hello, world! it's 42.", "你好 🎤 <|im_end|>"):
        assert a.encode(text) == b.encode(text) == tok.encode(text, add_special_tokens=False).ids
    a.close(); b.close()
    with pytest.raises(q3.Q3Error):
        q3.Tokenizer(str(tmp_path / "missing.json"))
    d3 = json.loads(json.dumps(d)); d3["model"]["type"] = "WordPiece"
    p3 = str(tmp_path / "wp.json"); json.dump(d3, open(p3, "w"))
    with pytest.raises(q3.Q3Error):
        q3.Tokenizer(p3)
    d4 = json.loads(json.dumps(d)); d4["pre_tokenizer"]["pretokenizers"][0]["pattern"]["Regex"] = r"\w+|\s+"
    p4 = str(tmp_path / "rx.json"); json.dump(d4, open(p4, "w"))
    with pytest.raises(q3.Q3Error):
        q3.Tokenizer(p4)
