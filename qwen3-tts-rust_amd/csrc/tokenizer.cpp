// tokenizer.cpp -- byte-level BPE tokenizer reading HuggingFace `tokenizer.json` files.
//
// Replaces what the reference gets from the `tokenizers` crate: Tokenizer::load = HfTokenizer::from_file(<model_dir>/tokenizer/tokenizer.json),
// encode = inner.encode(text, false).get_ids(), decode = inner.decode(ids, false)  (/root/reference/src/utils/tokenizer.rs:9-38; call sites
// src/tts/engine.rs:257,415).  The Qwen tokenizer family is: added (special) tokens split out of the raw text first; NFC normaliser; a Split
// pre-tokeniser with the pattern below (behaviour "isolated") followed by ByteLevel (no prefix space, no regex); a BPE model over the
// byte-level alphabet (vocab + ranked merges); ByteLevel decoder.  Parity is PINNED: tests/test_tokenizer_cpu.py builds tokenizer.json
// files with the Python `tokenizers` package (0.22.2, the crate's version) and compares ids string by string.
// Not implemented (rejected at load or documented): normalisers other than NFC / none,
// added tokens with lstrip / rstrip / single_word, byte_fallback, dropout, unk_token.
#include "tokenizer.h"
#include "q3_common.h"
#include "unicode_tables.h"
#include <algorithm>
#include <fstream>
#include <functional>
#include <map>
#include <tuple>
#include <queue>
#include <sstream>

namespace q3 {
namespace {

// ---------------- small JSON DOM (objects, arrays, strings with the full escape set, numbers, literals) ----------------
struct JVal {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false; double num = 0; std::string str; std::vector<JVal> arr; std::vector<std::pair<std::string, JVal>> obj;
    const JVal* get(const char* k) const { if (kind != Obj) return nullptr; for (auto& kv : obj) if (kv.first == k) return &kv.second; return nullptr; }
    bool truthy(const char* k) const { const JVal* v = get(k); return v && v->kind == Bool && v->b; }
    std::string s(const char* k, const std::string& def = "") const { const JVal* v = get(k); return (v && v->kind == Str) ? v->str : def; }
};
struct JParser {
    const std::string& s; size_t i = 0;
    [[noreturn]] void fail(const char* m) { throw Error(std::string("tokenizer.json: ") + m + " at byte " + std::to_string(i)); }
    void ws() { while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) i++; }
    static void utf8(std::string& o, uint32_t cp) {
        if (cp < 0x80) o += (char)cp;
        else if (cp < 0x800) { o += (char)(0xC0 | (cp >> 6)); o += (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { o += (char)(0xE0 | (cp >> 12)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
        else { o += (char)(0xF0 | (cp >> 18)); o += (char)(0x80 | ((cp >> 12) & 0x3F)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
    }
    uint32_t hex4() {
        if (i + 4 > s.size()) fail("truncated \\u escape");
        uint32_t v = 0;
        for (int k = 0; k < 4; k++) { const char c = s[i++]; v = v * 16 + (c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : (fail("bad hex digit"), 0)); }
        return v;
    }
    std::string str() {
        if (s[i] != '"') fail("string expected");
        i++;
        std::string o;
        while (i < s.size() && s[i] != '"') {
            if (s[i] == '\\') {
                if (++i >= s.size()) fail("truncated escape");
                const char c = s[i++];
                switch (c) {
                    case 'n': o += '\n'; break; case 't': o += '\t'; break; case 'r': o += '\r'; break; case 'b': o += '\b'; break; case 'f': o += '\f'; break;
                    case 'u': {
                        uint32_t cp = hex4();
                        if (cp >= 0xD800 && cp <= 0xDBFF && i + 1 < s.size() && s[i] == '\\' && s[i + 1] == 'u') { i += 2; const uint32_t lo = hex4(); cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00); }
                        utf8(o, cp);
                        break;
                    }
                    default: o += c;
                }
            } else o += s[i++];
        }
        if (i >= s.size()) fail("unterminated string");
        i++;
        return o;
    }
    JVal val() {
        ws();
        if (i >= s.size()) fail("unexpected end");
        JVal v;
        const char c = s[i];
        if (c == '{') {
            v.kind = JVal::Obj; i++; ws();
            if (i < s.size() && s[i] == '}') { i++; return v; }
            for (;;) {
                ws(); std::string k = str(); ws();
                if (i >= s.size() || s[i] != ':') fail("':' expected");
                i++;
                v.obj.emplace_back(std::move(k), val());
                ws();
                if (i < s.size() && s[i] == ',') { i++; continue; }
                if (i < s.size() && s[i] == '}') { i++; return v; }
                fail("',' or '}' expected");
            }
        }
        if (c == '[') {
            v.kind = JVal::Arr; i++; ws();
            if (i < s.size() && s[i] == ']') { i++; return v; }
            for (;;) {
                v.arr.push_back(val());
                ws();
                if (i < s.size() && s[i] == ',') { i++; continue; }
                if (i < s.size() && s[i] == ']') { i++; return v; }
                fail("',' or ']' expected");
            }
        }
        if (c == '"') { v.kind = JVal::Str; v.str = str(); return v; }
        if (!s.compare(i, 4, "true")) { v.kind = JVal::Bool; v.b = true; i += 4; return v; }
        if (!s.compare(i, 5, "false")) { v.kind = JVal::Bool; i += 5; return v; }
        if (!s.compare(i, 4, "null")) { i += 4; return v; }
        size_t st = i;
        while (i < s.size() && (isdigit((unsigned char)s[i]) || strchr("+-.eE", s[i]))) i++;
        if (st == i) fail("value expected");
        v.kind = JVal::Num; v.num = std::stod(s.substr(st, i - st));
        return v;
    }
};

// ---------------- Unicode helpers ----------------
bool in_ranges(const uint32_t (*tab)[2], size_t n, uint32_t cp) {
    size_t lo = 0, hi = n;
    while (lo < hi) { const size_t mid = (lo + hi) / 2; if (cp < tab[mid][0]) hi = mid; else if (cp > tab[mid][1]) lo = mid + 1; else return true; }
    return false;
}
bool is_L(uint32_t cp) { return in_ranges(kUnicodeL, sizeof(kUnicodeL) / sizeof(kUnicodeL[0]), cp); }
bool is_N(uint32_t cp) { return in_ranges(kUnicodeN, sizeof(kUnicodeN) / sizeof(kUnicodeN[0]), cp); }
bool is_S(uint32_t cp) { // \s of the regex crate: Unicode White_Space
    return (cp >= 9 && cp <= 13) || cp == 0x20 || cp == 0x85 || cp == 0xA0 || cp == 0x1680 || (cp >= 0x2000 && cp <= 0x200A) || cp == 0x2028 || cp == 0x2029 ||
           cp == 0x202F || cp == 0x205F || cp == 0x3000;
}
bool is_nl(uint32_t cp) { return cp == '\r' || cp == '\n'; }
struct Cp { uint32_t cp; uint32_t off; }; // code point + byte offset in the text
std::vector<Cp> decode_utf8(const std::string& t) {
    std::vector<Cp> out;
    size_t i = 0;
    while (i < t.size()) {
        const unsigned char c = (unsigned char)t[i];
        uint32_t cp = 0xFFFD; int n = 1;
        if (c < 0x80) cp = c;
        else if ((c >> 5) == 6 && i + 1 < t.size()) { cp = ((c & 0x1F) << 6) | ((unsigned char)t[i + 1] & 0x3F); n = 2; }
        else if ((c >> 4) == 14 && i + 2 < t.size()) { cp = ((c & 0x0F) << 12) | (((unsigned char)t[i + 1] & 0x3F) << 6) | ((unsigned char)t[i + 2] & 0x3F); n = 3; }
        else if ((c >> 3) == 30 && i + 3 < t.size()) { cp = ((c & 0x07) << 18) | (((unsigned char)t[i + 1] & 0x3F) << 12) | (((unsigned char)t[i + 2] & 0x3F) << 6) | ((unsigned char)t[i + 3] & 0x3F); n = 4; }
        out.push_back(Cp{cp, (uint32_t)i});
        i += n;
    }
    out.push_back(Cp{0, (uint32_t)t.size()}); // sentinel: end offset
    return out;
}

// ---- NFC (the Qwen tokenizer's normaliser; UAX #15) over the generated tables of unicode_tables.h ----
static uint32_t nfc_ccc(uint32_t cp) {
    if (cp < 0x300) return 0;
    size_t lo = 0, hi = sizeof(kNfcCcc) / sizeof(kNfcCcc[0]);
    while (lo < hi) { const size_t mid = (lo + hi) / 2; if (kNfcCcc[mid][1] < cp) lo = mid + 1; else hi = mid; }
    return (lo < sizeof(kNfcCcc) / sizeof(kNfcCcc[0]) && kNfcCcc[lo][0] <= cp) ? kNfcCcc[lo][2] : 0;
}
static void nfc_decompose(uint32_t cp, std::vector<uint32_t>& out) {
    if (cp >= 0xAC00 && cp <= 0xD7A3) { // Hangul syllable -> L V (T)
        const uint32_t s = cp - 0xAC00, t = s % 28;
        out.push_back(0x1100 + s / 588); out.push_back(0x1161 + (s % 588) / 28);
        if (t) out.push_back(0x11A7 + t);
        return;
    }
    if (cp >= 0xC0) {
        size_t lo = 0, hi = sizeof(kNfcDecompIdx) / sizeof(kNfcDecompIdx[0]);
        while (lo < hi) { const size_t mid = (lo + hi) / 2; if (kNfcDecompIdx[mid][0] < cp) lo = mid + 1; else hi = mid; }
        if (lo < sizeof(kNfcDecompIdx) / sizeof(kNfcDecompIdx[0]) && kNfcDecompIdx[lo][0] == cp) {
            for (uint32_t k = 0; k < kNfcDecompIdx[lo][2]; k++) out.push_back(kNfcDecompData[kNfcDecompIdx[lo][1] + k]);
            return;
        }
    }
    out.push_back(cp);
}
static uint32_t nfc_compose(uint32_t a, uint32_t b) { // 0 = no primary composite
    if (a >= 0x1100 && a <= 0x1112 && b >= 0x1161 && b <= 0x1175) return 0xAC00 + ((a - 0x1100) * 21 + (b - 0x1161)) * 28;
    if (a >= 0xAC00 && a <= 0xD7A3 && (a - 0xAC00) % 28 == 0 && b >= 0x11A8 && b <= 0x11C2) return a + (b - 0x11A7);
    size_t lo = 0, hi = sizeof(kNfcComp) / sizeof(kNfcComp[0]);
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (kNfcComp[mid][0] < a || (kNfcComp[mid][0] == a && kNfcComp[mid][1] < b)) lo = mid + 1; else hi = mid;
    }
    return (lo < sizeof(kNfcComp) / sizeof(kNfcComp[0]) && kNfcComp[lo][0] == a && kNfcComp[lo][1] == b) ? kNfcComp[lo][2] : 0;
}
std::string nfc_utf8_impl(const std::string& text) {
    bool ascii = true;
    for (unsigned char c : text) if (c >= 0x80) { ascii = false; break; }
    if (ascii) return text; // ASCII is its own NFC
    const std::vector<Cp> cps = decode_utf8(text);
    std::vector<uint32_t> d;
    d.reserve(cps.size() + 8);
    for (size_t i = 0; i + 1 < cps.size(); i++) nfc_decompose(cps[i].cp, d);
    // canonical ordering: stable sort of every run of non-starters by combining class
    for (size_t i = 0; i < d.size();) {
        if (nfc_ccc(d[i]) == 0) { i++; continue; }
        size_t e = i;
        while (e < d.size() && nfc_ccc(d[e]) != 0) e++;
        std::stable_sort(d.begin() + (long)i, d.begin() + (long)e, [](uint32_t x, uint32_t y) { return nfc_ccc(x) < nfc_ccc(y); });
        i = e;
    }
    // canonical composition (UAX #15 reference form)
    if (!d.empty()) {
        size_t starter_pos = 0, comp_pos = 1;
        uint32_t starter_ch = d[0];
        int last_class = (int)nfc_ccc(starter_ch);
        if (last_class != 0) last_class = 256; // a string that starts with a combining mark: nothing composes onto it
        for (size_t pos = 1; pos < d.size(); pos++) {
            const uint32_t ch = d[pos];
            const int ch_class = (int)nfc_ccc(ch);
            const uint32_t composite = nfc_compose(starter_ch, ch);
            if (composite && (last_class < ch_class || last_class == 0)) { d[starter_pos] = composite; starter_ch = composite; }
            else {
                if (ch_class == 0) { starter_pos = comp_pos; starter_ch = ch; }
                last_class = ch_class;
                d[comp_pos++] = ch;
            }
        }
        d.resize(comp_pos);
    }
    std::string out;
    out.reserve(text.size());
    for (uint32_t cp : d) JParser::utf8(out, cp);
    return out;
}

// The Qwen2 / GPT-4-style pre-tokenisation pattern, matched the way a backtracking leftmost-first engine does (alternatives in order):
//   (?i:'s|'t|'re|'ve|'m|'ll|'d) | [^\r\n\p{L}\p{N}]?\p{L}+ | \p{N} | ?[^\s\p{L}\p{N}]+[\r\n]* | \s*[\r\n]+ | \s+(?!\S) | \s+
const char* kQwenPattern = "(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\\r\\n\\p{L}\\p{N}]?\\p{L}+|\\p{N}| ?[^\\s\\p{L}\\p{N}]+[\\r\\n]*|\\s*[\\r\\n]+|\\s+(?!\\S)|\\s+";
size_t match_qwen(const std::vector<Cp>& c, size_t i, size_t n) { // returns the end index (> i) of the token starting at code point i
    auto lower = [](uint32_t x) { return (x >= 'A' && x <= 'Z') ? x + 32 : x; };
    // 1. contractions, case-insensitive.  (?i) also folds U+017F (long s) to 's' and U+212A (Kelvin) to 'k' in a Unicode-aware engine;
    //    only the former can matter here
    if (c[i].cp == '\'' && i + 1 < n) {
        const uint32_t a = lower(c[i + 1].cp), b = i + 2 < n ? lower(c[i + 2].cp) : 0;
        if (a == 's' || a == 't' || c[i + 1].cp == 0x17F) return i + 2;
        if ((a == 'r' && b == 'e') || (a == 'v' && b == 'e')) return i + 3;
        if (a == 'm') return i + 2;
        if (a == 'l' && b == 'l') return i + 3;
        if (a == 'd') return i + 2;
    }
    // 2. [^\r\n\p{L}\p{N}]?\p{L}+
    {
        size_t j = i;
        const uint32_t x = c[j].cp;
        if (!is_nl(x) && !is_L(x) && !is_N(x) && j + 1 < n && is_L(c[j + 1].cp)) j++;
        if (is_L(c[j].cp)) { while (j < n && is_L(c[j].cp)) j++; return j; }
    }
    // 3. \p{N}
    if (is_N(c[i].cp)) return i + 1;
    // 4.  ?[^\s\p{L}\p{N}]+[\r\n]*
    {
        size_t j = i;
        if (c[j].cp == ' ' && j + 1 < n) { const uint32_t y = c[j + 1].cp; if (!is_S(y) && !is_L(y) && !is_N(y)) j++; }
        const size_t st = j;
        while (j < n && !is_S(c[j].cp) && !is_L(c[j].cp) && !is_N(c[j].cp)) j++;
        if (j > st) { while (j < n && is_nl(c[j].cp)) j++; return j; }
    }
    // 5..7: whitespace runs
    size_t e = i;
    while (e < n && is_S(c[e].cp)) e++;
    if (e == i) return i + 1; // (cannot happen: every code point is a letter, a number, whitespace or "other"; stay safe)
    // 5. \s*[\r\n]+ : through the LAST newline of the run
    for (size_t k = e; k > i; k--) if (is_nl(c[k - 1].cp)) return k;
    // 6. \s+(?!\S) : the whole run at the end of the text, else all but its last character (if that leaves something)
    if (e == n) return e;
    if (e - i >= 2) return e - 1;
    // 7. \s+
    return e;
}

// GPT-2 byte <-> unicode alphabet of the ByteLevel pre-tokeniser: printable bytes map to themselves, the others to U+0100...
struct ByteMap {
    uint32_t b2u[256]; std::map<uint32_t, uint8_t> u2b;
    ByteMap() {
        int n = 0;
        for (int b = 0; b < 256; b++) {
            const bool keep = (b >= 33 && b <= 126) || (b >= 161 && b <= 172) || (b >= 174 && b <= 255);
            b2u[b] = keep ? (uint32_t)b : (uint32_t)(256 + n++);
            u2b[b2u[b]] = (uint8_t)b;
        }
    }
};
const ByteMap& bytemap() { static const ByteMap m; return m; }

} // namespace

struct Tokenizer::Impl {
    std::map<std::string, int32_t> vocab; std::vector<std::string> id_to_tok;
    std::map<std::pair<int32_t, int32_t>, std::pair<int32_t, int32_t>> merges; // (left id, right id) -> (rank, merged id)
    std::vector<std::pair<std::string, int32_t>> added;                        // content, id (matched leftmost-longest on the raw text)
    std::map<int32_t, std::string> added_by_id;
    bool ignore_merges = false; bool qwen_split = false; bool byte_level = true; bool nfc = false;
    std::map<std::string, std::vector<int32_t>> cache;

    void bpe(const std::string& piece, std::vector<int32_t>& out) { // piece: byte-level string (UTF-8 of the mapped alphabet)
        auto hit = cache.find(piece);
        if (hit != cache.end()) { out.insert(out.end(), hit->second.begin(), hit->second.end()); return; }
        std::vector<int32_t> ids;
        if (ignore_merges) { auto it = vocab.find(piece); if (it != vocab.end()) { out.push_back(it->second); return; } }
        // symbols = the piece's characters; a linked list merged by (rank, position) priority like the crate's Word::merge_all
        struct Sym { int32_t id; int prev, next; bool alive; };
        std::vector<Sym> sy;
        for (size_t i = 0; i < piece.size();) {
            const unsigned char c = (unsigned char)piece[i];
            const int n = c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : 4;
            auto it = vocab.find(piece.substr(i, n));
            if (it == vocab.end()) throw Error("tokenizer: symbol missing from the vocabulary (no unk_token / byte_fallback configured)");
            sy.push_back(Sym{it->second, (int)sy.size() - 1, (int)sy.size() + 1, true});
            i += n;
        }
        if (!sy.empty()) sy.back().next = -1;
        typedef std::tuple<int32_t, int, int32_t, int32_t> Cand; // rank, position, left id, right id (min-heap)
        std::priority_queue<Cand, std::vector<Cand>, std::greater<Cand>> pq;
        auto push = [&](int p) {
            if (p < 0 || sy[p].next < 0) return;
            auto it = merges.find({sy[p].id, sy[sy[p].next].id});
            if (it != merges.end()) pq.push(Cand{it->second.first, p, sy[p].id, sy[sy[p].next].id});
        };
        for (int p = 0; p < (int)sy.size(); p++) push(p);
        while (!pq.empty()) {
            const Cand c = pq.top(); pq.pop();
            const int p = std::get<1>(c);
            if (!sy[p].alive || sy[p].next < 0) continue;
            const int q = sy[p].next;
            if (sy[p].id != std::get<2>(c) || sy[q].id != std::get<3>(c)) continue; // stale entry
            sy[p].id = merges.at({std::get<2>(c), std::get<3>(c)}).second;
            sy[q].alive = false;
            sy[p].next = sy[q].next;
            if (sy[q].next >= 0) sy[sy[q].next].prev = p;
            push(sy[p].prev); push(p);
        }
        for (int p = sy.empty() ? -1 : 0; p >= 0; p = sy[p].next) ids.push_back(sy[p].id);
        if (cache.size() < 100000) cache[piece] = ids;
        out.insert(out.end(), ids.begin(), ids.end());
    }
    void encode_section(const std::string& raw, std::vector<int32_t>& out) {
        if (raw.empty()) return;
        const std::string text = nfc ? nfc_utf8_impl(raw) : raw; // the normaliser runs on the text between added tokens, before the pre-tokeniser
        const std::vector<Cp> cps = decode_utf8(text);
        const size_t n = cps.size() - 1;
        const ByteMap& bm = bytemap();
        size_t i = 0;
        while (i < n) {
            const size_t e = qwen_split ? match_qwen(cps, i, n) : n;
            std::string piece;
            for (size_t b = cps[i].off; b < cps[e].off; b++) {
                if (byte_level) JParser::utf8(piece, bm.b2u[(unsigned char)text[b]]); else piece += text[b];
            }
            bpe(piece, out);
            i = e;
        }
    }
};

Tokenizer::Tokenizer(const std::string& path) : impl_(new Impl()) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw Error("Failed to load tokenizer: cannot open " + path);
    std::stringstream ss; ss << f.rdbuf();
    const std::string txt = ss.str();
    JParser jp{txt};
    const JVal root = jp.val();
    const JVal* model = root.get("model");
    if (!model || model->s("type", "BPE") != "BPE") throw Error("Failed to load tokenizer: only BPE models are supported");
    if (model->truthy("byte_fallback")) throw Error("Failed to load tokenizer: byte_fallback is not supported");
    if (const JVal* d = model->get("dropout")) if (d->kind == JVal::Num && d->num > 0) throw Error("Failed to load tokenizer: dropout is not supported");
    Impl& m = *impl_;
    m.ignore_merges = model->truthy("ignore_merges");
    const JVal* vocab = model->get("vocab");
    if (!vocab || vocab->kind != JVal::Obj) throw Error("Failed to load tokenizer: model.vocab missing");
    for (auto& kv : vocab->obj) {
        const int32_t id = (int32_t)kv.second.num;
        m.vocab[kv.first] = id;
        if ((size_t)id >= m.id_to_tok.size()) m.id_to_tok.resize((size_t)id + 1);
        m.id_to_tok[(size_t)id] = kv.first;
    }
    if (const JVal* mg = model->get("merges")) {
        int32_t rank = 0;
        for (auto& e : mg->arr) { // "left right" (legacy) or ["left", "right"]
            std::string a, b;
            if (e.kind == JVal::Str) { const size_t sp = e.str.find(' '); if (sp == std::string::npos) throw Error("Failed to load tokenizer: bad merge entry"); a = e.str.substr(0, sp); b = e.str.substr(sp + 1); }
            else if (e.kind == JVal::Arr && e.arr.size() == 2) { a = e.arr[0].str; b = e.arr[1].str; }
            else throw Error("Failed to load tokenizer: bad merge entry");
            auto ia = m.vocab.find(a), ib = m.vocab.find(b), ic = m.vocab.find(a + b);
            if (ia == m.vocab.end() || ib == m.vocab.end() || ic == m.vocab.end()) throw Error("Failed to load tokenizer: merge refers to a token outside the vocabulary");
            m.merges[{ia->second, ib->second}] = {rank++, ic->second};
        }
    }
    if (const JVal* at = root.get("added_tokens")) for (auto& t : at->arr) {
        if (t.truthy("lstrip") || t.truthy("rstrip") || t.truthy("single_word")) throw Error("Failed to load tokenizer: added tokens with lstrip / rstrip / single_word are not supported");
        const JVal* idv = t.get("id");
        const std::string content = t.s("content");
        if (!idv || content.empty()) continue;
        m.added.emplace_back(content, (int32_t)idv->num);
        m.added_by_id[(int32_t)idv->num] = content;
    }
    std::sort(m.added.begin(), m.added.end(), [](const std::pair<std::string, int32_t>& a, const std::pair<std::string, int32_t>& b) { return a.first.size() > b.first.size(); });
    if (const JVal* nz = root.get("normalizer")) if (nz->kind == JVal::Obj) {
        if (nz->s("type") != "NFC") throw Error("Failed to load tokenizer: normalizer " + nz->s("type") + " is not supported (NFC or none)");
        m.nfc = true;
    }
    // pre-tokeniser: ByteLevel alone (GPT-2 family with its own regex: not supported), or Sequence[Split(pattern, isolated), ByteLevel(use_regex = false)]
    m.byte_level = false;
    std::function<void(const JVal&)> scan = [&](const JVal& p) {
        const std::string ty = p.s("type");
        if (ty == "Sequence") { if (const JVal* l = p.get("pretokenizers")) for (auto& q : l->arr) scan(q); }
        else if (ty == "Split") {
            const JVal* pat = p.get("pattern");
            const std::string rx = pat ? (pat->get("Regex") ? pat->get("Regex")->str : pat->get("String") ? pat->get("String")->str : "") : "";
            if (rx != kQwenPattern) throw Error("Failed to load tokenizer: unsupported Split pattern (only the Qwen2 pattern is implemented)");
            if (p.s("behavior") != "Isolated" || p.truthy("invert")) throw Error("Failed to load tokenizer: Split must be Isolated, not inverted");
            m.qwen_split = true;
        } else if (ty == "ByteLevel") {
            if (p.truthy("add_prefix_space")) throw Error("Failed to load tokenizer: ByteLevel add_prefix_space is not supported");
            if (p.truthy("use_regex")) throw Error("Failed to load tokenizer: ByteLevel use_regex (GPT-2 pattern) is not supported");
            m.byte_level = true;
        } else if (!ty.empty()) throw Error("Failed to load tokenizer: pre-tokenizer " + ty + " is not supported");
    };
    if (const JVal* pt = root.get("pre_tokenizer")) if (pt->kind == JVal::Obj) scan(*pt);
}
Tokenizer::~Tokenizer() = default;

std::vector<int32_t> Tokenizer::encode(const std::string& text) const { // encode(text, add_special_tokens = false): utils/tokenizer.rs:17-25
    Impl& m = *impl_;
    std::vector<int32_t> out;
    size_t pos = 0, sec = 0;
    while (pos < text.size()) { // added tokens are cut out of the raw text first (leftmost, longest)
        const std::pair<std::string, int32_t>* best = nullptr;
        for (auto& a : m.added) if (text.compare(pos, a.first.size(), a.first) == 0) { best = &a; break; } // sorted by length: first hit = longest
        if (best) {
            m.encode_section(text.substr(sec, pos - sec), out);
            out.push_back(best->second);
            pos += best->first.size(); sec = pos;
        } else pos++;
    }
    m.encode_section(text.substr(sec), out);
    return out;
}

std::string Tokenizer::decode(const std::vector<int32_t>& ids) const { // decode(ids, skip_special_tokens = false)
    const Impl& m = *impl_;
    const ByteMap& bm = bytemap();
    std::string out, pending; // pending: byte-level characters of ordinary tokens, flushed around added tokens
    auto flush = [&]() {
        if (!m.byte_level) { out += pending; pending.clear(); return; }
        const std::vector<Cp> cps = decode_utf8(pending);
        for (size_t i = 0; i + 1 < cps.size(); i++) { auto it = bm.u2b.find(cps[i].cp); if (it != bm.u2b.end()) out += (char)it->second; }
        pending.clear();
    };
    for (int32_t id : ids) {
        auto a = m.added_by_id.find(id);
        if (a != m.added_by_id.end()) { flush(); out += a->second; continue; }
        if (id >= 0 && (size_t)id < m.id_to_tok.size()) pending += m.id_to_tok[(size_t)id];
    }
    flush();
    return out;
}
std::string nfc_utf8(const std::string& text) { return nfc_utf8_impl(text); }
int32_t Tokenizer::vocab_size() const { return (int32_t)std::max(impl_->id_to_tok.size(), impl_->added_by_id.empty() ? (size_t)0 : (size_t)impl_->added_by_id.rbegin()->first + 1); }

} // namespace q3
