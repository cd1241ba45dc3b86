// tokenizer.h -- HuggingFace tokenizer.json (byte-level BPE, Qwen2 family) reader + encoder / decoder; see tokenizer.cpp.
// Mirrors /root/reference/src/utils/tokenizer.rs: Tokenizer::{load, encode, decode}.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace q3 {

class Tokenizer {
public:
    explicit Tokenizer(const std::string& tokenizer_json_path);   // throws q3::Error ("Failed to load tokenizer: ...")
    ~Tokenizer();
    Tokenizer(const Tokenizer&) = delete; Tokenizer& operator=(const Tokenizer&) = delete;
    std::vector<int32_t> encode(const std::string& text) const;   // add_special_tokens = false (utils/tokenizer.rs:18)
    std::string decode(const std::vector<int32_t>& ids) const;    // skip_special_tokens = false (utils/tokenizer.rs:29)
    int32_t vocab_size() const;
private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

// NFC of a UTF-8 string (UAX #15; tables of unicode_tables.h): what the tokenizer's normaliser applies between added tokens
std::string nfc_utf8(const std::string& text);

} // namespace q3
