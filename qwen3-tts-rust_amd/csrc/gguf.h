// gguf.h -- product-side GGUF reader (C++), independent of the oracle's C reader.
// Container layout as parsed by the reference's own mini reader (/root/reference/src/assets_manager.rs:33-148)
// plus array-typed metadata per the public GGUF spec [EXT] (needed for llama.cpp-format model files).
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace q3 {

struct GgufTensor {
    std::string name;
    int n_dims = 0;
    int64_t ne[4] = {1, 1, 1, 1};
    int type = 0;
    uint64_t offset = 0;
    const uint8_t* data = nullptr;
    size_t nbytes = 0;
    int64_t rows() const { return ne[1] * ne[2] * ne[3]; }
};

struct GgufValue {
    int type = -1;               // gguf value type
    uint64_t u = 0; int64_t i = 0; double f = 0; std::string s;
    int arr_type = -1; std::vector<int64_t> arr_i; std::vector<double> arr_f; uint64_t arr_n = 0;
    int64_t as_int(int64_t def) const;
    double as_float(double def) const;
};

class Gguf {
public:
    explicit Gguf(const std::string& path);   // throws q3::Error
    ~Gguf();
    Gguf(const Gguf&) = delete; Gguf& operator=(const Gguf&) = delete;
    const GgufTensor* find(const std::string& name) const;
    const GgufTensor& need(const std::string& name) const;
    const GgufValue* kv(const std::string& key) const;
    int64_t kv_int(const std::string& key, int64_t def) const;
    double kv_float(const std::string& key, double def) const;
    std::string kv_str(const std::string& key, const std::string& def) const;
    uint32_t version = 0;
    std::vector<GgufTensor> tensors;
    std::map<std::string, GgufValue> kvs;
    static size_t row_bytes(int type, int64_t k);
private:
    uint8_t* map_ = nullptr; size_t size_ = 0; int fd_ = -1;
    std::map<std::string, size_t> index_;
};

} // namespace q3
