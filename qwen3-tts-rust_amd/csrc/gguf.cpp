// gguf.cpp -- see gguf.h
#include "gguf.h"
#include "q3_common.h"
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace q3 {

static thread_local std::string g_last_error;
void set_last_error(const std::string& s) { g_last_error = s; }
const char* last_error() { return g_last_error.c_str(); }

namespace {
struct Cursor {
    const uint8_t* p; const uint8_t* end;
    template <typename T> T get() {
        if (sizeof(T) > (size_t)(end - p)) throw Error("truncated GGUF header");
        T v; std::memcpy(&v, p, sizeof(T)); p += sizeof(T); return v;
    }
    std::string str() {
        uint64_t n = get<uint64_t>();
        if (n > (uint64_t)(end - p)) throw Error("truncated GGUF string"); // (compare sizes: p + n could wrap)
        std::string s((const char*)p, (size_t)n); p += n; return s;
    }
};
int scalar_size(int t) {
    switch (t) { case 0: case 1: case 7: return 1; case 2: case 3: return 2; case 4: case 5: case 6: return 4;
                 case 10: case 11: case 12: return 8; default: return -1; }
}
void read_scalar(Cursor& c, int t, GgufValue& v, bool into_array) {
    double f = 0; int64_t i = 0; bool is_f = false;
    switch (t) {
        case 0: i = c.get<uint8_t>(); break;   case 1: i = c.get<int8_t>(); break;
        case 2: i = c.get<uint16_t>(); break;  case 3: i = c.get<int16_t>(); break;
        case 4: i = c.get<uint32_t>(); break;  case 5: i = c.get<int32_t>(); break;
        case 6: f = c.get<float>(); is_f = true; break;
        case 7: i = c.get<uint8_t>(); break;
        case 10: i = (int64_t)c.get<uint64_t>(); break; case 11: i = c.get<int64_t>(); break;
        case 12: f = c.get<double>(); is_f = true; break;
        default: throw Error("Unknown GGUF value type: " + std::to_string(t));
    }
    if (into_array) { if (is_f) v.arr_f.push_back(f); else v.arr_i.push_back(i); }
    else { v.i = i; v.u = (uint64_t)i; v.f = is_f ? f : (double)i; if (is_f) v.i = (int64_t)f; }
}
} // namespace

int64_t GgufValue::as_int(int64_t def) const { return (type >= 0 && type != 8 && type != 9) ? i : def; }
double GgufValue::as_float(double def) const { return (type >= 0 && type != 8 && type != 9) ? f : def; }

size_t Gguf::row_bytes(int type, int64_t k) {
    const int64_t blk = (type == Q3_T_Q8_0) ? 32 : (type == Q3_T_Q5_K || type == Q3_T_Q6_K) ? 256 : 1;
    if (k <= 0 || k % blk != 0) return 0; // a partial block would silently truncate the row
    switch (type) {
        case Q3_T_F32: return (size_t)k * 4;
        case Q3_T_F16: case Q3_T_BF16: return (size_t)k * 2;
        case Q3_T_Q8_0: return (size_t)(k / 32) * 34;
        case Q3_T_Q5_K: return (size_t)(k / 256) * 176;
        case Q3_T_Q6_K: return (size_t)(k / 256) * 210;
        default: return 0;
    }
}

Gguf::Gguf(const std::string& path) {
    fd_ = ::open(path.c_str(), O_RDONLY);
    if (fd_ < 0) throw Error("cannot open " + path);
    struct stat st;
    if (fstat(fd_, &st) != 0 || st.st_size <= 0) { ::close(fd_); fd_ = -1; throw Error("cannot stat " + path); }
    size_ = (size_t)st.st_size;
    void* m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
    if (m == MAP_FAILED) { ::close(fd_); fd_ = -1; throw Error("mmap failed: " + path); }
    map_ = (uint8_t*)m;
    try {
        if (size_ < 24 || std::memcmp(map_, "GGUF", 4) != 0) throw Error("Not a GGUF file");
        Cursor c{map_ + 4, map_ + size_};
        version = c.get<uint32_t>();
        if (version < 2) throw Error("Unsupported GGUF version: " + std::to_string(version));
        uint64_t nt = c.get<uint64_t>(), nkv = c.get<uint64_t>();
        if (nt > (1u << 20) || nkv > (1u << 20)) throw Error("implausible GGUF counts");
        uint64_t alignment = 32;
        for (uint64_t k = 0; k < nkv; k++) {
            std::string key = c.str();
            GgufValue v; v.type = (int)c.get<uint32_t>();
            if (v.type == 8) v.s = c.str();
            else if (v.type == 9) {
                v.arr_type = (int)c.get<uint32_t>(); v.arr_n = c.get<uint64_t>();
                if (v.arr_type == 8) { for (uint64_t j = 0; j < v.arr_n; j++) (void)c.str(); }
                else { if (scalar_size(v.arr_type) < 0) throw Error("Unknown GGUF array type"); for (uint64_t j = 0; j < v.arr_n; j++) read_scalar(c, v.arr_type, v, true); }
            } else read_scalar(c, v.type, v, false);
            if (key == "general.alignment" && v.type == 4) {
                alignment = v.u;
                if (alignment == 0 || (alignment & (alignment - 1)) != 0 || alignment > (1u << 20)) throw Error("general.alignment must be a power of two");
            }
            kvs[key] = std::move(v);
        }
        tensors.resize((size_t)nt);
        for (auto& t : tensors) {
            t.name = c.str();
            t.n_dims = (int)c.get<uint32_t>();
            if (t.n_dims > 4) throw Error("tensor " + t.name + ": too many dims");
            for (int d = 0; d < t.n_dims; d++) {
                const uint64_t ne = c.get<uint64_t>();
                if (ne == 0 || ne > ((uint64_t)1 << 40)) throw Error("tensor " + t.name + ": implausible dimension");
                t.ne[d] = (int64_t)ne;
            }
            t.type = (int)c.get<uint32_t>();
            t.offset = c.get<uint64_t>();
        }
        size_t pos = (size_t)(c.p - map_);
        size_t data_start = pos + (alignment - pos % alignment) % alignment;
        for (size_t i = 0; i < tensors.size(); i++) {
            auto& t = tensors[i];
            size_t rb = row_bytes(t.type, t.ne[0]);
            if (!rb) throw Error("Unsupported tensor type or row length: type " + std::to_string(t.type) + ", ne0 " + std::to_string(t.ne[0]) + " (" + t.name + ")");
            size_t rows = 1;
            for (int d = 1; d < 4; d++) if (__builtin_mul_overflow(rows, (size_t)t.ne[d], &rows)) throw Error("tensor " + t.name + ": size overflow");
            size_t end_off = 0;
            if (__builtin_mul_overflow(rb, rows, &t.nbytes) || __builtin_add_overflow((size_t)t.offset, t.nbytes, &end_off) ||
                __builtin_add_overflow(end_off, data_start, &end_off) || end_off > size_ || data_start > size_)
                throw Error("tensor " + t.name + " out of file");
            t.data = map_ + data_start + t.offset;
            index_[t.name] = i;
        }
    } catch (...) {
        munmap(map_, size_); ::close(fd_); map_ = nullptr; fd_ = -1;
        throw;
    }
}
Gguf::~Gguf() { if (map_) munmap(map_, size_); if (fd_ >= 0) ::close(fd_); }
const GgufTensor* Gguf::find(const std::string& name) const { auto it = index_.find(name); return it == index_.end() ? nullptr : &tensors[it->second]; }
const GgufTensor& Gguf::need(const std::string& name) const { auto* t = find(name); if (!t) throw Error("missing tensor " + name); return *t; }
const GgufValue* Gguf::kv(const std::string& key) const { auto it = kvs.find(key); return it == kvs.end() ? nullptr : &it->second; }
int64_t Gguf::kv_int(const std::string& key, int64_t def) const { auto* v = kv(key); return v ? v->as_int(def) : def; }
double Gguf::kv_float(const std::string& key, double def) const { auto* v = kv(key); return v ? v->as_float(def) : def; }
std::string Gguf::kv_str(const std::string& key, const std::string& def) const { auto* v = kv(key); return (v && v->type == 8) ? v->s : def; }

} // namespace q3
