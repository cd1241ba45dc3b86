// q3_common.h -- shared host-side helpers for the MI355X-native Qwen3-TTS engine (product code).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <stdexcept>
#include <mutex>
#include "../../include/q3tts_spec.h"

namespace q3 {

// thread-local last-error string surfaced through q3tts_last_error() (C ABI never throws / aborts)
void set_last_error(const std::string& s);
const char* last_error();

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };

// Process-wide lock around frame-graph construction (stream capture).  On ROCm 7.2 a synchronous hipMemcpy / hipMalloc / hipDeviceSynchronize on
// ANY thread while another thread captures fails ("would make the legacy stream depend on a capturing blocking stream") and invalidates the
// capture, thread-local capture mode notwithstanding; code that must allocate, copy or synchronise synchronously while engines may be running
// (voice registration, codec state export / import / reset, the ONNX executor) takes this lock.  Defined in engine.cpp.
std::mutex& capture_mutex();

#define Q3_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    throw ::q3::Error(std::string(#expr) + ": " + hipGetErrorString(e_) + " at " __FILE__ ":" + std::to_string(__LINE__)); } } while (0)
// launch-configuration failures (bad grid, LDS opt-in missing on this device, ...) are reported by hipGetLastError only
// (hipErrorNotReady is what a polling hipEventQuery / hipStreamQuery leaves behind on this thread: not a launch failure)
#define Q3_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess && e_ != hipErrorNotReady) { \
    throw ::q3::Error(std::string("kernel launch failed: ") + hipGetErrorString(e_) + " at " __FILE__ ":" + std::to_string(__LINE__)); } } while (0)
#define Q3_CHECK(cond, msg) do { if (!(cond)) throw ::q3::Error(std::string(msg) + " (" #cond ") at " __FILE__ ":" + std::to_string(__LINE__)); } while (0)

template <typename T> struct DevBuf {
    T* p = nullptr; size_t n = 0;
    DevBuf() = default;
    explicit DevBuf(size_t n_) { alloc(n_); }
    DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept { if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
    ~DevBuf() { release(); }
    void alloc(size_t n_) { release(); n = n_; if (n) Q3_HIP(hipMalloc((void**)&p, n * sizeof(T))); }
    void zero() { if (n) Q3_HIP(hipMemset(p, 0, n * sizeof(T))); }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
    void upload(const T* h, size_t cnt) { Q3_CHECK(cnt <= n, "upload overflow"); Q3_HIP(hipMemcpy(p, h, cnt * sizeof(T), hipMemcpyHostToDevice)); }
    void download(T* h, size_t cnt) const { Q3_CHECK(cnt <= n, "download overflow"); Q3_HIP(hipMemcpy(h, p, cnt * sizeof(T), hipMemcpyDeviceToHost)); }
};

// HIP-event timer for individual launches (bench.py's instrumented leg: roofline.achieved of the GEMV kernel)
struct LaunchTimer {
    std::vector<hipEvent_t> ev; size_t used = 0; double bytes = 0; long launches = 0;
    ~LaunchTimer() { for (auto e : ev) (void)hipEventDestroy(e); }
    void begin(hipStream_t st) {
        if (used + 2 > ev.size()) { size_t old = ev.size(); ev.resize(old + 2048); for (size_t i = old; i < ev.size(); i++) Q3_HIP(hipEventCreate(&ev[i])); }
        Q3_HIP(hipEventRecord(ev[used], st));
    }
    void end(hipStream_t st, double nbytes) { Q3_HIP(hipEventRecord(ev[used + 1], st)); used += 2; bytes += nbytes; launches++; }
    double collect_ms() { // call after the stream is synchronised
        double ms = 0;
        for (size_t i = 0; i + 1 < used; i += 2) { float t = 0; Q3_HIP(hipEventElapsedTime(&t, ev[i], ev[i + 1])); ms += t; }
        used = 0;
        return ms;
    }
};

} // namespace q3
