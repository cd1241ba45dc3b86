// codec.hip -- placeholder until the HIP codec decoder lands in this round (fails loudly, never falls back).
#include "codec.h"
namespace q3 {
struct CodecDecoder::Impl {};
CodecDecoder::CodecDecoder(const std::string&, int, int) { throw Error("HIP codec decoder not built yet"); }
CodecDecoder::~CodecDecoder() {}
int CodecDecoder::samples_per_frame() const { return 0; }
void CodecDecoder::reset(int) {}
int CodecDecoder::decode(hipStream_t, int, const int64_t*, int, bool, float*) { return -1; }
double CodecDecoder::flops_per_frame() const { return 0; }
}
