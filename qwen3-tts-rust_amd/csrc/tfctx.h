// tfctx.h -- one model + one single-sequence context: the object behind q3tts_tf_* and the llama_* shim.
#pragma once
#include "transformer.h"
#include <memory>

namespace q3 {

class TfContext {
public:
    TfContext(std::shared_ptr<Transformer> model, int n_ctx) : model_(std::move(model)), n_ctx_(n_ctx) {
        const auto& hp = model_->hp();
        const int pages = (n_ctx + 63) / 64;
        kv_.reset(new KvPool(hp.n_layer, hp.n_kv, pages, 1, pages));
        const int mt = model_->max_tok();
        d_x_.alloc((size_t)mt * hp.n_embd); d_hid_.alloc((size_t)mt * hp.n_embd);
        d_seq_.alloc(mt); d_slot_.alloc(mt); d_pos_.alloc((size_t)4 * mt);
        Q3_HIP(hipStreamCreate(&st_));
    }
    ~TfContext() { if (st_) (void)hipStreamDestroy(st_); }
    void clear() { n_past_ = 0; }
    int n_past() const { return n_past_; }
    Transformer& model() { return *model_; }
    // x [ntok][n_embd] host, pos4 [ntok][4] host; outputs host (nullable)
    // want_rows (optional, [ntok]): logits are computed only for tokens whose flag is non-zero (rows of the others are left untouched)
    void eval(const float* x, const int32_t* pos4, int ntok, float* hidden_out, float* logits_out, int row0, int row1, const int8_t* want_rows = nullptr) {
        const auto& hp = model_->hp();
        Q3_CHECK(n_past_ + ntok <= n_ctx_, "context overflow");
        kv_->ensure(0, n_past_ + ntok);
        const int mt = model_->max_tok();
        for (int t0 = 0; t0 < ntok; t0 += mt) {
            const int n = std::min(mt, ntok - t0);
            std::vector<int32_t> seq(n, 0), slot(n);
            for (int i = 0; i < n; i++) slot[i] = n_past_ + i;
            d_x_.upload(x + (size_t)t0 * hp.n_embd, (size_t)n * hp.n_embd);
            d_seq_.upload(seq.data(), n); d_slot_.upload(slot.data(), n); d_pos_.upload(pos4 + (size_t)4 * t0, (size_t)4 * n);
            TokMeta tm{d_seq_.p, d_slot_.p, d_pos_.p};
            Transformer::Input in; in.x = d_x_.p; in.x_stride = hp.n_embd;
            model_->set_same_seq_tokens(true);
            model_->forward(st_, in, n, tm, kv_->view(), nullptr);
            const bool want_logits = logits_out && row1 > row0;
            const int r0 = row0 & ~31, nr = want_logits ? row1 - r0 : 0;
            if (want_logits && d_logits_.n < (size_t)n * nr) d_logits_.alloc((size_t)n * nr);
            if (!want_rows) model_->head(st_, 0, n, want_logits ? r0 : 0, nr, d_logits_.p, nr, nullptr, -1, d_hid_.p);
            else { // llama.cpp computes the output matrix for flagged rows only (batch.logits[i]); hidden rows come from one call
                model_->head(st_, 0, n, 0, 0, nullptr, 0, nullptr, -1, d_hid_.p);
                if (want_logits)
                    for (int i = 0; i < n; i++)
                        if (want_rows[t0 + i]) model_->head(st_, i, 1, r0, nr, d_logits_.p + (size_t)i * nr, nr, nullptr, -1, nullptr);
            }
            Q3_HIP(hipStreamSynchronize(st_));
            if (want_logits) {
                std::vector<float> tmp((size_t)n * nr);
                d_logits_.download(tmp.data(), tmp.size());
                for (int i = 0; i < n; i++)
                    if (!want_rows || want_rows[t0 + i])
                        std::copy(tmp.begin() + (size_t)i * nr + (row0 - r0), tmp.begin() + (size_t)i * nr + (row0 - r0) + (row1 - row0),
                                  logits_out + (size_t)(t0 + i) * (row1 - row0));
            }
            if (hidden_out) d_hid_.download(hidden_out + (size_t)t0 * hp.n_embd, (size_t)n * hp.n_embd);
            n_past_ += n;
        }
    }
private:
    std::shared_ptr<Transformer> model_;
    int n_ctx_, n_past_ = 0;
    std::unique_ptr<KvPool> kv_;
    DevBuf<float> d_x_, d_hid_, d_logits_;
    DevBuf<int32_t> d_seq_, d_slot_, d_pos_;
    hipStream_t st_ = nullptr;
};

} // namespace q3
