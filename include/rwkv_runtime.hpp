// rwkv_runtime.hpp — header-only C++17 mirror of the `web_rwkv` items ai00-core uses, on top of rwkv_abi.h.
// Same names and argument meaning as the reference's Rust call sites (crates/ai00-core/src):
//   Loader::info                    lib.rs:587            ModelBuilder(...).quant().lora().build()   lib.rs:484-516
//   Runtime::infer(RnnInput&)       run.rs:1143           RnnInput / RnnInputBatch / RnnOption        run.rs:1128-1132
//   State::{init,load,back,read,write}  run.rs:477,1099-1106     softmax(runtime, rows)               run.rs:1179
//   State::{embed,embed_async,sync}  the `/embeddings` read-back (docs/doc-api/openai.md:376-437): one layer's WKV rows of a slot
//   Tokenizer::{encode,decode}      run.rs:157,856
// Errors surface as rwkv::Error (std::runtime_error) carrying the rwkv_status — the analogue of `anyhow ?`.
#pragma once
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "rwkv_abi.h"

namespace rwkv {

struct Error : std::runtime_error {
    rwkv_status code;
    Error(rwkv_status c) : std::runtime_error(std::string("rwkv: ") + rwkv_last_error()), code(c) {}
};
inline void check(rwkv_status s) { if (s != RWKV_OK) throw Error(s); }

enum class RnnOption : int32_t { Last = RWKV_OPTION_LAST, Full = RWKV_OPTION_FULL, None = RWKV_OPTION_NONE };
enum class Quant : int32_t { None = RWKV_QUANT_NONE, Int8 = RWKV_QUANT_INT8, NF4 = RWKV_QUANT_NF4 };
// reload.rs:89-94; Fp16Raw is this library's extension (ABI 7): f16 operands on EVERY launch — the fastest mode, outside 1e-3 at 32 layers
enum class Precision : int32_t { Fp16 = RWKV_PRECISION_FP16, Fp32 = RWKV_PRECISION_FP32, Fp16Raw = RWKV_PRECISION_FP16_RAW };
using ModelInfo = rwkv_model_info;

struct Loader {
    static ModelInfo info(const uint8_t *st, size_t len) { ModelInfo i{}; check(rwkv_model_info_from_st(st, len, &i)); return i; }
};

struct RnnInputBatch {                      // RnnInputBatch::new(tokens, option)
    std::vector<uint32_t> tokens;
    RnnOption option = RnnOption::Last;
};
struct RnnInput {                           // RnnInput::new(batches, token_chunk_size)
    std::vector<RnnInputBatch> batches;
    size_t num_token() const { size_t n = 0; for (auto &b : batches) n += b.tokens.size(); return n; }
};
using RnnOutputBatch = std::vector<float>;  // n_rows * num_vocab floats (empty: nothing emitted)

class Runtime;
class TensorGpu {                            // device-resident state snapshot (`state.read`)
   public:
    explicit TensorGpu(rwkv_dstate *h) : h_(h, rwkv_dstate_free) {}
    const rwkv_dstate *get() const { return h_.get(); }
   private:
    std::shared_ptr<rwkv_dstate> h_;         // clone == share, like the Rust `backed.clone()` (run.rs:962)
};

// float32 block in pinned host memory (rwkv_host_alloc): where asynchronous read-backs (State::embed_async) land
class PinnedBuffer {
   public:
    explicit PinnedBuffer(size_t n) : n_(n) {
        void *p = nullptr;
        check(rwkv_host_alloc((n ? n : 1) * sizeof(float), &p));
        p_.reset((float *)p, rwkv_host_free);
    }
    float *data() const { return p_.get(); }
    size_t size() const { return n_; }
   private:
    std::shared_ptr<float> p_;
    size_t n_;
};

class State {
   public:
    explicit State(rwkv_engine *e) : e_(e) {}
    std::vector<size_t> shape() const { std::vector<size_t> s(4); rwkv_state_shape(e_, s.data()); return s; }
    std::vector<float> init() const { std::vector<float> v(rwkv_state_len(e_)); check(rwkv_state_init(e_, v.data())); return v; }
    void load(const std::vector<float> &t, int batch) {
        if (t.size() != rwkv_state_len(e_)) throw std::invalid_argument("state tensor has the wrong size");
        check(rwkv_state_load(e_, batch, t.data()));
    }
    std::vector<float> back(int batch) { std::vector<float> v(rwkv_state_len(e_)); check(rwkv_state_back(e_, batch, v.data())); return v; }
    TensorGpu read(int batch) { rwkv_dstate *h = nullptr; check(rwkv_state_read(e_, batch, &h)); return TensorGpu(h); }
    void write(const TensorGpu &t, int batch) { check(rwkv_state_write(e_, batch, t.get())); }
    // one layer's WKV rows of a slot, [head_size][num_emb] floats (rwkv_state_shape = [C, N + 2, L, 1]: the N rows between the two
    // token-shift rows) — what `/embeddings` returns for the chosen layer
    size_t layer_len() const { auto s = shape(); return s[0] * (s[1] - 2); }
    void embed(int layer, int batch, float *dst) { check(rwkv_state_back_layer(e_, batch, layer, dst)); }
    // not waited for: `dst` is pinned memory (PinnedBuffer), valid after sync(); the slot may take its next request at once
    void embed_async(int layer, int batch, float *dst) { check(rwkv_state_back_layer_async(e_, batch, layer, dst)); }
    void sync() { check(rwkv_state_sync(e_)); }
   private:
    rwkv_engine *e_;
};

class Runtime {
   public:
    explicit Runtime(rwkv_engine *e) : e_(e, rwkv_engine_destroy), state(e) {
        check(rwkv_engine_info(e, &info));
        max_batch = rwkv_engine_max_batch(e);
        token_chunk_size = rwkv_engine_token_chunk_size(e);      // what the engine was built with: sizes the buffers of Full requests
    }
    // runtime.infer(input) -> output; `input` is consumed in place (tokens drained by n_consumed)
    std::vector<RnnOutputBatch> infer(RnnInput &input) {
        if ((int)input.batches.size() != max_batch) throw std::invalid_argument("RnnInput must have max_batch entries");
        std::vector<rwkv_slot_input> in(max_batch);
        std::vector<rwkv_slot_output> out(max_batch);
        std::vector<RnnOutputBatch> bufs(max_batch);
        for (int b = 0; b < max_batch; ++b) {
            auto &ib = input.batches[b];
            // one call emits at most token_chunk_size rows for a slot however long its pending token list is (an 8k-token Full
            // request must not allocate 8k x V floats per call); slots without tokens and state-only slots need no buffer
            const size_t rows = ib.tokens.empty() || ib.option == RnnOption::None ? 0
                               : ib.option == RnnOption::Full ? std::min(ib.tokens.size(), (size_t)token_chunk_size) : 1;
            bufs[b].resize(rows * (size_t)info.num_vocab);
            in[b] = rwkv_slot_input{ib.tokens.data(), ib.tokens.size(), (int32_t)ib.option, 0};
            out[b] = rwkv_slot_output{rows ? bufs[b].data() : nullptr, rows, 0, 0};
        }
        check(rwkv_infer(e_.get(), in.data(), out.data()));
        for (int b = 0; b < max_batch; ++b) {
            auto &t = input.batches[b].tokens;
            t.erase(t.begin(), t.begin() + (long)out[b].n_consumed);
            bufs[b].resize(out[b].n_rows * (size_t)info.num_vocab);
        }
        return bufs;
    }
    // rwkv_infer_sample: the on-device sampling front-end.  Slots whose pending tokens this call exhausts get a
    // sampled token instead of a logits row; `sp[b]` comes from a host-side sampler (include/rwkv_sampler.hpp).  `input` is
    // consumed in place like infer().
    struct Sampled { bool emitted = false; uint32_t token = 0; float prob = 0.f; };
    std::vector<Sampled> infer_sample(RnnInput &input, const std::vector<rwkv_sample_params> &sp) {
        if ((int)input.batches.size() != max_batch || (int)sp.size() != max_batch) throw std::invalid_argument("infer_sample: need max_batch entries");
        std::vector<rwkv_slot_input> in((size_t)max_batch);
        for (int b = 0; b < max_batch; ++b) {
            auto &ib = input.batches[(size_t)b];
            in[(size_t)b] = rwkv_slot_input{ib.tokens.data(), ib.tokens.size(), (int32_t)RnnOption::Last, 0};
        }
        std::vector<uint32_t> tok((size_t)max_batch);
        std::vector<float> prob((size_t)max_batch);
        std::vector<uint8_t> emitted((size_t)max_batch);
        std::vector<size_t> consumed((size_t)max_batch);
        check(rwkv_infer_sample(e_.get(), in.data(), sp.data(), tok.data(), prob.data(), emitted.data(), consumed.data()));
        std::vector<Sampled> out((size_t)max_batch);
        for (int b = 0; b < max_batch; ++b) {
            auto &t = input.batches[(size_t)b].tokens;
            t.erase(t.begin(), t.begin() + (long)consumed[(size_t)b]);
            out[(size_t)b] = Sampled{emitted[(size_t)b] != 0, tok[(size_t)b], prob[(size_t)b]};
        }
        return out;
    }
    rwkv_engine *raw() const { return e_.get(); }
    ModelInfo info{};
    int max_batch = 0;
    int token_chunk_size = 128;                  // tokens one infer call consumes at most (bounds the rows of a Full request)
   private:
    std::shared_ptr<rwkv_engine> e_;
   public:
    State state;
};

class ModelBuilder {
   public:
    ModelBuilder(const uint8_t *st, size_t len, int adapter = RWKV_ADAPTER_AUTO) { d_.st_bytes = st; d_.st_len = len; d_.adapter = adapter; }
    ModelBuilder &quant(int layers, Quant q) { d_.quant_layers = layers; d_.quant_type = (int32_t)q; return *this; }
    ModelBuilder &lora(const uint8_t *st, size_t len, float alpha) { lora_.push_back({st, len, alpha}); return *this; }
    Runtime build(int max_batch = 8, int token_chunk_size = 128, Precision p = Precision::Fp16) {
        d_.max_batch = max_batch; d_.token_chunk_size = token_chunk_size; d_.precision = (int32_t)p;
        d_.lora = lora_.empty() ? nullptr : lora_.data(); d_.n_lora = lora_.size();
        rwkv_engine *e = nullptr;
        check(rwkv_engine_create(&d_, &e));
        return Runtime(e);
    }
   private:
    rwkv_load_desc d_{};
    std::vector<rwkv_lora_desc> lora_;
};

inline std::vector<std::vector<float>> softmax(Runtime &rt, const std::vector<std::vector<float>> &rows) {
    for (const auto &r : rows)                    // rwkv_softmax reads num_vocab floats per row: a short row would be read out of bounds
        if (r.size() != (size_t)rt.info.num_vocab) throw std::invalid_argument("softmax: every row must hold num_vocab values");
    std::vector<std::vector<float>> out(rows.size());
    std::vector<const float *> pi(rows.size());
    std::vector<float *> po(rows.size());
    for (size_t i = 0; i < rows.size(); ++i) { out[i].resize(rows[i].size()); pi[i] = rows[i].data(); po[i] = out[i].data(); }
    check(rwkv_softmax(rt.raw(), pi.data(), po.data(), rows.size()));
    return out;
}

class Tokenizer {
   public:
    explicit Tokenizer(const std::string &vocab_json) {
        rwkv_tokenizer *t = nullptr;
        check(rwkv_tokenizer_create(vocab_json.data(), vocab_json.size(), &t));
        t_.reset(t, rwkv_tokenizer_destroy);
    }
    std::vector<uint32_t> encode(const std::string &text) const {
        int64_t n = rwkv_tokenizer_encode(t_.get(), (const uint8_t *)text.data(), text.size(), nullptr, 0);
        if (n < 0) throw Error((rwkv_status)n);
        std::vector<uint32_t> v((size_t)n);
        rwkv_tokenizer_encode(t_.get(), (const uint8_t *)text.data(), text.size(), v.data(), v.size());
        return v;
    }
    std::string decode(const std::vector<uint32_t> &toks) const {
        int64_t n = rwkv_tokenizer_decode(t_.get(), toks.data(), toks.size(), nullptr, 0);
        if (n < 0) throw Error((rwkv_status)n);
        std::string s((size_t)n, '\0');
        rwkv_tokenizer_decode(t_.get(), toks.data(), toks.size(), (uint8_t *)s.data(), s.size());
        return s;
    }
   private:
    std::shared_ptr<rwkv_tokenizer> t_;
};

}  // namespace rwkv
