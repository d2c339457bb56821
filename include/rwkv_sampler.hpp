// rwkv_sampler.hpp — host-side STATE of ai00-core's samplers (crates/ai00-core/src/sampler/*.rs) for the on-device sampling
// front-end (rwkv_infer_sample).  The device does what `Sampler::sample` does to a probability row (sort,
// top-k, top-p / tau / max_surprise, temperature, inverse CDF); what survives on the host is what the reference keeps between
// tokens:
//   NucleusSampler   nucleus.rs:13-122    penalty map: init (:49-59), transform (:61-67), the update at the end of sample (:104-119)
//   TypicalSampler   typical.rs:11-140    the same penalty state machine (:47-58, :62-68, :122-133); tau / top_k / temperature
//   MirostatSampler  mirostat.rs:11-90    max_surprise, updated from the token surprise the device returns (:85-87)
// `params_for()` fills the rwkv_sample_params of one slot: `-penalty` (transform) and `bias` (run.rs:681-683) merged into one
// sparse adjustment list, plus the uniform draw `fastrand::f32()` would make.  All arithmetic is f32, as in the reference.
// Header-only; needs rwkv_abi.h only for the struct.
#pragma once
#include <cmath>
#include <cstdint>
#include <map>
#include <vector>

#include "rwkv_abi.h"

namespace rwkv {

struct SamplerAdjust {                 // keeps the arrays rwkv_sample_params points into alive until the call returns
    std::vector<uint32_t> tokens;
    std::vector<float> values;
};

class NucleusSampler {
   public:
    float top_p = 0.5f;                // NucleusParams defaults, nucleus.rs:13-26
    int32_t top_k = 128;
    float temperature = 1.0f, presence_penalty = 0.3f, frequency_penalty = 0.3f, penalty_decay = 0.99654026f;
    std::map<uint32_t, float> penalties;             // NucleusState (a HashMap there; ordered here so adjustment lists are deterministic)
    std::map<uint32_t, float> bias;                  // GenerateRequest::bias (run.rs:681-683)

    virtual ~NucleusSampler() = default;
    // nucleus.rs:49-59: walk the prompt backwards, the most recent token first
    void init(const std::vector<uint32_t> &model_tokens) {
        float index = 0.f;
        for (auto it = model_tokens.rbegin(); it != model_tokens.rend(); ++it, index += 1.f) {
            auto f = penalties.find(*it);
            float penalty = f == penalties.end() ? presence_penalty : f->second;
            penalty += frequency_penalty * std::pow(penalty_decay, index);
            penalties[*it] = penalty;
        }
    }
    // what `transform` (nucleus.rs:61-67) and the bias loop (run.rs:681-683) add to the logits, duplicates merged
    SamplerAdjust adjustments() const {
        std::map<uint32_t, float> adj;
        for (auto &kv : penalties) adj[kv.first] = -kv.second;
        for (auto &kv : bias) adj[kv.first] += kv.second;
        SamplerAdjust a;
        for (auto &kv : adj) { a.tokens.push_back(kv.first); a.values.push_back(kv.second); }
        return a;
    }
    // the tail of `sample` (nucleus.rs:104-119) with the token the device picked
    void update(uint32_t token) {
        for (auto &kv : penalties) kv.second *= penalty_decay;
        auto f = penalties.find(token);
        penalties[token] = f == penalties.end() ? presence_penalty : f->second + frequency_penalty;
    }
    virtual rwkv_sample_params params_for(float uniform, const SamplerAdjust &adj, const uint8_t *allow = nullptr) const {
        return rwkv_sample_params{top_p, top_k, temperature, uniform, adj.tokens.empty() ? nullptr : adj.tokens.data(),
                                  adj.values.empty() ? nullptr : adj.values.data(), adj.tokens.size(), RWKV_SAMPLER_NUCLEUS, 0.f, allow};
    }
};

class TypicalSampler : public NucleusSampler {       // typical.rs: TypicalParams defaults tau 0.5, top_k 128, temperature 1.0
   public:
    float tau = 0.5f;
    rwkv_sample_params params_for(float uniform, const SamplerAdjust &adj, const uint8_t *allow = nullptr) const override {
        rwkv_sample_params p = NucleusSampler::params_for(uniform, adj, allow);
        p.top_p = 0.f;
        p.kind = RWKV_SAMPLER_TYPICAL;
        p.tau = tau;
        return p;
    }
};

class MirostatSampler {                              // mirostat.rs:11-36: tau (target surprise) 3.0, rate 0.1, max_surprise starts at 2 tau
   public:
    explicit MirostatSampler(float tau = 3.0f, float rate_ = 0.1f) : target(tau), rate(rate_), max_surprise(2.0f * tau) {}
    float target, rate, max_surprise;
    void init(const std::vector<uint32_t> &) {}      // mirostat.rs:38
    SamplerAdjust adjustments() const { return {}; } // transform is a no-op (mirostat.rs:40)
    // mirostat.rs:85-87 with the token surprise rwkv_infer_sample returns in out_probs
    void update(float token_surprise) {
        const float error = token_surprise - target;
        max_surprise = std::fmin(max_surprise - rate * error, 4.0f * target);
    }
    rwkv_sample_params params_for(float uniform, const SamplerAdjust &, const uint8_t *allow = nullptr) const {
        return rwkv_sample_params{0.f, 1, 1.0f, uniform, nullptr, nullptr, 0, RWKV_SAMPLER_MIROSTAT, max_surprise, allow};
    }
};

}  // namespace rwkv
