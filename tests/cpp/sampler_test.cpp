// CPU test of include/rwkv_sampler.hpp: a fixed scenario per sampler, printed value by value; tests/test_scheduler_cpp.py replays the
// same scenario through the Python mirrors (ai00_server_amd/harness.py, themselves restatements of sampler/*.rs) and compares.
#include <cstdio>

#include "../../include/rwkv_sampler.hpp"

static void dump(const char *tag, const rwkv::SamplerAdjust &a) {
    std::printf("%s", tag);
    for (size_t i = 0; i < a.tokens.size(); ++i) std::printf(" %u:%.9g", a.tokens[i], (double)a.values[i]);
    std::printf("\n");
}

int main() {
    const std::vector<uint32_t> prompt = {5, 9, 5, 3, 9, 9, 120, 5};
    const uint32_t picks[] = {9, 44, 5, 44, 44, 7, 120, 9};
    {
        rwkv::NucleusSampler s;
        s.presence_penalty = 0.4f; s.frequency_penalty = 0.25f; s.penalty_decay = 0.99f;
        s.bias = {{44, 1.5f}, {3, -2.0f}};
        s.init(prompt);
        dump("nucleus", s.adjustments());
        for (uint32_t t : picks) { s.update(t); dump("nucleus", s.adjustments()); }
        const rwkv::SamplerAdjust a = s.adjustments();
        const rwkv_sample_params p = s.params_for(0.25f, a);
        std::printf("params %g %d %g %g %zu %d %g\n", (double)p.top_p, p.top_k, (double)p.temperature, (double)p.uniform, p.n_adj, p.kind, (double)p.tau);
    }
    {
        rwkv::TypicalSampler s;                      // defaults: 0.3 / 0.3 / 0.99654026, tau 0.5
        s.tau = 0.7f;
        s.init(prompt);
        dump("typical", s.adjustments());
        for (uint32_t t : picks) { s.update(t); dump("typical", s.adjustments()); }
        const rwkv::SamplerAdjust a = s.adjustments();
        const rwkv_sample_params p = s.params_for(0.5f, a);
        std::printf("params %g %d %g %g %zu %d %g\n", (double)p.top_p, p.top_k, (double)p.temperature, (double)p.uniform, p.n_adj, p.kind, (double)p.tau);
    }
    {
        rwkv::MirostatSampler s(3.0f, 0.1f);
        const float surprises[] = {2.5f, 7.25f, 0.125f, 3.0f, 12.0f, 1.0f, 0.5f, 0.25f, 0.0f, 0.0f, 0.0f, 0.0f};
        std::printf("mirostat %.9g", (double)s.max_surprise);
        for (float x : surprises) { s.update(x); std::printf(" %.9g", (double)s.max_surprise); }
        std::printf("\n");
        const rwkv_sample_params p = s.params_for(0.75f, s.adjustments());
        std::printf("params %g %d %g %g %zu %d %.9g\n", (double)p.top_p, p.top_k, (double)p.temperature, (double)p.uniform, p.n_adj, p.kind, (double)p.tau);
    }
    return 0;
}
