// decode_loop.cpp — C++ mirror of ai00-core's `infer` task + greedy `process` loop (run.rs:1072-1162, 788-1020)
// on top of include/rwkv_runtime.hpp.  Usage:
//   decode_loop <model.st> <quant_layers> <quant_type> <max_batch> <chunk> <n_new> <prompt tokens of slot 0> [/ <slot 1> ...]
// Prints one line of greedy token ids per slot (arg-max == Nucleus top_k=1, nucleus.rs:77-89; token 0 stops, run.rs:855).
// With RWKV_DECODE_SAMPLER=nucleus|typical|mirostat the same loop samples on the device instead (rwkv_infer_sample) with the
// host-side sampler state of include/rwkv_sampler.hpp: init(prompt), then per token adjustments -> infer_sample -> update; the
// uniform draw of (step, slot) is frac(0.137 + 0.618034 (step + 1) + 0.31 slot) so that a test can replay it.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>

#include <cmath>
#include <string>

#include "../include/rwkv_runtime.hpp"
#include "../include/rwkv_sampler.hpp"

int main(int argc, char **argv) {
    if (argc < 8) { std::fprintf(stderr, "usage: see header\n"); return 2; }
    try {
        std::ifstream f(argv[1], std::ios::binary);
        std::vector<uint8_t> st((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        const int ql = std::atoi(argv[2]), qt = std::atoi(argv[3]), B = std::atoi(argv[4]), chunk = std::atoi(argv[5]), n_new = std::atoi(argv[6]);
        std::vector<std::vector<uint32_t>> prompts(1);
        for (int i = 7; i < argc; ++i) {
            if (!std::strcmp(argv[i], "/")) prompts.emplace_back();
            else prompts.back().push_back((uint32_t)std::strtoul(argv[i], nullptr, 10));
        }
        auto rt = rwkv::ModelBuilder(st.data(), st.size()).quant(ql, (rwkv::Quant)qt).build(B, chunk, rwkv::Precision::Fp16);
        const size_t V = (size_t)rt.info.num_vocab;
        std::vector<std::vector<uint32_t>> pending(B), gen(B);
        std::vector<bool> live(B, false);
        for (size_t s = 0; s < prompts.size() && (int)s < B; ++s) {
            pending[s] = prompts[s].empty() ? std::vector<uint32_t>{0} : prompts[s];      // run.rs:489-492
            live[s] = true;
        }
        const char *mode_env = std::getenv("RWKV_DECODE_SAMPLER");
        const std::string mode = mode_env ? mode_env : "";
        if (!mode.empty()) {
            std::vector<rwkv::NucleusSampler> nuc((size_t)B);
            std::vector<rwkv::TypicalSampler> typ((size_t)B);
            std::vector<rwkv::MirostatSampler> mir((size_t)B);
            for (int b = 0; b < B; ++b) if (live[b]) { nuc[(size_t)b].init(pending[b]); typ[(size_t)b].init(pending[b]); }
            for (int step = 0; step < n_new; ++step) {
                rwkv::RnnInput input;
                input.batches.resize((size_t)B);
                std::vector<rwkv::SamplerAdjust> adj((size_t)B);
                std::vector<rwkv_sample_params> sp((size_t)B);
                for (int b = 0; b < B; ++b) {
                    const float u = std::fmod(0.137f + 0.618034f * (float)(step + 1) + 0.31f * (float)b, 1.0f);
                    if (live[b]) input.batches[(size_t)b].tokens = pending[b];
                    if (mode == "typical") { adj[(size_t)b] = typ[(size_t)b].adjustments(); sp[(size_t)b] = typ[(size_t)b].params_for(u, adj[(size_t)b]); }
                    else if (mode == "mirostat") sp[(size_t)b] = mir[(size_t)b].params_for(u, adj[(size_t)b]);
                    else { adj[(size_t)b] = nuc[(size_t)b].adjustments(); sp[(size_t)b] = nuc[(size_t)b].params_for(u, adj[(size_t)b]); }
                }
                std::vector<rwkv::Runtime::Sampled> got((size_t)B);
                while (input.num_token() > 0) {
                    auto out = rt.infer_sample(input, sp);
                    for (int b = 0; b < B; ++b) if (out[(size_t)b].emitted) got[(size_t)b] = out[(size_t)b];
                }
                for (int b = 0; b < B; ++b) {
                    if (!live[b]) continue;
                    const uint32_t tok = got[(size_t)b].token;
                    if (mode == "typical") typ[(size_t)b].update(tok);
                    else if (mode == "mirostat") mir[(size_t)b].update(got[(size_t)b].prob);       // the token surprise
                    else nuc[(size_t)b].update(tok);
                    gen[b].push_back(tok);
                    pending[b] = {tok};
                }
            }
            for (size_t s = 0; s < prompts.size() && (int)s < B; ++s) {
                for (auto t : gen[s]) std::printf("%u ", t);
                std::printf("\n");
            }
            return 0;
        }
        bool any = true;
        while (any) {
            rwkv::RnnInput input;                                   // one request per slot (run.rs:1121-1132)
            input.batches.resize(B);
            for (int b = 0; b < B; ++b) if (live[b]) input.batches[b].tokens = pending[b];
            std::vector<rwkv::RnnOutputBatch> last(B);
            while (input.num_token() > 0) {                          // run.rs:1134-1156
                auto out = rt.infer(input);
                for (int b = 0; b < B; ++b) if (!out[b].empty()) last[b] = std::move(out[b]);
            }
            any = false;
            for (int b = 0; b < B; ++b) {
                if (!live[b]) continue;
                const float *lg = last[b].data() + (last[b].size() / V - 1) * V;
                const uint32_t tok = (uint32_t)(std::max_element(lg, lg + V) - lg);
                if (tok == 0 || (int)gen[b].size() >= n_new) { live[b] = false; continue; }
                gen[b].push_back(tok);
                pending[b] = {tok};
                if ((int)gen[b].size() >= n_new) live[b] = false; else any = true;
            }
        }
        for (size_t s = 0; s < prompts.size() && (int)s < B; ++s) {
            for (auto t : gen[s]) std::printf("%u ", t);
            std::printf("\n");
        }
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
