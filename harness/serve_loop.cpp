// serve_loop.cpp — the C++ scheduling core (include/rwkv_scheduler.hpp: slot choice, prefix cache, continuous batching)
// driving the real engine through include/rwkv_runtime.hpp, with the arg-max sampler (Nucleus top_k = 1, nucleus.rs:77-89).
// Scenario (printed as one line of token ids per stage, checked against the oracle by tests/test_gpu_parity.py):
//   1. request A = prompt0 is queued and runs alone for two device steps;
//   2. request B = prompt1 is queued while A is mid-flight and rides the next step (continuous batching);
//   3. both decode `n_new` greedy tokens and finish (states cached under their contents);
//   4. request C = prompt0 + A's tokens + `tail` continues from the cached state (Continue, prefix = all of A) and
//      decodes `n_new` more;
//   5. C's slot scores three choices (GenerateKind::Choose, calibrated) and gets its state back.
// Usage: serve_loop <model.st> <quant_layers> <quant_type> <max_batch> <chunk> <n_new> <prompt0 ...> / <prompt1 ...> / <tail ...>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>

#include "../include/rwkv_scheduler.hpp"

static uint32_t argmax(const std::vector<float> &lg) { return (uint32_t)(std::max_element(lg.begin(), lg.end()) - lg.begin()); }

int main(int argc, char **argv) {
    if (argc < 10) { std::fprintf(stderr, "usage: see header\n"); return 2; }
    try {
        std::ifstream f(argv[1], std::ios::binary);
        std::vector<uint8_t> st((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        const int ql = std::atoi(argv[2]), qt = std::atoi(argv[3]), B = std::atoi(argv[4]), chunk = std::atoi(argv[5]), n_new = std::atoi(argv[6]);
        std::vector<rwkv::Tokens> parts(1);
        for (int i = 7; i < argc; ++i) {
            if (!std::strcmp(argv[i], "/")) parts.emplace_back();
            else parts.back().push_back((uint32_t)std::strtoul(argv[i], nullptr, 10));
        }
        if (parts.size() != 3) throw std::invalid_argument("need prompt0 / prompt1 / tail");
        auto rt = rwkv::ModelBuilder(st.data(), st.size()).quant(ql, (rwkv::Quant)qt).build(B, chunk, rwkv::Precision::Fp16);
        rwkv::Scheduler<rwkv::Runtime> sched(rt);

        int a = -1, b = -1;
        if (sched.queue(parts[0], a) != rwkv::SlotResult::Success) throw std::runtime_error("queue A");
        int riders_first = sched.step();                                 // A alone
        if (sched.queue(parts[1], b) != rwkv::SlotResult::Success) throw std::runtime_error("queue B");
        int riders_second = sched.pending() ? sched.step() : 0;          // A (if it still has tokens) and B together
        while (sched.pending()) sched.step();
        rwkv::Tokens gen_a, gen_b;
        for (int i = 0; i < n_new; ++i) {                                // decode both slots in lock step
            const uint32_t ta = argmax(sched.request(a).output), tb = argmax(sched.request(b).output);
            gen_a.push_back(ta); gen_b.push_back(tb);
            sched.push(a, ta); sched.push(b, tb);
            if (sched.step() != 2) throw std::runtime_error("decode step did not carry both slots");
        }
        sched.finish(a); sched.finish(b);

        rwkv::Tokens c_tokens = parts[0];
        c_tokens.insert(c_tokens.end(), gen_a.begin(), gen_a.end());
        c_tokens.insert(c_tokens.end(), parts[2].begin(), parts[2].end());
        int c = -1;
        const rwkv::SlotResult rc = sched.queue(c_tokens, c);
        const size_t c_prefix = sched.request(c).prefix.size();
        while (sched.pending()) sched.step();
        rwkv::Tokens gen_c;
        for (int i = 0; i < n_new; ++i) {
            const uint32_t tc = argmax(sched.request(c).output);
            gen_c.push_back(tc);
            sched.push(c, tc);
            sched.step();
        }
        // 5. GenerateKind::Choose on C where it stands (run.rs:936-979): choices = the first three tokens of prompt1, the tail, and an
        //    empty one; calibrated against the initial state; the slot's state must come back bit-identical (device snapshot)
        const std::vector<float> before = sched.state(c);
        const std::vector<rwkv::Tokens> choices = {rwkv::Tokens(parts[1].begin(), parts[1].begin() + std::min<size_t>(3, parts[1].size())), parts[2], {}};
        const std::vector<float> ppl = sched.choose(c, choices, true);
        const bool restored = sched.state(c) == before;
        sched.finish(c);
        for (auto t : gen_a) std::printf("%u ", t);
        std::printf("\n");
        for (auto t : gen_b) std::printf("%u ", t);
        std::printf("\n");
        for (auto t : gen_c) std::printf("%u ", t);
        std::printf("\n");
        std::printf("meta %d %d %d %d %d %zu\n", riders_first, riders_second, a, c, (int)rc, c_prefix);
        std::printf("choose %.6f %.6f %s restored %d\n", ppl[0], ppl[1], std::isinf(ppl[2]) ? "inf" : "finite", (int)restored);
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
