// A deterministic stand-in for rwkv::Runtime (no GPU, no HIP) for the CPU tests of the scheduling core and the router.
#pragma once
#include <atomic>
#include <chrono>
#include <cmath>
#include <thread>
#include <stdexcept>

#include "../../include/rwkv_scheduler.hpp"

struct FakeState {
    std::vector<std::vector<float>> slots;
    std::vector<float> init() const { return {1.0f, 0.0f}; }
    void load(const std::vector<float> &t, int b) { slots.at((size_t)b) = t; }
    std::vector<float> back(int b) { return slots.at((size_t)b); }
    // one "layer" of the state: (hash + layer, count)
    size_t layer_len() const { return 2; }
    bool embed_fails = false;                     // the read-back throws (steps keep succeeding): a replica that is broken only on this path
    void embed(int layer, int b, float *dst) {
        if (embed_fails) throw std::runtime_error("fake read-back failure");
        dst[0] = slots.at((size_t)b)[0] + (float)layer;
        dst[1] = slots.at((size_t)b)[1];
    }
};
// state = (hash, count); a token updates hash = fmod(hash * 31 + tok + 1, 65521); logits[i] = fmod(hash + 7 i, 13)
struct FakeEngine {
    int max_batch;
    rwkv::ModelInfo info{};
    FakeState state;
    int chunk;            // tokens a slot may consume per infer call (like token_chunk_size / active slots)
    std::atomic<int> calls{0};   // read by the test thread while a replica thread may be stepping
    std::vector<int> riders;
    std::atomic<int> fail_at{-1};   // infer call number that throws (a device error in the middle of a serving loop); -1: never
    std::atomic<int> fail_from{-1}; // every infer call from this number on throws (a STICKY device fault: HIP errors do not go away); -1: never
    std::atomic<bool> hold{false};  // while set, infer() does not return (a test submits several requests "at the same instant")
    FakeEngine(int B, int chunk_) : max_batch(B), chunk(chunk_) { info.num_vocab = 8; state.slots.assign((size_t)B, state.init()); }
    std::vector<rwkv::RnnOutputBatch> infer(rwkv::RnnInput &in) {
        while (hold.load()) std::this_thread::sleep_for(std::chrono::microseconds(20));
        const int call = ++calls;
        if (call == fail_at.load()) throw std::runtime_error("fake device error");
        if (fail_from.load() >= 0 && call >= fail_from.load()) throw std::runtime_error("fake sticky device fault");
        int n = 0;
        std::vector<rwkv::RnnOutputBatch> out((size_t)max_batch);
        for (int b = 0; b < max_batch; ++b) {
            auto &t = in.batches[(size_t)b].tokens;
            if (t.empty()) continue;
            ++n;
            const size_t take = std::min<size_t>(t.size(), (size_t)chunk);
            for (size_t i = 0; i < take; ++i) {
                auto &s = state.slots[(size_t)b];
                s[0] = std::fmod(s[0] * 31.0f + (float)t[i] + 1.0f, 65521.0f);
                s[1] += 1.0f;
                const bool last = i + 1 == t.size();
                if (in.batches[(size_t)b].option == rwkv::RnnOption::None) continue;         // state only: no rows
                if (in.batches[(size_t)b].option == rwkv::RnnOption::Full || last)
                    for (int v = 0; v < 8; ++v) out[(size_t)b].push_back(std::fmod(s[0] + 7.0f * v, 13.0f));
            }
            t.erase(t.begin(), t.begin() + (long)take);
        }
        riders.push_back(n);
        return out;
    }
};
