// CPU test of include/rwkv_scheduler.hpp with a fake engine (no GPU, no HIP): slot choice, prefix cache, continuous batching.
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "../../include/rwkv_scheduler.hpp"

#include "fake_engine.hpp"

namespace {
std::vector<float> run_alone(const rwkv::Tokens &toks) {
    FakeEngine e(1, 1000);
    rwkv::Scheduler<FakeEngine> s(e);
    int b = -1;
    assert(s.queue(toks, b) == rwkv::SlotResult::Success && b == 0);
    while (s.pending()) s.step();
    auto out = s.request(0).output;
    s.finish(0);
    return out;
}
}  // namespace

#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main() {
    using namespace rwkv;
    // --- slot choice priority: continue (longest) > empty > back (oldest); all busy -> Failure
    {
        FakeEngine e(3, 1000);
        Scheduler<FakeEngine> s(e);
        int a = -1, b = -1, c = -1, d = -1;
        CHECK(s.queue({1, 2, 3}, a) == SlotResult::Success);            // empty slot (ties: the last one, like max_by)
        CHECK(s.queue({4, 5}, b) == SlotResult::Success && b != a);
        CHECK(s.queue({6}, c) == SlotResult::Success && c != a && c != b);
        CHECK(s.queue({7}, d) == SlotResult::Failure);                  // all busy
        while (s.pending()) s.step();
        s.finish(a); s.finish(b); s.finish(c);
        CHECK(s.slot(a).content == Tokens({1, 2, 3}));
        int x = -1;
        CHECK(s.queue({1, 2, 3, 9, 9}, x) == SlotResult::Success && x == a);   // continue beats everything
        CHECK(s.request(x).prefix == Tokens({1, 2, 3}) && s.request(x).suffix == Tokens({9, 9}));   // state checked out of the cache
        int y = -1;
        CHECK(s.queue({8, 8}, y) == SlotResult::Fault);                 // no empty, no match: back the OLDEST idle slot
        CHECK(y == b);                                                   // b went idle before c
    }
    // --- prefix cache: continuing from a cached prefix gives the same logits as processing the whole sequence alone
    {
        const Tokens full = {3, 1, 4, 1, 5, 9, 2, 6};
        const auto want = run_alone(full);
        FakeEngine e(2, 3);                                              // small chunk: several infer calls per request
        Scheduler<FakeEngine> s(e);
        int b = -1;
        CHECK(s.queue(Tokens(full.begin(), full.begin() + 5), b) == SlotResult::Success);
        while (s.pending()) s.step();
        s.finish(b);
        CHECK(s.cache().size() == 1);
        int b2 = -1;
        CHECK(s.queue(full, b2) == SlotResult::Success && b2 == b);
        CHECK(s.request(b2).prefix.size() == 5 && s.request(b2).suffix.size() == 3);
        const int calls0 = e.calls;
        while (s.pending()) s.step();
        CHECK(e.calls - calls0 == 1);                                    // only the 3 new tokens were fed
        CHECK(s.request(b2).output == want);
        s.finish(b2);
        // the whole request cached: nothing is fed, the cached output row is the answer (run.rs:809-811)
        int b3 = -1;
        const int calls1 = e.calls;
        CHECK(s.queue(full, b3) != SlotResult::Failure);
        CHECK(s.request(b3).suffix.empty() && s.request(b3).prefix == full && !s.pending());
        CHECK(s.request(b3).output == want && e.calls == calls1);
        s.finish(b3);
    }
    // --- continuous batching: a request queued while another is mid-flight rides the very next step
    {
        FakeEngine e(4, 2);
        Scheduler<FakeEngine> s(e);
        int a = -1, b = -1;
        CHECK(s.queue({1, 1, 1, 1, 1, 1}, a) == SlotResult::Success);   // 3 steps of 2 tokens
        CHECK(s.step() == 1);
        CHECK(s.queue({2, 2}, b) == SlotResult::Success);
        CHECK(s.step() == 2);                                            // both ride
        CHECK(!s.request(b).output.empty() && s.request(b).suffix.empty());
        CHECK(s.step() == 1);
        CHECK(!s.pending());
        // decode: push the "sampled" token and step again
        s.push(a, 5); s.push(b, 6);
        CHECK(s.step() == 2);
        s.finish(a); s.finish(b);
        CHECK(s.slot(a).content.size() == 7 && s.slot(b).content.size() == 3);
    }
    // --- cache bound: the stalest items go first
    {
        PrefixCache c(2);
        c.insert({1}, {1.f}, {1.f}, 1);
        c.insert({2}, {2.f}, {2.f}, 2);
        (void)c.checkout({1, 7}, 3);                                     // refreshes {1}
        c.insert({3}, {3.f}, {3.f}, 4);                                  // evicts {2}
        CHECK(c.size() == 2 && c.checkout({2}, 5).hit == false && c.checkout({1}, 6).hit && c.checkout({3}, 7).hit);
    }
    // --- eviction prunes the trie: after thousands of long keys through a 4-item cache only the live key paths remain
    {
        PrefixCache c(4);
        for (uint32_t i = 0; i < 2000; ++i) {
            Tokens k(100, i + 1);                                        // 100 tokens per key, disjoint paths
            k.back() = 7;
            c.insert(k, {1.f}, {1.f}, i + 1);
        }
        CHECK(c.size() == 4 && c.nodes() == 4 * 100);
        Tokens shared(50, 9u), a = shared, b = shared;
        a.push_back(1); b.push_back(2);
        c.insert(a, {1.f}, {1.f}, 5000); c.insert(b, {1.f}, {1.f}, 5001);
        c.insert({3}, {1.f}, {1.f}, 5002); c.insert({4}, {1.f}, {1.f}, 5003);   // pushes the four old keys out
        CHECK(c.size() == 4 && c.nodes() == 50 + 2 + 2);                 // the shared stem is kept once
        c.insert({5}, {1.f}, {1.f}, 5004);                               // evicts `a`: its leaf goes, the stem stays for `b`
        CHECK(!c.contains(a) && c.contains(b) && c.nodes() == 50 + 1 + 3);
    }
    // --- prompts longer than MIN_PROMPT_CACHE_TOKENS (32) are cached when they have been read in (run.rs:794-838), short ones
    //     only at finish; a second request with the same long prompt skips its prefill while the first one still decodes
    {
        FakeEngine e(2, 16);
        Scheduler<FakeEngine> s(e);
        Tokens longp(40, 5u), shortp(8, 6u);
        int a = -1, b = -1;
        CHECK(s.queue(longp, a) == SlotResult::Success);
        CHECK(s.queue(shortp, b) == SlotResult::Success);
        while (s.pending()) s.step();
        CHECK(s.cache().size() == 1 && s.cache().contains(longp) && !s.cache().contains(shortp));
        s.push(a, 9);                                                    // a keeps decoding
        s.step();
        s.finish(b);
        CHECK(s.cache().size() == 2);
        int c2 = -1;
        CHECK(s.queue(longp, c2) != SlotResult::Failure && c2 == b);
        CHECK(s.request(c2).prefix.size() == 40 && s.request(c2).suffix.empty() && !s.request(c2).output.empty());   // no prefill at all
    }
    // --- per-request initial states (`check_in_state`, run.rs:376-437): own cache per state id, misses start from the init slab
    {
        FakeEngine e(2, 100);
        Scheduler<FakeEngine> s(e);
        s.check_in_state(7, {123.0f, 0.0f});
        int a = -1, b = -1;
        CHECK(s.queue({1, 2, 3}, a, RnnOption::Last, 7) == SlotResult::Success);
        CHECK(s.queue({1, 2, 3}, b) == SlotResult::Success);
        while (s.pending()) s.step();
        CHECK(s.request(a).output != s.request(b).output);               // different starting states
        FakeEngine e2(1, 100);
        e2.state.slots[0] = {123.0f, 0.0f};
        rwkv::RnnInput in; in.batches.resize(1); in.batches[0].tokens = {1, 2, 3};
        CHECK(e2.infer(in)[0] == s.request(a).output);
        s.finish(a); s.finish(b);
        CHECK(s.cache(7).size() == 1 && s.cache(0).size() == 1);
        int c = -1;
        CHECK(s.queue({1, 2, 3, 4}, c, RnnOption::Last, 7) != SlotResult::Failure);
        CHECK(s.request(c).prefix.size() == 3);                          // continues from state 7's cache, not the default one
        bool threw = false;
        try { int d; s.queue({1}, d, RnnOption::Last, 99); } catch (const std::invalid_argument &) { threw = true; }
        CHECK(threw);
    }
    // --- stop strings (run.rs:899-932)
    {
        auto bytes = [](const char *t) { return std::vector<uint8_t>(t, t + std::strlen(t)); };
        StopScan r = scan_stops(bytes("hello\n\nUser"), {"\n\nUser", "</s>"});
        CHECK(r.matched && r.head == 5);                                 // emit "hello", stop
        r = scan_stops(bytes("hello\n\nUs"), {"\n\nUser"});
        CHECK(!r.matched && r.head == 5);                                // possible match pending: hold the tail back
        r = scan_stops(bytes("hello world"), {"\n\nUser"});
        CHECK(!r.matched && r.head == 11);                               // nothing pending: everything is safe to emit
        r = scan_stops(bytes("abc"), {});
        CHECK(!r.matched && r.head == 3);
        r = scan_stops(bytes("ab</s>cd"), {"zzz", "</s>"});
        CHECK(r.matched && r.head == 2);                                 // a matched stop wins over an unmatched one
    }
    // --- a very long cached context: one trie node per token, torn down (destructor, eviction) without recursing per token
    {
        Tokens longkey(300000);
        for (size_t i = 0; i < longkey.size(); ++i) longkey[i] = (uint32_t)(i * 2654435761u % 65536u);
        {
            PrefixCache c(2);
            c.insert(longkey, {1.0f}, {2.0f}, 1);
            CHECK(c.nodes() == longkey.size() && c.match_len(longkey) == longkey.size());
            Tokens other = longkey;
            other[150000] ^= 1u;                                         // shares half of the chain
            c.insert(other, {3.0f}, {4.0f}, 2);
            CHECK(c.nodes() == longkey.size() + (longkey.size() - 150000));
            c.insert({7, 7, 7}, {5.0f}, {6.0f}, 3);                      // third item: the oldest (longkey) is evicted, its private tail pruned
            CHECK(c.size() == 2 && c.match_len(longkey) == 0 && c.nodes() == longkey.size() + 3);
        }                                                                // destructor over a 300k-deep chain
    }
    // --- GenerateKind::Choose / ::State and `perplexity` (run.rs:699-755, 936-989) against a stand-alone replay of the fake engine
    {
        // replay: state (hash, count) after `toks` from `st`, and the fake logits row of every token
        auto replay = [](std::vector<float> st, const Tokens &toks, std::vector<std::vector<float>> &rows) {
            for (uint32_t t : toks) {
                st[0] = std::fmod(st[0] * 31.0f + (float)t + 1.0f, 65521.0f);
                st[1] += 1.0f;
                std::vector<float> lg(8);
                for (int v = 0; v < 8; ++v) lg[(size_t)v] = std::fmod(st[0] + 7.0f * v, 13.0f);
                rows.push_back(lg);
            }
            return st;
        };
        auto prob = [](const std::vector<float> &lg, uint32_t tok) {       // exp / sum, no max subtraction (run.rs:737-740)
            float sum = 0.f;
            for (float x : lg) sum += std::exp(x);
            return std::exp(lg[tok]) / sum;
        };
        auto ref_ppl = [&](const std::vector<float> &st, const Tokens &toks, const float *head) {
            Tokens all = toks;
            std::vector<float> p;
            if (head) p.push_back(*head); else all.insert(all.begin(), 0u);
            std::vector<std::vector<float>> rows;
            replay(st, all, rows);
            for (size_t j = 1; j < all.size(); ++j) p.push_back(prob(rows[j - 1], all[j]));
            double acc = 0.0;
            for (float x : p) acc += std::log((double)x);
            return (float)(-acc / (double)all.size());
        };
        FakeEngine e(2, 2);                                              // 2 tokens per slot per call: several steps per evaluation
        Scheduler<FakeEngine> s(e);
        const Tokens prompt = {5, 1, 2, 6, 3};
        int b = -1, other = -1;
        CHECK(s.queue(prompt, b) == SlotResult::Success);
        while (s.pending()) s.step();
        std::vector<std::vector<float>> prow;
        const std::vector<float> after_prompt = replay(e.state.init(), prompt, prow);
        CHECK(e.state.back(b) == after_prompt && s.request(b).output == prow.back());
        // another request is mid-prefill while the choices are scored: it must ride the same device steps
        CHECK(s.queue({7, 7, 7, 7, 7, 7, 7, 7, 7}, other) == SlotResult::Success && other != b);
        const std::vector<Tokens> choices = {{1, 2, 3}, {}, {4}, {6, 6, 0, 1, 2}};
        // probabilities handed in by the caller (what `sample()` returned), here: a plain softmax of the last logits
        std::vector<float> probs(8);
        { float mx = *std::max_element(prow.back().begin(), prow.back().end()); double sum = 0; for (int v = 0; v < 8; ++v) { probs[(size_t)v] = std::exp(prow.back()[(size_t)v] - mx); sum += probs[(size_t)v]; } for (float &x : probs) x = (float)(x / sum); }
        const int calls_before = e.calls;
        std::vector<float> got = s.choose(b, choices, false, probs);
        CHECK(got.size() == 4 && std::isinf(got[1]) && got[1] > 0);      // the empty choice keeps +inf
        for (size_t i : {0u, 2u, 3u}) {
            const float head = probs[choices[i][0]];
            const float want = ref_ppl(after_prompt, choices[i], &head);
            CHECK(std::fabs(got[i] - want) <= 1e-6f * std::max(1.0f, std::fabs(want)));
        }
        CHECK(e.calls > calls_before);
        CHECK(e.state.back(b) == after_prompt);                          // the slot is back where the prompt left it
        CHECK(s.request(other).suffix.empty() && s.request(other).prefix.size() == 9);   // the other request rode along
        // default probabilities (softmax of the request's last logits) and calibration against the initial state
        std::vector<float> cal = s.choose(b, choices, true);
        for (size_t i : {0u, 2u, 3u}) {
            const float head = probs[choices[i][0]];
            const float want = -ref_ppl(e.state.init(), choices[i], nullptr) + ref_ppl(after_prompt, choices[i], &head);
            CHECK(std::fabs(cal[i] - want) <= 2e-6f * std::max(1.0f, std::fabs(want)));
        }
        CHECK(e.state.back(b) == after_prompt);
        // perplexity on its own: no head -> token 0 is prepended and counted in the denominator
        const float pp = s.perplexity(b, {2, 2}, nullptr);
        CHECK(std::fabs(pp - ref_ppl(after_prompt, {2, 2}, nullptr)) <= 1e-6f);
        // State kind: the slab as it stands
        std::vector<std::vector<float>> tmp;
        CHECK(s.state(b) == replay(after_prompt, {0, 2, 2}, tmp));
        // the request can go on afterwards
        s.push(b, 4);
        while (s.pending()) s.step();
        s.finish(b); s.finish(other);
        // misuse: choices before the prompt has been read in
        int c = -1;
        CHECK(s.queue({1, 1, 1, 1}, c) != SlotResult::Failure);
        bool threw = false;
        try { s.choose(c, choices, false); } catch (const std::logic_error &) { threw = true; }
        CHECK(threw);
    }
    // --- embed_documents: more documents than slots, ragged lengths (some longer than a step's chunk), an empty one, and a slot held by
    //     another request throughout.  Every document's rows == the fake engine run over that document alone from the initial state.
    {
        FakeEngine e(3, 4);
        Scheduler<FakeEngine> s(e);
        int held = -1;
        CHECK(s.queue({9, 9}, held) == SlotResult::Success);
        while (s.pending()) s.step();
        const std::vector<float> held_state = s.state(held);
        std::vector<Tokens> docs;
        for (uint32_t d = 0; d < 9; ++d) {
            Tokens t;
            for (uint32_t i = 0; i < (d * 5) % 13; ++i) t.push_back((d * 7 + i * 3) % 8);
            docs.push_back(t);                                               // lengths 0, 5, 10, 2, 7, 12, 4, 9, 1
        }
        std::vector<float> out(docs.size() * 2, -1.f);
        const int calls_before = e.calls.load();
        const size_t steps = s.embed_documents(docs, 3, out.data());
        CHECK((int)steps == e.calls.load() - calls_before && steps >= 9);    // two free slots, 50 tokens + the empty document's [0], 4 per slot and step
        for (size_t d = 0; d < docs.size(); ++d) {
            float h = 1.0f, n = 0.0f;
            Tokens t = docs[d].empty() ? Tokens{0} : docs[d];                 // run.rs:489-492
            for (uint32_t tok : t) { h = std::fmod(h * 31.0f + (float)tok + 1.0f, 65521.0f); n += 1.0f; }
            CHECK(out[2 * d] == h + 3.0f && out[2 * d + 1] == n);
        }
        CHECK(s.state(held) == held_state);                                  // the other request's slot was not touched
        CHECK(s.cache().size() == 0);                                        // documents leave nothing in the prefix cache
        for (int b = 0; b < 3; ++b) CHECK(b == held || (s.slot(b).kind == SlotKind::Idle && s.slot(b).content.empty()));
        // every slot held: the job refuses instead of spinning
        int x = -1, y = -1;
        CHECK(s.queue({1}, x) != SlotResult::Failure && s.queue({2}, y) != SlotResult::Failure);
        bool threw = false;
        try { s.embed_documents(docs, 0, out.data()); } catch (const std::runtime_error &) { threw = true; }
        CHECK(threw);
    }
    // --- a state-only job must not poison the prefix cache (advisor, round 4): a chat request caches its long prompt P; the embeddings
    //     job then runs P + X (a cache hit on P, whose logits ride along in `output`).  The document must NOT be cached under P's row:
    //     a later completion of P + X has to go through the engine and get its own logits.
    {
        FakeEngine e(2, 64);
        Scheduler<FakeEngine> s(e);
        Tokens P;
        for (uint32_t i = 0; i < 40; ++i) P.push_back(i % 7 + 1);             // > MIN_PROMPT_CACHE_TOKENS
        int b = -1;
        CHECK(s.queue(P, b) == SlotResult::Success);
        while (s.pending()) s.step();
        s.finish(b);
        const size_t cached = s.cache().size();
        CHECK(cached >= 1 && s.cache().contains(P));
        Tokens PX = P;
        for (uint32_t i = 0; i < 5; ++i) PX.push_back(i + 1);
        std::vector<float> out(2, -1.f);
        s.embed_documents({PX}, 0, out.data());
        CHECK(s.cache().size() == cached && !s.cache().contains(PX));
        int c = -1;
        const int calls = e.calls.load();
        CHECK(s.queue(PX, c) != SlotResult::Failure);
        while (s.pending()) s.step();
        CHECK(e.calls.load() > calls);                                       // the suffix X went through the engine
        float h = 1.0f;
        for (uint32_t tok : PX) h = std::fmod(h * 31.0f + (float)tok + 1.0f, 65521.0f);
        CHECK(!s.request(c).output.empty() && s.request(c).output[0] == std::fmod(h, 13.0f));   // its own logits, not P's
        s.finish(c);
    }
    std::printf("scheduler_test: ok\n");
    return 0;
}
