// CPU test of include/rwkv_router.hpp over fake engines (no GPU, no HIP): prefix affinity, least-busy placement, full
// replicas skipped, per-replica threads, results identical to a single-engine run.
#include <cassert>
#include <cmath>
#include <chrono>
#include <thread>
#include <cstdio>

#include "../../include/rwkv_router.hpp"
#include "fake_engine.hpp"

#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

static rwkv::Tokens greedy_alone(const rwkv::Tokens &prompt, int n_new) {
    FakeEngine e(1, 1000);
    std::vector<FakeEngine *> es{&e};
    rwkv::ReplicaRouter<FakeEngine> r(es);
    rwkv::RoutedRequest q;
    q.tokens = prompt; q.max_new = n_new;
    assert(r.submit(&q) == 0);
    r.drain();
    return q.generated;
}

int main() {
    using namespace rwkv;
    // --- 8 replicas x 4 slots, 64 requests: every request completes, answers equal the single-engine answers, load spreads
    {
        std::vector<std::unique_ptr<FakeEngine>> engines;
        std::vector<FakeEngine *> es;
        for (int i = 0; i < 8; ++i) { engines.emplace_back(new FakeEngine(4, 3)); es.push_back(engines.back().get()); }
        // eight long requests first (tens of thousands of fake steps each, submitted microseconds apart): every later one sees the
        // earlier ones still in flight, so least-busy walks the replicas in index order — the placement rule, checked exactly
        std::vector<RoutedRequest> longs(8);
        ReplicaRouter<FakeEngine> router(es);
        for (int i = 0; i < 8; ++i) {
            longs[(size_t)i].tokens = Tokens{(uint32_t)(100 + i)};
            longs[(size_t)i].max_new = 30000;
            CHECK(router.submit(&longs[(size_t)i]) == i);
        }
        std::vector<RoutedRequest> reqs(64);
        for (int i = 0; i < 64; ++i) {
            reqs[(size_t)i].tokens = Tokens{(uint32_t)(i % 7 + 1), (uint32_t)(i % 5 + 2), (uint32_t)(i + 3)};
            reqs[(size_t)i].max_new = 5 + i % 3;
        }
        size_t next = 0;
        while (next < reqs.size()) {                                     // `enqueue`: retry while every replica is full
            if (router.submit(&reqs[next]) >= 0) ++next;
            else std::this_thread::yield();
        }
        router.drain();
        for (int r = 0; r < 8; ++r) CHECK(router.steps(r) >= 30000);     // every replica ran its long request to the end
        for (auto &q : longs) CHECK(q.done && (int)q.generated.size() == q.max_new);
        for (auto &q : reqs) {
            CHECK(q.done && q.replica >= 0 && (int)q.generated.size() == q.max_new);
            CHECK(q.generated == greedy_alone(q.tokens, q.max_new));
        }
    }
    // --- prefix affinity: a follow-up that extends a finished request goes to the replica that cached it, even if that
    //     replica is busier than the others, and continues from the cached state
    {
        FakeEngine e0(4, 100), e1(4, 100), e2(4, 100);
        std::vector<FakeEngine *> es{&e0, &e1, &e2};
        ReplicaRouter<FakeEngine> router(es);
        RoutedRequest a, b, c;
        a.tokens = {9, 8, 7, 6}; a.max_new = 3000;                        // long enough to still be in flight when b is routed
        b.tokens = {1, 2}; b.max_new = 2;
        CHECK(router.submit(&a) == 0);                                   // all idle: lowest index
        CHECK(router.submit(&b) == 1);                                   // replica 0 has one in flight: least busy is 1
        router.drain();
        Tokens follow = a.tokens;
        follow.insert(follow.end(), a.generated.begin(), a.generated.end());       // what replica 0 cached at finish: prompt + every fed token
        follow.push_back(42);
        RoutedRequest hold;                                              // make replica 0 the busiest
        hold.tokens = {5, 5, 5}; hold.max_new = 200;
        CHECK(router.submit(&hold) == 0);
        const auto where = router.route(follow);
        CHECK(where.first == 0 && where.second == follow.size() - 1);
        c.tokens = follow; c.max_new = 4;
        const int calls0 = e0.calls;
        CHECK(router.submit(&c) == 0);
        router.drain();
        CHECK(c.done && c.generated == greedy_alone(follow, 4));
        CHECK(e0.calls > calls0);
    }
    // --- a full replica is skipped; with every replica full submit() reports -1
    {
        FakeEngine e0(1, 100), e1(1, 100);
        std::vector<FakeEngine *> es{&e0, &e1};
        RoutedRequest a, b, c;                                           // declared before the router: they outlive its threads
        a.tokens = {1}; a.max_new = 100000; b.tokens = {2}; b.max_new = 100000; c.tokens = {3}; c.max_new = 1;
        ReplicaRouter<FakeEngine> router(es);
        CHECK(router.submit(&a) == 0 && router.submit(&b) == 1);
        CHECK(router.submit(&c) == -1);
    }   // the router's destructor stops the replica threads with requests still running
    // --- a replica whose engine throws fails ITS requests with the message and keeps serving; the process survives, drain()
    //     returns, the other replica is untouched
    {
        FakeEngine e0(2, 100), e1(2, 100);
        e0.fail_at = 3;                                                  // third infer call of replica 0 throws
        std::vector<FakeEngine *> es{&e0, &e1};
        RoutedRequest a, b, c, d;
        a.tokens = {1, 2, 3}; a.max_new = 50; b.tokens = {4, 5}; b.max_new = 50; c.tokens = {6}; c.max_new = 10; d.tokens = {7, 7}; d.max_new = 4;
        ReplicaRouter<FakeEngine> router(es);
        e0.hold = true;                                                  // replica 0 cannot fail (and empty) before b and c are placed
        CHECK(router.submit(&a) == 0 && router.submit(&b) == 1 && router.submit(&c) >= 0);
        e0.hold = false;
        router.drain();
        CHECK(a.done && a.failed && a.error.find("fake device error") != std::string::npos);
        CHECK(b.done && !b.failed && b.generated == greedy_alone(b.tokens, 50));
        CHECK(c.done && (c.failed || c.generated == greedy_alone(c.tokens, 10)));     // failed only if it rode replica 0's bad step
        CHECK(router.busy(0) == 0 && router.busy(1) == 0);
        CHECK(router.submit(&d) == 0);                                   // replica 0 serves again (its slots were given up, not leaked)
        router.drain();
        CHECK(d.done && !d.failed && d.generated == greedy_alone(d.tokens, 4));
        RoutedRequest t;                                                 // a throwing sample callback is contained the same way
        t.tokens = {9}; t.max_new = 3; t.sample = [](const std::vector<float> &) -> uint32_t { throw std::runtime_error("sampler blew up"); };
        CHECK(router.submit(&t) >= 0);
        router.drain();
        CHECK(t.done && t.failed && t.error == "sampler blew up");
    }
    // --- a STICKY engine fault (every step throws from some call on, like a HIP device fault): after `eject_after` consecutive
    //     failures the replica is ejected — it no longer wins the least-busy tie, new traffic goes to the healthy replica and
    //     completes, requests it had accepted but not started are re-routed, and with no healthy replica left submit() says -1
    {
        FakeEngine e0(2, 100), e1(2, 100);
        e0.fail_from = 2;                                                // replica 0: first step fine, then broken for good
        std::vector<FakeEngine *> es{&e0, &e1};
        std::vector<RoutedRequest> first(2), later(12);
        std::deque<RoutedRequest> probes;                                // stable addresses; declared before the router: they outlive its threads
        first[0].tokens = {1, 2}; first[0].max_new = 30; first[1].tokens = {3}; first[1].max_new = 30;
        ReplicaRouter<FakeEngine> router(es, 256, 2);
        CHECK(router.submit(&first[0]) == 0 && router.submit(&first[1]) == 1);
        for (int spin = 0; spin < 20000 && router.healthy(0); ++spin) {  // keep feeding replica 0 until its failures add up
            probes.emplace_back();
            RoutedRequest *probe = &probes.back();
            probe->tokens = {(uint32_t)(100 + spin)}; probe->max_new = 2;
            if (router.submit(probe) < 0) std::this_thread::yield();
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        CHECK(!router.healthy(0) && router.healthy(1) && router.healthy_count() == 1);
        router.drain();
        CHECK(first[0].done && first[0].failed && first[0].error.find("sticky") != std::string::npos);
        CHECK(first[1].done && !first[1].failed && first[1].generated == greedy_alone(first[1].tokens, 30));
        const int calls_before = e0.calls.load();
        for (int i = 0; i < 12; ++i) {                                   // an empty, broken replica must not absorb this traffic
            later[(size_t)i].tokens = {(uint32_t)(7 + i), 1}; later[(size_t)i].max_new = 3;
            int where;
            while ((where = router.submit(&later[(size_t)i])) < 0) std::this_thread::yield();
            CHECK(where == 1);
        }
        router.drain();
        for (auto &q : later) CHECK(q.done && !q.failed && q.replica == 1 && q.generated == greedy_alone(q.tokens, 3));
        CHECK(e0.calls.load() == calls_before);                          // nothing was even tried on the ejected engine
        CHECK(router.route({42}).first == 1);
        // revive: the operator reset the device
        e0.fail_from = -1;
        router.revive(0);
        RoutedRequest back;
        back.tokens = {5, 5}; back.max_new = 4;
        CHECK(router.healthy(0) && router.submit(&back) == 0);
        router.drain();
        CHECK(back.done && !back.failed && back.generated == greedy_alone(back.tokens, 4));
    }
    {   // every replica broken: submit() returns -1 instead of queueing onto a dead engine
        FakeEngine e0(1, 100);
        e0.fail_from = 1;
        std::vector<FakeEngine *> es{&e0};
        RoutedRequest a, b;
        a.tokens = {1}; a.max_new = 5; b.tokens = {2}; b.max_new = 5;
        ReplicaRouter<FakeEngine> router(es, 256, 1);
        CHECK(router.submit(&a) == 0);
        router.drain();
        CHECK(a.done && a.failed && !router.healthy(0));
        CHECK(router.submit(&b) == -1 && !b.done);
    }
    // --- many threads submitting at once: nobody gets a slot twice, in-flight never exceeds capacity, everything completes
    {
        std::vector<std::unique_ptr<FakeEngine>> engines;
        std::vector<FakeEngine *> es;
        for (int i = 0; i < 4; ++i) { engines.emplace_back(new FakeEngine(2, 5)); es.push_back(engines.back().get()); }
        std::vector<RoutedRequest> reqs(96);
        for (int i = 0; i < 96; ++i) { reqs[(size_t)i].tokens = Tokens{(uint32_t)(i + 1), (uint32_t)(i % 3)}; reqs[(size_t)i].max_new = 3 + i % 4; }
        ReplicaRouter<FakeEngine> router(es);
        std::atomic<int> over{0};
        std::vector<std::thread> subs;
        for (int t = 0; t < 6; ++t)
            subs.emplace_back([&, t] {
                for (int i = t; i < 96; i += 6) {
                    while (router.submit(&reqs[(size_t)i]) < 0) std::this_thread::yield();
                    for (int r = 0; r < 4; ++r) if (router.busy(r) > 2) ++over;
                }
            });
        for (auto &th : subs) th.join();
        router.drain();
        CHECK(over.load() == 0);
        for (auto &q : reqs) CHECK(q.done && !q.failed && q.generated == greedy_alone(q.tokens, q.max_new));
    }
    // --- the `/embeddings` batch job over replicas: 40 ragged documents (an empty one among them) over 3 replicas x 2 slots while ordinary
    //     requests are in flight; every document's rows equal the fake engine run over that document alone; every replica took part;
    //     nothing is left in a slot or a prefix cache
    {
        std::vector<std::unique_ptr<FakeEngine>> engines;
        std::vector<FakeEngine *> es;
        for (int i = 0; i < 3; ++i) { engines.emplace_back(new FakeEngine(2, 4)); es.push_back(engines.back().get()); }
        auto expect = [](const Tokens &doc, int layer, float *o) {
            float h = 1.0f, n = 0.0f;
            const Tokens t = doc.empty() ? Tokens{0} : doc;
            for (uint32_t tok : t) { h = std::fmod(h * 31.0f + (float)tok + 1.0f, 65521.0f); n += 1.0f; }
            o[0] = h + (float)layer; o[1] = n;
        };
        std::vector<Tokens> docs;
        for (uint32_t d = 0; d < 40; ++d) {
            Tokens t;
            for (uint32_t i = 0; i < (d * 7) % 23; ++i) t.push_back((d * 5 + i * 3) % 8);
            docs.push_back(t);
        }
        {
            ReplicaRouter<FakeEngine> router(es);
            RoutedRequest other;
            other.tokens = Tokens{3, 1, 4}; other.max_new = 200;
            CHECK(router.submit(&other) >= 0);
            std::vector<float> out(docs.size() * 2, -1.f);
            CHECK(router.embed_documents(docs, 5, out.data(), 2) == 0);
            for (size_t d = 0; d < docs.size(); ++d) {
                float want[2];
                expect(docs[d], 5, want);
                CHECK(out[2 * d] == want[0] && out[2 * d + 1] == want[1]);
            }
            CHECK(other.done && !other.failed && other.generated == greedy_alone(other.tokens, 200));
            for (int i = 0; i < 3; ++i) CHECK(router.steps(i) > 0 && router.busy(i) == 0);
        }
        // a replica dies under the job (sticky fault from its third call on): it is ejected, its documents are handed out again, the
        // job still returns every embedding
        {
            std::vector<std::unique_ptr<FakeEngine>> e2;
            std::vector<FakeEngine *> es2;
            for (int i = 0; i < 3; ++i) { e2.emplace_back(new FakeEngine(2, 4)); es2.push_back(e2.back().get()); }
            e2[1]->fail_from = 3;
            ReplicaRouter<FakeEngine> router(es2, 256, 2);
            std::vector<float> out(docs.size() * 2, -1.f);
            const size_t retried = router.embed_documents(docs, 0, out.data(), 2, 4);
            CHECK(retried > 0 && !router.healthy(1) && router.healthy_count() == 2);
            for (size_t d = 0; d < docs.size(); ++d) {
                float want[2];
                expect(docs[d], 0, want);
                CHECK(out[2 * d] == want[0] && out[2 * d + 1] == want[1]);
            }
        }
        // no healthy replica: the job reports it instead of waiting forever
        {
            FakeEngine bad(2, 4);
            bad.fail_from = 1;
            std::vector<FakeEngine *> es3{&bad};
            ReplicaRouter<FakeEngine> router(es3, 256, 1);
            std::vector<float> out(docs.size() * 2, -1.f);
            bool threw = false;
            try { router.embed_documents(docs, 0, out.data(), 2); } catch (const std::runtime_error &) { threw = true; }
            CHECK(threw);
        }
    }
    // --- drain() is router-wide (advisor, round 4): a request re-routed off an ejected replica may land on a replica a per-replica walk
    //     has already seen idle.  Replica 1 is blocked inside its step with b in a slot and c still in its inbox; it then fails for good
    //     (ejected at once): b fails, c moves to replica 0, whose engine is held.  drain() must not return before c has completed there.
    {
        FakeEngine e0(3, 100), e1(2, 100);
        std::vector<FakeEngine *> es{&e0, &e1};
        ReplicaRouter<FakeEngine> router(es, 256, 1);
        // every CHECK of this block runs with the holds released again: a failed CHECK must fail the test, not hang its threads
        RoutedRequest warm1, warm2, b, c;
        warm1.tokens = {9}; warm2.tokens = {8}; b.tokens = {3, 4}; b.max_new = 3; c.tokens = {5, 6}; c.max_new = 3;
        e0.hold = true; e1.hold = true;
        const int w1 = router.submit(&warm1);                                // replica 0 (least busy, lowest index)
        const int pb = router.submit(&b);                                    // replica 1 (0 in flight against 1)
        const int w2 = router.submit(&warm2);                                // tie 1 : 1 -> replica 0
        std::this_thread::sleep_for(std::chrono::milliseconds(30));         // replica 1's thread now sits inside infer() with b in a slot
        const int pc = router.submit(&c);                                    // 2 : 1 -> replica 1, where it stays in the inbox
        e0.hold = false;                                                     // replica 0 finishes its two requests and is IDLE when drain() looks at it
        for (int i = 0; i < 200 && router.busy(0) > 0; ++i) std::this_thread::sleep_for(std::chrono::milliseconds(5));
        const bool idle0 = router.busy(0) == 0;
        e0.hold = true;                                                      // whatever arrives there from now on stays in flight until released
        std::thread later([&] {
            std::this_thread::sleep_for(std::chrono::milliseconds(50));     // drain() is by now waiting on replica 1
            e1.fail_from = 1;
            e1.hold = false;                                                 // the step throws: replica 1 is ejected at once, c is re-routed to replica 0 (held)
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
            e0.hold = false;
        });
        router.drain();
        const bool c_done_at_drain = c.done, w_done_at_drain = warm1.done && warm2.done;
        later.join();
        router.drain();
        CHECK(idle0);
        CHECK(w1 == 0 && pb == 1 && w2 == 0 && pc == 1);
        CHECK(b.done && b.failed && !router.healthy(1));
        CHECK(c_done_at_drain && w_done_at_drain);                           // drain() returned only after the re-routed request had completed
        CHECK(c.done && !c.failed && c.replica == 0 && c.generated == greedy_alone(c.tokens, 3));
    }
    // --- a replica whose STEPS succeed but whose embedding read-back throws every time is ejected too (the failure count is reset only by
    //     an iteration that went through completely), and the job finishes on the healthy replica
    {
        FakeEngine good(2, 8), flaky(2, 8);
        flaky.state.embed_fails = true;
        std::vector<FakeEngine *> es{&good, &flaky};
        ReplicaRouter<FakeEngine> router(es, 256, 2);
        std::vector<Tokens> docs;
        for (uint32_t d = 0; d < 12; ++d) docs.push_back(Tokens{d % 7 + 1, (d * 3) % 5 + 1, 2});
        std::vector<float> out(docs.size() * 2, -1.f);
        router.embed_documents(docs, 1, out.data(), 2, 6);
        CHECK(!router.healthy(1) && router.healthy(0));
        for (size_t d = 0; d < docs.size(); ++d) {
            float h = 1.0f, n = 0.0f;
            for (uint32_t tok : docs[d]) { h = std::fmod(h * 31.0f + (float)tok + 1.0f, 65521.0f); n += 1.0f; }
            CHECK(out[2 * d] == h + 1.0f && out[2 * d + 1] == n);
        }
    }
    std::printf("router_test: ok\n");
    return 0;
}
