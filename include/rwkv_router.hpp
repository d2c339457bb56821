// rwkv_router.hpp — request-level sharding over N independent engines of one process (one engine per GPU): SURVEY §8 row (e).
//
// The path shards by REQUEST: a slot owns its recurrent state, weights are read-only, nothing is exchanged between
// replicas (the reference itself is single-GPU: docs/doc-guide/FAQ.md:14-16; `run.rs:1121-1130` builds each batch from
// independent slots).  So there is no collective and no RCCL here — only a routing decision on the host:
//
//   1. the replica whose prefix cache holds the LONGEST prefix of the request's tokens (the cache of `run.rs:443-485` is
//      per replica, because a cached state can only be checked out into a slot of the engine that produced its layout and
//      sits in that replica's host memory) — ties and misses fall through to
//   2. the replica with the fewest busy slots (then the lowest index), skipping replicas that are full.
//
// Health: a replica whose ENGINE throws (a HIP device fault is sticky: every later step on that device throws again) would empty
// at once, win every least-busy tie and absorb — and fail — most new traffic.  So engine failures are counted per replica; after
// `eject_after` consecutive ones (a successful step resets the count) the replica is EJECTED: route() / submit() skip it, the
// requests it had accepted but not started are re-routed to healthy replicas, and submit() returns -1 when none is left.
// `revive()` puts a replica back (after the operator reset the device).  A throwing `sample` callback is the REQUEST's fault: it
// fails that request alone and does not count against the replica.
//
// Threading follows the reference's contract per engine (two long-lived caller threads: `infer` and `softmax`,
// run.rs:1072-1190): every replica is driven by its OWN thread (`Replica::run`), which is the only thread that touches that
// engine's scheduler; `submit` only appends to the replica's inbox under its mutex.  Header-only, no HIP: `Engine` is what
// rwkv::Scheduler needs (include/rwkv_scheduler.hpp), so the CPU test runs it over fake engines.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>

#include "rwkv_scheduler.hpp"

namespace rwkv {

struct RoutedRequest {
    Tokens tokens;                                // the prompt (run.rs:489-492: empty => [0])
    int max_new = 0;                              // greedy tokens to generate after the prompt (0: prefill only)
    std::function<uint32_t(const std::vector<float> &)> sample;   // logits -> token; default arg-max (Nucleus top_k = 1)
    // GenerateKind::State (run.rs:980-989) for the `/embeddings` route: embed_layer >= 0 makes this a state-only request (no logits, nothing
    // sampled, nothing cached); when its tokens have been read in, that layer's WKV rows ([head_size][num_emb] floats) are written to
    // `embed_dst` (pinned memory for rwkv::Runtime engines) before the request is reported done
    int embed_layer = -1;
    float *embed_dst = nullptr;
    // filled in by the replica thread
    Tokens generated;
    std::vector<float> last_output;
    int replica = -1;
    bool done = false;
    bool failed = false;                          // the replica's engine (or the sample callback) threw while this request was in flight
    std::string error;                            // what it threw
};

template <class Engine>
class ReplicaRouter {
   public:
    explicit ReplicaRouter(std::vector<Engine *> engines, size_t max_cached = 256, int eject_after = 2) {
        for (Engine *e : engines) reps_.emplace_back(new Replica(*e, max_cached));
        for (auto &r : reps_) {
            r->eject_after = std::max(1, eject_after);
            r->reroute = [this](RoutedRequest *rq) { return submit(rq); };
            r->freed = [rm = room_] { { std::lock_guard<std::mutex> g(rm->mu); ++rm->epoch; --rm->inflight; } rm->cv.notify_all(); };
            r->moved = [rm = room_] { { std::lock_guard<std::mutex> g(rm->mu); --rm->inflight; } rm->cv.notify_all(); };
            r->thread = std::thread([p = r.get()] { p->run(); });
        }
    }
    ~ReplicaRouter() {
        for (auto &r : reps_) { { std::lock_guard<std::mutex> g(r->mu); r->stop = true; } r->cv.notify_all(); }
        for (auto &r : reps_) if (r->thread.joinable()) r->thread.join();
    }
    size_t size() const { return reps_.size(); }

    // Routing decision for `tokens` (exposed for tests): {replica, matched prefix length}.
    std::pair<int, size_t> route(const Tokens &tokens) {
        int best = -1;
        size_t best_len = 0;
        for (size_t i = 0; i < reps_.size(); ++i) {                      // 1. prefix affinity
            std::lock_guard<std::mutex> g(reps_[i]->mu);
            if (!reps_[i]->healthy || reps_[i]->load() >= reps_[i]->capacity) continue;
            const size_t len = reps_[i]->sched.match_len(tokens);          // thread-safe probe: never inserts, see Scheduler::match_len
            if (len > best_len) { best_len = len; best = (int)i; }
        }
        if (best >= 0) return {best, best_len};
        int least = -1, least_busy = 0;
        for (size_t i = 0; i < reps_.size(); ++i) {                      // 2. least busy
            std::lock_guard<std::mutex> g(reps_[i]->mu);
            const int busy = reps_[i]->load();
            if (!reps_[i]->healthy || busy >= reps_[i]->capacity) continue;
            if (least < 0 || busy < least_busy) { least = (int)i; least_busy = busy; }
        }
        return {least, 0};
    }

    // Hand a request to a replica; returns the replica index, or -1 when every replica is full (the caller queues it, as the
    // reference's `enqueue` task does with `SlotResult::Failure`, run.rs:1030-1062).  `req` must outlive its completion.
    int submit(RoutedRequest *req) {
        // Any number of threads may submit: the capacity test is repeated under the lock that appends to the inbox, so two callers
        // racing for the last free slot cannot both get it (the loser re-routes; -1 when nobody has room).
        for (size_t attempt = 0; attempt <= reps_.size(); ++attempt) {
            const auto where = route(req->tokens);
            if (where.first < 0) return -1;
            Replica &r = *reps_[(size_t)where.first];
            {
                std::lock_guard<std::mutex> g(r.mu);
                if (r.inflight >= r.capacity || r.stop || !r.healthy) continue;
                req->replica = where.first;
                // router-wide count first (drain() waits on it): the replica thread may complete the request the moment it is in the inbox
                { std::lock_guard<std::mutex> gr(room_->mu); ++room_->inflight; }
                r.inbox.push_back(req);
                ++r.inflight;
            }
            r.cv.notify_all();
            return where.first;
        }
        return -1;
    }
    // Block until every submitted request has completed — on WHATEVER replica it ends up: an ejected replica hands its unstarted requests
    // to healthy ones (engine_failed -> reroute), possibly to a replica a per-replica walk has already found idle, so completion is
    // counted router-wide: submit() adds one, complete() takes one off, and a re-routed request is submitted to its new replica BEFORE
    // its old count is given back (the count never touches zero while the request lives).
    void drain() {
        std::unique_lock<std::mutex> g(room_->mu);
        room_->cv.wait(g, [&] { return room_->inflight == 0; });
    }
    // The `/embeddings` batch job over replicas (documents handed out request by request, results gathered on the host; no
    // collective): every document becomes a State-kind request routed like any other (least busy replica with room), `out` receives
    // document d's rows at out + d * layer_len.  Documents whose replica failed under them are handed out again while a healthy replica is
    // left (at most `retries` times each); throws when a document cannot be placed or keeps failing.  Other traffic may be submitted
    // concurrently.  Returns the number of documents that needed a retry.
    size_t embed_documents(const std::vector<Tokens> &docs, int layer, float *out, size_t layer_len, int retries = 2) {
        std::vector<RoutedRequest> reqs(docs.size());
        std::vector<size_t> todo(docs.size());
        for (size_t d = 0; d < docs.size(); ++d) todo[d] = d;
        size_t retried = 0;
        for (int round = 0; !todo.empty(); ++round) {
            for (size_t d : todo) {
                RoutedRequest &rq = reqs[d];
                rq = RoutedRequest();
                rq.tokens = docs[d];
                rq.embed_layer = layer;
                rq.embed_dst = out + d * layer_len;
                for (;;) {
                    uint64_t seen;
                    { std::lock_guard<std::mutex> g(room_->mu); seen = room_->epoch; }
                    if (submit(&rq) >= 0) break;
                    if (healthy_count() == 0) { drain(); throw std::runtime_error("embed_documents(): no healthy replica left"); }
                    std::unique_lock<std::mutex> g(room_->mu);                      // every replica full: wait until a request completes somewhere
                    // (system_clock: pthread_cond_timedwait, which thread sanitizers of this toolchain intercept; steady_clock's clockwait they do not)
                    room_->cv.wait_until(g, std::chrono::system_clock::now() + std::chrono::milliseconds(50), [&] { return room_->epoch != seen; });
                }
            }
            drain();
            std::vector<size_t> again;
            std::string why;
            for (size_t d : todo) if (reqs[d].failed) { again.push_back(d); why = reqs[d].error; }
            if (!again.empty() && round >= retries) throw std::runtime_error("embed_documents(): " + std::to_string(again.size()) + " document(s) failed: " + why);
            retried += again.size();
            todo.swap(again);
        }
        return retried;
    }
    int busy(int replica) { std::lock_guard<std::mutex> g(reps_[(size_t)replica]->mu); return reps_[(size_t)replica]->load(); }
    bool healthy(int replica) { std::lock_guard<std::mutex> g(reps_[(size_t)replica]->mu); return reps_[(size_t)replica]->healthy; }
    int healthy_count() { int n = 0; for (size_t i = 0; i < reps_.size(); ++i) n += healthy((int)i) ? 1 : 0; return n; }
    // Put an ejected replica back into rotation (its engine was reset / replaced by the operator).
    void revive(int replica) {
        Replica &r = *reps_[(size_t)replica];
        std::lock_guard<std::mutex> g(r.mu);
        r.healthy = true;
        r.engine_failures = 0;
    }
    uint64_t steps(int replica) const { return reps_[(size_t)replica]->steps.load(); }

   private:
    struct Replica {
        Scheduler<Engine> sched;
        int capacity;
        std::mutex mu;
        std::condition_variable cv, idle_cv;
        std::deque<RoutedRequest *> inbox;
        int inflight = 0;                         // submitted and not yet completed (inbox + slots)
        bool stop = false;
        bool healthy = true;                      // false: ejected after `eject_after` consecutive engine failures (guarded by mu)
        int engine_failures = 0;                  // consecutive; a successful step resets it (guarded by mu)
        int eject_after = 2;
        std::function<int(RoutedRequest *)> reroute;   // the router's submit(): where unstarted requests of an ejected replica go
        std::function<void()> freed;                   // tells the router that a request completed here (someone may be waiting for room)
        std::function<void()> moved;                   // a request of this replica was re-routed: the router-wide count it held here is released
        std::atomic<uint64_t> steps{0};
        std::thread thread;
        Replica(Engine &e, size_t max_cached) : sched(e, max_cached), capacity(e.max_batch) {}
        int load() const { return inflight; }

        // The replica's `infer` thread: admit what is in the inbox (slot choice + cache checkout, Scheduler::queue), run one
        // device step over every slot with tokens pending, sample the slots whose prompt / token has been consumed.
        void run() {
            std::vector<RoutedRequest *> owner((size_t)capacity, nullptr);
            std::deque<RoutedRequest *> parked;                                    // admitted by submit() but no slot yet: retried after a step
            for (;;) {
                std::deque<RoutedRequest *> fresh;
                {
                    std::unique_lock<std::mutex> g(mu);
                    cv.wait(g, [&] { return stop || !inbox.empty() || !parked.empty() || active(owner); });
                    if (stop && inbox.empty() && parked.empty() && !active(owner)) return;
                    fresh.swap(inbox);
                }
                fresh.insert(fresh.begin(), parked.begin(), parked.end());
                parked.clear();
                // One replica's failure (a device error, a malformed request) must not take the process down (an exception leaving a
                // std::thread is std::terminate) nor wedge drain(): everything this replica holds in a slot is failed with the
                // message and its slots are given up.  Engine failures count towards ejection (see the header comment).
                bool engine_ok = true, stepped = false;
                try {
                    while (!fresh.empty()) {
                        RoutedRequest *rq = fresh.front();
                        int b = -1;
                        const RnnOption opt = rq->embed_layer >= 0 ? RnnOption::None : RnnOption::Last;
                        if (sched.queue(rq->tokens, b, opt) == SlotResult::Failure) break;   // every slot busy: park the rest until one frees
                        fresh.pop_front();
                        owner[(size_t)b] = rq;
                    }
                    parked.swap(fresh);
                    if (sched.pending()) { sched.step(); ++steps; stepped = true; }
                } catch (const std::exception &ex) {
                    engine_ok = false;
                    engine_failed(owner, fresh, parked, ex.what());
                } catch (...) {
                    engine_ok = false;
                    engine_failed(owner, fresh, parked, "unknown exception in the replica thread");
                }
                if (!engine_ok) continue;
                // State-kind requests whose tokens have been read in: the rows leave (asynchronously where the engine can), one wait for
                // all of them, then the slots are given up without caching and the requests are done
                try {
                    std::vector<int> leaving;
                    for (int b = 0; b < capacity; ++b) {
                        RoutedRequest *rq = owner[(size_t)b];
                        if (!rq || rq->embed_layer < 0 || !sched.request(b).suffix.empty()) continue;
                        sched.embed(b, rq->embed_layer, rq->embed_dst);
                        leaving.push_back(b);
                    }
                    if (!leaving.empty()) sched.embed_sync();
                    for (int b : leaving) {
                        RoutedRequest *rq = owner[(size_t)b];
                        sched.abort(b);
                        owner[(size_t)b] = nullptr;
                        complete(rq, nullptr);
                    }
                } catch (const std::exception &ex) {
                    engine_failed(owner, fresh, parked, ex.what());
                    continue;
                } catch (...) {
                    engine_failed(owner, fresh, parked, "unknown exception while reading embeddings back");
                    continue;
                }
                // the failure count is consecutive ITERATIONS that failed: reset only when the step AND the read-back went through (a replica
                // whose steps succeed but whose embed / embed_sync throws every time must still reach `eject_after`)
                if (stepped) { std::lock_guard<std::mutex> g(mu); engine_failures = 0; }
                for (int b = 0; b < capacity; ++b) {
                    RoutedRequest *rq = owner[(size_t)b];
                    if (!rq || rq->embed_layer >= 0) continue;
                    try {
                        auto &r = sched.request(b);
                        if (!r.suffix.empty() || r.output.empty()) continue;           // still reading tokens in
                        if ((int)rq->generated.size() < rq->max_new) {
                            const uint32_t t = rq->sample ? rq->sample(r.output)
                                                          : (uint32_t)(std::max_element(r.output.begin(), r.output.end()) - r.output.begin());
                            rq->generated.push_back(t);
                            sched.push(b, t);
                            continue;
                        }
                        rq->last_output = r.output;
                        sched.finish(b);
                        owner[(size_t)b] = nullptr;
                        complete(rq, nullptr);
                    } catch (const std::exception &ex) {                               // the request's own callback (or its bookkeeping) threw:
                        fail_slot(owner, b, ex.what());                                // that request fails, the replica is not to blame
                    } catch (...) {
                        fail_slot(owner, b, "unknown exception while sampling");
                    }
                }
            }
        }
        void fail_slot(std::vector<RoutedRequest *> &owner, int b, const char *what) {
            try { sched.abort(b); } catch (...) {}
            RoutedRequest *rq = owner[(size_t)b];
            owner[(size_t)b] = nullptr;
            if (rq) complete(rq, what);
        }
        // The engine threw: requests that sit in a slot have lost their state and fail; the failure counts.  On ejection the requests
        // that had not started yet (no slot, nothing consumed) go back to the router, which places them on a healthy replica or fails
        // them when none is left; below the threshold they stay parked here and ride the next step.
        void engine_failed(std::vector<RoutedRequest *> &owner, std::deque<RoutedRequest *> &fresh, std::deque<RoutedRequest *> &parked, const char *what) {
            bool ejected = false;
            std::deque<RoutedRequest *> orphans;
            {                                                                      // health first: whoever sees a request of this step
                std::lock_guard<std::mutex> g(mu);                                 // completed also sees the replica's new standing
                if (++engine_failures >= eject_after) {
                    healthy = false;
                    ejected = true;
                    orphans.swap(inbox);
                }
            }
            for (int b = 0; b < capacity; ++b) {
                try { sched.abort(b); } catch (...) {}                             // Idle, nothing cached from a step that failed
                if (owner[(size_t)b]) { complete(owner[(size_t)b], what); owner[(size_t)b] = nullptr; }
            }
            for (auto *q : {&fresh, &parked}) { for (RoutedRequest *rq : *q) orphans.push_back(rq); q->clear(); }
            if (!ejected) { parked.swap(orphans); return; }                        // transient so far: the unstarted requests wait for the next step
            for (RoutedRequest *rq : orphans) {
                { std::lock_guard<std::mutex> g(mu); --inflight; }
                rq->replica = -1;
                if (!reroute || reroute(rq) < 0) {
                    { std::lock_guard<std::mutex> g(mu); ++inflight; }             // complete() takes it off again (and the router-wide count)
                    complete(rq, (std::string(what) + " (replica ejected; no healthy replica has room)").c_str());
                } else if (moved) {
                    moved();                                                       // placed elsewhere (counted there): give this replica's router-wide count back
                }
            }
            idle_cv.notify_all();
        }
        void complete(RoutedRequest *rq, const char *err) {
            {
                std::lock_guard<std::mutex> g(mu);
                if (err) { rq->failed = true; rq->error = err; }
                rq->done = true;
                --inflight;
            }
            idle_cv.notify_all();
            if (freed) freed();
        }
        static bool active(const std::vector<RoutedRequest *> &o) { for (auto *p : o) if (p) return true; return false; }
    };
    std::vector<std::unique_ptr<Replica>> reps_;
    struct Room { std::mutex mu; std::condition_variable cv; uint64_t epoch = 0; long inflight = 0; };   // epoch advances whenever a request completes on any replica; inflight: submitted, not completed, router-wide
    std::shared_ptr<Room> room_ = std::make_shared<Room>();                           // shared with the replicas' `freed` callbacks
};

}  // namespace rwkv
