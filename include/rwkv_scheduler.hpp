// rwkv_scheduler.hpp — C++ mirror of the scheduling core of ai00-core's runtime (crates/ai00-core/src/run.rs), on top
// of anything shaped like rwkv::Runtime (include/rwkv_runtime.hpp).  SURVEY §8 row f-3.
//
//   SlotState / SlotChoice     run.rs:289-331   which idle slot takes a new request: continue > empty > back, ties by idle age
//   PrefixCache::checkout      run.rs:441-485   longest cached token prefix -> (state, last output); miss -> initial state
//   Scheduler::queue           run.rs:488-626   pick a slot, check the state out of the cache, load it, split prefix / suffix
//   Scheduler::step            run.rs:1113-1157 one `infer` over EVERY slot that has tokens pending
//   Scheduler::finish          run.rs:629-662   busy slot -> Idle(content), state + output cached under the content
//   Scheduler::perplexity      run.rs:699-755   Full rows of a token list -> -mean ln p of the realised tokens
//   Scheduler::choose / state  run.rs:936-989   GenerateKind::Choose (perplexity per choice, optional calibration) / ::State
//
// Differences from the reference, on purpose: (1) synchronous — the caller owns the thread (the reference spreads this over
// tokio tasks and channels; the decisions are the same); (2) `step()` re-collects the pending tokens of all busy slots on
// every call, so a request queued while others are mid-flight rides the very next device step (the reference only forms a
// batch from what is already waiting when the infer task wakes up: run.rs:1120-1132, the "opportunistic batch gap");
// (3) the cache is bounded by item count (`max_cached`), evicting the stalest items.
//
// Header-only, no HIP: `Engine` needs  int max_batch;  ModelInfo info;  State state (init/load/back);
//                                       std::vector<RnnOutputBatch> infer(RnnInput &)  (consumes tokens in place).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "rwkv_runtime.hpp"

namespace rwkv {

using Tokens = std::vector<uint32_t>;

struct CachedItem {                              // run.rs:201-223
    // Immutable once cached and shared by reference: a checkout hands out the pointers (the reference clones an `Arc`'d tensor the
    // same way, `backed.clone()` run.rs:962,979), so no lock is ever held across the copy of a 21 MB slab.
    std::shared_ptr<const std::vector<float>> state;    // public slab [C, N+2, L, 1]
    std::shared_ptr<const std::vector<float>> output;   // logits of the last token of the cached prefix
    uint64_t stamp = 0;                          // logical clock (the reference uses Instant)
};

class PrefixCache {                              // `Trie<Tokens, CachedItem>` keyed by whole token sequences
   public:
    explicit PrefixCache(size_t max_cached = 256) : max_cached_(max_cached) {}      // MAX_CACHE_ITEMS, run.rs:41
    PrefixCache(const PrefixCache &) = delete;
    PrefixCache &operator=(const PrefixCache &) = delete;
    // A key is one trie node per token, so a cached 100k-token context is a 100k-deep chain of owning pointers: letting the
    // default destructors unwind it recurses once per token and overflows the stack.  Tear the trie down iteratively.
    ~PrefixCache() {
        std::vector<std::unique_ptr<Node>> work;
        for (auto &kv : root_.next) work.push_back(std::move(kv.second));
        root_.next.clear();
        while (!work.empty()) {
            std::unique_ptr<Node> n = std::move(work.back());
            work.pop_back();
            for (auto &kv : n->next) work.push_back(std::move(kv.second));
            n->next.clear();
        }                                                                           // n dies here with no children left
    }
    struct Checkout { size_t prefix_len = 0; std::shared_ptr<const std::vector<float>> state, output; bool hit = false; };
    // longest cached key that is a prefix of `tokens` (run.rs:447-455); refreshes the item's stamp (CachedItem::update)
    Checkout checkout(const Tokens &tokens, uint64_t now) {
        std::lock_guard<std::mutex> g(mu_);
        Checkout c;
        Node *n = &root_;
        Node *best = nullptr;
        size_t best_len = 0;
        for (size_t i = 0; i < tokens.size(); ++i) {
            auto it = n->next.find(tokens[i]);
            if (it == n->next.end()) break;
            n = it->second.get();
            if (n->item) { best = n; best_len = i + 1; }
        }
        if (best) {
            touch(best, now);
            c.prefix_len = best_len; c.state = best->item->state; c.output = best->item->output; c.hit = true;
        }
        return c;
    }
    // length of the longest cached key that is a prefix of `tokens`, without touching stamps (the router's affinity probe)
    size_t match_len(const Tokens &tokens) const {
        std::lock_guard<std::mutex> g(mu_);
        const Node *n = &root_;
        size_t best = 0;
        for (size_t i = 0; i < tokens.size(); ++i) {
            auto it = n->next.find(tokens[i]);
            if (it == n->next.end()) break;
            n = it->second.get();
            if (n->item) best = i + 1;
        }
        return best;
    }
    bool contains(const Tokens &tokens) const {                                     // `cache.contains_key`, run.rs:795
        std::lock_guard<std::mutex> g(mu_);
        const Node *n = &root_;
        for (uint32_t t : tokens) {
            auto it = n->next.find(t);
            if (it == n->next.end()) return false;
            n = it->second.get();
        }
        return n != &root_ && n->item != nullptr;
    }
    void insert(const Tokens &tokens, std::vector<float> state, std::vector<float> output, uint64_t now) {
        if (tokens.empty()) return;
        std::lock_guard<std::mutex> g(mu_);
        Node *n = &root_;
        for (uint32_t t : tokens) {
            auto &slot = n->next[t];
            if (!slot) { slot.reset(new Node()); slot->parent = n; slot->tok = t; }
            n = slot.get();
        }
        if (!n->item) { n->item.reset(new CachedItem()); ++count_; n->item->stamp = now; n->age = by_age_.emplace(now, n); }
        n->item->state = std::make_shared<const std::vector<float>>(std::move(state));
        n->item->output = std::make_shared<const std::vector<float>>(std::move(output));
        touch(n, now);
        while (count_ > max_cached_) evict_oldest();                                // `Cache::maintain`, run.rs:237-257
    }
    size_t size() const { std::lock_guard<std::mutex> g(mu_); return count_; }
    size_t nodes() const { std::lock_guard<std::mutex> g(mu_); return count_nodes(&root_) - 1; }   // trie nodes alive (tests: eviction prunes)

   private:
    struct Node {
        std::map<uint32_t, std::unique_ptr<Node>> next;
        std::unique_ptr<CachedItem> item;
        Node *parent = nullptr;
        uint32_t tok = 0;
        std::multimap<uint64_t, Node *>::iterator age;                              // valid while `item` is set
    };
    static size_t count_nodes(const Node *root) {                                   // iterative for the same reason as the destructor
        size_t c = 0;
        std::vector<const Node *> work{root};
        while (!work.empty()) {
            const Node *n = work.back();
            work.pop_back();
            ++c;
            for (auto &kv : n->next) work.push_back(kv.second.get());
        }
        return c;
    }
    void touch(Node *n, uint64_t now) {
        by_age_.erase(n->age);
        n->item->stamp = now;
        n->age = by_age_.emplace(now, n);
    }
    // the oldest item goes, and with it every trie node that now leads nowhere: the cache of a long-running server holds at most
    // `max_cached` items AND at most the nodes on their key paths (an item-only eviction would leak one node per token ever cached)
    void evict_oldest() {
        if (by_age_.empty()) return;
        Node *n = by_age_.begin()->second;
        by_age_.erase(by_age_.begin());
        n->item.reset();
        --count_;
        while (n != &root_ && !n->item && n->next.empty()) {
            Node *parent = n->parent;
            parent->next.erase(n->tok);                                             // destroys n
            n = parent;
        }
    }
    // one engine's scheduler owns the cache, but the router thread probes it (match_len) while that replica's thread inserts:
    // every public method takes the lock (the critical sections are a trie walk plus, for checkout, one state copy)
    mutable std::mutex mu_;
    Node root_;
    std::multimap<uint64_t, Node *> by_age_;                                       // stamp -> node: eviction is O(log n), not a trie walk
    size_t count_ = 0, max_cached_;
};

enum class SlotKind { Idle, Busy };
struct SlotState {                               // run.rs:289-302 (Locked only exists between awaits there)
    SlotKind kind = SlotKind::Idle;
    Tokens content;                              // Idle: the tokens whose state the slot holds; Busy: prefix consumed so far
    uint64_t since = 0;                          // Idle: when it became idle
};

struct SlotChoice {                              // run.rs:304-331
    enum Kind { Back = 0, Empty = 1, Continue = 2 } kind;
    int batch;
    size_t len;                                  // Continue: length of the matching content
    // priority: continue (longer match first) > empty > back
    static int cmp(const SlotChoice &a, const SlotChoice &b) {
        if (a.kind == Continue && b.kind == Continue) return a.len < b.len ? -1 : a.len > b.len ? 1 : 0;
        return a.kind < b.kind ? -1 : a.kind > b.kind ? 1 : 0;
    }
};

// Stop-string scan of the `process` loop (run.rs:899-932), byte for byte: for every stop string walk the text buffer keeping
// `index_safe` (everything before it can no longer be part of a match) and report (index_safe, fully matched); the winner is a
// matched stop before an unmatched one, then the smallest index.  The caller emits buffer[..head] and keeps the tail.
struct StopScan { size_t head = 0; bool matched = false; };
inline StopScan scan_stops(const std::vector<uint8_t> &buffer, const std::vector<std::string> &stops) {
    bool have = false;
    StopScan best{buffer.size(), false};                              // no stop strings: everything is safe (`unwrap_or`)
    for (const std::string &stop : stops) {
        size_t index_safe = 0, index_unsafe = 0;
        StopScan cur;
        bool done = false;
        while (index_unsafe < buffer.size()) {
            const size_t index_stop = index_unsafe - index_safe;
            if (index_stop >= stop.size()) { cur = StopScan{index_safe, true}; done = true; break; }
            const uint8_t out = buffer[index_unsafe], st = (uint8_t)stop[index_stop];
            ++index_unsafe;
            if (out != st) index_safe = index_unsafe;
        }
        if (!done) cur = StopScan{index_safe, index_unsafe - index_safe >= stop.size()};
        const bool better = !have || (cur.matched && !best.matched) || (cur.matched == best.matched && cur.head < best.head);
        if (better) { best = cur; have = true; }
    }
    return best;
}

enum class SlotResult { Success, Fault, Failure };   // run.rs: Success(batch) / Fault(batch) (had to back a slot) / Failure (all busy)

template <class Engine>
class Scheduler {
   public:
    struct Request {
        Tokens prefix, suffix;                   // consumed / still to feed (GenerateContext::{prefix, suffix})
        std::vector<float> output;               // logits after the last consumed token (empty until one exists)
        RnnOption option = RnnOption::Last;
        std::vector<std::vector<float>> rows;    // Full: one entry per emitted row
        uint64_t state_id = 0;                   // `request.state.id()`: which initial state / cache the request lives in (run.rs:443-447)
        size_t prompt_len = 0;                   // tokens of the request as queued (the "prompt", run.rs:794)
        bool cache_prompt = false;               // CachedPrompt::Future: cache the state when the prompt has been consumed
    };

    static constexpr size_t kMinPromptCacheTokens = 32;                          // MIN_PROMPT_CACHE_TOKENS, run.rs:40
    explicit Scheduler(Engine &e, size_t max_cached = 256) : e_(e), slots_(e.max_batch), reqs_(e.max_batch), max_cached_(max_cached) {
        cache_of(0);                                                               // the default cache exists from the start
    }

    // `check_in_state` (run.rs:376-437): a state-tuned initial state registered under an id gets its OWN prefix cache
    // (`Cache { state: Some(state), cache: Trie::new() }`); requests that name the id start from it instead of the zero state.
    // `slab` is what rwkv_read_init_state / `vN::read_state` produced (lib.rs:378-389).  Id 0 is the default (no init state).
    void check_in_state(uint64_t id, std::vector<float> slab) {
        if (id == 0) throw std::invalid_argument("state id 0 is the default state");
        init_states_[id] = std::move(slab);
        std::lock_guard<std::mutex> g(cache_mu_);
        caches_.erase(id);
    }

    // run.rs:488-626.  On Success / Fault `batch` is the slot now Busy with the request.
    SlotResult queue(Tokens tokens, int &batch, RnnOption option = RnnOption::Last, uint64_t state_id = 0) {
        if (state_id != 0 && !init_states_.count(state_id)) throw std::invalid_argument("unknown state id");
        if (tokens.empty()) tokens = {0};                              // run.rs:489-492
        ++clock_;
        bool have = false;
        SlotChoice best{SlotChoice::Back, -1, 0};
        uint64_t best_idle = 0;
        for (int b = 0; b < (int)slots_.size(); ++b) {
            const SlotState &s = slots_[b];
            if (s.kind != SlotKind::Idle) continue;
            SlotChoice c{SlotChoice::Back, b, 0};
            if (s.content.empty()) c.kind = SlotChoice::Empty;
            else if (s.content.size() <= tokens.size() && std::equal(s.content.begin(), s.content.end(), tokens.begin())) {
                c.kind = SlotChoice::Continue; c.len = s.content.size();
            }
            const uint64_t idle = clock_ - s.since;                    // `instant.elapsed()`
            const int k = have ? SlotChoice::cmp(c, best) : 1;
            // max_by keeps the LAST maximum on ties (Iterator::max_by), so `>=` on the secondary key
            if (!have || k > 0 || (k == 0 && idle >= best_idle)) { best = c; best_idle = idle; have = true; }
        }
        if (!have) return SlotResult::Failure;                         // every slot is busy: hand the request back
        batch = best.batch;
        // check the state out of the cache (longest cached prefix, else the initial state) and load it into the slot.
        // (The reference does this for all three choices, Continue included: run.rs:548-626.)
        // A request that is cached whole starts with an empty suffix and the cached output row: the process loop samples from
        // it without touching the engine (`(0, Some(output)) => output`, run.rs:809-811).
        // `cache_mu_` guards the MAP of caches only and is held just long enough to find this request's trie (map entries are
        // node-stable, and only this thread — the engine's one driving thread — ever erases one, in check_in_state); the trie has its
        // own mutex, and a checkout copies two pointers under it, not a slab: a router thread probing `match_len` on this replica
        // never waits behind a 21 MB memcpy.
        PrefixCache *cache = nullptr;
        { std::lock_guard<std::mutex> g(cache_mu_); cache = &cache_of(state_id); }
        PrefixCache::Checkout co = cache->checkout(tokens, clock_);
        const bool cached_whole = cache->contains(tokens);
        const size_t len = co.hit ? co.prefix_len : 0;
        if (co.hit) e_.state.load(*co.state, batch);
        else if (state_id != 0) e_.state.load(init_states_[state_id], batch);       // `state.unwrap_or_else(|| self.state.init())`, run.rs:476-477
        else load_init(batch);
        Request r;
        r.prefix.assign(tokens.begin(), tokens.begin() + (long)len);
        r.suffix.assign(tokens.begin() + (long)len, tokens.end());
        r.output = co.hit ? *co.output : std::vector<float>();
        r.option = option;
        r.state_id = state_id;
        r.prompt_len = tokens.size();
        // run.rs:794-803: prompts longer than MIN_PROMPT_CACHE_TOKENS that are not cached yet get a cache entry as soon as
        // they have been read in (so a second request with the same long prompt skips its prefill even while this one decodes)
        r.cache_prompt = tokens.size() > kMinPromptCacheTokens && !cached_whole;
        // A state-only request (RnnOption::None, the `/embeddings` job) never receives an output row from the engine: the row copied
        // from the cache above belongs to the SHORTER cached prefix, and caching the whole document under it would hand a later
        // completion with the same prompt the wrong logits (and cost a full slab read-back per document).  The reference runs State
        // requests with `Last`, so its cache entries always carry their own row; here such requests neither carry nor create one.
        if (option == RnnOption::None) { r.output.clear(); r.cache_prompt = false; }
        reqs_[batch] = std::move(r);
        const bool back = best.kind == SlotChoice::Back;
        slots_[batch].kind = SlotKind::Busy;
        slots_[batch].content.clear();
        return back ? SlotResult::Fault : SlotResult::Success;
    }

    // feed more tokens to a busy slot (the decode loop appends the sampled token: run.rs:1004-1010)
    void push(int batch, uint32_t token) { need_busy(batch); reqs_[batch].suffix.push_back(token); }
    bool pending() const {
        for (size_t b = 0; b < slots_.size(); ++b) if (slots_[b].kind == SlotKind::Busy && !reqs_[b].suffix.empty()) return true;
        return false;
    }

    // one device step over every busy slot with tokens pending; returns the number of slots that rode it
    int step() { return step_impl(nullptr); }

    // `perplexity` (run.rs:699-755).  `tokens` ride the engine with RnnOption::Full on slot `batch` (whose request must have
    // nothing pending); row j (j = 1, 2, …) yields softmax(row)[tokens'[j]] — exp / sum WITHOUT max subtraction, as the
    // reference computes it — where tokens' = tokens when the probability of tokens[0] is known (`head`, the sampled
    // distribution of the prompt's last token), else [0] ++ tokens; result = -sum(ln p) / len(tokens').  The slot's state
    // advances by tokens'; every other busy slot with tokens pending rides the same device steps.
    float perplexity(int batch, const Tokens &tokens, const float *head) {
        need_busy(batch);
        if (!reqs_[(size_t)batch].suffix.empty()) throw std::logic_error("perplexity(): the slot still has tokens pending");
        Probe pr;
        pr.batch = batch;
        if (head) { pr.p.push_back(*head); pr.all = tokens; }
        else { pr.all.push_back(0); pr.all.insert(pr.all.end(), tokens.begin(), tokens.end()); }
        pr.left = pr.all;
        pr.want = tokens.size();
        while (!pr.left.empty()) step_impl(&pr);
        double acc = 0.0;
        for (float x : pr.p) acc += std::log((double)x);
        return (float)(-acc / (double)pr.all.size());
    }

    // GenerateKind::Choose (run.rs:936-979): once the prompt has been read in, score every non-empty choice by the perplexity
    // of its tokens continuing the prompt (`head` = probs[choice[0]], `probs` being the distribution `sample()` returned for
    // the prompt's last token; empty `probs` = plain softmax of the request's last logits); with `calibrate`, first add minus
    // the perplexity of the choice on its own from the request's initial state.  The slot's state is snapshotted before and put
    // back after every evaluation (`read` / `write`, run.rs:938, 960, 976), so the request can go on afterwards.  Empty
    // choices score +inf.
    std::vector<float> choose(int batch, const std::vector<Tokens> &choices, bool calibrate, std::vector<float> probs = {}) {
        need_busy(batch);
        Request &r = reqs_[(size_t)batch];
        if (!r.suffix.empty() || r.output.empty()) throw std::logic_error("choose(): the prompt has not been read in yet");
        if (probs.empty()) {
            probs = r.output;
            float mx = probs[0];
            for (float x : probs) mx = std::max(mx, x);
            double sum = 0.0;
            for (float &x : probs) { x = std::exp(x - mx); sum += x; }
            for (float &x : probs) x = (float)(x / sum);
        }
        auto backed = snapshot(e_.state, batch, 0);
        std::vector<float> ppl(choices.size(), std::numeric_limits<float>::infinity());
        if (calibrate) {
            const std::vector<float> init = r.state_id != 0 ? init_states_.at(r.state_id) : e_.state.init();
            for (size_t i = 0; i < choices.size(); ++i) {
                if (choices[i].empty()) continue;
                e_.state.load(init, batch);
                ppl[i] = -perplexity(batch, choices[i], nullptr);
            }
            restore(e_.state, backed, batch, 0);
        }
        for (size_t i = 0; i < choices.size(); ++i) {
            if (choices[i].empty()) continue;
            const float head = probs.at(choices[i][0]);
            const float p = perplexity(batch, choices[i], &head);
            ppl[i] = calibrate ? ppl[i] + p : p;
            restore(e_.state, backed, batch, 0);
        }
        return ppl;
    }

    // GenerateKind::State (run.rs:980-985): the slot's state slab as it stands (`Token::Embed(embed, shape)`).
    std::vector<float> state(int batch) { need_busy(batch); return e_.state.back(batch); }

    Request &request(int batch) { need_busy(batch); return reqs_[batch]; }

    // run.rs:629-662 + the cache write at the end of `process` (run.rs:1012-1020): the slot becomes Idle(content) and its
    // state + last output are cached under the content, so a later request with that prefix continues from it.
    void finish(int batch) {
        need_busy(batch);
        Request &r = reqs_[batch];
        if (!r.suffix.empty()) throw std::logic_error("finish(): tokens still pending");
        ++clock_;
        if (!r.prefix.empty() && !r.output.empty()) {
            auto slab = e_.state.back(batch);                                      // device round trip OUTSIDE any lock
            PrefixCache *cache = nullptr;
            { std::lock_guard<std::mutex> g(cache_mu_); cache = &cache_of(r.state_id); }
            cache->insert(r.prefix, std::move(slab), r.output, clock_);
        }
        slots_[batch].kind = SlotKind::Idle;
        slots_[batch].content = r.prefix;
        slots_[batch].since = clock_;
    }

    // The `/embeddings` route (docs/doc-api/openai.md:376-437; `GenerateKind::State` run.rs:980-989, api/oai/state.rs:29-40) fed a
    // LIST of documents, as a batch job with slot turnover: every document is queued as a request of its own (RnnOption::None: the
    // state is all it wants — no head GEMM, nothing for the prefix cache), an idle slot takes the next waiting one, all busy slots ride
    // the same device steps, and a slot whose document has been read in hands over layer `layer`'s WKV rows ([head_size][num_emb]
    // floats at out + doc * layer_len) and is free again.  With an engine whose State has embed_async / sync (rwkv::State) the rows
    // leave on the copy stream into `out` (PINNED memory then) while the next documents prefill; otherwise the copy is waited for.
    // Slots busy with other requests keep riding the steps.  Returns the number of device steps.  (Python twin: harness.StateJob.)
    size_t embed_documents(const std::vector<Tokens> &docs, int layer, float *out) {
        const size_t L = e_.state.layer_len();
        std::vector<long> owner(slots_.size(), -1);
        size_t next = 0, busy = 0, calls = 0;
        try {
            while (next < docs.size() || busy) {
                while (next < docs.size()) {
                    int b = -1;
                    if (queue(docs[next], b, RnnOption::None) == SlotResult::Failure) break;
                    owner[(size_t)b] = (long)next++;
                    ++busy;
                }
                if (!busy) throw std::runtime_error("embed_documents(): every slot is held by another request");
                step();
                ++calls;
                for (size_t b = 0; b < owner.size(); ++b) {
                    if (owner[b] < 0 || !reqs_[b].suffix.empty()) continue;
                    embed_out(e_.state, layer, (int)b, out + (size_t)owner[b] * L, 0);
                    abort((int)b);                                                  // Idle with no content: the next document finds it Empty
                    owner[b] = -1;
                    --busy;
                }
            }
            embed_sync(e_.state, 0);
        } catch (...) {
            for (size_t b = 0; b < owner.size(); ++b) if (owner[b] >= 0) abort((int)b);
            throw;
        }
        return calls;
    }

    // One layer's WKV rows of a busy slot to `dst` — through the engine's asynchronous read-back when it has one (valid after
    // embed_sync(); `dst` pinned then), else copied before the call returns.  What the router's State-kind requests use.
    void embed(int batch, int layer, float *dst) { need_busy(batch); embed_out(e_.state, layer, batch, dst, 0); }
    void embed_sync() { embed_sync(e_.state, 0); }
    size_t embed_len() const { return e_.state.layer_len(); }

    // give a busy slot up without caching anything (its engine call failed, or the request was cancelled): Idle, no content
    void abort(int batch) {
        if (batch < 0 || batch >= (int)slots_.size()) return;
        reqs_[(size_t)batch] = Request();
        slots_[(size_t)batch].kind = SlotKind::Idle;
        slots_[(size_t)batch].content.clear();
        slots_[(size_t)batch].since = ++clock_;
    }

    const SlotState &slot(int batch) const { return slots_.at((size_t)batch); }
    // single-threaded access (tests, serve_loop); other threads ask through match_len()
    PrefixCache &cache(uint64_t state_id = 0) { return cache_of(state_id); }
    // length of the longest cached prefix of `tokens` — the router's affinity query.  Safe from any thread while the scheduler's
    // own thread runs: never inserts a cache, and every mutation of the tries happens under the same mutex.
    size_t match_len(const Tokens &tokens, uint64_t state_id = 0) const {
        std::lock_guard<std::mutex> g(cache_mu_);
        auto it = caches_.find(state_id);
        return it == caches_.end() ? 0 : it->second.match_len(tokens);
    }

   private:
    struct Probe {                                // a Full run on one slot that is not part of its request (perplexity)
        int batch = -1;
        Tokens all, left;                         // tokens' and what of it is still to feed
        std::vector<float> p;
        size_t index = 1, want = 0;
    };
    int step_impl(Probe *probe) {
        RnnInput in;
        in.batches.resize(slots_.size());
        int riders = 0;
        for (size_t b = 0; b < slots_.size(); ++b) {
            if (probe && (int)b == probe->batch) {
                in.batches[b].tokens = probe->left;
                in.batches[b].option = RnnOption::Full;
                ++riders;
                continue;
            }
            if (slots_[b].kind != SlotKind::Busy || reqs_[b].suffix.empty()) continue;
            in.batches[b].tokens = reqs_[b].suffix;
            in.batches[b].option = reqs_[b].option;
            ++riders;
        }
        if (!riders) return 0;
        std::vector<RnnOutputBatch> out = e_.infer(in);
        const size_t V = (size_t)e_.info.num_vocab;
        for (size_t b = 0; b < slots_.size(); ++b) {
            if (probe && (int)b == probe->batch) {
                probe->left = in.batches[b].tokens;                                 // what the engine did not consume yet
                for (size_t row = 0; row + 1 <= out[b].size() / V; ++row, ++probe->index) {
                    if (probe->p.size() >= probe->want || probe->index >= probe->all.size()) continue;
                    const float *lg = out[b].data() + row * V;
                    float sum = 0.f;
                    for (size_t v = 0; v < V; ++v) sum += std::exp(lg[v]);
                    probe->p.push_back(std::exp(lg[probe->all[probe->index]]) / sum);
                }
                continue;
            }
            if (slots_[b].kind != SlotKind::Busy || reqs_[b].suffix.empty()) continue;
            Request &r = reqs_[b];
            const size_t consumed = r.suffix.size() - in.batches[b].tokens.size();
            r.prefix.insert(r.prefix.end(), r.suffix.begin(), r.suffix.begin() + (long)consumed);
            r.suffix.erase(r.suffix.begin(), r.suffix.begin() + (long)consumed);
            for (size_t row = 0; row + 1 <= out[b].size() / V; ++row) {
                std::vector<float> lg(out[b].begin() + (long)(row * V), out[b].begin() + (long)((row + 1) * V));
                if (r.option == RnnOption::Full) r.rows.push_back(lg);
                r.output = std::move(lg);
            }
            if (r.cache_prompt && r.suffix.empty() && r.prefix.size() == r.prompt_len && !r.output.empty()) {   // run.rs:829-838
                auto slab = e_.state.back((int)b);
                PrefixCache *cache = nullptr;
                { std::lock_guard<std::mutex> g(cache_mu_); cache = &cache_of(r.state_id); }
                cache->insert(r.prefix, std::move(slab), r.output, ++clock_);
                r.cache_prompt = false;
            }
        }
        return riders;
    }
    // the initial state into a slot: uploaded once, device-resident afterwards (`state.write(backed.clone(), batch)`, run.rs:1104) — a
    // request that misses the cache does not pay a slab over PCIe
    void load_init(int batch) {
        if (!init_snap_) {
            e_.state.load(e_.state.init(), batch);
            init_snap_ = std::make_unique<InitSnap>(snapshot(e_.state, batch, 0));
        } else restore(e_.state, *init_snap_, batch, 0);
    }
    template <class S> static auto embed_out(S &s, int layer, int b, float *dst, int) -> decltype(s.embed_async(layer, b, dst)) { return s.embed_async(layer, b, dst); }
    template <class S> static void embed_out(S &s, int layer, int b, float *dst, long) { s.embed(layer, b, dst); }
    template <class S> static auto embed_sync(S &s, int) -> decltype(s.sync()) { return s.sync(); }
    template <class S> static void embed_sync(S &, long) {}
    // device-resident snapshot when the engine's State has read / write (rwkv::State), host round trip otherwise
    template <class S> static auto snapshot(S &s, int b, int) -> decltype(s.read(b)) { return s.read(b); }
    template <class S> static std::vector<float> snapshot(S &s, int b, long) { return s.back(b); }
    template <class S, class T> static auto restore(S &s, const T &t, int b, int) -> decltype(s.write(t, b)) { return s.write(t, b); }
    template <class S> static void restore(S &s, const std::vector<float> &t, int b, long) { s.load(t, b); }
    using InitSnap = decltype(snapshot(std::declval<decltype(std::declval<Engine &>().state) &>(), 0, 0));
    std::unique_ptr<InitSnap> init_snap_;
    void need_busy(int batch) const {
        if (batch < 0 || batch >= (int)slots_.size() || slots_[(size_t)batch].kind != SlotKind::Busy) throw std::invalid_argument("slot is not busy");
    }
    Engine &e_;
    std::vector<SlotState> slots_;
    std::vector<Request> reqs_;
    PrefixCache &cache_of(uint64_t id) {             // `caches.fetch(id)`: one trie per initial state (run.rs:262-287)
        auto it = caches_.find(id);
        if (it == caches_.end()) it = caches_.emplace(std::piecewise_construct, std::forward_as_tuple(id), std::forward_as_tuple(max_cached_)).first;
        return it->second;
    }
    size_t max_cached_;
    mutable std::mutex cache_mu_;                    // guards the MAP caches_ (lookup / erase); each trie is guarded by its own PrefixCache::mu_
    std::map<uint64_t, PrefixCache> caches_;
    std::map<uint64_t, std::vector<float>> init_states_;
    uint64_t clock_ = 0;
};

}  // namespace rwkv
