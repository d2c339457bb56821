// router_loop.cpp — include/rwkv_router.hpp over REAL engines: N replicas of one model, replica r on device devices[r % len] —
// `0,1,2,3,4,5,6,7` puts one replica on each GPU of a node (the serving path of SURVEY §8e: router + one driving thread per
// engine, no collective), `0` or `0,0` puts them all on one device so the N>1 path can be exercised on a one-GPU box.
// Requests are routed by prefix affinity then least busy.  Prints one line per request: "<replica> <generated token ids...>"
// (tests/test_gpu_parity.py compares with the oracle), then the follow-up request, "meta ..." and, on stderr, the wall time and
// the aggregate rate of the first wave; then the prompts once more as an `/embeddings` batch job over the replicas ("emb ..." lines).
// Usage: router_loop <model.st> <n_replicas> <device[,device...]> <max_batch> <chunk> <n_new> <prompt ...> [/ <prompt ...>]...
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>

#include "../include/rwkv_router.hpp"

int main(int argc, char **argv) {
    if (argc < 8) { std::fprintf(stderr, "usage: see header\n"); return 2; }
    try {
        std::ifstream f(argv[1], std::ios::binary);
        std::vector<uint8_t> st((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        const int n_rep = std::atoi(argv[2]), B = std::atoi(argv[4]), chunk = std::atoi(argv[5]), n_new = std::atoi(argv[6]);
        std::vector<int> devices;
        for (const char *p = argv[3]; *p;) {
            char *end = nullptr;
            devices.push_back((int)std::strtol(p, &end, 10));
            if (end == p) { std::fprintf(stderr, "bad device list\n"); return 2; }
            p = *end == ',' ? end + 1 : end;
        }
        if (devices.empty() || n_rep < 1) { std::fprintf(stderr, "bad device list\n"); return 2; }
        std::vector<rwkv::Tokens> prompts(1);
        for (int i = 7; i < argc; ++i) {
            if (!std::strcmp(argv[i], "/")) prompts.emplace_back();
            else prompts.back().push_back((uint32_t)std::strtoul(argv[i], nullptr, 10));
        }
        std::vector<rwkv::Runtime> rts;
        for (int r = 0; r < n_rep; ++r)
            rts.push_back(rwkv::ModelBuilder(st.data(), st.size(), devices[(size_t)r % devices.size()]).build(B, chunk, rwkv::Precision::Fp16));
        std::vector<rwkv::Runtime *> es;
        for (auto &r : rts) es.push_back(&r);
        std::vector<rwkv::RoutedRequest> reqs(prompts.size());
        {
            rwkv::ReplicaRouter<rwkv::Runtime> router(es);
            const auto t0 = std::chrono::steady_clock::now();
            for (size_t i = 0; i < prompts.size(); ++i) {
                reqs[i].tokens = prompts[i];
                reqs[i].max_new = n_new;
                while (router.submit(&reqs[i]) < 0) std::this_thread::yield();   // every replica full: retry, like `enqueue`
            }
            router.drain();
            {
                const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                size_t toks = 0;
                for (auto &q : reqs) toks += q.tokens.size() + q.generated.size();
                std::fprintf(stderr, "first wave: %zu requests over %d replicas on %zu device(s): %.3f s, %.0f tokens/s (prompt + generated)\n",
                             reqs.size(), n_rep, devices.size(), sec, (double)toks / sec);
            }
            // second wave: request 0 again, extended by its own output — must return to the replica that cached it
            rwkv::RoutedRequest again;
            again.tokens = prompts[0];
            again.tokens.insert(again.tokens.end(), reqs[0].generated.begin(), reqs[0].generated.end());
            again.tokens.push_back(reqs[0].generated.empty() ? 1u : reqs[0].generated.back());
            again.max_new = n_new;
            const auto where = router.route(again.tokens);
            router.submit(&again);
            router.drain();
            for (auto &q : reqs) {
                std::printf("%d", q.replica);
                for (auto t : q.generated) std::printf(" %u", t);
                std::printf("\n");
            }
            std::printf("%d", again.replica);
            for (auto t : again.generated) std::printf(" %u", t);
            std::printf("\nmeta %d %zu %d\n", where.first, where.second, reqs[0].replica);
            // the `/embeddings` batch job over the replicas: the prompts as documents, State-kind requests routed like any other, the last
            // layer's WKV rows read back asynchronously into one pinned block.  One line per document: "emb <sum> <v[0]> <v[L/3]> <v[L-1]>"
            const size_t L = rts[0].state.layer_len();
            rwkv::PinnedBuffer emb(prompts.size() * L);
            const uint64_t steps_before[2] = {router.steps(0), router.steps(n_rep > 1 ? 1 : 0)};
            router.embed_documents(prompts, rts[0].info.num_layer - 1, emb.data(), L);
            for (size_t d = 0; d < prompts.size(); ++d) {
                const float *v = emb.data() + d * L;
                double sum = 0.0;
                for (size_t i = 0; i < L; ++i) sum += v[i];
                std::printf("emb %.9e %.9e %.9e %.9e\n", sum, v[0], v[L / 3], v[L - 1]);
            }
            std::printf("embsteps %llu %llu\n", (unsigned long long)(router.steps(0) - steps_before[0]),
                        (unsigned long long)(router.steps(n_rep > 1 ? 1 : 0) - steps_before[1]));
        }
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
