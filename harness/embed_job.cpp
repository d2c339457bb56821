// embed_job.cpp — the `/embeddings` batch job (docs/doc-api/openai.md:376-437; GenerateKind::State, run.rs:980-989) on the C++ side of the
// boundary: rwkv::Scheduler::embed_documents (include/rwkv_scheduler.hpp) over the real engine — documents queued as state-only requests,
// slot turnover, one layer's WKV rows read back asynchronously into pinned memory while the next documents prefill.
// Usage: embed_job <model.st> <quant_layers> <quant_type> <max_batch> <chunk> <layer> <docs.bin> <out.bin> [repeat]
//   docs.bin  u32 n_docs, then per document u32 len + len x u32 token ids
//   out.bin   float32 [n_docs][head_size][num_emb]
// Prints "steps S docs N seconds T docs_per_s R" for the last of `repeat` runs (tests/test_gpu_embeddings.py compares out.bin and S with
// the Python twin, harness.StateJob, bit for bit).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>

#include "../include/rwkv_scheduler.hpp"

int main(int argc, char **argv) {
    if (argc < 9) { std::fprintf(stderr, "usage: see header\n"); return 2; }
    try {
        std::ifstream f(argv[1], std::ios::binary);
        std::vector<uint8_t> st((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        const int ql = std::atoi(argv[2]), qt = std::atoi(argv[3]), B = std::atoi(argv[4]), chunk = std::atoi(argv[5]), layer = std::atoi(argv[6]);
        const int repeat = argc > 9 ? std::atoi(argv[9]) : 1;
        std::ifstream df(argv[7], std::ios::binary);
        std::vector<uint8_t> raw((std::istreambuf_iterator<char>(df)), std::istreambuf_iterator<char>());
        if (raw.size() < 4 || raw.size() % 4) throw std::invalid_argument("docs.bin: not a list of u32");
        const uint32_t *w = (const uint32_t *)raw.data(), *end = w + raw.size() / 4;
        std::vector<rwkv::Tokens> docs(*w++);
        for (auto &d : docs) {
            if (w >= end || (size_t)(end - w - 1) < *w) throw std::invalid_argument("docs.bin: truncated");
            const uint32_t n = *w++;
            d.assign(w, w + n);
            w += n;
        }
        auto rt = rwkv::ModelBuilder(st.data(), st.size()).quant(ql, (rwkv::Quant)qt).build(B, chunk, rwkv::Precision::Fp16);
        rwkv::Scheduler<rwkv::Runtime> sched(rt);
        const size_t L = rt.state.layer_len();
        rwkv::PinnedBuffer out(docs.size() * L);
        size_t steps = 0;
        double secs = 0.0;
        for (int r = 0; r < repeat; ++r) {
            const auto t0 = std::chrono::steady_clock::now();
            steps = sched.embed_documents(docs, layer, out.data());
            secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        std::ofstream of(argv[8], std::ios::binary);
        of.write((const char *)out.data(), (std::streamsize)(out.size() * sizeof(float)));
        std::printf("steps %zu docs %zu seconds %.6f docs_per_s %.2f\n", steps, docs.size(), secs, (double)docs.size() / secs);
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "embed_job: %s\n", e.what());
        return 1;
    }
}
