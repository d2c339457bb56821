// Mutation fuzzer for the entry points of librwkv_hip that run on a CPU and parse caller-supplied bytes:
//   rwkv_model_info_from_st   (safetensors header / prefab sniffing: `Loader::info`, lib.rs:587)
//   rwkv_tokenizer_create / _encode / _decode   (vocabulary JSON, byte strings, token ids)
//   rwkv_plan_chunk
// "Nothing aborts" is the ABI's promise (include/rwkv_abi.h): whatever the bytes, a status comes back.  Meant to be linked against a
// build of the library whose host code was compiled with -fsanitize=address,undefined (tests/test_abi_cpu.py does that when the
// toolchain allows it; scripts/README.md has the manual recipe).
// Usage: fuzz_cpu_entry_points <iterations> <vocab.json> <model.st> [<model.st> ...]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "../../include/rwkv_abi.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static std::vector<uint8_t> slurp(const char *p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

static void mutate(std::vector<uint8_t> &b, size_t window) {
    if (b.empty()) return;
    window = window < b.size() ? window : b.size();
    switch (rnd() % 6) {
        case 0:                                                                      // bit flips
            for (int k = 0, n = 1 + (int)(rnd() % 4); k < n; ++k) b[rnd() % window] ^= (uint8_t)(1u << (rnd() % 8));
            break;
        case 1:                                                                      // random bytes
            for (int k = 0, n = 1 + (int)(rnd() % 8); k < n; ++k) b[rnd() % window] = (uint8_t)rnd();
            break;
        case 2:                                                                      // truncate anywhere
            b.resize(rnd() % (b.size() + 1));
            break;
        case 3: {                                                                    // header length field
            const uint64_t v[] = {0, 1, 7, 8, 0xFFFFFFFFull, 0x7FFFFFFFFFFFFFFFull, 0xFFFFFFFFFFFFFFFFull, (uint64_t)b.size(), (uint64_t)b.size() * 2};
            const uint64_t x = v[rnd() % 9];
            if (b.size() >= 8) std::memcpy(b.data(), &x, 8);
            break;
        }
        case 4: {                                                                    // JSON-ish garbage
            const size_t at = rnd() % window, n = 1 + rnd() % 16;
            const char *digits = "0123456789-eE.[]{},:\"";
            for (size_t i = at; i < at + n && i < b.size(); ++i) b[i] = (uint8_t)digits[rnd() % 21];
            break;
        }
        case 5: {                                                                    // long numbers
            const size_t at = rnd() % window;
            b.insert(b.begin() + (long)at, (size_t)(1 + rnd() % 32), (uint8_t)'9');
            break;
        }
    }
}

// A small but complete prefab image in the documented layout (rwkv_engine.cpp, "Prefab"): header, then entries
// { head; name; pad 16; data; pad 16; scales; pad 16 } — one fp32 vector, one raw fp16 tensor, one fp16 matrix, one Int8 and one NF4 matrix.
struct PfHead { uint32_t kind; int32_t fmt, rows, K; uint32_t counted, name_len; uint64_t n_elems, data_bytes, scale_bytes; };
struct PfHdr { char magic[8]; uint32_t version, n_entries; rwkv_model_info info; int32_t quant_layers, quant_type; };
static std::vector<uint8_t> make_prefab() {
    std::vector<uint8_t> b;
    auto put = [&](const void *p, size_t n) { const uint8_t *q = (const uint8_t *)p; b.insert(b.end(), q, q + n); };
    auto pad = [&] { while (b.size() % 16) b.push_back(0); };
    PfHdr h{};
    std::memcpy(h.magic, "RWKVHIP", 8);
    h.version = 1; h.n_entries = 5;
    h.info = rwkv_model_info{6, 2, 64, 128, 256, 1, 64, 0};
    h.quant_layers = 2; h.quant_type = 1;
    put(&h, sizeof(h));
    auto entry = [&](uint32_t kind, int32_t fmt, int32_t rows, int32_t K, uint64_t n_elems, uint64_t db, uint64_t sb, const char *name) {
        PfHead e{kind, fmt, rows, K, 1u, (uint32_t)std::strlen(name), n_elems, db, sb};
        put(&e, sizeof(e));
        put(name, std::strlen(name)); pad();
        b.insert(b.end(), (size_t)db, (uint8_t)0x3c); pad();
        if (sb) { b.insert(b.end(), (size_t)sb, (uint8_t)0x01); pad(); }
    };
    entry(0, 0, 0, 0, 64, 256, 0, "blocks.0.ln1.weight");
    entry(1, 0, 0, 0, 64 * 256, 2 * 64 * 256, 0, "emb.weight");
    entry(2, 0, 64, 64, 0, 64 * 64 * 2, 0, "blocks.0.att.key.weight");
    entry(2, 1, 64, 256, 0, 64 * 256, 64 * 2 * 4, "blocks.0.ffn.key.weight");
    entry(2, 2, 64, 256, 0, 64 * 256 / 2, 64 * 4 * 2, "blocks.0.ffn.value.weight");
    return b;
}

int main(int argc, char **argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: fuzz_cpu_entry_points <iterations> <vocab.json> <model.st>...\n"); return 2; }
    const long iters = std::atol(argv[1]);
    const std::vector<uint8_t> vocab = slurp(argv[2]);
    std::vector<std::vector<uint8_t>> models;
    for (int i = 3; i < argc; ++i) models.push_back(slurp(argv[i]));
    long ok_info = 0, ok_tok = 0;
    // the unmodified inputs must be accepted
    for (auto &m : models) { rwkv_model_info mi{}; if (rwkv_model_info_from_st(m.data(), m.size(), &mi) != RWKV_OK) { std::printf("valid model rejected: %s\n", rwkv_last_error()); return 1; } }
    // prefab images: the seed must be accepted whole, mutated ones must come back with a status
    const std::vector<uint8_t> prefab = make_prefab();
    {
        rwkv_model_info mi{};
        if (rwkv_model_info_from_st(prefab.data(), prefab.size(), &mi) != RWKV_OK || mi.num_emb != 64) { std::printf("valid prefab rejected: %s\n", rwkv_last_error()); return 1; }
        long ok_pf = 0;
        for (long it = 0; it < iters; ++it) {
            std::vector<uint8_t> m = prefab;
            for (int k = 0, n = 1 + (int)(rnd() % 3); k < n; ++k) {
                mutate(m, m.size());
                if (m.size() >= 8 && rnd() % 2) std::memcpy(m.data(), "RWKVHIP", 8);      // keep most of them on the prefab path
            }
            if (rwkv_model_info_from_st(m.data(), m.size(), &mi) == RWKV_OK) ++ok_pf;
        }
        std::printf("prefab: %ld of %ld mutated images still accepted\n", ok_pf, iters);
    }
    // deeply nested JSON (every level used to be a stack frame of the header parser) and absurd chunk plans
    for (const char *open : {"[", "{\"a\":", "{\"1\":[", "\"", "{\"__metadata__\":[[", "9"}) {
        std::string s;
        for (int i = 0; i < 300000; ++i) s += open;
        rwkv_tokenizer *t = nullptr;
        if (rwkv_tokenizer_create(s.data(), s.size(), &t) == RWKV_OK && t) rwkv_tokenizer_destroy(t);
        std::vector<uint8_t> b(8 + s.size());
        const uint64_t n = s.size();
        std::memcpy(b.data(), &n, 8);
        std::memcpy(b.data() + 8, s.data(), s.size());
        rwkv_model_info mi{};
        if (rwkv_model_info_from_st(b.data(), b.size(), &mi) == RWKV_OK) { std::printf("nonsense header accepted\n"); return 1; }
    }
    {
        const size_t huge[4] = {(size_t)-1, (size_t)1 << 63, 5, ((size_t)1 << 62) + 3};
        int32_t consumed[4] = {0, 0, 0, 0};
        if (rwkv_plan_chunk(4, 2147483647, huge, consumed) != RWKV_OK) { std::printf("plan_chunk rejected a large plan\n"); return 1; }
        long total = 0;
        for (int c : consumed) { if (c < 0) { std::printf("negative share\n"); return 1; } total += c; }
        if (total != 2147483647L) { std::printf("budget not spent: %ld\n", total); return 1; }
    }
    for (long it = 0; it < iters; ++it) {
        // ---- safetensors: mutate inside the header (the first 8 + header_len bytes) or cut the file
        std::vector<uint8_t> m = models[rnd() % models.size()];
        uint64_t hl = 0;
        std::memcpy(&hl, m.data(), 8);
        if (rnd() % 4 == 0) m.resize(8 + (size_t)hl + rnd() % 64);               // keep only header (+ a little): data offsets point outside
        for (int k = 0, n = 1 + (int)(rnd() % 3); k < n; ++k) mutate(m, 8 + (size_t)hl);
        rwkv_model_info mi{};
        if (rwkv_model_info_from_st(m.data(), m.size(), &mi) == RWKV_OK) ++ok_info;
        // ---- tokenizer: the vocabulary is large, fuzz it every 16th round
        if (it % 16 == 0) {
            std::vector<uint8_t> v = vocab;
            for (int k = 0, n = 1 + (int)(rnd() % 3); k < n; ++k) mutate(v, v.size());
            rwkv_tokenizer *t = nullptr;
            if (rwkv_tokenizer_create((const char *)v.data(), v.size(), &t) == RWKV_OK && t) {
                ++ok_tok;
                std::vector<uint8_t> text(rnd() % 200);
                for (auto &c : text) c = (uint8_t)rnd();
                std::vector<uint32_t> ids(text.size() / 2 + 1);               // sometimes too small: the count may exceed cap
                (void)rwkv_tokenizer_encode(t, text.data(), text.size(), ids.data(), ids.size());
                std::vector<uint32_t> rid(rnd() % 64);
                for (auto &x : rid) x = (uint32_t)(rnd() % 70000);
                std::vector<uint8_t> outb(16);
                (void)rwkv_tokenizer_decode(t, rid.data(), rid.size(), outb.data(), outb.size());           // too small a buffer on purpose
                (void)rwkv_tokenizer_token_bytes(t, (uint32_t)(rnd() % 70000), outb.data(), rnd() % 17);
                (void)rwkv_tokenizer_vocab_size(t);
                rwkv_tokenizer_destroy(t);
            }
        }
        // ---- plan_chunk
        {
            const int B = 1 + (int)(rnd() % 40);
            std::vector<size_t> nt((size_t)B);
            for (auto &x : nt) x = rnd() % 5 == 0 ? (size_t)rnd() : (size_t)(rnd() % 3000);
            std::vector<int32_t> consumed((size_t)B);
            (void)rwkv_plan_chunk(B, (int32_t)(rnd() % 5000) - 5, nt.data(), consumed.data());
        }
    }
    std::printf("fuzz_cpu_entry_points: %ld iterations, %ld mutated models still accepted, %ld mutated vocabularies still accepted, no crash\n", iters, ok_info, ok_tok);
    return 0;
}
