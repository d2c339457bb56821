// tokenizer.cpp — RWKV "World" tokenizer: greedy longest-match over a byte trie.
// Replaces web-rwkv `Tokenizer::{new, encode, decode, token_index_to_bytes}` as used at
// crates/ai00-core/src/lib.rs:375, run.rs:157-168,856 and sampler/bnf.rs:14-27.
// Vocab JSON (assets/tokenizer/rwkv_vocab_v20230424.json, produced by convert_tokenizer.py:22-32):
//   { "<id>": "string" | [byte, byte, ...], ... }   ids 1..65529, id 0 reserved (EOS, run.rs:855).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rwkv_abi.h"

namespace {

// Byte trie in two forms.  While the vocabulary is read, nodes keep a small unsorted child list; `freeze()` then lays the
// trie out as flat arrays — per node a token id and a [begin, end) range into byte-sorted edge arrays, plus a 256-entry
// table for the root (the only wide node).  The World vocabulary (65 529 tokens, ~116 k nodes) takes ~2 MB this way; a
// node of 256 int32 children, the first version, took 120 MB per tokenizer handle.
struct Tok {
    std::vector<std::string> id2bytes;            // index = token id
    std::vector<uint8_t> present;
    struct Build { std::vector<std::pair<uint8_t, int32_t>> kids; int32_t token = -1; };
    std::vector<Build> build;                     // emptied by freeze()
    std::vector<int32_t> tok, ebeg;               // per node: token id (-1: none), first edge; ebeg has one extra entry
    std::vector<uint8_t> ebyte;                   // per edge, sorted by byte within a node
    std::vector<int32_t> enode;
    int32_t root[256];

    int new_node() { build.emplace_back(); return (int)build.size() - 1; }
    void insert(const std::string &b, int id) {
        int n = 0;
        for (unsigned char c : b) {
            int m = -1;
            for (auto &k : build[n].kids) if (k.first == c) { m = k.second; break; }
            if (m < 0) { m = new_node(); build[n].kids.emplace_back((uint8_t)c, m); }
            n = m;
        }
        build[n].token = id;
    }
    void freeze() {
        const size_t N = build.size();
        tok.resize(N); ebeg.resize(N + 1);
        size_t ne = 0;
        for (auto &b : build) ne += b.kids.size();
        ebyte.reserve(ne); enode.reserve(ne);
        for (size_t n = 0; n < N; ++n) {
            auto &k = build[n].kids;
            std::sort(k.begin(), k.end());
            tok[n] = build[n].token;
            ebeg[n] = (int32_t)ebyte.size();
            for (auto &e : k) { ebyte.push_back(e.first); enode.push_back(e.second); }
        }
        ebeg[N] = (int32_t)ebyte.size();
        std::memset(root, 0xff, sizeof(root));
        if (N) for (auto &e : build[0].kids) root[e.first] = e.second;
        std::vector<Build>().swap(build);
    }
    int child(int n, uint8_t c) const {
        if (n == 0) return root[c];
        int lo = ebeg[n], hi = ebeg[n + 1];
        while (hi - lo > 4) { const int mid = (lo + hi) >> 1; if (ebyte[mid] <= c) lo = mid; else hi = mid; }
        for (; lo < hi; ++lo) if (ebyte[lo] == c) return enode[lo];
        return -1;
    }
};

struct P {
    const char *p, *e;
    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p; }
    char peek() { return p < e ? *p : '\0'; }
    char next() { if (p >= e) throw std::runtime_error("tokenizer: truncated json"); return *p++; }
    void expect(char c) { if (next() != c) throw std::runtime_error(std::string("tokenizer: expected '") + c + "'"); }
    static void utf8(std::string &s, unsigned v) {
        if (v < 0x80) s += (char)v;
        else if (v < 0x800) { s += (char)(0xC0 | (v >> 6)); s += (char)(0x80 | (v & 63)); }
        else if (v < 0x10000) { s += (char)(0xE0 | (v >> 12)); s += (char)(0x80 | ((v >> 6) & 63)); s += (char)(0x80 | (v & 63)); }
        else { s += (char)(0xF0 | (v >> 18)); s += (char)(0x80 | ((v >> 12) & 63)); s += (char)(0x80 | ((v >> 6) & 63)); s += (char)(0x80 | (v & 63)); }
    }
    unsigned hex4() {
        unsigned v = 0;
        for (int i = 0; i < 4; ++i) {
            char h = next();
            unsigned d = h >= '0' && h <= '9' ? h - '0' : h >= 'a' && h <= 'f' ? h - 'a' + 10 : h >= 'A' && h <= 'F' ? h - 'A' + 10 : 99;
            if (d == 99) throw std::runtime_error("tokenizer: bad \\u escape");
            v = v * 16 + d;
        }
        return v;
    }
    std::string str() {
        expect('"');
        std::string s;
        for (;;) {
            char c = next();
            if (c == '"') break;
            if (c != '\\') { s += c; continue; }
            char d = next();
            switch (d) {
                case 'n': s += '\n'; break;
                case 't': s += '\t'; break;
                case 'r': s += '\r'; break;
                case 'b': s += '\b'; break;
                case 'f': s += '\f'; break;
                case 'u': {
                    unsigned v = hex4();
                    if (v >= 0xD800 && v < 0xDC00 && p + 1 < e && p[0] == '\\' && p[1] == 'u') {
                        p += 2;
                        unsigned lo = hex4();
                        v = 0x10000 + ((v - 0xD800) << 10) + (lo - 0xDC00);
                    }
                    utf8(s, v);
                    break;
                }
                default: s += d;
            }
        }
        return s;
    }
    long integer() {
        long v = 0;
        bool any = false;
        while (peek() >= '0' && peek() <= '9') {
            if (v > (1L << 40)) throw std::runtime_error("tokenizer: integer out of range");   // token ids and byte values only
            v = v * 10 + (next() - '0');
            any = true;
        }
        if (!any) throw std::runtime_error("tokenizer: expected integer");
        return v;
    }
};

}  // namespace

struct rwkv_tokenizer { Tok t; };

// failures are reported through the library's one error slot (rwkv_last_error, rwkv_engine.cpp)
extern "C" void rwkv_set_last_error(const char *msg);

extern "C" {

rwkv_status rwkv_tokenizer_create(const char *json, size_t len, rwkv_tokenizer **out) {
    if (!json || !out) return RWKV_ERR_INVALID;
    *out = nullptr;
    try {
        std::unique_ptr<rwkv_tokenizer> tk(new rwkv_tokenizer());
        Tok &t = tk->t;
        t.new_node();
        P p{json, json + len};
        p.ws(); p.expect('{');
        p.ws();
        if (p.peek() != '}') {
            for (;;) {
                p.ws();
                std::string key = p.str();
                long id = std::stol(key);
                if (id < 0 || id > (1 << 24)) throw std::runtime_error("tokenizer: bad token id");
                p.ws(); p.expect(':'); p.ws();
                std::string bytes;
                if (p.peek() == '"') bytes = p.str();
                else {
                    p.expect('[');
                    p.ws();
                    if (p.peek() != ']') {
                        for (;;) {
                            p.ws();
                            bytes += (char)(unsigned char)p.integer();
                            p.ws();
                            if (p.peek() == ',') { p.next(); continue; }
                            break;
                        }
                    }
                    p.expect(']');
                }
                if ((size_t)id >= t.id2bytes.size()) { t.id2bytes.resize(id + 1); t.present.resize(id + 1, 0); }
                t.id2bytes[id] = bytes;
                t.present[id] = 1;
                if (!bytes.empty()) t.insert(bytes, (int)id);
                p.ws();
                if (p.peek() == ',') { p.next(); continue; }
                p.expect('}');
                break;
            }
        }
        t.freeze();
        *out = tk.release();
        return RWKV_OK;
    } catch (const std::exception &e) {
        rwkv_set_last_error(e.what());
        return RWKV_ERR_FORMAT;
    }
}

void rwkv_tokenizer_destroy(rwkv_tokenizer *t) { delete t; }

int64_t rwkv_tokenizer_encode(const rwkv_tokenizer *tk, const uint8_t *text, size_t len, uint32_t *out, size_t cap) {
    if (!tk || (!text && len)) { rwkv_set_last_error("tokenizer: null argument"); return RWKV_ERR_INVALID; }
    const Tok &t = tk->t;
    size_t i = 0, n = 0;
    while (i < len) {
        int node = 0, best = -1;
        size_t best_len = 0;
        for (size_t j = i; j < len; ++j) {
            node = t.child(node, text[j]);
            if (node < 0) break;
            if (t.tok[node] >= 0) { best = t.tok[node]; best_len = j - i + 1; }
        }
        if (best < 0) {                               // TokenizerError::NoMatchingTokenFound
            rwkv_set_last_error(("tokenizer: no token matches the input at byte " + std::to_string(i)).c_str());
            return RWKV_ERR_INVALID;
        }
        if (out && n < cap) out[n] = (uint32_t)best;
        ++n;
        i += best_len;
    }
    return (int64_t)n;
}

int64_t rwkv_tokenizer_token_bytes(const rwkv_tokenizer *tk, uint32_t token, uint8_t *out, size_t cap) {
    if (!tk) return RWKV_ERR_INVALID;
    const Tok &t = tk->t;
    if (token >= t.id2bytes.size() || !t.present[token]) { rwkv_set_last_error("tokenizer: token id out of range"); return RWKV_ERR_INVALID; }
    const std::string &b = t.id2bytes[token];
    if (out) std::memcpy(out, b.data(), b.size() < cap ? b.size() : cap);
    return (int64_t)b.size();
}

int64_t rwkv_tokenizer_decode(const rwkv_tokenizer *tk, const uint32_t *tokens, size_t n, uint8_t *out, size_t cap) {
    if (!tk || (!tokens && n)) return RWKV_ERR_INVALID;
    const Tok &t = tk->t;
    size_t w = 0;
    for (size_t i = 0; i < n; ++i) {
        if (tokens[i] >= t.id2bytes.size() || !t.present[tokens[i]]) {                            // TokenizerError::OutOfRange
            rwkv_set_last_error(("tokenizer: token id " + std::to_string(tokens[i]) + " is not in the vocabulary").c_str());
            return RWKV_ERR_INVALID;
        }
        const std::string &b = t.id2bytes[tokens[i]];
        for (unsigned char c : b) {
            if (out && w < cap) out[w] = c;
            ++w;
        }
    }
    return (int64_t)w;
}

int64_t rwkv_tokenizer_vocab_size(const rwkv_tokenizer *tk) { return tk ? (int64_t)tk->t.id2bytes.size() : RWKV_ERR_INVALID; }

}  // extern "C"
