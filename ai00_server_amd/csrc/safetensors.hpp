// safetensors.hpp — minimal zero-copy safetensors header parser (host only).
// Format: u64 LE header length, JSON header {name: {dtype, shape, data_offsets}}, payload.
// Mirrors what `SafeTensors::deserialize` gives ai00-core (crates/ai00-core/src/lib.rs:465, 583-591).
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace rwkv {

struct StTensor {
    std::string dtype;              // "F16", "F32", "BF16"
    std::vector<int64_t> shape;
    const uint8_t *data = nullptr;
    size_t nbytes = 0;
    int64_t numel() const {
        int64_t n = 1;
        for (auto d : shape) n *= d;
        return n;
    }
};

class SafeTensors {
   public:
    std::map<std::string, StTensor> tensors;

    static SafeTensors parse(const uint8_t *bytes, size_t len) {
        if (!bytes || len < 8) throw std::runtime_error("safetensors: buffer too small");
        uint64_t hl;
        std::memcpy(&hl, bytes, 8);
        if (hl > len - 8 || hl < 2) throw std::runtime_error("safetensors: bad header length");
        Parser p{(const char *)bytes + 8, (const char *)bytes + 8 + hl};
        SafeTensors st;
        const uint8_t *payload = bytes + 8 + hl;
        const size_t payload_len = len - 8 - hl;
        p.ws();
        p.expect('{');
        p.ws();
        if (p.peek() == '}') return st;
        for (;;) {
            p.ws();
            std::string name = p.string();
            p.ws();
            p.expect(':');
            p.ws();
            if (name == "__metadata__") {
                p.skip_value();
            } else {
                StTensor t;
                int64_t o0 = -1, o1 = -1;
                p.expect('{');
                for (;;) {
                    p.ws();
                    std::string key = p.string();
                    p.ws();
                    p.expect(':');
                    p.ws();
                    if (key == "dtype") t.dtype = p.string();
                    else if (key == "shape") t.shape = p.int_array();
                    else if (key == "data_offsets") {
                        auto v = p.int_array();
                        if (v.size() != 2) throw std::runtime_error("safetensors: bad data_offsets");
                        o0 = v[0]; o1 = v[1];
                    } else p.skip_value();
                    p.ws();
                    if (p.peek() == ',') { p.next(); continue; }
                    p.expect('}');
                    break;
                }
                if (o0 < 0 || o1 < o0 || (uint64_t)o1 > payload_len)
                    throw std::runtime_error("safetensors: tensor '" + name + "' out of bounds");
                t.data = payload + o0;
                t.nbytes = (size_t)(o1 - o0);
                size_t es = t.dtype == "F32" ? 4 : (t.dtype == "F16" || t.dtype == "BF16") ? 2 : 0;
                if (es == 0) throw std::runtime_error("safetensors: unsupported dtype " + t.dtype);
                {   // dims come from the file: non-negative, and their product must not wrap before it is compared
                    uint64_t prod = 1;
                    for (auto d : t.shape) {
                        if (d < 0 || (d != 0 && prod > (uint64_t)1 << 46) || (uint64_t)d > ((uint64_t)1 << 46))
                            throw std::runtime_error("safetensors: bad shape for '" + name + "'");
                        prod *= (uint64_t)d;
                        if (prod > ((uint64_t)1 << 46)) throw std::runtime_error("safetensors: bad shape for '" + name + "'");
                    }
                }
                if ((size_t)t.numel() * es != t.nbytes)
                    throw std::runtime_error("safetensors: size mismatch for '" + name + "'");
                st.tensors.emplace(std::move(name), std::move(t));
            }
            p.ws();
            if (p.peek() == ',') { p.next(); continue; }
            p.expect('}');
            break;
        }
        return st;
    }

    const StTensor *find(const std::string &n) const {
        auto it = tensors.find(n);
        return it == tensors.end() ? nullptr : &it->second;
    }
    const StTensor &get(const std::string &n) const {
        auto *t = find(n);
        if (!t) throw std::runtime_error("safetensors: missing tensor '" + n + "'");
        return *t;
    }

   private:
    struct Parser {
        const char *p, *e;
        char peek() const { return p < e ? *p : '\0'; }
        char next() {
            if (p >= e) throw std::runtime_error("safetensors: truncated header");
            return *p++;
        }
        void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
        void expect(char c) {
            if (next() != c) throw std::runtime_error(std::string("safetensors: expected '") + c + "'");
        }
        std::string string() {
            expect('"');
            std::string s;
            for (;;) {
                char c = next();
                if (c == '"') break;
                if (c == '\\') {
                    char d = next();
                    switch (d) {
                        case 'n': s += '\n'; break;
                        case 't': s += '\t'; break;
                        case 'r': s += '\r'; break;
                        case 'b': s += '\b'; break;
                        case 'f': s += '\f'; break;
                        case 'u': {
                            unsigned v = 0;
                            for (int i = 0; i < 4; ++i) {
                                char h = next();
                                v = v * 16 + (h >= '0' && h <= '9' ? h - '0' : (h | 32) - 'a' + 10);
                            }
                            if (v < 0x80) s += (char)v;
                            else if (v < 0x800) { s += (char)(0xC0 | (v >> 6)); s += (char)(0x80 | (v & 63)); }
                            else { s += (char)(0xE0 | (v >> 12)); s += (char)(0x80 | ((v >> 6) & 63)); s += (char)(0x80 | (v & 63)); }
                            break;
                        }
                        default: s += d;
                    }
                } else s += c;
            }
            return s;
        }
        std::vector<int64_t> int_array() {
            std::vector<int64_t> v;
            expect('[');
            ws();
            if (peek() == ']') { next(); return v; }
            for (;;) {
                ws();
                int64_t x = 0;
                bool any = false;
                while (peek() >= '0' && peek() <= '9') {
                    if (x > (INT64_MAX - 9) / 10) throw std::runtime_error("safetensors: integer out of range");
                    x = x * 10 + (next() - '0');
                    any = true;
                }
                if (!any) throw std::runtime_error("safetensors: expected integer");
                v.push_back(x);
                ws();
                if (peek() == ',') { next(); continue; }
                expect(']');
                break;
            }
            return v;
        }
        // values the loader has no use for (`__metadata__`, unknown keys).  Nesting is bounded: a header is attacker-controlled
        // bytes and every level is a stack frame.
        void skip_value(int depth = 0) {
            if (depth > 64) throw std::runtime_error("safetensors: header nested too deeply");
            ws();
            char c = peek();
            if (c == '"') { string(); return; }
            if (c == '{' || c == '[') {
                char close = c == '{' ? '}' : ']';
                next();
                ws();
                if (peek() == close) { next(); return; }
                for (;;) {
                    ws();
                    if (c == '{') { string(); ws(); expect(':'); }
                    skip_value(depth + 1);
                    ws();
                    if (peek() == ',') { next(); continue; }
                    expect(close);
                    return;
                }
            }
            while (p < e && *p != ',' && *p != '}' && *p != ']') ++p;   // number / literal
        }
    };
};

}  // namespace rwkv
