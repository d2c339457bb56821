// rwkv_engine.cpp — host side of librwkv_hip.so: loader, engine, step planner, C ABI.
//
// Replaces, behind include/rwkv_abi.h, the web-rwkv objects ai00-core holds:
//   Context + ModelBuilder + vN::Bundle + TokioRuntime<Rnn>  (crates/ai00-core/src/lib.rs:391-516)
//   State {init,load,back,read,write}                         (run.rs:477, 1099-1106)
//   softmax::softmax                                          (run.rs:1179)
// No CPU fallback: every entry point that computes needs a gfx950 device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rwkv_abi.h"
#include "rwkv_kernels.h"
#include "safetensors.hpp"

using namespace rwkv;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
struct RwkvError : std::runtime_error {
    rwkv_status code;
    RwkvError(rwkv_status c, const std::string &m) : std::runtime_error(m), code(c) {}
};
#define HIP_CHECK(x)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (x);                                                                           \
        if (_e != hipSuccess)                                                                          \
            throw RwkvError(_e == hipErrorOutOfMemory ? RWKV_ERR_OOM : RWKV_ERR_DEVICE,                \
                            std::string(#x) + ": " + hipGetErrorString(_e));                           \
    } while (0)

template <class F>
static rwkv_status guard(F &&f) {
    try {
        f();
        return RWKV_OK;
    } catch (const RwkvError &e) {
        g_err = e.what();
        return e.code;
    } catch (const std::bad_alloc &) {
        g_err = "host out of memory";
        return RWKV_ERR_OOM;
    } catch (const std::exception &e) {
        g_err = e.what();
        return std::strncmp(e.what(), "safetensors:", 12) == 0 ? RWKV_ERR_FORMAT : RWKV_ERR_INVALID;
    }
}

// ------------------------------------------------------------------------------------------------
// model info (Loader::info, lib.rs:587)
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// Prefab: the loaded (re-tiled / quantised / converted) model as one binary image, so that a later load skips the
// safetensors parse, the LoRA blend and the GPU re-tile / quantise passes.  Counterpart of `ModelSerialize::serialize`
// (lib.rs:131-154) and `LoadType::Prefab` (lib.rs:517-553, sniffed at lib.rs:585-588); the reference's CBOR image of
// web-rwkv's internal tensors is not reproducible here, this is our own versioned format:
//   header  { char magic[8] "RWKVHIP\0"; u32 version; u32 n_entries; rwkv_model_info info; i32 quant_layers, quant_type; }
//   entries { u32 kind (0 f32 vector, 1 raw f16, 2 tiled matrix); i32 fmt, rows, K; u32 counted; u32 name_len;
//             u64 n_elems, data_bytes, scale_bytes; name[name_len]; pad to 16; data; pad to 16; scales; pad to 16 }
// ------------------------------------------------------------------------------------------------
static const char kPrefabMagic[8] = {'R', 'W', 'K', 'V', 'H', 'I', 'P', 0};
constexpr uint32_t kPrefabVersion = 1;
struct PfEntryHead { uint32_t kind; int32_t fmt, rows, K; uint32_t counted, name_len; uint64_t n_elems, data_bytes, scale_bytes; };
struct PfHeader { char magic[8]; uint32_t version, n_entries; rwkv_model_info info; int32_t quant_layers, quant_type; };
struct PfEntry { PfEntryHead h; const uint8_t *data = nullptr, *scales = nullptr; };
// the dimension rules every loaded model must satisfy (kernel tiling), whether the header comes from tensor shapes or a prefab
static void validate_info(const rwkv_model_info &i) {
    if (i.version != 5 && i.version != 6 && i.version != 7) throw RwkvError(RWKV_ERR_UNSUPPORTED, "unsupported model version");
    if (i.num_layer <= 0 || i.num_layer > 4096 || i.num_emb <= 0 || i.num_hidden <= 0 || i.num_vocab <= 0 || i.num_head <= 0 ||
        i.num_vocab > (1 << 24) || i.num_hidden > (1 << 20))
        throw RwkvError(RWKV_ERR_FORMAT, "model dimensions out of range");
    if (i.head_size != 64 || (int64_t)i.num_emb != (int64_t)i.num_head * 64) throw RwkvError(RWKV_ERR_UNSUPPORTED, "head size must be 64");
    if (i.num_emb % 64 || i.num_hidden % 32 || i.num_vocab % 16)
        throw RwkvError(RWKV_ERR_UNSUPPORTED, "dims must satisfy C%64==0, F%32==0, V%16==0");
    if (i.num_emb > 8192) throw RwkvError(RWKV_ERR_UNSUPPORTED, "num_emb > 8192");
}
static bool prefab_sniff(const uint8_t *b, size_t n) { return b && n >= sizeof(PfHeader) && std::memcmp(b, kPrefabMagic, 8) == 0; }
struct Prefab {
    PfHeader hdr{};
    std::map<std::string, PfEntry> entries;
    static Prefab parse(const uint8_t *b, size_t n) {
        Prefab p;
        if (!prefab_sniff(b, n)) throw RwkvError(RWKV_ERR_FORMAT, "not a prefab image");
        std::memcpy(&p.hdr, b, sizeof(PfHeader));
        if (p.hdr.version != kPrefabVersion) throw RwkvError(RWKV_ERR_UNSUPPORTED, "prefab version mismatch");
        validate_info(p.hdr.info);
        size_t off = sizeof(PfHeader);
        // every length comes from the file: compare against what is LEFT (len > n - off), never off + len (which can wrap)
        auto take = [&](uint64_t len) -> const uint8_t * {            // `len` bytes at `off`, then `off` up to the next 16-byte FILE offset
            if (off > n || len > (uint64_t)(n - off)) throw RwkvError(RWKV_ERR_FORMAT, "prefab truncated");
            const uint8_t *q = b + off;
            const size_t end = off + (size_t)len;                     // <= n (an in-memory size): neither this nor + 15 can wrap
            off = std::min(n, (end + 15) & ~(size_t)15);              // padding cut off at the end of the file is harmless
            return q;
        };
        for (uint32_t i = 0; i < p.hdr.n_entries; ++i) {
            PfEntry e;
            const uint8_t *hp = take(sizeof(PfEntryHead));
            std::memcpy(&e.h, hp, sizeof(PfEntryHead));
            // the 16-byte padding applies after name / data / scales, not after the fixed head
            off = (size_t)(hp - b) + sizeof(PfEntryHead);
            if (e.h.name_len > 4096) throw RwkvError(RWKV_ERR_FORMAT, "prefab: entry name too long");
            std::string name((const char *)take(e.h.name_len), e.h.name_len);
            e.data = take(e.h.data_bytes);
            e.scales = e.h.scale_bytes ? take(e.h.scale_bytes) : nullptr;
            // payload sizes must be the ones save_prefab computes from the entry's own description
            const uint64_t ne = e.h.n_elems;
            if (e.h.kind == 0) { if (ne > (1ull << 40) || e.h.data_bytes != ne * 4 || e.h.scale_bytes) throw RwkvError(RWKV_ERR_FORMAT, "prefab: bad vector entry " + name); }
            else if (e.h.kind == 1) { if (ne > (1ull << 40) || e.h.data_bytes != ne * 2 || e.h.scale_bytes) throw RwkvError(RWKV_ERR_FORMAT, "prefab: bad fp16 entry " + name); }
            else if (e.h.kind == 2) {
                const int64_t rows = e.h.rows, K = e.h.K;
                bool ok = rows > 0 && K > 0 && rows % 16 == 0 && rows <= (1 << 24) && K <= (1 << 24);
                uint64_t db = 0, sb = 0;
                if (ok && e.h.fmt == W_F16) { ok = K % 32 == 0; db = (uint64_t)rows * K * 2; }
                else if (ok && e.h.fmt == W_INT8) { ok = K % 256 == 0; db = (uint64_t)rows * K; sb = (uint64_t)rows * (K / 128) * 4; }
                else if (ok && e.h.fmt == W_NF4) { ok = K % 256 == 0; db = (uint64_t)rows * K / 2; sb = (uint64_t)rows * (K / 64) * 2; }
                else ok = false;
                if (!ok || e.h.data_bytes != db || e.h.scale_bytes != sb) throw RwkvError(RWKV_ERR_FORMAT, "prefab: bad matrix entry " + name);
            } else throw RwkvError(RWKV_ERR_FORMAT, "prefab: unknown entry kind");
            p.entries.emplace(std::move(name), e);
        }
        return p;
    }
    const PfEntry &get(const std::string &name, uint32_t kind) const {
        auto it = entries.find(name);
        if (it == entries.end() || it->second.h.kind != kind) throw RwkvError(RWKV_ERR_FORMAT, "prefab: missing entry " + name);
        return it->second;
    }
    bool has(const std::string &name) const { return entries.count(name) != 0; }
};

static rwkv_model_info detect_info(const SafeTensors &st) {
    rwkv_model_info i{};
    if (st.find("blocks.0.att.x_r")) i.version = 7;
    else if (st.find("blocks.0.att.time_mix_x")) i.version = 6;
    else if (st.find("blocks.0.att.ln_x.weight") && st.find("blocks.0.att.gate.weight")) {
        const StTensor &td = st.get("blocks.0.att.time_decay");
        if (td.shape.size() < 2 || td.shape.back() <= 1)
            throw RwkvError(RWKV_ERR_UNSUPPORTED, "RWKV v5.0/v5.1 checkpoints are not supported (need v5.2)");
        i.version = 5;
    } else
        throw RwkvError(RWKV_ERR_UNSUPPORTED, "unsupported model version (v4 or unknown tensor naming)");
    int L = 0;
    while (st.find("blocks." + std::to_string(L) + ".ln1.weight")) ++L;
    const StTensor &emb = st.get("emb.weight");
    if (emb.shape.size() != 2) throw RwkvError(RWKV_ERR_FORMAT, "emb.weight must be 2-D");
    i.num_layer = L;
    i.num_vocab = (int)emb.shape[0];
    i.num_emb = (int)emb.shape[1];
    auto dim0 = [&](const char *name, size_t min_rank) -> int {
        const StTensor &t = st.get(name);
        if (t.shape.size() < min_rank || t.shape[0] <= 0 || t.shape[0] > (1 << 24)) throw RwkvError(RWKV_ERR_FORMAT, std::string(name) + ": unexpected shape");
        return (int)t.shape[0];
    };
    if (emb.shape[0] <= 0 || emb.shape[0] > (1 << 24) || emb.shape[1] <= 0 || emb.shape[1] > (1 << 20)) throw RwkvError(RWKV_ERR_FORMAT, "emb.weight: unexpected shape");
    i.num_hidden = dim0("blocks.0.ffn.key.weight", 2);
    i.num_head = i.version == 7 ? dim0("blocks.0.att.r_k", 2) : dim0("blocks.0.att.time_first", 2);
    i.head_size = i.num_emb / std::max(1, i.num_head);
    validate_info(i);
    return i;
}

// ------------------------------------------------------------------------------------------------
// engine
// ------------------------------------------------------------------------------------------------
struct Opd {                        // f16 hi/lo activation operand, B-tiled (rwkv_kernels.hip opd_off): ceil16(Tmax) x ld
    _Float16 *hi = nullptr, *lo = nullptr;
    int ld = 0;
};

struct ProbSpec {
    const DMat *W = nullptr;
    Opd x;
    int xoff = 0;                   // column offset into x (multiple of 32)
    int act = ACT_NONE, post = POST_NONE;
    const float *bias = nullptr, *m0 = nullptr, *m1 = nullptr;
    int ldm = 0;
    float *out = nullptr;
    int ldo = 0;
    bool partial = false;           // out = partial-sum buffer, K may be split across blocks
    Opd oh;                         // optional operand output
};

enum Family { FAM_GEMM = 0, FAM_HEAD = 1, FAM_ROW = 2, FAM_WKV = 3, FAM_SAMPLE = 4, FAM_COPY = 5 };
static const char *kFamilyNames[RWKV_PROFILE_FAMILIES] = {"gemm_layers", "gemm_head", "row_ln_shift", "wkv", "sample", "copy", "", ""};

struct LayerW {
    const float *ln1w, *ln1b, *ln2w, *ln2b, *lnxw, *lnxb;
    const float *mu[6];             // att mixes (v5: k,v,r,g; v6: x,w,k,v,r,g; v7: r,w,k,v,a,g)
    const float *fmu[2];            // ffn mixes (v5/v6: k,r; v7: k)
    const DMat *Wr, *Wk, *Wv, *Wg, *Wo, *Fk, *Fv, *Fr;
    // v5/v6
    const float *wdec;              // v5: exp(-exp(decay)); v6: raw time_decay
    const float *u;
    const DMat *W1;                 // v6 time_mix_w1 [5Dm x C]
    const DMat *W2[5];              // v6 time_mix_w2 [C x Dm] x5
    const DMat *D1;                 // v6 time_decay_w1 [Dd x C]
    const _Float16 *D2;             // v6 time_decay_w2 raw [C][Dd]
    // v7
    const DMat *w1, *w2, *a1, *a2, *v1, *v2, *g1, *g2;
    const float *w0, *a0, *v0, *k_k, *k_a, *r_k;
};

struct StepPlan {
    bool dense = false;             // slots 0..T-1 active with one token each, every row emitted: row index == slot index
    int T = 0, n_seq = 0, n_out = 0;
    std::vector<int> token, slot, prev, last, seq_slot, seq_begin, seq_len, out_rows;
    std::vector<int> slot_consumed, slot_out_begin, slot_out_rows;   // per slot
    uint64_t id = 0;                // plans with the same id have identical metadata apart from the token ids
};

struct rwkv_dstate {
    int device = 0;
    float *sxa = nullptr, *sxf = nullptr, *wkv = nullptr;
};

struct rwkv_engine {
    rwkv_model_info info{};
    int device = 0;
    int max_batch = 8, chunk = 128;
    bool hilo = false;
    // Operand promotion: in Precision::Fp16 the GEMM launches of the classes whose bit is set read hi + lo f16 operands like Precision::Fp32
    // does everywhere — the price of exactness is paid only where a model's error comes from (profiles/r5_fp16_error_attribution_*.jsonl:
    // V7's Wr / Wk / Wv inputs carry 4.7e-3 of its 4.9e-3 at 32 layers).  The mask is chosen from the model version (promote_for_version):
    // that IS RWKV_PRECISION_FP16 since ABI 7; RWKV_PRECISION_FP16_RAW is mask 0; RWKV_PROMOTE=<mask> is a dev override of either.
    enum OpdClass { CLS_ATT = 0, CLS_LORA2 = 1, CLS_WO = 2, CLS_FFN1 = 3, CLS_FV = 4, CLS_HEAD = 5, CLS_NONE = 31 };
    static int promote_for_version(int version) { return version == 7 ? 7 : 1; }   // V5 / V6: CLS_ATT; V7: CLS_ATT | CLS_LORA2 | CLS_WO
    int promote = 0;
    bool wide(int cls) const { return hilo || (cls < 31 && ((promote >> cls) & 1)); }
    int quant_layers = 0, quant_type = 0;
    hipStream_t s_main = nullptr, s_soft = nullptr;
    FILE *launch_log = nullptr;                                // RWKV_LAUNCH_LOG=<path> (dev): one JSON line per GEMM launch of layer 0 / the head
    int cur_layer = -1;                                        //   (grid, rows, K, stored bytes, flops): scripts/roofline_table.py joins it with rocprofv3's CSV
    hipStream_t s_copy = nullptr;                              // rwkv_state_back_layer_async: pack + device-to-host copy beside the compute stream
    hipEvent_t ev_copy_a = nullptr, ev_copy_b = nullptr;
    float *emb_stage = nullptr;                                // [max_batch][64 * C]: one packed layer slice per slot
    std::vector<void *> allocs;
    std::map<std::string, DMat> mats;
    std::map<std::string, float *> vecs;
    std::map<std::string, _Float16 *> raws;
    std::map<std::string, std::pair<size_t, bool>> vec_meta, raw_meta;   // element count, counted-in-weight_bytes (prefab save)
    void save_prefab(const char *path);
    std::vector<LayerW> layers;
    const float *ln0w, *ln0b, *lnow, *lnob;
    const _Float16 *emb = nullptr;
    const DMat *head = nullptr;
    uint64_t weight_bytes = 0;
    int Dm = 0, Dd = 0, Dl = 0;     // LoRA dims (v6: Dm, Dd; v7: max of w/a/v/g dims)

    // state (internal layout)
    float *sxa = nullptr, *sxf = nullptr, *wkv = nullptr;
    long sx_slot_stride = 0, wkv_slot_stride = 0;
    float *slab_dev = nullptr;      // staging for pack/unpack
    float *slab_host = nullptr;     // pinned

    // scratch
    float *xA = nullptr, *xB = nullptr, *P = nullptr, *xx = nullptr, *dx = nullptr;
    float *fr = nullptr, *fk = nullptr, *fv = nullptr, *fg = nullptr, *ftd = nullptr, *frr = nullptr;
    float *fw7 = nullptr, *fa7 = nullptr, *fvg7 = nullptr, *vfirst = nullptr;
    float *logits = nullptr, *logits_host = nullptr;
    float *soft_in = nullptr, *soft_out = nullptr, *soft_host = nullptr;
    size_t soft_rows_cap = 0;
    Opd opA[6], opM, opY, opK, opO, opL[4];
    long pstride = 0;
    // row meta (device + pinned host)
    int *d_meta = nullptr, *h_meta = nullptr;                  // h_meta: a ring of META_RING pinned staging buffers (state-only steps are not waited for)
    static constexpr int META_RING = 4;
    hipEvent_t meta_ev[META_RING] = {nullptr, nullptr, nullptr, nullptr};
    int meta_next = 0;
    int *h_tok = nullptr, *dv_tok = nullptr;               // pinned token ids of a dense step and their device-visible alias
    StepPlan last_plan;                                    // plan cache: a serving loop repeats the same dense decode pattern
    std::vector<size_t> last_ntok;
    std::vector<int> last_opt;
    uint64_t plan_counter = 0, uploaded_id = 0;
    size_t meta_cap = 0;
    int *d_tok_feedback = nullptr, *d_hist = nullptr, *d_amax_i = nullptr;
    // sampling front-end: per-row params, sparse adjustments (row, token, value), outputs; pinned host mirror
    int *d_adj_row = nullptr, *d_adj_tok = nullptr;
    float *d_adj_val = nullptr;
    unsigned char *d_allow = nullptr, *h_allow = nullptr;   // formatter masks of the rows that carry one: [n][V] bytes, pinned mirror
    int *d_allow_row = nullptr;
    unsigned char *h_samp = nullptr;
    static constexpr size_t ADJ_CAP = 1 << 16;
    float *d_amax_v = nullptr;
    size_t hist_cap = 0;

    // profiling
    bool profiling = false;
    float prof_ms[RWKV_PROFILE_FAMILIES] = {0};
    int prof_n[RWKV_PROFILE_FAMILIES] = {0};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    // graph cache for decode-shaped steps
    struct GraphEntry { hipGraphExec_t exec = nullptr; uint64_t used = 0; };
    uint64_t graph_clock = 0;
    std::map<uint64_t, GraphEntry> graphs;
    std::map<int, hipGraphExec_t> greedy_graphs;               // rwkv_decode_greedy: step + arg-max feedback, keyed by slot count
    std::set<uint64_t> graph_seen;
    bool use_graphs = true;
    Knobs kn;                                                    // experiment switches, frozen at creation (rwkv_kernels.h)

    template <class T>
    T *dalloc(size_t n) {
        void *p = nullptr;
        HIP_CHECK(hipMalloc(&p, std::max<size_t>(n * sizeof(T), 16)));
        allocs.push_back(p);
        return (T *)p;
    }
    Opd alloc_opd(int ld) {
        Opd o;
        ld = (ld + 31) / 32 * 32;                                      // whole k-tiles
        o.ld = ld;
        const size_t cap = (size_t)((chunk + 15) / 16 * 16) * ld;      // whole (16 token x 32 k) tiles
        o.hi = dalloc<_Float16>(cap);
        o.lo = (hilo || promote) ? dalloc<_Float16>(cap) : nullptr;
        HIP_CHECK(hipMemset(o.hi, 0, cap * 2));                        // lanes of a partly filled token tile are read (never used)
        if (o.lo) HIP_CHECK(hipMemset(o.lo, 0, cap * 2));
        return o;
    }
    ~rwkv_engine() {
        (void)hipSetDevice(device);
        if (s_main) (void)hipStreamSynchronize(s_main);
        if (s_soft) (void)hipStreamSynchronize(s_soft);
        if (s_copy) (void)hipStreamSynchronize(s_copy);
        for (auto &g : graphs) if (g.second.exec) (void)hipGraphExecDestroy(g.second.exec);
        for (auto &g : greedy_graphs) if (g.second) (void)hipGraphExecDestroy(g.second);
        for (auto ev : prof_ev) (void)hipEventDestroy(ev);
        for (void *p : allocs) (void)hipFree(p);
        if (slab_host) (void)hipHostFree(slab_host);
        if (logits_host) (void)hipHostFree(logits_host);
        if (soft_host) (void)hipHostFree(soft_host);
        if (h_meta) (void)hipHostFree(h_meta);
        for (auto ev : meta_ev) if (ev) (void)hipEventDestroy(ev);
        if (h_tok) (void)hipHostFree(h_tok);
        if (h_samp) (void)hipHostFree(h_samp);
        if (h_allow) (void)hipHostFree(h_allow);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (s_main) (void)hipStreamDestroy(s_main);
        if (s_soft) (void)hipStreamDestroy(s_soft);
        if (s_copy) (void)hipStreamDestroy(s_copy);
        if (launch_log) std::fclose(launch_log);
        if (ev_copy_a) (void)hipEventDestroy(ev_copy_a);
        if (ev_copy_b) (void)hipEventDestroy(ev_copy_b);
        if (emb_stage) (void)hipFree(emb_stage);
    }

    // ---- launch wrapper: optional per-family hipEvent timing on the compute stream.  Event pairs are recorded
    // around every launch WITHOUT host synchronisation (the stream stays busy like in the captured step); the
    // elapsed times are read back once the step has drained (prof_collect).
    std::vector<hipEvent_t> prof_ev;
    std::vector<int> prof_fam;
    template <class F>
    void launch(int fam, F &&f) {
        if (profiling) {
            const size_t i = prof_fam.size();
            if (prof_ev.size() < 2 * (i + 1)) {
                hipEvent_t a, b;
                HIP_CHECK(hipEventCreate(&a));
                HIP_CHECK(hipEventCreate(&b));
                prof_ev.push_back(a);
                prof_ev.push_back(b);
            }
            HIP_CHECK(hipEventRecord(prof_ev[2 * i], s_main));
            f();
            HIP_CHECK(hipEventRecord(prof_ev[2 * i + 1], s_main));
            prof_fam.push_back(fam);
        } else {
            f();
        }
    }
    void prof_collect() {
        HIP_CHECK(hipStreamSynchronize(s_main));
        for (size_t i = 0; i < prof_fam.size(); ++i) {
            float ms = 0;
            HIP_CHECK(hipEventElapsedTime(&ms, prof_ev[2 * i], prof_ev[2 * i + 1]));
            prof_ms[prof_fam[i]] += ms;
            prof_n[prof_fam[i]] += 1;
        }
        prof_fam.clear();
    }

    void load(const rwkv_load_desc &d);
    int gemm(std::vector<ProbSpec> &ps, int T, int fam, const ShiftCommit *commit = nullptr, int cls = CLS_NONE);
    void log_gemm(const std::vector<ProbSpec> &ps, int T, int fam, const char *kind, int variant, int grid, int ksplit, int threads);
    void log_row(const char *kernel, int T, long grid, double bytes);   // RWKV_LAUNCH_LOG: algorithmic bytes of a non-GEMM launch of layer 0 / 1
    int step_max_rows = 0;                                // most rows any sequence has in the step being enqueued (run_plan)
    float *lnp_xx_att = nullptr;                          // normalised rows published by the V6 mix's LN-prologue launch (for the commit)
    void plan_step(const rwkv_slot_input *in, StepPlan &pl);
    void upload_plan(const StepPlan &pl);
    void run_layers(int T, int n_seq, int n_out, const int *d_token, bool dense);
    void infer(const rwkv_slot_input *in, rwkv_slot_output *out);
    void run_plan(const StepPlan &pl);
    void infer_sample(const rwkv_slot_input *in, const rwkv_sample_params *sp, uint32_t *out_tokens, float *out_probs,
                      uint8_t *emitted, size_t *n_consumed);
    RowMeta meta_ptrs(int T) const;
    const int *d_seq_slot, *d_seq_begin, *d_seq_len, *d_out_rows;
};

// ------------------------------------------------------------------------------------------------
// loader
// ------------------------------------------------------------------------------------------------
// the tensor names `LoraBlend::full` matches: blocks.<digits>.<anything>
static bool lora_scope(const std::string &name) {
    if (name.compare(0, 7, "blocks.") != 0) return false;
    size_t i = 7;
    while (i < name.size() && name[i] >= '0' && name[i] <= '9') ++i;
    return i > 7 && i + 1 < name.size() && name[i] == '.';
}

static bool is_quant_target(int version, const std::string &suffix) {
    static const char *att56[] = {"att.receptance.weight", "att.key.weight", "att.value.weight", "att.output.weight", "att.gate.weight"};
    static const char *att7[] = {"att.receptance.weight", "att.key.weight", "att.value.weight", "att.output.weight"};
    static const char *ffn56[] = {"ffn.key.weight", "ffn.value.weight", "ffn.receptance.weight"};
    static const char *ffn7[] = {"ffn.key.weight", "ffn.value.weight"};
    if (version == 7) {
        for (auto s : att7) if (suffix == s) return true;
        for (auto s : ffn7) if (suffix == s) return true;
    } else {
        for (auto s : att56) if (suffix == s) return true;
        for (auto s : ffn56) if (suffix == s) return true;
    }
    return false;
}

void rwkv_engine::save_prefab(const char *path) {
    FILE *f = std::fopen(path, "wb");
    if (!f) throw RwkvError(RWKV_ERR_INVALID, std::string("cannot open ") + path);
    struct Closer { FILE *f; ~Closer() { std::fclose(f); } } closer{f};
    auto put = [&](const void *p, size_t n) { if (n && std::fwrite(p, 1, n, f) != n) throw RwkvError(RWKV_ERR_INVALID, "short write"); };
    size_t pos = 0;
    auto putp = [&](const void *p, size_t n) {                 // payload + pad to 16
        static const char zeros[16] = {0};
        put(p, n);
        pos += n;
        const size_t padn = ((pos + 15) & ~(size_t)15) - pos;
        put(zeros, padn);
        pos += padn;
    };
    PfHeader h{};
    std::memcpy(h.magic, kPrefabMagic, 8);
    h.version = kPrefabVersion;
    h.n_entries = (uint32_t)(vecs.size() + raws.size() + mats.size());
    h.info = info; h.quant_layers = quant_layers; h.quant_type = quant_type;
    put(&h, sizeof(h));
    pos = sizeof(h);
    std::vector<uint8_t> host;
    auto entry = [&](const std::string &name, uint32_t kind, int fmt, int rows, int K, bool counted, uint64_t n_elems,
                     const void *dev_data, size_t data_bytes, const void *dev_scales, size_t scale_bytes) {
        PfEntryHead e{};
        e.kind = kind; e.fmt = fmt; e.rows = rows; e.K = K; e.counted = counted ? 1 : 0; e.name_len = (uint32_t)name.size();
        e.n_elems = n_elems; e.data_bytes = data_bytes; e.scale_bytes = scale_bytes;
        put(&e, sizeof(e));
        pos += sizeof(e);
        putp(name.data(), name.size());
        host.resize(std::max(data_bytes, scale_bytes));
        HIP_CHECK(hipMemcpy(host.data(), dev_data, data_bytes, hipMemcpyDeviceToHost));
        putp(host.data(), data_bytes);
        if (scale_bytes) {
            HIP_CHECK(hipMemcpy(host.data(), dev_scales, scale_bytes, hipMemcpyDeviceToHost));
            putp(host.data(), scale_bytes);
        }
    };
    HIP_CHECK(hipSetDevice(device));
    for (auto &kv : vecs) entry(kv.first, 0, 0, 0, 0, true, vec_meta.at(kv.first).first, kv.second, vec_meta.at(kv.first).first * 4, nullptr, 0);
    for (auto &kv : raws) entry(kv.first, 1, 0, 0, 0, raw_meta.at(kv.first).second, raw_meta.at(kv.first).first, kv.second, raw_meta.at(kv.first).first * 2, nullptr, 0);
    for (auto &kv : mats) {
        const DMat &m = kv.second;
        const size_t db = m.fmt == W_F16 ? (size_t)m.rows * m.K * 2 : m.fmt == W_INT8 ? (size_t)m.rows * m.K : (size_t)m.rows * m.K / 2;
        const size_t sb = m.fmt == W_F16 ? 0 : m.fmt == W_INT8 ? (size_t)m.rows * (m.K / 128) * 4 : (size_t)m.rows * (m.K / 64) * 2;
        entry(kv.first, 2, m.fmt, m.rows, m.K, true, m.bytes, m.data, db, m.scales, sb);
    }
}

void rwkv_engine::load(const rwkv_load_desc &d) {
    if (const char *lp = std::getenv("RWKV_LAUNCH_LOG")) if (*lp) launch_log = std::fopen(lp, "a");
    const bool pf = prefab_sniff(d.st_bytes, d.st_len);      // lib.rs:585-588: the file is sniffed, not named
    Prefab prefab;
    SafeTensors st;
    if (pf) { prefab = Prefab::parse(d.st_bytes, d.st_len); info = prefab.hdr.info; }
    else { st = SafeTensors::parse(d.st_bytes, d.st_len); info = detect_info(st); }
    const int L = info.num_layer, C = info.num_emb, F = info.num_hidden, V = info.num_vocab, H = info.num_head;
    max_batch = d.max_batch > 0 ? d.max_batch : 8;
    chunk = d.token_chunk_size > 0 ? d.token_chunk_size : 128;
    kn = Knobs::from_env();                                  // frozen for the engine's lifetime (and for every graph it captures)
    use_knobs(kn);
    if (d.precision != RWKV_PRECISION_FP16 && d.precision != RWKV_PRECISION_FP32 && d.precision != RWKV_PRECISION_FP16_RAW)
        throw RwkvError(RWKV_ERR_INVALID, "precision must be RWKV_PRECISION_FP16, _FP32 or _FP16_RAW");
    hilo = d.precision == RWKV_PRECISION_FP32;
    promote = hilo ? 0 : (d.precision == RWKV_PRECISION_FP16 ? promote_for_version(info.version) : 0);
    if (!hilo && kn.promote >= 0) promote = kn.promote & 63;   // dev override (A/B runs, the error-attribution experiments)
    quant_layers = std::max(0, std::min(d.quant_layers, L));
    quant_type = d.quant_type;
    if (pf) {                                                // a prefab is already quantised / blended: its settings win
        quant_layers = prefab.hdr.quant_layers; quant_type = prefab.hdr.quant_type;
        if (d.n_lora) throw RwkvError(RWKV_ERR_UNSUPPORTED, "LoRA adapters cannot be applied to a prefab image");
    }
    if (quant_type != RWKV_QUANT_NONE && quant_type != RWKV_QUANT_INT8 && quant_type != RWKV_QUANT_NF4)
        throw RwkvError(RWKV_ERR_UNSUPPORTED, "quant_type must be None, Int8 or NF4 (SF4 is not supported)");
    if (quant_type != RWKV_QUANT_NONE && quant_layers > 0 && (C % 256 || F % 256))
        throw RwkvError(RWKV_ERR_UNSUPPORTED, "quantisation needs num_emb and num_hidden to be multiples of 256");

    HIP_CHECK(hipStreamCreateWithFlags(&s_main, hipStreamNonBlocking));
    HIP_CHECK(hipStreamCreateWithFlags(&s_soft, hipStreamNonBlocking));
    HIP_CHECK(hipEventCreate(&ev0));
    HIP_CHECK(hipEventCreate(&ev1));

    // LoRA adapters parsed once (LoraBlend::full(alpha), lib.rs:466-482): W += alpha * B * A^T for every
    // matrix that has `<name minus .weight>.lora.0` ([in, r], transposed on conversion) and `.lora.1` ([out, r]).
    std::vector<std::pair<SafeTensors, float>> loras;
    for (size_t i = 0; i < d.n_lora; ++i)
        loras.emplace_back(SafeTensors::parse(d.lora[i].st_bytes, d.lora[i].st_len), d.lora[i].alpha);

    size_t raw_cap = 0;
    if (!pf) for (auto &kv : st.tensors) raw_cap = std::max(raw_cap, kv.second.nbytes);
    _Float16 *raw = nullptr;                                   // temp upload buffer
    HIP_CHECK(hipMalloc((void **)&raw, std::max<size_t>(raw_cap, 16)));
    struct RawGuard { void *p; ~RawGuard() { (void)hipFree(p); } } rg{raw};
    _Float16 *lbuf = nullptr;
    size_t lbuf_cap = 0;
    struct LGuard { _Float16 **p; ~LGuard() { if (*p) (void)hipFree(*p); } } lg{&lbuf};

    auto upload_raw = [&](const StTensor &t) {
        if (t.dtype != "F16") throw RwkvError(RWKV_ERR_UNSUPPORTED, "tensor dtype must be F16 (convert_safetensors.py:64)");
        HIP_CHECK(hipMemcpyAsync(raw, t.data, t.nbytes, hipMemcpyHostToDevice, s_main));
    };
    // `want` = the element count the kernels will read (0: unchecked): a short tensor must fail the load, not a kernel
    auto load_vec = [&](const std::string &name, int op = 0, size_t want = 0) -> const float * {
        if (pf) {
            const PfEntry &e = prefab.get(name, 0);
            if (want && e.h.n_elems != want) throw RwkvError(RWKV_ERR_FORMAT, name + ": unexpected size");
            float *v = dalloc<float>((size_t)e.h.n_elems);
            HIP_CHECK(hipMemcpy(v, e.data, e.h.n_elems * 4, hipMemcpyHostToDevice));
            vecs[name] = v; vec_meta[name] = {(size_t)e.h.n_elems, true};
            weight_bytes += (uint64_t)e.h.n_elems * 2;
            return v;
        }
        const StTensor &t = st.get(name);
        if (want && (size_t)t.numel() != want) throw RwkvError(RWKV_ERR_FORMAT, name + ": unexpected size");
        upload_raw(t);
        float *v = dalloc<float>((size_t)t.numel());
        // `LoraBlend::full(alpha)` (lib.rs:466-482).  web-rwkv is not vendored in the reference tree; restated from its loader
        // (runtime/loader.rs, 0.10.x) and UNPINNED: `full` is the single pattern `blocks\.([0-9]+)\.([0-9a-zA-Z\.\_]+)`, so only
        // per-block tensors blend (emb, head, ln_out do not; blocks.0.ln0 does), and a VECTOR the LoRA file holds under the model's
        // own name (time_mix_*, time_decay, time_first, LayerNorm weights ...: fine-tuned whole, not low-rank) is blended
        // v = alpha * l + (1 - alpha) * v — `factor = [alpha, 1 - alpha]` of its `load_vector_*`; alpha = 1 replaces it —
        // in fp32 before the load-time transform.  Matrices: W += alpha * B A^T (`factor = [alpha, 1]`), load_mat below.
        bool blended = false;
        for (auto &lp : loras) {
            const StTensor *l = lora_scope(name) ? lp.first.find(name) : nullptr;
            if (!l) continue;
            if (l->dtype != "F16" || l->numel() != t.numel()) throw RwkvError(RWKV_ERR_FORMAT, name + ": LoRA tensor does not match the model's");
            if (!blended) { launch_f16_to_f32(raw, v, t.numel(), 0, s_main); blended = true; }
            HIP_CHECK(hipStreamSynchronize(s_main));                                // `raw` is free again
            HIP_CHECK(hipMemcpyAsync(raw, l->data, l->nbytes, hipMemcpyHostToDevice, s_main));
            launch_vec_blend(v, raw, t.numel(), lp.second, s_main);
        }
        if (blended) { if (op) launch_vec_op(v, t.numel(), op, s_main); }
        else launch_f16_to_f32(raw, v, t.numel(), op, s_main);
        HIP_CHECK(hipStreamSynchronize(s_main));
        vecs[name] = v; vec_meta[name] = {(size_t)t.numel(), true};
        weight_bytes += (uint64_t)t.numel() * 2;
        return v;
    };
    auto load_raw16 = [&](const std::string &name, bool count, size_t want = 0) -> const _Float16 * {
        if (pf) {
            const PfEntry &e = prefab.get(name, 1);
            if (want && e.h.n_elems != want) throw RwkvError(RWKV_ERR_FORMAT, name + ": unexpected size");
            _Float16 *v = dalloc<_Float16>((size_t)e.h.n_elems);
            HIP_CHECK(hipMemcpy(v, e.data, e.h.n_elems * 2, hipMemcpyHostToDevice));
            raws[name] = v; raw_meta[name] = {(size_t)e.h.n_elems, count};
            if (count) weight_bytes += (uint64_t)e.h.n_elems * 2;
            return v;
        }
        const StTensor &t = st.get(name);
        if (t.dtype != "F16") throw RwkvError(RWKV_ERR_UNSUPPORTED, "tensor dtype must be F16");
        if (want && (size_t)t.numel() != want) throw RwkvError(RWKV_ERR_FORMAT, name + ": unexpected size");
        _Float16 *v = dalloc<_Float16>((size_t)t.numel());
        HIP_CHECK(hipMemcpy(v, t.data, t.nbytes, hipMemcpyHostToDevice));
        raws[name] = v; raw_meta[name] = {(size_t)t.numel(), count};
        if (count) weight_bytes += (uint64_t)t.nbytes;
        return v;
    };
    // matrix [.., rows, K] (leading dims folded into `index`)
    // want_rows / want_K: the shape the forward pass assumes (0: taken from the tensor, e.g. LoRA ranks); rows are checked
    // after padding to whole 16-row strips
    auto load_mat = [&](const std::string &name, int fmt, int index = -1, const std::string &key = "", int want_rows = 0, int want_K = 0) -> const DMat * {
        auto check_shape = [&](int rows16, int K) {
            if ((want_rows && rows16 != (want_rows + 15) / 16 * 16) || (want_K && K != want_K))
                throw RwkvError(RWKV_ERR_FORMAT, name + ": unexpected matrix shape");
        };
        if (pf) {
            const std::string &k = key.empty() ? name : key;
            const PfEntry &e = prefab.get(k, 2);
            check_shape(e.h.rows, e.h.K);
            if (e.h.fmt != fmt) throw RwkvError(RWKV_ERR_FORMAT, name + ": stored format differs from the image header's quantisation");
            DMat m;
            m.fmt = e.h.fmt; m.rows = e.h.rows; m.K = e.h.K; m.bytes = e.h.n_elems;
            void *p = dalloc<uint8_t>((size_t)e.h.data_bytes);
            HIP_CHECK(hipMemcpy(p, e.data, e.h.data_bytes, hipMemcpyHostToDevice));
            m.data = p;
            if (e.h.scale_bytes) {
                void *sc = dalloc<uint8_t>((size_t)e.h.scale_bytes);
                HIP_CHECK(hipMemcpy(sc, e.scales, e.h.scale_bytes, hipMemcpyHostToDevice));
                m.scales = sc;
            }
            weight_bytes += m.bytes;
            return &mats.emplace(k, m).first->second;
        }
        const StTensor &t = st.get(name);
        if (t.shape.size() < 2) throw RwkvError(RWKV_ERR_FORMAT, name + ": expected a matrix");
        if (t.shape.back() <= 0 || t.shape.back() > (1 << 24) || t.shape[t.shape.size() - 2] <= 0 || t.shape[t.shape.size() - 2] > (1 << 24))
            throw RwkvError(RWKV_ERR_FORMAT, name + ": unexpected matrix shape");
        const int K = (int)t.shape.back(), rows = (int)t.shape[t.shape.size() - 2];
        check_shape((rows + 15) / 16 * 16, K);
        {
            size_t lead = 1;
            for (size_t i = 0; i + 2 < t.shape.size(); ++i) lead *= (size_t)t.shape[i];
            if (index >= 0 ? (size_t)index >= lead : lead != 1) throw RwkvError(RWKV_ERR_FORMAT, name + ": unexpected leading dimensions");
        }
        upload_raw(t);
        const _Float16 *src = raw + (index >= 0 ? (size_t)index * rows * K : 0);
        if (index < 0) {
            // LoRA blend on the raw fp16 matrix (fp32 math, one rounding back to fp16)
            const std::string stem = name.size() > 7 && name.substr(name.size() - 7) == ".weight" ? name.substr(0, name.size() - 7) : name;
            for (auto &lp : loras) {
                if (!lora_scope(name)) break;                                   // LoraBlend::full matches `blocks.N.*` only (see load_vec)
                const StTensor *A = lp.first.find(stem + ".lora.0"), *B = lp.first.find(stem + ".lora.1");
                if (!A || !B) continue;
                if (A->shape.size() != 2 || B->shape.size() != 2 || A->shape[0] != K || B->shape[0] != rows || A->shape[1] != B->shape[1])
                    throw RwkvError(RWKV_ERR_FORMAT, "LoRA shape mismatch for " + stem);
                const int r = (int)A->shape[1];
                size_t need = (size_t)(K + rows) * r * 2;
                if (need > lbuf_cap) {
                    if (lbuf) (void)hipFree(lbuf);
                    lbuf = nullptr;
                    HIP_CHECK(hipMalloc((void **)&lbuf, need));
                    lbuf_cap = need;
                }
                HIP_CHECK(hipMemcpyAsync(lbuf, A->data, A->nbytes, hipMemcpyHostToDevice, s_main));
                HIP_CHECK(hipMemcpyAsync(lbuf + (size_t)K * r, B->data, B->nbytes, hipMemcpyHostToDevice, s_main));
                launch_lora_blend(raw, lbuf + (size_t)K * r, lbuf, rows, K, r, lp.second, s_main);
            }
        }
        if (K % 32) throw RwkvError(RWKV_ERR_UNSUPPORTED, name + ": inner dim must be a multiple of 32");
        DMat m;
        m.fmt = fmt;
        m.K = K;
        m.rows = (rows + 15) / 16 * 16;
        if (fmt != W_F16 && (K % 256 || rows % 16)) throw RwkvError(RWKV_ERR_UNSUPPORTED, name + ": cannot quantise (dims)");
        if (fmt == W_F16) {
            size_t n = (size_t)m.rows * K * 2;
            void *p = dalloc<uint8_t>(n);
            launch_tile_f16(src, rows, m.rows, K, p, s_main);
            m.data = p;
            m.bytes = (uint64_t)rows * K * 2;
        } else if (fmt == W_INT8) {
            void *p = dalloc<uint8_t>((size_t)m.rows * K);
            void *sc = dalloc<uint8_t>((size_t)m.rows * (K / 128) * 4);
            launch_quant_int8(src, m.rows, K, p, sc, s_main);
            m.data = p; m.scales = sc;
            m.bytes = (uint64_t)m.rows * K + (uint64_t)m.rows * (K / 128) * 4;
        } else {
            void *p = dalloc<uint8_t>((size_t)m.rows * K / 2);
            void *sc = dalloc<uint8_t>((size_t)m.rows * (K / 64) * 2);
            launch_quant_nf4(src, m.rows, K, p, sc, s_main);
            m.data = p; m.scales = sc;
            m.bytes = (uint64_t)m.rows * K / 2 + (uint64_t)m.rows * (K / 64) * 2;
        }
        HIP_CHECK(hipStreamSynchronize(s_main));
        weight_bytes += m.bytes;
        auto it = mats.emplace(key.empty() ? name : key, m).first;
        return &it->second;
    };

    const size_t nC = (size_t)C;
    emb = load_raw16("emb.weight", false, (size_t)V * C);     // embedding table: only B rows touched per step
    ln0w = load_vec("blocks.0.ln0.weight", 0, nC);
    ln0b = load_vec("blocks.0.ln0.bias", 0, nC);
    lnow = load_vec("ln_out.weight", 0, nC);
    lnob = load_vec("ln_out.bias", 0, nC);
    head = load_mat("head.weight", W_F16, -1, "", V, C);
    if (head->rows != V) throw RwkvError(RWKV_ERR_FORMAT, "head.weight rows != vocab");

    layers.resize(L);
    for (int l = 0; l < L; ++l) {
        LayerW &w = layers[l];
        std::memset(&w, 0, sizeof(w));
        const std::string p = "blocks." + std::to_string(l) + ".";
        const int qf = (quant_type != RWKV_QUANT_NONE && l < quant_layers) ? (quant_type == RWKV_QUANT_INT8 ? W_INT8 : W_NF4) : W_F16;
        auto qfmt = [&](const char *suffix) { return is_quant_target(info.version, suffix) ? qf : W_F16; };
        // every vector is [C] (time_decay / time_first / r_k are [H, 64]); every matrix is checked against the shape the
        // forward pass assumes, so a truncated or inconsistent checkpoint fails here with RWKV_ERR_FORMAT
        w.ln1w = load_vec(p + "ln1.weight", 0, nC); w.ln1b = load_vec(p + "ln1.bias", 0, nC);
        w.ln2w = load_vec(p + "ln2.weight", 0, nC); w.ln2b = load_vec(p + "ln2.bias", 0, nC);
        w.lnxw = load_vec(p + "att.ln_x.weight", 0, nC); w.lnxb = load_vec(p + "att.ln_x.bias", 0, nC);
        w.Wr = load_mat(p + "att.receptance.weight", qfmt("att.receptance.weight"), -1, "", C, C);
        w.Wk = load_mat(p + "att.key.weight", qfmt("att.key.weight"), -1, "", C, C);
        w.Wv = load_mat(p + "att.value.weight", qfmt("att.value.weight"), -1, "", C, C);
        w.Wo = load_mat(p + "att.output.weight", qfmt("att.output.weight"), -1, "", C, C);
        w.Fk = load_mat(p + "ffn.key.weight", qfmt("ffn.key.weight"), -1, "", F, C);
        w.Fv = load_mat(p + "ffn.value.weight", qfmt("ffn.value.weight"), -1, "", C, F);
        if (info.version != 7) {
            w.Wg = load_mat(p + "att.gate.weight", qfmt("att.gate.weight"), -1, "", C, C);
            w.Fr = load_mat(p + "ffn.receptance.weight", qfmt("ffn.receptance.weight"), -1, "", C, C);
            w.fmu[0] = load_vec(p + "ffn.time_mix_k", 0, nC);
            w.fmu[1] = load_vec(p + "ffn.time_mix_r", 0, nC);
            w.u = load_vec(p + "att.time_first", 0, nC);
        }
        if (info.version == 5) {
            const char *n4[] = {"k", "v", "r", "g"};
            for (int i = 0; i < 4; ++i) w.mu[i] = load_vec(p + "att.time_mix_" + n4[i], 0, nC);
            w.wdec = load_vec(p + "att.time_decay", 1, nC);  // exp(-exp(decay)), static per channel
        } else if (info.version == 6) {
            const char *n6[] = {"x", "w", "k", "v", "r", "g"};
            for (int i = 0; i < 6; ++i) w.mu[i] = load_vec(p + "att.time_mix_" + n6[i], 0, nC);
            w.wdec = load_vec(p + "att.time_decay", 0, nC);
            int dm_l;
            if (!pf) {
                const StTensor &w2 = st.get(p + "att.time_mix_w2");
                if (w2.shape.size() != 3 || w2.shape[0] != 5 || w2.shape[1] != C || w2.shape[2] <= 0 || w2.shape[2] > 4096)
                    throw RwkvError(RWKV_ERR_FORMAT, "time_mix_w2 must be [5,C,Dm]");
                dm_l = (int)w2.shape[2];
            } else {
                dm_l = prefab.get(p + "att.time_mix_w2#0", 2).h.K;
            }
            if (l > 0 && dm_l != Dm) throw RwkvError(RWKV_ERR_FORMAT, "time_mix LoRA dim differs between layers");
            Dm = dm_l;
            w.W1 = load_mat(p + "att.time_mix_w1", W_F16, -1, "", 5 * Dm, C);
            for (int c = 0; c < 5; ++c) w.W2[c] = load_mat(p + "att.time_mix_w2", W_F16, c, p + "att.time_mix_w2#" + std::to_string(c), C, Dm);
            w.D1 = load_mat(p + "att.time_decay_w1", W_F16, -1, "", 0, C);
            if (l > 0 && w.D1->rows != Dd) throw RwkvError(RWKV_ERR_FORMAT, "time_decay LoRA dim differs between layers");
            Dd = w.D1->rows;
            if (Dd > 128 || Dd % 4) throw RwkvError(RWKV_ERR_UNSUPPORTED, "time_decay LoRA dim must be <=128");
            w.D2 = load_raw16(p + "att.time_decay_w2", true, (size_t)C * Dd);
        } else {
            const char *n7[] = {"r", "w", "k", "v", "a", "g"};
            for (int i = 0; i < 6; ++i) w.mu[i] = load_vec(p + "att.x_" + n7[i], 0, nC);
            w.fmu[0] = load_vec(p + "ffn.x_k", 0, nC);
            w.w0 = load_vec(p + "att.w0", 0, nC); w.a0 = load_vec(p + "att.a0", 0, nC);
            w.k_k = load_vec(p + "att.k_k", 0, nC); w.k_a = load_vec(p + "att.k_a", 0, nC); w.r_k = load_vec(p + "att.r_k", 0, nC);
            // LoRA pairs: stage 1 is [D, C] with a free rank D, stage 2 must be [C, D] of the same rank
            auto lora_pair = [&](const char *n1, const char *n2, const DMat *&m1, const DMat *&m2) {
                m1 = load_mat(p + n1, W_F16, -1, "", 0, C);
                m2 = load_mat(p + n2, W_F16, -1, "", C, 0);
                if (m2->K > m1->rows || m2->K % 32) throw RwkvError(RWKV_ERR_FORMAT, p + n2 + ": LoRA rank mismatch");
            };
            lora_pair("att.w1", "att.w2", w.w1, w.w2);
            lora_pair("att.a1", "att.a2", w.a1, w.a2);
            lora_pair("att.g1", "att.g2", w.g1, w.g2);
            if (l > 0 || (pf ? prefab.has(p + "att.v0") : st.find(p + "att.v0") != nullptr)) {
                w.v0 = load_vec(p + "att.v0", 0, nC);
                lora_pair("att.v1", "att.v2", w.v1, w.v2);
            }
            for (const DMat *m : {w.w1, w.a1, w.v1, w.g1}) if (m) Dl = std::max(Dl, m->rows);
        }
    }

    // ---- state + scratch
    sx_slot_stride = (long)L * C;
    wkv_slot_stride = (long)L * H * 4096;
    sxa = dalloc<float>((size_t)max_batch * sx_slot_stride);
    sxf = dalloc<float>((size_t)max_batch * sx_slot_stride);
    wkv = dalloc<float>((size_t)max_batch * wkv_slot_stride);
    HIP_CHECK(hipMemset(sxa, 0, (size_t)max_batch * sx_slot_stride * 4));
    HIP_CHECK(hipMemset(sxf, 0, (size_t)max_batch * sx_slot_stride * 4));
    HIP_CHECK(hipMemset(wkv, 0, (size_t)max_batch * wkv_slot_stride * 4));
    const size_t slab = (size_t)L * 66 * C;
    slab_dev = dalloc<float>(slab);
    HIP_CHECK(hipHostMalloc((void **)&slab_host, slab * 4, hipHostMallocDefault));

    const size_t TC = (size_t)chunk * C;
    pstride = (long)TC;
    xA = dalloc<float>(TC); xB = dalloc<float>(TC); P = dalloc<float>(TC * 8);
    xx = dalloc<float>(TC); dx = dalloc<float>(TC);
    lnp_xx_att = dalloc<float>((size_t)LNP_MAX_T * C);
    fr = dalloc<float>(TC); fk = dalloc<float>(TC); fv = dalloc<float>(TC); fg = dalloc<float>(TC); frr = dalloc<float>(TC);
    ftd = dalloc<float>((size_t)chunk * 128);
    if (info.version == 7) {
        fw7 = dalloc<float>(TC); fa7 = dalloc<float>(TC); fvg7 = dalloc<float>(TC); vfirst = dalloc<float>(TC);
    }
    for (auto &o : opA) o = alloc_opd(C);
    opY = alloc_opd(C);
    opO = alloc_opd(C);
    opK = alloc_opd(F);
    opM = alloc_opd(std::max(16, info.version == 6 ? 5 * Dm : 16));
    for (auto &o : opL) o = alloc_opd(std::max(16, Dl));
    logits = dalloc<float>((size_t)chunk * V);
    HIP_CHECK(hipHostMalloc((void **)&logits_host, (size_t)chunk * V * 4, hipHostMallocDefault));
    meta_cap = (size_t)chunk * 5 + (size_t)max_batch * 3 + 16;
    d_meta = dalloc<int>(meta_cap);
    HIP_CHECK(hipHostMalloc((void **)&h_meta, meta_cap * 4 * META_RING, hipHostMallocDefault));
    for (auto &ev : meta_ev) HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIP_CHECK(hipHostMalloc((void **)&h_tok, (size_t)max_batch * 4, hipHostMallocMapped | hipHostMallocCoherent));   // uncached on the device: always the host's latest store
    HIP_CHECK(hipHostGetDevicePointer((void **)&dv_tok, h_tok, 0));
    d_tok_feedback = dalloc<int>(chunk);
    soft_rows_cap = (size_t)std::max(1, max_batch);
    soft_in = dalloc<float>(soft_rows_cap * V);
    soft_out = dalloc<float>(soft_rows_cap * V);
    HIP_CHECK(hipHostMalloc((void **)&soft_host, soft_rows_cap * V * 4, hipHostMallocDefault));
    d_adj_row = dalloc<int>(ADJ_CAP); d_adj_tok = dalloc<int>(ADJ_CAP); d_adj_val = dalloc<float>(ADJ_CAP);
    d_allow = dalloc<unsigned char>((size_t)max_batch * V); d_allow_row = dalloc<int>(max_batch);
    HIP_CHECK(hipHostMalloc((void **)&h_allow, (size_t)max_batch * V + (size_t)max_batch * 4 + 16, hipHostMallocDefault));
    HIP_CHECK(hipHostMalloc((void **)&h_samp, (size_t)chunk * (sizeof(SampleRow) + 8) + ADJ_CAP * 12, hipHostMallocDefault));
    d_amax_v = dalloc<float>((size_t)chunk * 32);
    d_amax_i = dalloc<int>((size_t)chunk * 32);
    HIP_CHECK(hipDeviceSynchronize());
}

// ------------------------------------------------------------------------------------------------
// GEMM planning: choose the K split per problem, the X chunk and launch
// ------------------------------------------------------------------------------------------------

// Decomposition of one launch (DESIGN.md "GEMM planning"): every wave owns KW = KSW*32 k of the block's K range;
// linear ("partial") problems may split K across `ksb` blocks (the consumer row kernel sums the partials);
// a block walks `spb` strips.  Aim: >= ~1.5 blocks per CU in flight, whole matrix in flight at once.
static int plan_gemm(GemmLaunch &Lh, std::vector<ProbSpec> &ps, int T, bool hilo, long pstride, int force_spb = 0) {
    if (ps.empty() || ps.size() > GEMM_MAXP) throw RwkvError(RWKV_ERR_INVALID, "gemm: bad problem count");
    Lh = GemmLaunch{};
    Lh.nprob = (int)ps.size();
    Lh.T = T;
    int NT, KSW;
    gemm_variant(T, hilo, NT, KSW);
    const int KW = KSW * 32;
    long total_strips = 0;
    for (auto &s : ps) total_strips += s.W->rows / 16;
    int blocks = 0, np = 1, max_nw = 1, lds_items = 1;
    bool shot = true, tail = false;
    for (size_t i = 0; i < ps.size(); ++i) {
        const ProbSpec &s = ps[i];
        GemmProb &g = Lh.p[i];
        const int K = s.W->K, strips = s.W->rows / 16;
        const int align = s.W->fmt == W_F16 ? 32 : 256;
        // K split across blocks: mandatory when the range needs more than 16 waves, optional (linear epilogues)
        // to spread small matrices over more CUs
        int ksb = 1;
        auto valid = [&](int b) { return K % b == 0 && (K / b) % align == 0; };
        if (s.partial) {
            // smallest split that gives >= 1.5 blocks per CU (one strip per block), else the largest valid one <= 8
            int best = 0;
            // (round 6: capping the split at 1 or 2 — one or two partial slabs for the next row kernel to sum instead of five — costs 5 % of a 32-slot
            // step: 2.274 -> 2.395 / 2.385 ms, profiles/r6_exp_decode_ab.log)
            for (int b = 1; b <= 8; ++b) {
                if (!valid(b)) continue;
                best = b;
                if ((long)strips * b >= 384) break;
            }
            if (!best) throw RwkvError(RWKV_ERR_UNSUPPORTED, "cannot split inner dimension");
            ksb = best;
        }
        const int Kb = K / ksb;
        const int nslice = (Kb + KW - 1) / KW;                 // balanced: every wave owns the same number of slices
        const int maxw = gemm_variant_max_waves(NT, KSW, hilo), per_wave = (nslice + maxw - 1) / maxw;
        int nw = (nslice + per_wave - 1) / per_wave;
        // two-tile hi + lo launches run 512-thread blocks (8 waves): ten 256-k slices balance as five waves with two slices each, but what a CU
        // pulls from HBM grows with its waves — eight waves, two of them with a second slice, stream the first 80 % of the block's bytes at once
        // (Precision::Fp32 at 32 slots 2.587 -> 2.490 ms per step, Fp16 + RWKV_PROMOTE=1 2.307 -> 2.256)
        if (hilo && NT == 2 && per_wave > 1) nw = std::min(maxw, nslice);
        // (Round 6: the two slices beyond the eight waves dealt out as six (slice, strip) items — one strip of a slice per wave instead of two waves
        // walking a whole second slice — is SLOWER: every item pulls its slice's whole operand for a third of the work; 32-slot step 2.274 -> 2.30 ms,
        // Fp32 2.51 -> 2.59, V7 2.38 -> 2.41; profiles/r6_exp_hilo_ragged_items.log.  Not kept.)
        // strips per block: the whole grid should be resident at once (~164 VGPRs -> 12 waves per CU), and a wave's
        // rounds should fit in registers so that every load is issued up-front (single shot); the head matrix is too
        // big for that and runs 8 strips per block, software-pipelined.
        const int sub = KSW / 8, maxr = gemm_max_rounds(s.W->fmt, NT, hilo);
        const long cap = 256L * std::max(1, 8 / nw);           // measured: 5-wave blocks are resident one per CU
        int spb = (int)((total_strips * ksb + cap - 1) / cap);
        if ((total_strips * ksb + spb - 1) / spb > 1024) spb = 8;            // huge matrices (head): long pipelined blocks
        else if (spb * sub > maxr && strips * ksb <= 64) spb = std::max(1, maxr / sub);   // tiny member of a group
        spb = std::max(1, std::min(spb, 8));
        if (force_spb) spb = force_spb;
        spb = std::min(spb, std::max(1, 150 / (nw * NT)));                     // LDS: spb*nw*NT KiB <= 150 KiB
        g.W = s.W->data; g.S = s.W->scales; g.fmt = s.W->fmt; g.rows = s.W->rows; g.K = K;
        g.xhi = s.x.hi + (s.xoff >> 5) * 512; g.xlo = s.x.lo ? s.x.lo + (s.xoff >> 5) * 512 : nullptr; g.ldx = s.x.ld;   // column offset = whole k-tiles
        g.spb = spb; g.nw = nw; g.ksb = ksb;
        g.Kb = Kb; g.nslice = nslice;
        g.nblk_strip = (strips + spb - 1) / spb;
        g.block_begin = blocks;
        blocks += g.nblk_strip * ksb;
        g.act = s.act; g.post = s.post; g.bias = s.bias; g.m0 = s.m0; g.m1 = s.m1; g.ldm = s.ldm;
        g.out_f32 = s.out; g.ldo = s.ldo; g.partial_stride = pstride;
        g.out_hi = s.oh.hi; g.out_lo = s.oh.lo; g.ldh = s.oh.ld;
        max_nw = std::max(max_nw, nw);
        if (spb * sub > maxr) shot = false;
        if (Kb % 256) tail = true;
        lds_items = std::max(lds_items, spb * nw);
        if (s.partial) np = ksb;
    }
    Lh.total_blocks = blocks;
    Lh.threads = max_nw * 64;
    Lh.lds_items = lds_items;
    // 17..32-row steps over quantised weights: a ring of two rounds in flight per wave instead of every load issued up-front.  A wave that
    // has issued its 16 operand tiles and 12 weight tiles sits in the issue queue for ~3 us (profiles/r4_trace_gemm_timeline_t1_t32.log)
    // and only then starts on a strip that landed long ago; with the ring its dequantisation starts a round earlier: r/k/v/g Int8 at
    // T = 32 10.07 -> 9.74 us, Fk / Fr 10.70 -> 10.48, fp16 and T <= 16 unchanged or slower (profiles/r4_exp_gemm_ring_vs_shot.log).
    {
        bool all_quant = true;
        for (auto &sp : ps) all_quant = all_quant && (sp.W->fmt != W_F16 || sp.W->rows <= 256);   // (the decay LoRA's 64 fp16 rows ride along)
        if (NT == 2 && all_quant && !hilo) shot = false;
    }
    Lh.single_shot = shot ? 1 : 0;
    Lh.tail = tail ? 1 : 0;
    return np;
}

int rwkv_engine::gemm(std::vector<ProbSpec> &ps, int T, int fam, const ShiftCommit *commit, int cls) {
    GemmLaunch Lh;
    const bool hilo = wide(cls);                                 // shadows the engine-wide flag: this launch's operand width
    if (hilo) for (auto &sp : ps) if (!sp.x.lo) throw RwkvError(RWKV_ERR_INVALID, "gemm: a hi + lo launch needs the lo part of every operand");
    // launches whose every matrix is short in K (V7's second LoRA stage): the output-stationary small-K kernel, whatever the step's rows
    // (a launch that has to carry a token-shift commit keeps the decode kernel: the commit rides on its extra block)
    // Decode-shaped steps only: at 32 rows 6.4 -> ~5.5 us (V7-2.9B: -1.0 / -1.7 / -1.7 % per step at 32 / 8 / 1 slots); at 256 and 2048 rows
    // the tile kernels, which share X through LDS, are as fast (profiles/r5_exp_smallk_ab.log).
    if (T <= 64 && !(commit && commit->src) && !ps.empty() && ps.size() <= GEMM_MAXP) {
        Lh = GemmLaunch{};
        Lh.nprob = (int)ps.size();
        Lh.T = T;
        int items = 0;
        bool ok = true;
        for (size_t i = 0; i < ps.size() && ok; ++i) {
            const ProbSpec &sp = ps[i];
            GemmProb &g = Lh.p[i];
            ok = !sp.partial && sp.xoff == 0;
            g.W = sp.W->data; g.S = sp.W->scales; g.fmt = sp.W->fmt; g.rows = sp.W->rows; g.K = sp.W->K;
            g.xhi = sp.x.hi; g.xlo = sp.x.lo; g.ldx = sp.x.ld;
            g.spb = 1; g.nw = 1; g.ksb = 1; g.Kb = g.K; g.nslice = 1; g.nblk_strip = g.rows / 16;
            g.block_begin = items;
            items += g.rows / 16;
            g.act = sp.act; g.post = sp.post; g.bias = sp.bias; g.m0 = sp.m0; g.m1 = sp.m1; g.ldm = sp.ldm;
            g.out_f32 = sp.out; g.ldo = sp.ldo; g.partial_stride = pstride;
            g.out_hi = sp.oh.hi; g.out_lo = sp.oh.lo; g.ldh = sp.oh.ld;
        }
        Lh.total_blocks = items;
        if (ok && smallk_supported(Lh)) {
            log_gemm(ps, T, fam, "smallk", 0, (items + 3) / 4, 1, 256);
            launch(fam, [&] { launch_smallk(Lh, hilo, s_main); });
            return 1;
        }
    }
    const int no_tile = kn.no_tile;
    if (T >= GEMM_TILE_MIN_T && !no_tile) {
        // prefill: LDS-tiled MFMA GEMM, no K split (partial problems write one slab)
        if (ps.empty() || ps.size() > GEMM_MAXP) throw RwkvError(RWKV_ERR_INVALID, "gemm: bad problem count");
        Lh = GemmLaunch{};
        Lh.nprob = (int)ps.size();
        Lh.T = T;
        // 64x64 tiles measured best everywhere (tile_bench): with 256-k chunks while the launch is latency-bound
        // (few blocks: one L2 round trip per chunk dominates), with 128-k chunks (more blocks per CU) once it is
        // throughput-bound.  RWKV_TILE_SHAPE overrides (0..10) for experiments.
        const int f_shape = kn.tile_shape;                          // the parity tests force every shape (one engine per shape)
        long tot64 = 0;
        for (auto &s : ps) tot64 += gemm_tile_blocks(3, s.W->rows, T);
        int shape = tot64 <= 1536 ? 4 : 3;
        // fp16 weights fill a wave's registers twice as fast as Int8: the 128-k chunks (more blocks per CU) win at every grid size
        // since the LDS image is in fragment order (rkvg fp16 T = 512: 489 -> 534 TFLOP/s, T = 256: 374 -> 456)
        bool all_f16 = true;
        for (auto &s : ps) all_f16 = all_f16 && s.W->fmt == W_F16;
        if (all_f16) shape = 3;
        // NF4 sits in between (a quarter of the bytes per weight, the most dequantisation work): the 128-k chunks win from ~500 tiles
        // (isolated, 3 B width, 256 rows: r/k/v/g 42.0 -> 40.2 us, Fk + Fr 50.8 -> 47.4; Wo / Fv with 160 tiles lose 40 %),
        // profiles/r3_exp_tile_128x64.log
        bool all_nf4 = true;
        for (auto &s : ps) all_nf4 = all_nf4 && s.W->fmt == W_NF4;
        if (all_nf4 && tot64 >= 512) shape = 3;
        // the direct-to-LDS 128x64 shape (7: two strips per wave, X tiles by global_load_lds) pays only for very large
        // grids: 7B fp16 prefill at chunk 1024 25.9 -> 27.5 k tok/s, but 21.3 -> 17.8 k at chunk 512; the 256x128
        // GLDS shape (9) wins isolated large fp16 GEMMs (404 -> 536 TFLOP/s) and loses the model (small matrices starve)
        // (not for launches whose matrices are all short in K — V7's second-stage LoRA, K = 64..320, four [T][C] outputs: 88 us on that
        // shape at 2048 rows; on the 64x64 shapes V7-2.9B NF4 prefill 74.4 -> 76.5 k tok/s, 64.7 -> 66.5 k at 1024; profiles/r3_exp_shape7_by_k.log)
        int maxK = 0;
        for (auto &s : ps) maxK = std::max(maxK, s.W->K);
        if (T >= 1024 && tot64 >= 2500 && maxK >= 1024) shape = 7;
        // The pipelined 128x128 kernel (shape 10) keeps two blocks per CU resident, 512 tiles a round, and runs ~820 TFLOP/s on
        // whole rounds against ~540 for the 64x64 shapes whatever the grid (scripts/tile_bench2.py); a partial last round costs
        // a whole one (blocks left alone on a CU are latency-bound), so it is used when its rounds are at least 60 % full and it has
        // at least 300 tiles: V6-3B chunk 2048 60.8 -> 69.5 k prefill tok/s, V6-7B chunk 1024 30.4 -> 32.7 k.  (Round 3 moved the bar
        // from 65 % / 400 tiles: Wo of the 3 B models at 2048 rows — 320 tiles, 62.5 % of a round — is 75 us on 64x64 tiles and one
        // round of this kernel, ~64 us: 76.4 -> 78.3 k, V7-2.9B NF4 69.2 -> 71.8 k; profiles/r3_exp_tile3_thresholds.log.)
        // RWKV_TILE3_FILL=<percent> (0 = never), RWKV_TILE3_MIN_TILES.
        bool ok3 = true;
        {
            long t3 = 0;
            for (auto &sp : ps) { t3 += gemm_tile_blocks(GEMM_TILE3, sp.W->rows, T); ok3 = ok3 && gemm_tile3_supported(hilo, sp.W->K); }
            const long fill_min = 60;
            const long rounds = (t3 + 511) / 512;
            if (ok3 && t3 >= 300 && t3 * 100 >= fill_min * rounds * 512) shape = GEMM_TILE3;
        }
        // The pipelined kernel on 128 x 64 tiles (shape 11, round 4) for the NON-linear launches of steps the 128-token tile cannot fill:
        // at 256 rows a 10304-row launch is 160 tiles of 128 x 128 (fewer than CUs) but 324 of 128 x 64, each prefetching four stages
        // ahead where the 64 x 64 shapes prefetch one chunk: r/k/v/g/decay 49.7 -> 40.8 us, Fk / Fr 45.7 -> 40.0 (Int8, 256 rows).
        // Where it pays, measured after the epilogue rewrite (profiles/r4_exp_tile3_128x64.log, last section; V6-3B Int8 / fp16, V7-2.9B
        // NF4, V6-7B fp16): quantised launches at 256 rows (Int8 +5.8 %, NF4 even) and NF4 at 1024 rows (+2.7 %); fp16 at 512 rows (7 B
        // +8 %, 3 B +1 %); everywhere else the 64 x 64 shapes or the 128 x 128 tile are as fast or faster (fp16 at 1024 rows -3 %).
        // The linear launches (Wo, Fv) stay on K copies of 64 x 64 tiles (22 us at 256 rows against 30).  RWKV_TILE3_64=0 turns the rule off.
        {
            const bool linear_launch = ps.size() == 1 && ps[0].partial;
            bool big_f16 = false, big_not_nf4 = false;
            for (auto &sp : ps) {
                if (sp.W->rows <= 512) continue;                       // (the fp16 LoRA stages — V6's decay, V7's w / a / g / v: 64..320 rows — ride along)
                big_f16 = big_f16 || sp.W->fmt == W_F16;
                big_not_nf4 = big_not_nf4 || sp.W->fmt != W_NF4;
            }
            const bool in_range = big_f16 ? (T > 320 && T <= 768) : (T <= 320 || (!big_not_nf4 && T > 768 && T <= 1280));
            if (ok3 && !linear_launch && in_range) shape = GEMM_TILE3_64;
        }
        // hi + lo operands (Precision::Fp32, and the launch classes Precision::Fp16 promotes): the software-pipelined 128 x 64 kernel whenever
        // every K is a multiple of 128 — it fetches and dequantises a weight once for both operand halves: r/k/v/g Int8 46.8 us against 76.5 on
        // the 64x64 shape at 256 rows, 262 against 547 at 2048 (profiles/r6_exp_tile4.log).  (V7's second-stage LoRAs, K = 64..320, stay on 64x64.)
        bool ok4 = hilo;
        for (auto &sp : ps) ok4 = ok4 && gemm_tile4_supported(hilo, sp.W->K);
        if (ok4) shape = GEMM_TILE4_HILO;
        if (f_shape >= 0 && f_shape < GEMM_TILE_SHAPES) {
            bool okf = true;
            for (auto &sp : ps) okf = okf && gemm_tile_shape_supported(f_shape, hilo, sp.W->K);
            if (okf) shape = f_shape;
            else if (gemm_tile_pipelined(f_shape) && ok4) shape = GEMM_TILE4_HILO;      // a forced pipelined shape means "the pipelined kernel of this operand form"
            else shape = 4;                                                              // (also shape 5 with hi + lo operands: its LDS image does not fit)
        }
        const bool wide_tile = shape == GEMM_TILE3;                                      // 128 x 128 pipelined tiles
        const bool narrow_tile = shape == GEMM_TILE3_64 || shape == GEMM_TILE4_HILO;    // 128 x 64
        const bool okp = ok3 || ok4;
        // K split of a linear launch on the pipelined kernel (Wo, Fv: one `partial` problem whose output the next row kernel sums
        // anyway): a grid of fewer than 512 tiles costs a whole round of the kernel, so the tiles are replicated over `ksb` K ranges
        // until the rounds are full — 3 x 320 tiles (V6-3B at 2048 rows) fill 94 % of two rounds a third as long (tg3_body).
        int ksplit = 1;
        if (okp && kn.tile_ksplit && ps.size() == 1 && ps[0].partial && ps[0].post != POST_MIX && ps[0].act == ACT_NONE && !ps[0].bias &&
            !ps[0].oh.hi && (f_shape < 0 || gemm_tile_pipelined(f_shape))) {
            const long t3 = gemm_tile_blocks(GEMM_TILE3, ps[0].W->rows, T);
            const int G = ps[0].W->K / 128;
            double best = wide_tile ? (double)t3 / (((t3 + 511) / 512) * 512) : 0.0;
            if (t3 >= 128 && ok3) {
                // a copy must keep >= 2048 k: the pipeline's ramp and the fp32 slab a tile writes are fixed costs per copy — measured
                // (V6-3B Int8, 2048 rows): Fv (K = 8960) in three copies 200 -> 173 us, Wo (K = 2560) in three copies 67 -> 86 us
                for (int b = 2; b <= 4 && G / b >= 16; ++b) {
                    const double fill = (double)(t3 * b) / (((t3 * b + 511) / 512) * 512);
                    if (fill > best + 0.10 && fill >= 0.80) { best = fill; ksplit = b; }
                }
            }
            if (narrow_tile) {
                // copies over K until the launch has about one block per CU, a copy keeping >= 768 k
                const long t11 = gemm_tile_blocks(GEMM_TILE3_64, ps[0].W->rows, T);
                ksplit = 1;
                for (int b = 2; b <= 4 && ps[0].W->K / b >= 768 && t11 * (b - 1) < 224; ++b) ksplit = b;
            } else if (ksplit > 1) shape = GEMM_TILE3;
            else if (!wide_tile) {
                // the 64x64 shapes on a step of a few hundred rows: Wo / Fv have fewer tiles than the chip has CUs (160 at 256 rows of
                // the 3 B model); copies over K fill it
                const long t64 = gemm_tile_blocks(shape, ps[0].W->rows, T);
                const int K = ps[0].W->K;
                // (only below one tile per CU: at 320 tiles — 512 rows — the copies cost more in slabs than they fill: 49.6 -> 47.1 k tok/s)
                if (t64 < 256) for (int b = 2; b <= 4 && K / b >= 768 && t64 * (b - 1) < 448; ++b) ksplit = b;
            }
        }
        int blocks = 0;
        for (size_t i = 0; i < ps.size(); ++i) {
            const ProbSpec &s = ps[i];
            GemmProb &g = Lh.p[i];
            g.W = s.W->data; g.S = s.W->scales; g.fmt = s.W->fmt; g.rows = s.W->rows; g.K = s.W->K;
            g.xhi = s.x.hi + (s.xoff >> 5) * 512; g.xlo = s.x.lo ? s.x.lo + (s.xoff >> 5) * 512 : nullptr; g.ldx = s.x.ld;   // column offset = whole k-tiles
            g.spb = 16; g.nw = 8; g.ksb = ksplit; g.nblk_strip = 0;
            g.block_begin = blocks;
            blocks += gemm_tile_blocks(shape, s.W->rows, T) * ksplit;
            g.act = s.act; g.post = s.post; g.bias = s.bias; g.m0 = s.m0; g.m1 = s.m1; g.ldm = s.ldm;
            g.out_f32 = s.out; g.ldo = s.ldo; g.partial_stride = pstride;
            g.out_hi = s.oh.hi; g.out_lo = s.oh.lo; g.ldh = s.oh.ld;
        }
        Lh.total_blocks = blocks;
        Lh.xcd_map = 1;                                             // XCD-banded tile numbering (rwkv_kernels.hip tg_body)
        // Order of the XCD bands (round 6, profiles/r6_exp_tile5_and_xcd_order.log).  Row-tile-major (1): every weight byte is fetched by ONE XCD, which walks
        // all token tiles for it.  Token-tile-major (2): an XCD keeps its own token tiles' X rows in its L2 and streams the weights past them.  An
        // estimate by bytes says (2) from ~640 rows on; measured, the blocks of an XCD walk K in near lock-step, so X streams through the L2 once either
        // way: no change for plain operands (r/k/v/g Int8 at 2048 rows 144.5 / 145.8 us), worse with fp16 weights (149 -> 155), and -8 % only where
        // the operand is doubled — hi + lo launches of 2048-row steps (258 -> 238 us).  RWKV_TILE_XCD overrides (dev).
        if (hilo && T >= 2048 && shape == GEMM_TILE4_HILO) Lh.xcd_map = 2;
        if (const char *e = std::getenv("RWKV_TILE_XCD")) { if (*e) Lh.xcd_map = std::atoi(e); }
        {   // block size of the tile shape (rwkv_kernels.hip TG_SH; the pipelined kernel runs 256 threads): the profile joins on it
            static const int waves[10] = {8, 8, 4, 4, 4, 8, 4, 4, 4, 8};
            log_gemm(ps, T, fam, "tile", shape, blocks, ksplit, shape < 10 ? waves[shape] * 64 : 256);
        }
        launch(fam, [&] { launch_gemm_tile(Lh, shape, hilo, s_main); });
        return ksplit;
    }
    const int np = plan_gemm(Lh, ps, T, hilo, pstride);
    if (commit) Lh.commit = *commit;
    log_gemm(ps, T, fam, "decode", Lh.single_shot, Lh.total_blocks + (Lh.commit.src ? 1 : 0), np, Lh.threads);
    launch(fam, [&] { launch_gemm(Lh, hilo, s_main); });
    return np;
}

// RWKV_LAUNCH_LOG (dev): what a launch of layer 0 (or the head) streams and computes, so that a profile can be priced without guessing which
// grid size is which launch: {"kind","variant","T","grid","ksplit","rows","bytes","flops","mats"}
void rwkv_engine::log_row(const char *kernel, int T, long grid, double bytes) {
    if (!launch_log || cur_layer > 1) return;
    std::fprintf(launch_log, "{\"kind\": \"row\", \"kernel\": \"%s\", \"T\": %d, \"grid\": %ld, \"bytes\": %.0f}\n", kernel, T, grid, bytes);
    std::fflush(launch_log);
}
void rwkv_engine::log_gemm(const std::vector<ProbSpec> &ps, int T, int fam, const char *kind, int variant, int grid, int ksplit, int threads) {
    if (!launch_log || (cur_layer > 1 && fam != FAM_HEAD)) return;       // layers 0 and 1 (V7's layer 0 has no value-residual LoRA) + the head
    uint64_t bytes = 0;
    double flops = 0;
    long rows = 0;
    std::string names;
    for (auto &sp : ps) {
        bytes += sp.W->bytes;
        rows += sp.W->rows;
        flops += 2.0 * T * (double)sp.W->rows * sp.W->K;
        for (auto &kv : mats)
            if (&kv.second == sp.W) { names += (names.empty() ? "" : "+") + kv.first; break; }
    }
    std::fprintf(launch_log, "{\"kind\": \"%s\", \"variant\": %d, \"T\": %d, \"grid\": %d, \"threads\": %d, \"ksplit\": %d, \"rows\": %ld, \"bytes\": %llu, \"flops\": %.0f, \"mats\": \"%s\"}\n",
                 kind, variant, T, grid, threads, ksplit, rows, (unsigned long long)bytes, flops, names.c_str());
    std::fflush(launch_log);
}

// ------------------------------------------------------------------------------------------------
// step planning (RnnInput::new(batches, token_chunk_size), run.rs:1132) — which tokens ride this call
// ------------------------------------------------------------------------------------------------
// water-filling of the chunk budget across slots with pending tokens: short (decode) requests are never starved by
// a long prefill (any split is result-equivalent: slots are independent, RNN chunking is exact)
static void water_fill(long budget, const std::vector<size_t> &pending, std::vector<int> &take_out) {
    const int B = (int)pending.size();
    take_out.assign(B, 0);
    std::vector<int> act;
    for (int b = 0; b < B; ++b) if (pending[b] > 0) act.push_back(b);
    while (budget > 0 && !act.empty()) {
        const long share = std::max<long>(1, budget / (long)act.size());
        std::vector<int> next;
        for (int b : act) {
            if (budget <= 0) break;
            const long pend = (long)std::min<size_t>(pending[b], (size_t)1 << 40);   // a count, whatever the caller put there
            const long want = pend - take_out[b];
            const long take = std::min({want, share, budget});
            take_out[b] += (int)take;
            budget -= take;
            if (take_out[b] < pend) next.push_back(b);
        }
        act.swap(next);
    }
}

void rwkv_engine::plan_step(const rwkv_slot_input *in, StepPlan &pl) {
    const int B = max_batch;
    // A decode loop hands the same pattern (the same slots, one token each, every row emitted) step after step: the plan of
    // the previous call is reused with the new token ids (dense: row index == slot index).
    if (last_plan.dense && (int)last_ntok.size() == B) {
        bool same = true;
        for (int b = 0; b < B && same; ++b) same = in[b].n_tokens == last_ntok[b] && (in[b].n_tokens == 0 || in[b].option == last_opt[b]);
        if (same) {
            pl = last_plan;
            for (int r = 0; r < pl.T; ++r) pl.token[r] = (int)in[r].tokens[0];
            return;
        }
    }
    pl.slot_consumed.assign(B, 0);
    pl.slot_out_begin.assign(B, 0);
    pl.slot_out_rows.assign(B, 0);
    std::vector<size_t> pending(B);
    for (int b = 0; b < B; ++b) pending[b] = in[b].n_tokens;
    water_fill(chunk, pending, pl.slot_consumed);
    pl.token.clear(); pl.slot.clear(); pl.prev.clear(); pl.last.clear();
    pl.seq_slot.clear(); pl.seq_begin.clear(); pl.seq_len.clear(); pl.out_rows.clear();
    for (int b = 0; b < B; ++b) {
        const int n = pl.slot_consumed[b];
        if (!n) continue;
        const int begin = (int)pl.token.size();
        pl.seq_slot.push_back(b); pl.seq_begin.push_back(begin); pl.seq_len.push_back(n);
        pl.slot_out_begin[b] = (int)pl.out_rows.size();
        const bool exhausted = (size_t)n == in[b].n_tokens;
        for (int i = 0; i < n; ++i) {
            pl.token.push_back((int)in[b].tokens[i]);
            pl.slot.push_back(b);
            pl.prev.push_back(i == 0 ? -1 : begin + i - 1);
            pl.last.push_back(i == 0 ? begin + n - 1 : -1);
            if (in[b].option == RWKV_OPTION_FULL || (in[b].option == RWKV_OPTION_LAST && exhausted && i == n - 1)) pl.out_rows.push_back(begin + i);
        }
        pl.slot_out_rows[b] = (int)pl.out_rows.size() - pl.slot_out_begin[b];
    }
    pl.T = (int)pl.token.size();
    pl.n_seq = (int)pl.seq_slot.size();
    pl.n_out = (int)pl.out_rows.size();
    const int no_dense = kn.no_dense;                           // A/B switch
    pl.dense = !no_dense && pl.T > 0 && pl.n_seq == pl.T && pl.n_out == pl.T;
    for (int i = 0; i < pl.n_seq && pl.dense; ++i) pl.dense = pl.seq_slot[i] == i;
    pl.id = ++plan_counter;
    last_plan = pl;
    last_ntok.assign(B, 0);
    last_opt.assign(B, 0);
    for (int b = 0; b < B; ++b) { last_ntok[b] = in[b].n_tokens; last_opt[b] = in[b].option; }
}

// meta layout in d_meta: token[chunk] slot[chunk] prev[chunk] last[chunk] out_rows[chunk] seq_slot[B] seq_begin[B] seq_len[B]
RowMeta rwkv_engine::meta_ptrs(int) const {
    RowMeta rm;
    rm.dense = 0;
    rm.token = d_meta;
    rm.slot = d_meta + chunk;
    rm.prev = d_meta + 2 * chunk;
    rm.last = d_meta + 3 * chunk;
    return rm;
}

void rwkv_engine::upload_plan(const StepPlan &pl) {
    // a staging buffer is free again when the copy that read it has run (a step that emits nothing returns without waiting for the device,
    // so the previous copies may still be queued: four buffers, each guarded by the event recorded behind its copy)
    const int slot = meta_next;
    meta_next = (meta_next + 1) % META_RING;
    HIP_CHECK(hipEventSynchronize(meta_ev[slot]));
    int *h = h_meta + (size_t)slot * meta_cap;
    std::memcpy(h, pl.token.data(), pl.T * 4);
    std::memcpy(h + chunk, pl.slot.data(), pl.T * 4);
    std::memcpy(h + 2 * chunk, pl.prev.data(), pl.T * 4);
    std::memcpy(h + 3 * chunk, pl.last.data(), pl.T * 4);
    std::memcpy(h + 4 * chunk, pl.out_rows.data(), pl.n_out * 4);
    std::memcpy(h + 5 * chunk, pl.seq_slot.data(), pl.n_seq * 4);
    std::memcpy(h + 5 * chunk + max_batch, pl.seq_begin.data(), pl.n_seq * 4);
    std::memcpy(h + 5 * chunk + 2 * max_batch, pl.seq_len.data(), pl.n_seq * 4);
    HIP_CHECK(hipMemcpyAsync(d_meta, h, meta_cap * 4, hipMemcpyHostToDevice, s_main));
    HIP_CHECK(hipEventRecord(meta_ev[slot], s_main));
    uploaded_id = pl.id;
}

// ------------------------------------------------------------------------------------------------
// the forward pass: enqueue every kernel of one step on s_main
// ------------------------------------------------------------------------------------------------
void rwkv_engine::run_layers(int T, int n_seq, int n_out, const int *d_token, bool dense) {
    const int C = info.num_emb, V = info.num_vocab, H = info.num_head, L = info.num_layer;
    RowMeta rm = meta_ptrs(T);
    rm.token = d_token;
    rm.dense = dense ? 1 : 0;
    const int *seq_slot = d_meta + 5 * chunk, *seq_begin = seq_slot + max_batch, *seq_len = seq_begin + max_batch;
    const int *out_rows = d_meta + 4 * chunk;

    float *cur = xA, *oth = xB;
    int np = 0;
    {
        EmbedArgs e{emb, ln0w, ln0b, d_token, cur, C, V};
        launch(FAM_ROW, [&] { launch_embed(e, T, s_main); });
    }
    for (int l = 0; l < L; ++l) {
        const LayerW &w = layers[l];
        cur_layer = l;
        // ---- time mix
        LnShiftArgs a{};
        a.x_in = cur; a.x_out = oth; a.P = P; a.np = np; a.pstride = pstride;
        a.lnw = w.ln1w; a.lnb = w.ln1b;
        a.sx = sxa + (long)l * C; a.sx_slot_stride = sx_slot_stride;
        a.rm = rm; a.C = C; a.ldh = C;
        std::vector<ProbSpec> ps;
        auto prob = [&](const DMat *W, const Opd &x, int act, float *out, int ldo) {
            ProbSpec s; s.W = W; s.x = x; s.act = act; s.out = out; s.ldo = ldo; return s;
        };
        // an operand as ITS CONSUMER's class sees it: the lo part exists only for launches that read hi + lo (Precision::Fp32, or promoted)
        auto as = [&](const Opd &o, int cls) { Opd r = o; if (!wide(cls)) r.lo = nullptr; return r; };
        const Opd aA[6] = {as(opA[0], CLS_ATT), as(opA[1], CLS_ATT), as(opA[2], CLS_ATT), as(opA[3], CLS_ATT), as(opA[4], CLS_ATT), as(opA[5], CLS_ATT)};
        const Opd aL[4] = {as(opL[0], CLS_LORA2), as(opL[1], CLS_LORA2), as(opL[2], CLS_LORA2), as(opL[3], CLS_LORA2)};
        const Opd aF[2] = {as(opA[0], CLS_FFN1), as(opA[1], CLS_FFN1)};
        const Opd aY = as(opY, CLS_WO), aK = as(opK, CLS_FV), aZ = as(opA[0], CLS_NONE), aM = as(opM, CLS_NONE);
        // single-token steps: the LayerNorm + token shift rides as a prologue of the GEMM that consumes it, and the launch
        // after that commits the shift state (LnProArgs / ShiftCommit in rwkv_kernels.h)
        auto ln_pro = [&](const LnShiftArgs &r, float *xx_pub) {
            LnProArgs lp{};
            lp.x_in = r.x_in; lp.x_out = r.x_out; lp.P = r.P; lp.np = r.np; lp.pstride = r.pstride;
            lp.lnw = r.lnw; lp.lnb = r.lnb; lp.sx = r.sx; lp.sx_slot_stride = r.sx_slot_stride; lp.rm = r.rm;
            lp.mode = r.mode; lp.xx_out = xx_pub; lp.C = C;
            return lp;
        };
        auto commit_of = [&](const LnShiftArgs &r, const float *xx_pub) {
            ShiftCommit cm{};
            cm.src = xx_pub; cm.sx = r.sx; cm.sx_slot_stride = r.sx_slot_stride; cm.rm = r.rm; cm.T = T; cm.C = C;
            return cm;
        };
        // RWKV_LAUNCH_LOG: what a row-kernel launch moves (its rows and their partial slabs in, state row in and out, residual and operands out)
        auto log_ln = [&](const LnShiftArgs &r) {
            int nlo = 0;
            for (int m = 0; m < r.nmix; ++m) nlo += r.olo[m] ? 1 : 0;
            log_row("ln_shift_kernel", T, T, (double)T * C * (4.0 * (4 + r.np + (r.xx_out ? 1 : 0) + (r.dx_out ? 1 : 0)) + 2.0 * (r.nmix + nlo)) + 4.0 * C * (2 + r.nmix));
        };
        bool att_fused = false;                                    // a commit of the time-mix shift state is pending
        if (info.version == 5) {
            a.mode = 0; a.nmix = 4;
            for (int i = 0; i < 4; ++i) { a.mu[i] = w.mu[i]; a.ohi[i] = aA[i].hi; a.olo[i] = aA[i].lo; }
            ps = {prob(w.Wk, aA[0], ACT_NONE, fk, C), prob(w.Wv, aA[1], ACT_NONE, fv, C),
                  prob(w.Wr, aA[2], ACT_NONE, fr, C), prob(w.Wg, aA[3], ACT_SILU, fg, C)};
            { launch(FAM_ROW, [&] { launch_ln_shift(a, T, s_main); }); log_ln(a); }
            gemm(ps, T, FAM_GEMM, nullptr, CLS_ATT);
        } else if (info.version == 6) {
            a.mode = 1; a.nmix = 1; a.mu[0] = w.mu[0]; a.ohi[0] = aZ.hi; a.olo[0] = aZ.lo;
            a.xx_out = xx; a.dx_out = dx;
            const int no_fuse = kn.no_v6_fuse, no_ln_fuse = kn.no_ln_fuse;
            att_fused = !no_fuse && !no_ln_fuse && v6_mix_ln_supported(T, C, Dm, hilo, np);
            if (!att_fused) { launch(FAM_ROW, [&] { launch_ln_shift(a, T, s_main); }); log_ln(a); }
            if ((v6_mix_supported(T, C, Dm) || v6_mix_wide_supported(T, C, Dm)) && !no_fuse) {
                // fused: x_c = xx + dx * (mu_c + W2_c tanh(W1_c z)) in one launch
                V6MixArgs m{};
                m.W1 = w.W1->data;
                for (int c = 0; c < 5; ++c) { m.W2[c] = w.W2[c]->data; m.mu[c] = w.mu[1 + c]; m.ohi[c] = aA[1 + c].hi; m.olo[c] = aA[1 + c].lo; }
                m.zhi = aZ.hi; m.zlo = aZ.lo; m.ldz = C;
                m.xx = xx; m.dx = dx; m.ldh = C; m.T = T; m.C = C; m.Dm = Dm;
                m.mg_hi = aM.hi; m.mg_lo = aM.lo;               // scratch of the two-launch form (T x 5 Dm halves, like the unfused path's operand)
                // fp32 partials of the K-sliced first stage: the partial-slab buffer, free between the ln_shift that summed the previous layer's
                // Fv slabs and this layer's Wo launch (<= 16 slices x 5 x T x Dm floats of its 8 x T x C)
                if ((long)16 * 5 * Dm <= (long)8 * C) { m.mp = P; m.ksp_max = kn.v6_ksp_max; m.ksp_blocks = kn.v6_ksp_blocks; m.ksp_min_t = kn.v6_ksp_min_t; }
                if (att_fused) { m.lnp = ln_pro(a, lnp_xx_att); m.mu_x = w.mu[0]; }
                launch(FAM_GEMM, [&] { launch_v6_mix(m, hilo, s_main); });
                // LoRA matrices once, z / xx / dx in, five operands out (+ the LayerNorm prologue's row traffic on single-token steps)
                if (const int nsl = v6_mix_split(m, hilo)) {
                    // two launches: phase 1 reads W1 and z and leaves m (f16, or fp32 partials per K slice); the apply launch reads m, W2, xx / dx, mu
                    const double mbytes = nsl > 1 ? (double)nsl * 5 * T * Dm * 4 : (double)5 * T * Dm * 2 * (m.mg_lo ? 2 : 1);
                    log_row("v6_mix_kernel", T, nsl * 5 * (T / 32), 5.0 * Dm * C * 2 + (double)T * C * 2.0 + mbytes);
                    const long ag = (long)((C / 16 + 7) / 8) * (T / 32);                // (launch_v6_mix: 16-token tiles below one block per CU)
                    log_row("v6_mix_apply_kernel", T, ag < 256 ? (long)((C / 16 + 7) / 8) * ((T + 15) / 16) : ag, 5.0 * Dm * C * 2 + mbytes + (double)T * C * (8.0 + 10.0 * (m.olo[0] ? 2 : 1)) + 20.0 * C);
                } else
                log_row("v6_mix_kernel", T, 0, 2.0 * (5.0 * Dm * C * 2) + (double)T * C * (2.0 + 8.0 + 10.0 * (m.olo[0] ? 2 : 1)) + 20.0 * C +
                                                   (att_fused ? (double)T * C * 4.0 * (4 + np) : 0.0));
            } else {
                {   // m = tanh(W1 z)  ->  operand [T][5*Dm]
                    ProbSpec s = prob(w.W1, aZ, ACT_TANH, nullptr, 0);
                    s.oh = aM;
                    ps = {s};
                    gemm(ps, T, FAM_GEMM);
                }
                ps.clear();
                for (int c = 0; c < 5; ++c) {   // x_c = xx + dx * (mu_c + W2_c m_c),  c in (w,k,v,r,g)
                    ProbSpec s = prob(w.W2[c], aM, ACT_NONE, nullptr, 0);
                    s.xoff = c * Dm;
                    s.bias = w.mu[1 + c]; s.post = POST_MIX; s.m0 = xx; s.m1 = dx; s.ldm = C;
                    s.oh = aA[1 + c];
                    ps.push_back(s);
                }
                gemm(ps, T, FAM_GEMM);
            }
            ps = {prob(w.Wk, aA[2], ACT_NONE, fk, C), prob(w.Wv, aA[3], ACT_NONE, fv, C),
                  prob(w.Wr, aA[4], ACT_NONE, fr, C), prob(w.Wg, aA[5], ACT_SILU, fg, C),
                  prob(w.D1, aA[1], ACT_TANH, ftd, Dd)};
            if (att_fused) { ShiftCommit cm = commit_of(a, lnp_xx_att); gemm(ps, T, FAM_GEMM, &cm, CLS_ATT); att_fused = false; }
            else gemm(ps, T, FAM_GEMM, nullptr, CLS_ATT);
        } else {
            a.mode = 1; a.nmix = 6;
            for (int i = 0; i < 6; ++i) { a.mu[i] = w.mu[i]; a.ohi[i] = aA[i].hi; a.olo[i] = aA[i].lo; }
            // opA: 0=r 1=w 2=k 3=v 4=a 5=g
            ps = {prob(w.Wr, aA[0], ACT_NONE, fr, C), prob(w.Wk, aA[2], ACT_NONE, fk, C), prob(w.Wv, aA[3], ACT_NONE, fv, C)};
            { ProbSpec s = prob(w.w1, aA[1], ACT_TANH, nullptr, 0); s.oh = aL[0]; ps.push_back(s); }
            { ProbSpec s = prob(w.a1, aA[4], ACT_NONE, nullptr, 0); s.oh = aL[1]; ps.push_back(s); }
            { ProbSpec s = prob(w.g1, aA[5], ACT_SIGMOID, nullptr, 0); s.oh = aL[2]; ps.push_back(s); }
            if (l > 0) { ProbSpec s = prob(w.v1, aA[3], ACT_NONE, nullptr, 0); s.oh = aL[3]; ps.push_back(s); }
            { launch(FAM_ROW, [&] { launch_ln_shift(a, T, s_main); }); log_ln(a); }
            gemm(ps, T, FAM_GEMM, nullptr, CLS_ATT);
            ps.clear();
            { ProbSpec s = prob(w.w2, aL[0], ACT_DECAY7, fw7, C); s.bias = w.w0; ps.push_back(s); }
            { ProbSpec s = prob(w.a2, aL[1], ACT_SIGMOID, fa7, C); s.bias = w.a0; ps.push_back(s); }
            { ProbSpec s = prob(w.g2, aL[2], ACT_NONE, fg, C); ps.push_back(s); }
            if (l > 0) { ProbSpec s = prob(w.v2, aL[3], ACT_SIGMOID, fvg7, C); s.bias = w.v0; ps.push_back(s); }
            gemm(ps, T, FAM_GEMM, nullptr, CLS_LORA2);
        }
        std::swap(cur, oth);
        {
            WkvArgs k{};
            k.version = info.version; k.H = H; k.C = C; k.n_seq = n_seq;
            k.seq_slot = seq_slot; k.seq_begin = seq_begin; k.seq_len = seq_len; k.dense = dense ? 1 : 0;
            k.state = wkv + (long)l * H * 4096; k.slot_stride = wkv_slot_stride;
            k.r = fr; k.k = fk; k.v = fv; k.g = fg;
            k.wdec_or_decay = w.wdec; k.u = w.u; k.td = ftd; k.D2 = w.D2; k.Dd = Dd;
            k.w7 = fw7; k.a7 = fa7; k.vg7 = fvg7; k.k_k = w.k_k; k.k_a = w.k_a; k.r_k = w.r_k;
            k.v_first = vfirst; k.layer = l;
            k.lnx_w = w.lnxw; k.lnx_b = w.lnxb;
            k.yhi = aY.hi; k.ylo = aY.lo; k.ldh = C;
            k.max_rows = step_max_rows;
            launch(FAM_WKV, [&] { launch_wkv(k, T > n_seq, s_main); });
            // state tiles in and out (2 x 16 KiB per (slot, head)), the projections of every row in, the gated output operand out
            log_row(T > n_seq ? "wkv_chunk_kernel" : "wkv_kernel", T, (long)n_seq * H,
                    (double)n_seq * H * 32768.0 + (double)T * C * 4.0 * (info.version == 7 ? (l > 0 ? 8 : 6) : 4) +
                        (info.version == 6 ? (double)T * Dd * 4.0 + (double)C * Dd * 2.0 : 0.0) + (double)T * C * 2.0 * (k.ylo ? 2 : 1));
        }
        {
            ProbSpec s = prob(w.Wo, aY, ACT_NONE, P, C);
            s.partial = true;
            ps = {s};
            np = gemm(ps, T, FAM_GEMM, nullptr, CLS_WO);
        }
        // ---- channel mix
        LnShiftArgs f{};
        f.x_in = cur; f.x_out = oth; f.P = P; f.np = np; f.pstride = pstride;
        f.lnw = w.ln2w; f.lnb = w.ln2b;
        f.sx = sxf + (long)l * C; f.sx_slot_stride = sx_slot_stride;
        f.rm = rm; f.C = C; f.ldh = C;
        f.mode = info.version == 5 ? 0 : 1;
        f.nmix = info.version == 7 ? 1 : 2;
        for (int i = 0; i < f.nmix; ++i) { f.mu[i] = w.fmu[i]; f.ohi[i] = aF[i].hi; f.olo[i] = aF[i].lo; }
        {
            ProbSpec s = prob(w.Fk, aF[0], ACT_RELU2, nullptr, 0);
            s.oh = aK;
            ps = {s};
            if (info.version != 7) { ProbSpec r = prob(w.Fr, aF[1], ACT_SIGMOID, frr, C); ps.push_back(r); }
            { launch(FAM_ROW, [&] { launch_ln_shift(f, T, s_main); }); log_ln(f); }
            gemm(ps, T, FAM_GEMM, nullptr, CLS_FFN1);
            std::swap(cur, oth);
        }
        {
            ProbSpec s = prob(w.Fv, aK, ACT_NONE, P, C);
            s.partial = true;
            if (info.version != 7) { s.post = POST_MUL; s.m0 = frr; s.ldm = C; }
            ps = {s};
            np = gemm(ps, T, FAM_GEMM, nullptr, CLS_FV);
        }
    }
    if (n_out > 0) {
        Opd aO = opO;
        if (!wide(CLS_HEAD)) aO.lo = nullptr;
        LnOutArgs o{cur, P, np, pstride, lnow, lnob, dense ? nullptr : out_rows, aO.hi, aO.lo, C, C};
        launch(FAM_ROW, [&] { launch_ln_out(o, n_out, s_main); });
        std::vector<ProbSpec> ps(1);
        ps[0].W = head; ps[0].x = aO; ps[0].out = logits; ps[0].ldo = V;
        gemm(ps, n_out, FAM_HEAD, nullptr, CLS_HEAD);
    } else if (np > 0) {
        // nothing consumes the pending partial sums: fine, the residual stream dies with the step
    }
}

// upload the row metadata and enqueue the step (graph replay when this shape was seen before)
void rwkv_engine::run_plan(const StepPlan &pl) {
    // Dense steps read their token ids from pinned host memory through its device-visible alias (one uncached read per row in
    // the embedding kernel) and nothing else changes between two steps of the same plan, so a repeated dense plan needs no
    // host-to-device copy at all; everything else uploads its metadata.
    const int *tok_ptr = d_meta;
    if (pl.dense) {
        std::memcpy(h_tok, pl.token.data(), (size_t)pl.T * 4);
        tok_ptr = dv_tok;
        if (pl.id != uploaded_id) upload_plan(pl);
    } else {
        upload_plan(pl);
    }
    step_max_rows = 0;
    for (int n : pl.seq_len) step_max_rows = std::max(step_max_rows, n);
    const bool short_rows = step_max_rows <= 8;                  // picks the WKV form: part of the graph's identity
    const uint64_t key = ((uint64_t)pl.dense << 63) | ((uint64_t)short_rows << 62) | ((uint64_t)pl.T << 40) | ((uint64_t)pl.n_seq << 20) | (uint64_t)pl.n_out;
    if (use_graphs && !profiling) {
        auto it = graphs.find(key);
        // a shape is captured the SECOND time it shows up: decode-shaped steps repeat at once, while the one-off shapes of
        // prefill tails would pay capture + instantiation (milliseconds) for a single replay and churn the cache
        if (it == graphs.end() && graph_seen.insert(key).second) {
            if (graph_seen.size() > 4096) graph_seen.clear();
            run_layers(pl.T, pl.n_seq, pl.n_out, tok_ptr, pl.dense);
            return;
        }
        if (it == graphs.end()) {
            hipGraph_t g = nullptr;
            HIP_CHECK(hipStreamBeginCapture(s_main, hipStreamCaptureModeThreadLocal));
            try {
                run_layers(pl.T, pl.n_seq, pl.n_out, tok_ptr, pl.dense);
            } catch (...) {
                (void)hipStreamEndCapture(s_main, &g);
                if (g) (void)hipGraphDestroy(g);
                throw;
            }
            HIP_CHECK(hipStreamEndCapture(s_main, &g));
            GraphEntry ge;
            HIP_CHECK(hipGraphInstantiate(&ge.exec, g, nullptr, nullptr, 0));
            HIP_CHECK(hipGraphDestroy(g));
            if (graphs.size() >= 64) {                             // full: the least recently replayed shape goes, the rest stay warm
                auto victim = graphs.begin();
                for (auto o = graphs.begin(); o != graphs.end(); ++o) if (o->second.used < victim->second.used) victim = o;
                (void)hipGraphExecDestroy(victim->second.exec);
                graphs.erase(victim);
            }
            it = graphs.emplace(key, ge).first;
        }
        it->second.used = ++graph_clock;
        HIP_CHECK(hipGraphLaunch(it->second.exec, s_main));
    } else {
        run_layers(pl.T, pl.n_seq, pl.n_out, tok_ptr, pl.dense);
    }
}

void rwkv_engine::infer_sample(const rwkv_slot_input *in, const rwkv_sample_params *sp, uint32_t *out_tokens, float *out_probs,
                               uint8_t *emitted, size_t *n_consumed) {
    HIP_CHECK(hipSetDevice(device));
    if (info.num_vocab > 65536) throw RwkvError(RWKV_ERR_UNSUPPORTED, "on-device sampling needs num_vocab <= 65536");
    std::vector<rwkv_slot_input> last(in, in + max_batch);
    for (int b = 0; b < max_batch; ++b) {
        last[b].option = RWKV_OPTION_LAST;
        if (emitted) emitted[b] = 0;
        if (n_consumed) n_consumed[b] = 0;
        if (in[b].n_tokens && !in[b].tokens) throw RwkvError(RWKV_ERR_INVALID, "slot has n_tokens>0 but tokens==NULL");
    }
    StepPlan pl;
    plan_step(last.data(), pl);
    if (pl.T == 0) return;
    // pack per-row sampler params + sparse adjustments (rows are in out_rows order == ascending slot order)
    SampleRow *hs = (SampleRow *)h_samp;
    int *h_row = (int *)(h_samp + (size_t)chunk * sizeof(SampleRow));
    int *h_tok = h_row + ADJ_CAP;
    float *h_val = (float *)(h_tok + ADJ_CAP);
    size_t nadj = 0;
    bool any_nt = false, any_miro = false;
    int n_allow = 0;
    int *h_allow_row = (int *)(h_allow + (size_t)max_batch * info.num_vocab);
    for (int b = 0; b < max_batch; ++b) {
        if (pl.slot_out_rows[b] == 0) continue;
        const int r = pl.slot_out_begin[b];
        const rwkv_sample_params &p = sp[b];
        if (p.kind != RWKV_SAMPLER_MIROSTAT && p.top_k > 256) throw RwkvError(RWKV_ERR_UNSUPPORTED, "on-device sampling supports top_k <= 256");
        if (p.kind != RWKV_SAMPLER_MIROSTAT && !(p.temperature > 0.f)) throw RwkvError(RWKV_ERR_INVALID, "temperature must be > 0");
        if (p.n_adj && (!p.adj_tokens || !p.adj_values)) throw RwkvError(RWKV_ERR_INVALID, "null adjustment arrays");
        if (nadj + p.n_adj > ADJ_CAP) throw RwkvError(RWKV_ERR_INVALID, "too many logit adjustments");
        if (p.kind < RWKV_SAMPLER_NUCLEUS || p.kind > RWKV_SAMPLER_MIROSTAT) throw RwkvError(RWKV_ERR_UNSUPPORTED, "unknown sampler kind");
        if (p.kind == RWKV_SAMPLER_MIROSTAT) any_miro = true; else any_nt = true;
        hs[r] = SampleRow{p.top_p, p.top_k, p.temperature, p.uniform, p.kind, p.tau};
        if (p.allow) {                                             // formatter mask: staged row by row, one H2D copy for all
            std::memcpy(h_allow + (size_t)n_allow * info.num_vocab, p.allow, (size_t)info.num_vocab);
            h_allow_row[n_allow++] = r;
        }
        for (size_t i = 0; i < p.n_adj; ++i, ++nadj) { h_row[nadj] = r; h_tok[nadj] = (int)p.adj_tokens[i]; h_val[nadj] = p.adj_values[i]; }
    }
    // The sampler kernel reads its per-row parameters from, and writes its 8 bytes per row to, pinned host memory directly
    // (device-visible mapping of h_samp): no copy operation sits in front of the step or between it and the host seeing its
    // tokens.  (Measured against an H2D copy of the parameters ahead of the step: no difference beyond run-to-run noise.)
    const SampleRow *rows_ptr = nullptr;
    HIP_CHECK(hipHostGetDevicePointer((void **)&rows_ptr, hs, 0));
    run_plan(pl);
    if (pl.n_out > 0) {
        int *ht = (int *)(h_samp + (size_t)chunk * sizeof(SampleRow) + (size_t)ADJ_CAP * 12);
        float *hp = (float *)(ht + chunk);
        int *dv_out_tok = nullptr;
        HIP_CHECK(hipHostGetDevicePointer((void **)&dv_out_tok, ht, 0));
        float *dv_prob = (float *)(dv_out_tok + chunk);
        if (nadj) {
            HIP_CHECK(hipMemcpyAsync(d_adj_row, h_row, nadj * 4, hipMemcpyHostToDevice, s_main));
            HIP_CHECK(hipMemcpyAsync(d_adj_tok, h_tok, nadj * 4, hipMemcpyHostToDevice, s_main));
            HIP_CHECK(hipMemcpyAsync(d_adj_val, h_val, nadj * 4, hipMemcpyHostToDevice, s_main));
            launch_logit_adjust(logits, info.num_vocab, d_adj_row, d_adj_tok, d_adj_val, (int)nadj, s_main);
        }
        if (n_allow) {                                             // after the adjustments: -inf + bias stays -inf (run.rs:676-683)
            HIP_CHECK(hipMemcpyAsync(d_allow, h_allow, (size_t)n_allow * info.num_vocab, hipMemcpyHostToDevice, s_main));
            HIP_CHECK(hipMemcpyAsync(d_allow_row, h_allow_row, (size_t)n_allow * 4, hipMemcpyHostToDevice, s_main));
            launch_logit_mask(logits, info.num_vocab, d_allow_row, d_allow, n_allow, s_main);
        }
        launch_nucleus(logits, pl.n_out, info.num_vocab, rows_ptr, any_nt, any_miro, dv_out_tok, dv_prob, s_main);
        HIP_CHECK(hipStreamSynchronize(s_main));
        for (int b = 0; b < max_batch; ++b) {
            if (pl.slot_out_rows[b] == 0) continue;
            const int r = pl.slot_out_begin[b];
            if (out_tokens) out_tokens[b] = (uint32_t)ht[r];
            if (out_probs) out_probs[b] = hp[r];
            if (emitted) emitted[b] = 1;
        }
    } else {
        HIP_CHECK(hipStreamSynchronize(s_main));
    }
    if (n_consumed) for (int b = 0; b < max_batch; ++b) n_consumed[b] = (size_t)pl.slot_consumed[b];
}

void rwkv_engine::infer(const rwkv_slot_input *in, rwkv_slot_output *out) {
    HIP_CHECK(hipSetDevice(device));
    (void)hipGetLastError();                                   // start clean: the poll behind a row-less step must only see THIS call's errors
                                                               // (a handled, non-sticky failure of an earlier call — a probe — stays recorded otherwise)
    for (int b = 0; b < max_batch; ++b) {
        out[b].n_rows = 0;
        out[b].n_consumed = 0;
        if (in[b].n_tokens && !in[b].tokens) throw RwkvError(RWKV_ERR_INVALID, "slot has n_tokens>0 but tokens==NULL");
        if (in[b].option != RWKV_OPTION_LAST && in[b].option != RWKV_OPTION_FULL && in[b].option != RWKV_OPTION_NONE)
            throw RwkvError(RWKV_ERR_INVALID, "bad RnnOption");
    }
    StepPlan pl;
    plan_step(in, pl);
    if (pl.T == 0) return;
    for (int b = 0; b < max_batch; ++b)
        if (pl.slot_out_rows[b] > 0 && (!out[b].logits || out[b].logits_capacity_rows < (size_t)pl.slot_out_rows[b]))
            throw RwkvError(RWKV_ERR_INVALID, "logits buffer too small for slot " + std::to_string(b));
    run_plan(pl);
    const int V = info.num_vocab;
    // Logits go straight into the caller's buffers when those are pinned host memory (rwkv_host_alloc, or anything registered
    // with the HIP runtime): destinations that are contiguous in output-row order are merged, so a caller that hands one pinned
    // block for all slots gets ONE device-to-host copy and no host-side memcpy.  Pageable destinations take the staged path.
    struct Seg { float *dst; size_t row0, rows; };
    std::vector<Seg> segs;
    for (int b = 0; b < max_batch; ++b) {
        out[b].n_consumed = (size_t)pl.slot_consumed[b];
        out[b].n_rows = (size_t)pl.slot_out_rows[b];
        if (pl.slot_out_rows[b] == 0) continue;
        const size_t r0 = (size_t)pl.slot_out_begin[b], n = (size_t)pl.slot_out_rows[b];
        if (!segs.empty() && segs.back().dst + segs.back().rows * V == out[b].logits && segs.back().row0 + segs.back().rows == r0) segs.back().rows += n;
        else segs.push_back(Seg{out[b].logits, r0, n});
    }
    bool direct = !segs.empty();
    for (const Seg &g : segs) {
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, g.dst) != hipSuccess) { (void)hipGetLastError(); direct = false; break; }
        if (at.type != hipMemoryTypeHost) { direct = false; break; }
    }
    if (direct) {
        for (const Seg &g : segs)
            HIP_CHECK(hipMemcpyAsync(g.dst, logits + g.row0 * V, g.rows * V * 4, hipMemcpyDeviceToHost, s_main));
        HIP_CHECK(hipStreamSynchronize(s_main));
        return;
    }
    // A step that emits no row (state-only requests: RWKV_OPTION_NONE, or `Last` slots that still have tokens pending) hands nothing back
    // but `n_consumed`, which the plan already knows: it is NOT waited for.  The next call's host work (plan, metadata staging, launch)
    // overlaps this step on the device; every call that reads device data (`state.back / read / write`, a step with rows) is ordered
    // behind it on the stream and waits as before.  A LAUNCH error (bad configuration, a sticky device fault from an earlier step) is reported
    // by this call: the error state is polled after the enqueue; an asynchronous fault of THIS step surfaces at the next wait, and the slot states
    // it touched are undefined from then on (the caller has already advanced by n_consumed: include/rwkv_abi.h).
    if (pl.n_out == 0) {
        HIP_CHECK(hipPeekAtLastError());
        return;
    }
    HIP_CHECK(hipMemcpyAsync(logits_host, logits, (size_t)pl.n_out * V * 4, hipMemcpyDeviceToHost, s_main));
    HIP_CHECK(hipStreamSynchronize(s_main));
    for (const Seg &g : segs) std::memcpy(g.dst, logits_host + g.row0 * V, g.rows * V * 4);
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

const char *rwkv_last_error(void) { return g_err.c_str(); }
void rwkv_set_last_error(const char *msg) { g_err = msg ? msg : ""; }   // internal: the tokenizer TU reports through the same slot
int32_t rwkv_abi_version(void) { return RWKV_ABI_VERSION; }

int32_t rwkv_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

rwkv_status rwkv_device_name(int32_t index, char *buf, size_t buf_len) {
    return guard([&] {
        if (!buf || !buf_len) throw RwkvError(RWKV_ERR_INVALID, "null buffer");
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
        if (index < 0 || index >= n) throw RwkvError(RWKV_ERR_DEVICE, "adapter index out of range (ContextError::RequestAdapterFailed)");
        hipDeviceProp_t p;
        HIP_CHECK(hipGetDeviceProperties(&p, index));
        std::snprintf(buf, buf_len, "%s (%s, HIP)", p.name, p.gcnArchName);
    });
}

rwkv_status rwkv_model_info_from_st(const uint8_t *st_bytes, size_t st_len, rwkv_model_info *out) {
    return guard([&] {
        if (!out) throw RwkvError(RWKV_ERR_INVALID, "null out");
        if (prefab_sniff(st_bytes, st_len)) {                   // lib.rs:585-588: safetensors or prefab, by content
            // the whole entry table is walked (headers only: no payload byte is touched), so a truncated or corrupt image is
            // refused here, where a caller asks "what is this file", rather than halfway through a load
            const Prefab p = Prefab::parse(st_bytes, st_len);
            *out = p.hdr.info;
            return;
        }
        SafeTensors st = SafeTensors::parse(st_bytes, st_len);
        *out = detect_info(st);
    });
}

rwkv_status rwkv_engine_save_prefab(rwkv_engine *e, const char *path) {
    return guard([&] {
        if (!e || !path) throw RwkvError(RWKV_ERR_INVALID, "null argument");
        e->save_prefab(path);
    });
}

rwkv_status rwkv_engine_create(const rwkv_load_desc *desc, rwkv_engine **out) {
    return guard([&] {
        if (!desc || !out || !desc->st_bytes) throw RwkvError(RWKV_ERR_INVALID, "null argument");
        *out = nullptr;
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
            throw RwkvError(RWKV_ERR_DEVICE, "no HIP device: librwkv_hip has no CPU fallback");
        if (desc->adapter < RWKV_ADAPTER_ECONOMICAL) throw RwkvError(RWKV_ERR_INVALID, "adapter must be RWKV_ADAPTER_AUTO, RWKV_ADAPTER_ECONOMICAL or a device index");
        int dev = desc->adapter >= 0 ? desc->adapter : 0;   // Auto / Economical: every MI355X is identical -> device 0
        if (dev >= n) throw RwkvError(RWKV_ERR_DEVICE, "adapter index out of range (ContextError::RequestAdapterFailed)");
        HIP_CHECK(hipSetDevice(dev));
        hipDeviceProp_t p;
        HIP_CHECK(hipGetDeviceProperties(&p, dev));
        if (std::strncmp(p.gcnArchName, "gfx950", 6) != 0)
            throw RwkvError(RWKV_ERR_DEVICE, std::string("device is ") + p.gcnArchName + ", this library is built for gfx950 only");
        std::unique_ptr<rwkv_engine> e(new rwkv_engine());
        e->device = dev;
        e->load(*desc);
        *out = e.release();
    });
}

void rwkv_engine_destroy(rwkv_engine *e) { delete e; }

rwkv_status rwkv_engine_info(const rwkv_engine *e, rwkv_model_info *out) {
    return guard([&] {
        if (!e || !out) throw RwkvError(RWKV_ERR_INVALID, "null argument");
        *out = e->info;
    });
}
int32_t rwkv_engine_device(const rwkv_engine *e) { return e ? e->device : -1; }
int32_t rwkv_engine_max_batch(const rwkv_engine *e) { return e ? e->max_batch : 0; }
int32_t rwkv_engine_token_chunk_size(const rwkv_engine *e) { return e ? e->chunk : 0; }
uint64_t rwkv_engine_weight_bytes(const rwkv_engine *e) { return e ? e->weight_bytes : 0; }

// Blocks handed out by rwkv_host_alloc, base -> bytes: the asynchronous read-back checks that the rows it is asked to write END inside the
// block they start in (a pinned block that is too small, or a large offset into it, must come back as RWKV_ERR_INVALID, not as a DMA
// over whatever follows the block in host memory).
static std::mutex g_host_mu;
static std::map<uintptr_t, size_t> g_host_blocks;
rwkv_status rwkv_host_alloc(size_t bytes, void **out) {
    return guard([&] {
        if (!out || bytes == 0) throw RwkvError(RWKV_ERR_INVALID, "bad arguments");
        void *p = nullptr;
        HIP_CHECK(hipHostMalloc(&p, bytes, hipHostMallocDefault));
        { std::lock_guard<std::mutex> g(g_host_mu); g_host_blocks[(uintptr_t)p] = bytes; }
        *out = p;
    });
}
void rwkv_host_free(void *p) {
    if (!p) return;
    { std::lock_guard<std::mutex> g(g_host_mu); g_host_blocks.erase((uintptr_t)p); }
    (void)hipHostFree(p);
}
// bytes from `p` to the end of the pinned block that holds it: from the registry above, else (memory the caller pinned itself) from the
// HIP runtime's address-range query; 0 = unknown
static size_t pinned_bytes_left(const void *p) {
    {
        std::lock_guard<std::mutex> g(g_host_mu);
        auto it = g_host_blocks.upper_bound((uintptr_t)p);
        if (it != g_host_blocks.begin()) {
            --it;
            if ((uintptr_t)p < it->first + it->second) return it->first + it->second - (uintptr_t)p;
        }
    }
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p) == hipSuccess && base && (uintptr_t)p >= (uintptr_t)base && (uintptr_t)p < (uintptr_t)base + size)
        return (uintptr_t)base + size - (uintptr_t)p;
    (void)hipGetLastError();
    return 0;
}

rwkv_status rwkv_infer(rwkv_engine *e, const rwkv_slot_input *in, rwkv_slot_output *out) {
    return guard([&] {
        if (!e || !in || !out) throw RwkvError(RWKV_ERR_INVALID, "null argument");
        use_knobs(e->kn);
        e->infer(in, out);
    });
}

rwkv_status rwkv_infer_sample(rwkv_engine *e, const rwkv_slot_input *in, const rwkv_sample_params *sp, uint32_t *out_tokens,
                              float *out_probs, uint8_t *emitted, size_t *n_consumed) {
    return guard([&] {
        if (!e || !in || !sp || !out_tokens || !emitted) throw RwkvError(RWKV_ERR_INVALID, "null argument");
        use_knobs(e->kn);
        e->infer_sample(in, sp, out_tokens, out_probs, emitted, n_consumed);
    });
}

rwkv_status rwkv_plan_chunk(int32_t max_batch, int32_t token_chunk_size, const size_t *n_tokens, int32_t *consumed) {
    return guard([&] {
        if (max_batch <= 0 || token_chunk_size <= 0 || !n_tokens || !consumed) throw RwkvError(RWKV_ERR_INVALID, "bad arguments");
        std::vector<size_t> pending(n_tokens, n_tokens + max_batch);
        std::vector<int> take;
        water_fill(token_chunk_size, pending, take);
        for (int b = 0; b < max_batch; ++b) consumed[b] = take[b];
    });
}

const char *rwkv_profile_family_name(int32_t f) { return f >= 0 && f < RWKV_PROFILE_FAMILIES ? kFamilyNames[f] : ""; }

rwkv_status rwkv_profile_infer(rwkv_engine *e, const rwkv_slot_input *in, rwkv_slot_output *out, float *ms, int32_t *launches) {
    return guard([&] {
        if (!e || !in || !out || !ms) throw RwkvError(RWKV_ERR_INVALID, "null argument");
        use_knobs(e->kn);
        std::fill(e->prof_ms, e->prof_ms + RWKV_PROFILE_FAMILIES, 0.f);
        std::fill(e->prof_n, e->prof_n + RWKV_PROFILE_FAMILIES, 0);
        e->profiling = true;
        e->prof_fam.clear();
        try {
            e->infer(in, out);
            e->prof_collect();
        } catch (...) {
            e->profiling = false;
            throw;
        }
        e->profiling = false;
        for (int i = 0; i < RWKV_PROFILE_FAMILIES; ++i) {
            ms[i] = e->prof_ms[i];
            if (launches) launches[i] = e->prof_n[i];
        }
    });
}

// ---- state ---------------------------------------------------------------------------------------
size_t rwkv_state_len(const rwkv_engine *e) { return e ? (size_t)e->info.num_layer * 66 * e->info.num_emb : 0; }
void rwkv_state_shape(const rwkv_engine *e, size_t shape[4]) {
    if (!e || !shape) return;
    shape[0] = (size_t)e->info.num_emb; shape[1] = 66; shape[2] = (size_t)e->info.num_layer; shape[3] = 1;
}
rwkv_status rwkv_state_init(const rwkv_engine *e, float *dst) {
    return guard([&] {
        if (!e || !dst) throw RwkvError(RWKV_ERR_INVALID, "null argument");
        std::memset(dst, 0, rwkv_state_len(e) * 4);         // v5/v6/v7 initial state is all-zero
    });
}
static StatePackArgs pack_args(rwkv_engine *e, float *sxa, float *sxf, float *wkv, int to_slab, int layer_only) {
    StatePackArgs a{};
    a.slab = e->slab_dev; a.sxa = sxa; a.sxf = sxf; a.wkv = wkv;
    a.L = e->info.num_layer; a.C = e->info.num_emb; a.H = e->info.num_head;
    a.transposed = e->info.version != 7; a.to_slab = to_slab; a.layer_only = layer_only;
    return a;
}
static void check_slot(const rwkv_engine *e, int slot) {
    if (!e) throw RwkvError(RWKV_ERR_INVALID, "null engine");
    if (slot < 0 || slot >= e->max_batch) throw RwkvError(RWKV_ERR_INVALID, "slot out of range");
}
rwkv_status rwkv_state_load(rwkv_engine *e, int32_t slot, const float *src) {
    return guard([&] {
        check_slot(e, slot);
        if (!src) throw RwkvError(RWKV_ERR_INVALID, "null src");
        HIP_CHECK(hipSetDevice(e->device));
        const size_t n = rwkv_state_len(e);
        std::memcpy(e->slab_host, src, n * 4);
        HIP_CHECK(hipMemcpyAsync(e->slab_dev, e->slab_host, n * 4, hipMemcpyHostToDevice, e->s_main));
        launch_state_pack(pack_args(e, e->sxa + slot * e->sx_slot_stride, e->sxf + slot * e->sx_slot_stride,
                                    e->wkv + slot * e->wkv_slot_stride, 0, -1), e->s_main);
        HIP_CHECK(hipStreamSynchronize(e->s_main));
    });
}
rwkv_status rwkv_state_back(rwkv_engine *e, int32_t slot, float *dst) {
    return guard([&] {
        check_slot(e, slot);
        if (!dst) throw RwkvError(RWKV_ERR_INVALID, "null dst");
        HIP_CHECK(hipSetDevice(e->device));
        const size_t n = rwkv_state_len(e);
        launch_state_pack(pack_args(e, e->sxa + slot * e->sx_slot_stride, e->sxf + slot * e->sx_slot_stride,
                                    e->wkv + slot * e->wkv_slot_stride, 1, -1), e->s_main);
        HIP_CHECK(hipMemcpyAsync(e->slab_host, e->slab_dev, n * 4, hipMemcpyDeviceToHost, e->s_main));
        HIP_CHECK(hipStreamSynchronize(e->s_main));
        std::memcpy(dst, e->slab_host, n * 4);
    });
}
rwkv_status rwkv_state_back_layer(rwkv_engine *e, int32_t slot, int32_t layer, float *dst) {
    return guard([&] {
        check_slot(e, slot);
        if (!dst || layer < 0 || layer >= e->info.num_layer) throw RwkvError(RWKV_ERR_INVALID, "bad layer/dst");
        HIP_CHECK(hipSetDevice(e->device));
        const size_t n = (size_t)64 * e->info.num_emb;
        launch_state_pack(pack_args(e, e->sxa + slot * e->sx_slot_stride, e->sxf + slot * e->sx_slot_stride,
                                    e->wkv + slot * e->wkv_slot_stride, 1, layer), e->s_main);
        HIP_CHECK(hipMemcpyAsync(e->slab_host, e->slab_dev, n * 4, hipMemcpyDeviceToHost, e->s_main));
        HIP_CHECK(hipStreamSynchronize(e->s_main));
        std::memcpy(dst, e->slab_host, n * 4);
    });
}
rwkv_status rwkv_state_back_layer_async(rwkv_engine *e, int32_t slot, int32_t layer, float *dst) {
    return guard([&] {
        check_slot(e, slot);
        if (!dst || layer < 0 || layer >= e->info.num_layer) throw RwkvError(RWKV_ERR_INVALID, "bad layer/dst");
        HIP_CHECK(hipSetDevice(e->device));
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, dst) != hipSuccess || at.type != hipMemoryTypeHost) {
            (void)hipGetLastError();
            throw RwkvError(RWKV_ERR_INVALID, "rwkv_state_back_layer_async: dst must be pinned host memory (rwkv_host_alloc)");
        }
        const size_t n = (size_t)64 * e->info.num_emb;
        {
            const size_t left = pinned_bytes_left(dst);
            if (left && left < n * 4) throw RwkvError(RWKV_ERR_INVALID, "rwkv_state_back_layer_async: the rows would end " + std::to_string(n * 4 - left) + " bytes past the pinned block that holds dst");
        }
        if (!e->s_copy) {
            HIP_CHECK(hipStreamCreateWithFlags(&e->s_copy, hipStreamNonBlocking));
            HIP_CHECK(hipEventCreateWithFlags(&e->ev_copy_a, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&e->ev_copy_b, hipEventDisableTiming));
            HIP_CHECK(hipMalloc((void **)&e->emb_stage, (size_t)e->max_batch * n * 4));
        }
        // the pack reads the slot as the compute stream left it ...
        HIP_CHECK(hipEventRecord(e->ev_copy_a, e->s_main));
        HIP_CHECK(hipStreamWaitEvent(e->s_copy, e->ev_copy_a, 0));
        StatePackArgs a = pack_args(e, e->sxa + slot * e->sx_slot_stride, e->sxf + slot * e->sx_slot_stride,
                                    e->wkv + slot * e->wkv_slot_stride, 1, layer);
        a.slab = e->emb_stage + (size_t)slot * n;                // per-slot staging: the copy stream is in order, so a slot's staging
        launch_state_pack(a, e->s_copy);                         // area is free again before its next pack runs
        // ... and whatever the compute stream does to the slot next waits for the pack (microseconds), not for the PCIe copy
        HIP_CHECK(hipEventRecord(e->ev_copy_b, e->s_copy));
        HIP_CHECK(hipStreamWaitEvent(e->s_main, e->ev_copy_b, 0));
        HIP_CHECK(hipMemcpyAsync(dst, a.slab, n * 4, hipMemcpyDeviceToHost, e->s_copy));
    });
}
rwkv_status rwkv_state_sync(rwkv_engine *e) {
    return guard([&] {
        if (!e) throw RwkvError(RWKV_ERR_INVALID, "null engine");
        HIP_CHECK(hipSetDevice(e->device));
        if (e->s_copy) HIP_CHECK(hipStreamSynchronize(e->s_copy));
    });
}
rwkv_status rwkv_state_read(rwkv_engine *e, int32_t slot, rwkv_dstate **snap) {
    return guard([&] {
        check_slot(e, slot);
        if (!snap) throw RwkvError(RWKV_ERR_INVALID, "null snap");
        HIP_CHECK(hipSetDevice(e->device));
        std::unique_ptr<rwkv_dstate> s(new rwkv_dstate());
        s->device = e->device;
        const size_t nsx = (size_t)e->sx_slot_stride * 4, nw = (size_t)e->wkv_slot_stride * 4;
        HIP_CHECK(hipMalloc((void **)&s->sxa, nsx));
        HIP_CHECK(hipMalloc((void **)&s->sxf, nsx));
        HIP_CHECK(hipMalloc((void **)&s->wkv, nw));
        HIP_CHECK(hipMemcpyAsync(s->sxa, e->sxa + slot * e->sx_slot_stride, nsx, hipMemcpyDeviceToDevice, e->s_main));
        HIP_CHECK(hipMemcpyAsync(s->sxf, e->sxf + slot * e->sx_slot_stride, nsx, hipMemcpyDeviceToDevice, e->s_main));
        HIP_CHECK(hipMemcpyAsync(s->wkv, e->wkv + slot * e->wkv_slot_stride, nw, hipMemcpyDeviceToDevice, e->s_main));
        HIP_CHECK(hipStreamSynchronize(e->s_main));
        *snap = s.release();
    });
}
rwkv_status rwkv_state_write(rwkv_engine *e, int32_t slot, const rwkv_dstate *s) {
    return guard([&] {
        check_slot(e, slot);
        if (!s) throw RwkvError(RWKV_ERR_INVALID, "null snap");
        HIP_CHECK(hipSetDevice(e->device));
        const size_t nsx = (size_t)e->sx_slot_stride * 4, nw = (size_t)e->wkv_slot_stride * 4;
        HIP_CHECK(hipMemcpyAsync(e->sxa + slot * e->sx_slot_stride, s->sxa, nsx, hipMemcpyDeviceToDevice, e->s_main));
        HIP_CHECK(hipMemcpyAsync(e->sxf + slot * e->sx_slot_stride, s->sxf, nsx, hipMemcpyDeviceToDevice, e->s_main));
        HIP_CHECK(hipMemcpyAsync(e->wkv + slot * e->wkv_slot_stride, s->wkv, nw, hipMemcpyDeviceToDevice, e->s_main));
        HIP_CHECK(hipStreamSynchronize(e->s_main));
    });
}
void rwkv_dstate_free(rwkv_dstate *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    (void)hipFree(s->sxa); (void)hipFree(s->sxf); (void)hipFree(s->wkv);
    delete s;
}

rwkv_status rwkv_read_init_state(const rwkv_engine *e, const uint8_t *st_bytes, size_t st_len, float *dst) {
    return guard([&] {
        if (!e || !dst) throw RwkvError(RWKV_ERR_INVALID, "null argument");
        SafeTensors st = SafeTensors::parse(st_bytes, st_len);
        const int L = e->info.num_layer, C = e->info.num_emb, H = e->info.num_head, N = 64;
        if (!st.find("blocks.0.att.time_state")) throw RwkvError(RWKV_ERR_NO_STATE, "no `time_state` tensors in file");
        std::memset(dst, 0, rwkv_state_len(e) * 4);
        for (int l = 0; l < L; ++l) {
            const StTensor &t = st.get("blocks." + std::to_string(l) + ".att.time_state");
            if (t.dtype != "F16" || t.numel() != (int64_t)H * N * N) throw RwkvError(RWKV_ERR_FORMAT, "bad time_state tensor");
            const _Float16 *ts = (const _Float16 *)t.data;     // stored [H][j][i] (converter transposed the last two dims)
            float *rows = dst + ((size_t)l * 66 + 1) * C;
            for (int h = 0; h < H; ++h)
                for (int j = 0; j < N; ++j)
                    for (int i = 0; i < N; ++i) rows[(size_t)i * C + h * N + j] = (float)ts[((size_t)h * N + j) * N + i];
        }
    });
}

// ---- softmax (second caller thread, own stream) ---------------------------------------------------
rwkv_status rwkv_softmax(rwkv_engine *e, const float *const *in, float *const *out, size_t n_rows) {
    return guard([&] {
        if (!e || (n_rows && (!in || !out))) throw RwkvError(RWKV_ERR_INVALID, "null argument");
        if (!n_rows) return;
        HIP_CHECK(hipSetDevice(e->device));
        const size_t V = (size_t)e->info.num_vocab;
        static std::mutex mu;                                  // one softmax task per engine by contract; guard the staging
        std::lock_guard<std::mutex> lk(mu);
        // staging for max_batch rows is allocated at load (never from this thread: the allocator belongs to the infer
        // thread); larger requests are processed in groups
        for (size_t r0 = 0; r0 < n_rows; r0 += e->soft_rows_cap) {
            const size_t n = std::min(e->soft_rows_cap, n_rows - r0);
            for (size_t r = 0; r < n; ++r) {
                if (!in[r0 + r] || !out[r0 + r]) throw RwkvError(RWKV_ERR_INVALID, "null row");
                std::memcpy(e->soft_host + r * V, in[r0 + r], V * 4);
            }
            HIP_CHECK(hipMemcpyAsync(e->soft_in, e->soft_host, n * V * 4, hipMemcpyHostToDevice, e->s_soft));
            launch_softmax(e->soft_in, e->soft_out, (int)n, (int)V, e->s_soft);
            HIP_CHECK(hipMemcpyAsync(e->soft_host, e->soft_out, n * V * 4, hipMemcpyDeviceToHost, e->s_soft));
            HIP_CHECK(hipStreamSynchronize(e->s_soft));
            for (size_t r = 0; r < n; ++r) std::memcpy(out[r0 + r], e->soft_host + r * V, V * 4);
        }
    });
}

// ---- device-resident greedy decode ----------------------------------------------------------------
rwkv_status rwkv_decode_greedy(rwkv_engine *e, int32_t n_slots, const uint32_t *first_tokens, int32_t n_steps,
                               uint32_t *out_tokens, float *elapsed_ms) {
    return guard([&] {
        if (!e || !first_tokens || !out_tokens || n_slots <= 0 || n_slots > e->max_batch || n_slots > e->chunk || n_steps <= 0)
            throw RwkvError(RWKV_ERR_INVALID, "bad arguments");
        HIP_CHECK(hipSetDevice(e->device));
        use_knobs(e->kn);
        StepPlan pl;
        std::vector<rwkv_slot_input> in(e->max_batch);
        std::vector<uint32_t> tk(first_tokens, first_tokens + n_slots);
        for (int b = 0; b < e->max_batch; ++b) {
            in[b] = rwkv_slot_input{b < n_slots ? &tk[b] : nullptr, (size_t)(b < n_slots ? 1 : 0), RWKV_OPTION_LAST, 0};
        }
        e->plan_step(in.data(), pl);
        e->upload_plan(pl);
        const size_t need = (size_t)n_steps * n_slots;
        if (need > e->hist_cap) {                                  // grow: the old buffer goes back (dalloc only frees at destroy)
            if (e->d_hist) {
                e->allocs.erase(std::remove(e->allocs.begin(), e->allocs.end(), (void *)e->d_hist), e->allocs.end());
                (void)hipFree(e->d_hist);
            }
            e->d_hist = e->dalloc<int>(need);
            e->hist_cap = need;
        }
        HIP_CHECK(hipMemcpyAsync(e->d_tok_feedback, tk.data(), n_slots * 4, hipMemcpyHostToDevice, e->s_main));
        HIP_CHECK(hipStreamSynchronize(e->s_main));
        // one graph = one decode step + arg-max feeding the next step's token ids on the device; kept per slot count (the plan of
        // "slots 0..n-1, one token each" is always the same and every buffer it names lives as long as the engine), so a
        // serving loop that calls this repeatedly pays capture + instantiation (~0.7 ms for 260 nodes) once
        hipGraphExec_t exec = nullptr;
        auto cached = e->greedy_graphs.find(n_slots);
        if (cached != e->greedy_graphs.end()) {
            exec = cached->second;
        } else {
            hipGraph_t g = nullptr;
            HIP_CHECK(hipStreamBeginCapture(e->s_main, hipStreamCaptureModeThreadLocal));
            try {
                e->run_layers(pl.T, pl.n_seq, pl.n_out, e->d_tok_feedback, pl.dense);
                launch_argmax(e->logits, n_slots, e->info.num_vocab, e->d_tok_feedback, e->d_amax_v, e->d_amax_i, e->s_main);
            } catch (...) {
                (void)hipStreamEndCapture(e->s_main, &g);
                if (g) (void)hipGraphDestroy(g);
                throw;
            }
            HIP_CHECK(hipStreamEndCapture(e->s_main, &g));
            HIP_CHECK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
            HIP_CHECK(hipGraphDestroy(g));
            e->greedy_graphs.emplace(n_slots, exec);
        }
        HIP_CHECK(hipEventRecord(e->ev0, e->s_main));
        for (int s = 0; s < n_steps; ++s) {
            HIP_CHECK(hipGraphLaunch(exec, e->s_main));
            HIP_CHECK(hipMemcpyAsync(e->d_hist + (size_t)s * n_slots, e->d_tok_feedback, n_slots * 4, hipMemcpyDeviceToDevice, e->s_main));
        }
        HIP_CHECK(hipEventRecord(e->ev1, e->s_main));
        HIP_CHECK(hipEventSynchronize(e->ev1));
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, e->ev0, e->ev1));
        if (elapsed_ms) *elapsed_ms = ms;
        HIP_CHECK(hipMemcpy(out_tokens, e->d_hist, need * 4, hipMemcpyDeviceToHost));
    });
}

// ---- kernel microbench (measurement hook): one [rows x K] problem, `nmat` distinct weight copies rotated so
// that the 256 MiB Infinity Cache cannot serve re-reads; returns average microseconds per launch.
rwkv_status rwkv_bench_gemm(int32_t rows, int32_t K, int32_t fmt, int32_t T, int32_t hilo, int32_t spb, int32_t nmat,
                            int32_t iters, float *us_per_launch, float *lds_kib) {
    return guard([&] {
        if (rows % 16 || K % 256 || T < 1 || nmat < 1 || iters < 1 || iters > 2000) throw RwkvError(RWKV_ERR_INVALID, "bad args");
        use_knobs(Knobs::from_env());                            // no engine here: the microbenchmark reads the environment per call
        hipStream_t st;
        HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        const size_t wbytes = fmt == W_F16 ? (size_t)rows * K * 2 : fmt == W_INT8 ? (size_t)rows * K : (size_t)rows * K / 2;
        const size_t sbytes = fmt == W_F16 ? 16 : fmt == W_INT8 ? (size_t)rows * (K / 128) * 4 : (size_t)rows * (K / 64) * 2;
        std::vector<void *> bufs;
        auto dal = [&](size_t n) { void *p = nullptr; HIP_CHECK(hipMalloc(&p, n)); bufs.push_back(p); return p; };
        std::vector<DMat> mats(nmat);
        // Random operands (round 6): zero or constant fills let the chip clock higher than real data does (the micro-architecture guide measures
        // +15...21 % on a GEMM), so a zero-filled microbenchmark flatters every number it prints.  Weights: random bytes (fp16: random halfs of
        // magnitude < 2; quantised: random codes, scale words near 0.01); X: uniform halfs in [-1, 1).
        uint64_t rs = 0x9E3779B97F4A7C15ull;
        auto rnd = [&]() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; };
        std::vector<uint16_t> hbuf(std::max(wbytes, sbytes) / 2 + 8);
        auto fill_halfs = [&](void *dst, size_t bytes, int kind) {      // 0: raw random bytes, 1: fp16 in (-2, 2), 2: fp16 scale ~0.01, 3: fp16 in [-1, 1)
            const size_t n = (bytes + 1) / 2;
            if (hbuf.size() < n) hbuf.resize(n);
            for (size_t j = 0; j < n; ++j) {
                const uint64_t r = rnd();
                hbuf[j] = kind == 0 ? (uint16_t)r : kind == 1 ? (uint16_t)((r & 0x8000) | (0x3000 + (r >> 20) % 0x1000))
                        : kind == 2 ? (uint16_t)(0x2100 + (r >> 20) % 0x100) : (uint16_t)((r & 0x8000) | (0x2C00 + (r >> 20) % 0x1000));
            }
            HIP_CHECK(hipMemcpy(dst, hbuf.data(), bytes, hipMemcpyHostToDevice));
        };
        for (int i = 0; i < nmat; ++i) {
            mats[i].data = dal(wbytes); mats[i].scales = dal(sbytes);
            if (i == 0) { fill_halfs((void *)mats[i].data, wbytes, fmt == W_F16 ? 1 : 0); fill_halfs((void *)mats[i].scales, sbytes, 2); }
            else {                                                       // the other copies: the same bytes rotated (device-to-device, cheap)
                const size_t rot = ((size_t)i * 4099 * 16) % wbytes & ~(size_t)15;
                HIP_CHECK(hipMemcpy((char *)mats[i].data, (const char *)mats[0].data + rot, wbytes - rot, hipMemcpyDeviceToDevice));
                if (rot) HIP_CHECK(hipMemcpy((char *)mats[i].data + (wbytes - rot), mats[0].data, rot, hipMemcpyDeviceToDevice));
                HIP_CHECK(hipMemcpy((void *)mats[i].scales, mats[0].scales, sbytes, hipMemcpyDeviceToDevice));
            }
            mats[i].fmt = fmt; mats[i].rows = rows; mats[i].K = K;
        }
        Opd x; x.ld = K;
        const size_t xcap = (size_t)((T + 15) / 16 * 16) * x.ld * 2;
        x.hi = (_Float16 *)dal(xcap); x.lo = (_Float16 *)dal(xcap);
        fill_halfs(x.hi, xcap, 3);
        HIP_CHECK(hipMemcpy(x.lo, x.hi, xcap, hipMemcpyDeviceToDevice));
        float *out = (float *)dal((size_t)T * rows * 4);
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
        std::vector<float *> parts;                                   // partial-sum target when K needs a block split
        float *pbuf = (float *)dal((size_t)8 * T * rows * 4);
        // RWKV_BENCH_DUAL=1 (dev): TWO dependent chains over the same matrices on two streams of one graph — what two half-batches
        // of a decode step would do (each chain's launch k streams matrix k; the chains are independent of each other)
        const bool dual = std::getenv("RWKV_BENCH_DUAL") && std::atoi(std::getenv("RWKV_BENCH_DUAL")) != 0;
        hipStream_t st2 = nullptr;
        Opd x2 = x;
        float *out2 = out, *pbuf2 = pbuf;
        if (dual) {
            HIP_CHECK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
            x2.hi = (_Float16 *)dal(xcap); x2.lo = (_Float16 *)dal(xcap);
            HIP_CHECK(hipMemset(x2.hi, 0, xcap)); HIP_CHECK(hipMemset(x2.lo, 0, xcap));
            out2 = (float *)dal((size_t)T * rows * 4);
            pbuf2 = (float *)dal((size_t)8 * T * rows * 4);
        }
        hipStream_t run_st = st;
        auto run = [&](int n) {
            const bool second = run_st != st;
            for (int i = 0; i < n; ++i) {
                std::vector<ProbSpec> ps(1);
                ps[0].W = &mats[i % nmat]; ps[0].x = second ? x2 : x; ps[0].out = K > 5120 ? (second ? pbuf2 : pbuf) : (second ? out2 : out); ps[0].ldo = rows;
                ps[0].partial = K > 5120;
                GemmLaunch Lh;
                if (T >= GEMM_TILE_MIN_T) {                      // prefill path; `spb` selects the tile shape (0..3), -1 = auto
                    int shape = spb;
                    if (shape < 0 || shape >= GEMM_TILE_SHAPES) {
                        shape = GEMM_TILE3;
                        for (int sh = 0; sh < GEMM_TILE_SHAPES; ++sh) if (gemm_tile_blocks(sh, rows, T) >= 1024) { shape = sh; break; }
                    }
                    if (!gemm_tile_shape_supported(shape, hilo != 0, K)) throw RwkvError(RWKV_ERR_INVALID, "bench_gemm: the pipelined shapes need K % 128 == 0; shapes 10 / 11 take plain operands, shape 12 hi + lo");
                    Lh = GemmLaunch{};
                    Lh.nprob = 1; Lh.T = T;
                    GemmProb &g = Lh.p[0];
                    g.W = ps[0].W->data; g.S = ps[0].W->scales; g.fmt = fmt; g.rows = rows; g.K = K;
                    g.xhi = x.hi; g.xlo = x.lo; g.ldx = K; g.ksb = 1; g.block_begin = 0;
                    g.out_f32 = out; g.ldo = rows;
                    if (const char *e = std::getenv("RWKV_BENCH_KSPLIT")) {      // K copies of the tile grid writing partial slabs (linear launches)
                        g.ksb = std::max(1, std::min(8, std::atoi(e)));
                        g.out_f32 = pbuf; g.partial_stride = (long)T * rows;
                    }
                    Lh.total_blocks = gemm_tile_blocks(shape, rows, T) * g.ksb;
                    Lh.xcd_map = 1;
                    if (const char *e = std::getenv("RWKV_TILE_XCD")) { if (*e) Lh.xcd_map = std::atoi(e); }     // 2: token-tile-major bands
                    if (lds_kib) *lds_kib = (float)Lh.total_blocks;
                    launch_gemm_tile(Lh, shape, hilo != 0, run_st);
                    continue;
                }
                plan_gemm(Lh, ps, T, hilo != 0, (long)T * rows, spb);
                if (lds_kib) *lds_kib = (float)Lh.total_blocks;
                launch_gemm(Lh, hilo != 0, run_st);
            }
        };
        run(nmat);
        HIP_CHECK(hipStreamSynchronize(st));
        hipEvent_t ev_fork = nullptr, ev_join = nullptr;
        if (dual) { HIP_CHECK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming)); HIP_CHECK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming)); }
        hipGraph_t g = nullptr;
        hipGraphExec_t ge = nullptr;
        HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        if (dual) {
            HIP_CHECK(hipEventRecord(ev_fork, st));
            HIP_CHECK(hipStreamWaitEvent(st2, ev_fork, 0));
            run_st = st2; run(iters); run_st = st;
            HIP_CHECK(hipEventRecord(ev_join, st2));
        }
        run(iters);
        if (dual) HIP_CHECK(hipStreamWaitEvent(st, ev_join, 0));
        HIP_CHECK(hipStreamEndCapture(st, &g));
        HIP_CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        HIP_CHECK(hipGraphLaunch(ge, st));
        HIP_CHECK(hipStreamSynchronize(st));
        HIP_CHECK(hipEventRecord(e0, st));
        HIP_CHECK(hipGraphLaunch(ge, st));
        HIP_CHECK(hipEventRecord(e1, st));
        HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
        if (us_per_launch) *us_per_launch = ms * 1e3f / iters;
        for (void *p : bufs) (void)hipFree(p);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(st);
        if (dual) { (void)hipEventDestroy(ev_fork); (void)hipEventDestroy(ev_join); (void)hipStreamDestroy(st2); }
    });
}

}  // extern "C"
