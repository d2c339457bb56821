/*
 * rwkv_abi.h — flat C ABI of librwkv_hip.so, the MI355X-native replacement for the
 * `web-rwkv` surface that ai00-core binds (ai00_server @ 2025-10-24).
 *
 * Every entry point names the reference call site it replaces (paths relative to the
 * reference root, crates/ai00-core/src/...).  Conventions:
 *   - plain pointers + sizes only; no C++/torch types cross this boundary;
 *   - every function returning `rwkv_status` returns 0 on success and a negative code on
 *     failure; `rwkv_last_error()` gives a thread-local message.  Nothing aborts — mirrors
 *     the `Result<_, RuntimeError|TensorError>` + `anyhow ?` convention (run.rs:1143);
 *   - threading contract (run.rs:1072-1190): per engine exactly two long-lived caller threads:
 *     the `infer` task (serialises rwkv_infer + all rwkv_state_* calls) and the `softmax`
 *     task (rwkv_softmax, own stream).  Tokenizer handles are immutable and thread-safe.
 *   - the library REQUIRES a gfx950 device: there is no CPU fallback; engine creation fails
 *     with RWKV_ERR_DEVICE when no HIP device is present.
 */
#ifndef RWKV_ABI_H
#define RWKV_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RWKV_ABI_VERSION 7   /* 2: rwkv_sample_params gained kind/tau; rwkv_engine_save_prefab
                              * 3: rwkv_sample_params gained allow (formatter mask); rwkv_host_alloc/free; RWKV_OPTION_NONE
                              * 4: rwkv_engine_token_chunk_size
                              * 5: rwkv_state_back_layer_async / rwkv_state_sync
                              * 6: no new symbol — rwkv_infer no longer waits for a step that emits no row; rwkv_state_back_layer_async checks
                              *    that the rows end inside the pinned block that holds `dst`
                              * 7: rwkv_load_desc.precision: RWKV_PRECISION_FP16 now holds 1e-3 at depth (the launches that carry a model's f16 operand
                              *    rounding read hi + lo operands); the old all-f16 behaviour is RWKV_PRECISION_FP16_RAW */

typedef int32_t rwkv_status;
enum {
    RWKV_OK = 0,
    RWKV_ERR_INVALID = -1,     /* bad argument / malformed input            */
    RWKV_ERR_FORMAT = -2,      /* not a safetensors file / missing tensor   */
    RWKV_ERR_UNSUPPORTED = -3, /* model version (v4, v5.0/5.1) or option    */
    RWKV_ERR_DEVICE = -4,      /* no HIP device / HIP runtime error         */
    RWKV_ERR_OOM = -5,
    RWKV_ERR_NO_STATE = -6     /* file has no `time_state` tensors (lib.rs:442) */
};

/* thread-local text of the last failure on this thread ("" if none). */
const char *rwkv_last_error(void);
int32_t rwkv_abi_version(void);

/* ---- adapters: `list_adapters` lib.rs:339-349, surfaced by /api/adapters (adapter.rs:8-14) */
int32_t rwkv_device_count(void);
rwkv_status rwkv_device_name(int32_t index, char *buf, size_t buf_len);

/* ---- `Loader::info(&SafeTensors)` lib.rs:587, api/file.rs:113-116 -> `ModelInfo` ---------- */
enum { RWKV_V5 = 5, RWKV_V6 = 6, RWKV_V7 = 7 };
typedef struct rwkv_model_info {
    int32_t version;     /* ModelVersion: 5 (=v5.2), 6, 7 */
    int32_t num_layer;
    int32_t num_emb;
    int32_t num_hidden;
    int32_t num_vocab;
    int32_t num_head;
    int32_t head_size;   /* num_emb / num_head (64) */
    int32_t reserved;
} rwkv_model_info;
rwkv_status rwkv_model_info_from_st(const uint8_t *st_bytes, size_t st_len, rwkv_model_info *out);
/* `st_bytes` here and in rwkv_load_desc may also be a PREFAB image written by rwkv_engine_save_prefab: the content is
 * sniffed exactly as lib.rs:585-588 does (safetensors, else prefab).  A prefab carries the re-tiled / quantised /
 * LoRA-blended weights, so loading it skips those passes; its quantisation settings override the descriptor's and
 * LoRA adapters are rejected (RWKV_ERR_UNSUPPORTED).  The format is this library's own (versioned), not web-rwkv's CBOR. */

/* ---- `create_context` lib.rs:351-368 + `load_runtime` lib.rs:391-516 ---------------------- */
enum { RWKV_QUANT_NONE = 0, RWKV_QUANT_INT8 = 1, RWKV_QUANT_NF4 = 2 };   /* `Quant` lib.rs:689-704 */
/* `Precision` reload.rs:89-94, passed by `load_runtime` lib.rs:503-515.  GEMM operands are f16 with fp32 accumulation in every mode:
 *   FP16 (the reference's default, and this library's): f16 operands, except that the launches whose input rounding carries a model's
 *        error at depth read the operand as a hi + lo f16 pair (V5 / V6: the time-mix projections and first-stage LoRAs; V7: those, the
 *        second-stage LoRAs and the output projection) — logits, state and embeddings within 1e-3 of an fp32 evaluation at 32 layers;
 *   FP32: every launch reads hi + lo operands (fp32-class: <= 2e-5);
 *   FP16_RAW: f16 operands everywhere — the fastest mode; relative error ~1e-3 of the row's magnitude, NOT within 1e-3 absolute at 32 layers
 *        (V7-2.9B NF4 measures 4.7e-3).  For callers that accept that. */
enum { RWKV_PRECISION_FP16 = 0, RWKV_PRECISION_FP32 = 1, RWKV_PRECISION_FP16_RAW = 2 };
enum { RWKV_ADAPTER_AUTO = -1, RWKV_ADAPTER_ECONOMICAL = -2 };           /* reload.rs AdapterOption; >=0 = Manual(n) */

typedef struct rwkv_lora_desc {    /* `reload::Lora{path, alpha}` + LoraBlend::full(alpha), lib.rs:466-482 */
    const uint8_t *st_bytes;
    size_t st_len;
    float alpha;
} rwkv_lora_desc;

typedef struct rwkv_load_desc {    /* the `ReloadRequest` fields that reach web-rwkv, lib.rs:200-231 */
    int32_t adapter;               /* RWKV_ADAPTER_* or device index (Manual(n))            */
    int32_t quant_layers;          /* `quant`: layers 0..quant are quantised (lib.rs:465)   */
    int32_t quant_type;            /* RWKV_QUANT_*                                          */
    int32_t precision;             /* RWKV_PRECISION_*: activation precision at GEMM inputs */
    int32_t max_batch;             /* state slots resident on the device (default 8)        */
    int32_t token_chunk_size;      /* max tokens consumed per rwkv_infer call (default 128) */
    const uint8_t *st_bytes;       /* model `.st` bytes (caller's mmap; released after return, lib.rs:446) */
    size_t st_len;
    const rwkv_lora_desc *lora;    /* may be NULL */
    size_t n_lora;
} rwkv_load_desc;

typedef struct rwkv_engine rwkv_engine;   /* = Context + Model + vN::Bundle + TokioRuntime<Rnn> + State */

rwkv_status rwkv_engine_create(const rwkv_load_desc *desc, rwkv_engine **out);
void rwkv_engine_destroy(rwkv_engine *e);                       /* Unload/drop lib.rs:652-656 */
/* `ModelSerialize::serialize(file)` lib.rs:131-154 (the "save model" admin call): write the loaded model as a prefab image. */
rwkv_status rwkv_engine_save_prefab(rwkv_engine *e, const char *path);
rwkv_status rwkv_engine_info(const rwkv_engine *e, rwkv_model_info *out);
int32_t rwkv_engine_device(const rwkv_engine *e);               /* HIP device ordinal in use */
int32_t rwkv_engine_max_batch(const rwkv_engine *e);
int32_t rwkv_engine_token_chunk_size(const rwkv_engine *e);     /* the load-time `token_chunk_size` (lib.rs:221-223): rows one rwkv_infer call emits at most per Full slot */
/* bytes of weights resident in HBM at their storage width (fp16 / int8+scales / nf4+absmax),
 * embedding table excluded: the W_q of SURVEY 8(d).  For roofline accounting. */
uint64_t rwkv_engine_weight_bytes(const rwkv_engine *e);

/* ---- `runtime.infer(RnnInput) -> (RnnInput, RnnOutput)` run.rs:1134-1156 ------------------ */
enum { RWKV_OPTION_LAST = 0, RWKV_OPTION_FULL = 1,              /* RnnOption, run.rs:716,819 */
       RWKV_OPTION_NONE = 2 };   /* extension: consume the tokens, emit no row (state-only jobs: the documented `/embeddings`
                                  * route, docs/doc-api/openai.md:376-437, reads back a state slice and never looks at logits);
                                  * a step in which no slot emits skips the final LayerNorm, the head GEMM and the logits copy */
typedef struct rwkv_slot_input {   /* RnnInputBatch::new(tokens, option) run.rs:1128 */
    const uint32_t *tokens;        /* remaining tokens of this slot (may be NULL if n_tokens==0) */
    size_t n_tokens;
    int32_t option;                /* RWKV_OPTION_* */
    int32_t reserved;
} rwkv_slot_input;
typedef struct rwkv_slot_output {  /* RnnOutputBatch, run.rs:1146-1155 */
    float *logits;                 /* caller buffer, >= logits_capacity_rows * num_vocab floats */
    size_t logits_capacity_rows;
    size_t n_rows;                 /* OUT: rows written (0 = nothing emitted this call)      */
    size_t n_consumed;             /* OUT: tokens of this slot consumed by this call          */
} rwkv_slot_output;
/* One forward step over <= token_chunk_size tokens spread across the `max_batch` slots
 * (arrays have max_batch entries).  The caller advances `tokens` by `n_consumed` and calls
 * again while any tokens remain (the `while input.num_token() > 0` loop, run.rs:1134).
 * Last: one row when the slot's tokens are exhausted by this call.  Full: one row per token
 * consumed.  State of each touched slot is updated in place on the device.  A call that emits rows returns when they are in the
 * caller's buffers.  A call that emits NO row (state-only slots, or `Last` slots whose tokens are not exhausted yet) returns as soon as
 * the step is queued: `n_consumed` is final, the device runs behind, and everything that reads device data afterwards (a later
 * rwkv_infer with rows, rwkv_state_back / _read / _write / _back_layer[_async]) is ordered behind it — host work between two steps of a
 * long prefill overlaps the device.  The `tokens` arrays may be reused as soon as the call returns (they are staged on return).  Errors of such a
 * call: a launch error is returned by the call itself; an asynchronous device fault of the queued step is returned by the NEXT call that waits
 * (a step with rows, rwkv_state_back / _read / _sync), and the states of the slots the step touched are undefined from then on. */
rwkv_status rwkv_infer(rwkv_engine *e, const rwkv_slot_input *in, rwkv_slot_output *out);

/* Pinned host memory for the `logits` buffers of rwkv_infer (the `TensorCpu<f32>` outputs of run.rs:1146-1155 are read back
 * through mapped staging buffers in web-rwkv; this is the equivalent on the HIP side).  When every destination of a call is
 * pinned, rows are copied device-to-host straight into it (destinations contiguous in slot order become one copy: hand the
 * slots consecutive pieces of one block); pageable buffers still work and take a staged copy.  Any thread; free with
 * rwkv_host_free. */
rwkv_status rwkv_host_alloc(size_t bytes, void **out);
void rwkv_host_free(void *p);

/* The chunk policy of rwkv_infer as a pure host function (no device needed): how many of each slot's pending tokens one
 * call consumes — water-filling of `token_chunk_size`, so decode slots are never starved by a long prefill
 * (web-rwkv's own split inside `RnnInput::new(batches, chunk)` run.rs:1132 is not visible; any split is
 * result-equivalent). */
rwkv_status rwkv_plan_chunk(int32_t max_batch, int32_t token_chunk_size, const size_t *n_tokens, int32_t *consumed);

/* ---- `State` trait: run.rs:477,950 (init) 1099 (load) 1101 (back) 1104 (write) 1106 (read) -- */
size_t rwkv_state_len(const rwkv_engine *e);                       /* floats in one slab      */
void rwkv_state_shape(const rwkv_engine *e, size_t shape[4]);      /* [C, N+2, L, 1] run.rs:987 */
rwkv_status rwkv_state_init(const rwkv_engine *e, float *dst);     /* zero slab (CPU)         */
rwkv_status rwkv_state_load(rwkv_engine *e, int32_t slot, const float *src);   /* H2D         */
rwkv_status rwkv_state_back(rwkv_engine *e, int32_t slot, float *dst);         /* D2H, blocks */
typedef struct rwkv_dstate rwkv_dstate;                            /* TensorGpu snapshot       */
rwkv_status rwkv_state_read(rwkv_engine *e, int32_t slot, rwkv_dstate **snap); /* D2D copy out */
rwkv_status rwkv_state_write(rwkv_engine *e, int32_t slot, const rwkv_dstate *snap); /* D2D in; snap reusable */
void rwkv_dstate_free(rwkv_dstate *snap);
/* f-2 (docs/doc-api/openai.md:376-437): one layer's WKV rows [N][C] of a slot, D2H */
rwkv_status rwkv_state_back_layer(rwkv_engine *e, int32_t slot, int32_t layer, float *dst);
/* The same read-back, NOT waited for: the layer's rows are packed on a second stream (ordered behind everything issued so far; later
 * work on the slot — rwkv_infer, rwkv_state_load / _write — is ordered behind the pack, a few microseconds) and copied to `dst`, which
 * must be pinned host memory (rwkv_host_alloc), while the engine goes on with the next rwkv_infer.  An embedding job (`/embeddings`,
 * docs/doc-api/openai.md:376-437; the reference's `state.back(batch).await` yields to the other tasks the same way, run.rs:1101) hands a
 * finished document's slot to the next document without waiting for PCIe.  `dst` is valid after rwkv_state_sync. */
rwkv_status rwkv_state_back_layer_async(rwkv_engine *e, int32_t slot, int32_t layer, float *dst);
rwkv_status rwkv_state_sync(rwkv_engine *e);                       /* wait for every pending rwkv_state_back_layer_async */

/* ---- `vN::read_state(context, info, model)` lib.rs:378-389 --------------------------------- */
rwkv_status rwkv_read_init_state(const rwkv_engine *e, const uint8_t *st_bytes, size_t st_len, float *dst);

/* ---- `softmax::softmax(&context, Vec<TensorCpu>)` run.rs:1178-1183 -------------------------
 * n rows of num_vocab floats each, host pointers in / out (may alias).  Own stream: may run
 * concurrently with rwkv_infer from the second caller thread. */
rwkv_status rwkv_softmax(rwkv_engine *e, const float *const *in, float *const *out, size_t n_rows);

/* ---- on-device sampling front-end: what `sample()` run.rs:664-697 + NucleusSampler::sample
 * (sampler/nucleus.rs:69-101) do with three PCIe hops and a 65,536-element CPU sort, done on the device.
 * The caller keeps the sampler STATE (penalty map, nucleus.rs:104-119) and passes its effect as sparse logit
 * adjustments (-penalty[token] + bias[token], duplicates merged) plus the uniform draw `fastrand::f32()` would make. */
typedef struct rwkv_sample_params {
    float top_p;                   /* NucleusParams defaults: 0.5 / 128 / 1.0 (nucleus.rs:13-26); top_k <= 256 */
    int32_t top_k;
    float temperature;
    float uniform;                 /* u in [0,1) */
    const uint32_t *adj_tokens;    /* may be NULL when n_adj == 0 */
    const float *adj_values;       /* added to logits[adj_tokens[i]] before the softmax */
    size_t n_adj;
    int32_t kind;                  /* RWKV_SAMPLER_NUCLEUS: top_p / top_k / temperature (nucleus.rs:69-101).                 */
    float tau;                     /* RWKV_SAMPLER_TYPICAL (typical.rs:70-120; defaults 0.5 / 128 / 1.0): keys |(-ln p) - H|  */
                                   /*   ascending, take top_k, keep while the cumulative probability before an element <= tau */
                                   /* RWKV_SAMPLER_MIROSTAT (mirostat.rs:44-90): tau = the sampler's current `max_surprise`;  */
                                   /*   top_p / top_k / temperature unused; out_probs[b] returns the TOKEN SURPRISE           */
                                   /*   log2(sum) - log2(p) the caller needs for `max_surprise -= rate * (surprise - tau)`.   */
                                   /*   Exact while max_surprise < 13 (<= 8192 candidates); beyond, the tail below 2^-13 is cut. */
    const uint8_t *allow;          /* NULL, or num_vocab bytes: allow[token] == 0 forbids the token (its logit becomes -inf before  */
                                   /*   the softmax).  This is what a `Formatter::transform` leaves behind (run.rs:676-679,         */
                                   /*   sampler/bnf.rs:35-38: kbnf's mask_logits): the grammar state machine stays on the host and  */
                                   /*   hands over its current allowed-token set, so BNF-constrained requests can sample on the     */
                                   /*   device too.                                                                                 */
} rwkv_sample_params;
enum { RWKV_SAMPLER_NUCLEUS = 0, RWKV_SAMPLER_TYPICAL = 1, RWKV_SAMPLER_MIROSTAT = 2 };
/* Like rwkv_infer with RWKV_OPTION_LAST on every slot, but slots whose pending tokens are exhausted by this call
 * get a sampled token id (out_tokens[b], emitted[b] = 1, out_probs[b] = its softmax probability) instead of a
 * logits row; only 8 bytes per slot cross PCIe.  n_consumed[b] as in rwkv_slot_output.  num_vocab <= 65536. */
rwkv_status rwkv_infer_sample(rwkv_engine *e, const rwkv_slot_input *in, const rwkv_sample_params *sp,
                              uint32_t *out_tokens, float *out_probs, uint8_t *emitted, size_t *n_consumed);

/* ---- `Tokenizer` lib.rs:375; run.rs:157-168,856; sampler/bnf.rs:14-27 ---------------------- */
typedef struct rwkv_tokenizer rwkv_tokenizer;
rwkv_status rwkv_tokenizer_create(const char *vocab_json, size_t len, rwkv_tokenizer **out);
void rwkv_tokenizer_destroy(rwkv_tokenizer *t);
/* returns number of tokens (may exceed cap: call again with a larger buffer), <0 on error */
int64_t rwkv_tokenizer_encode(const rwkv_tokenizer *t, const uint8_t *text, size_t len, uint32_t *out, size_t cap);
/* returns number of bytes (may exceed cap), <0 on error (unknown token id) */
int64_t rwkv_tokenizer_decode(const rwkv_tokenizer *t, const uint32_t *tokens, size_t n, uint8_t *out, size_t cap);
int64_t rwkv_tokenizer_token_bytes(const rwkv_tokenizer *t, uint32_t token, uint8_t *out, size_t cap);
int64_t rwkv_tokenizer_vocab_size(const rwkv_tokenizer *t);   /* highest id + 1 */

/* ---- measurement hooks (bench.py; not part of the reference surface) -----------------------
 * rwkv_profile_infer runs the same step as rwkv_infer but brackets every kernel launch with
 * hipEvents on the engine's compute stream and accumulates per-kernel-family milliseconds.
 * Families: see rwkv_profile_family_name.  `ms` has RWKV_PROFILE_FAMILIES entries. */
#define RWKV_PROFILE_FAMILIES 8
const char *rwkv_profile_family_name(int32_t family);
rwkv_status rwkv_profile_infer(rwkv_engine *e, const rwkv_slot_input *in, rwkv_slot_output *out,
                               float *ms, int32_t *launches);
/* device-resident greedy decode (f-1 front-end, arg-max only): runs `n_steps` decode steps for the
 * first `n_slots` slots, feeding each slot's arg-max token back on the device; only token ids
 * cross PCIe.  first_tokens[n_slots] in, out_tokens[n_steps*n_slots] (step-major) out.
 * Returns milliseconds of device time for the timed region in *elapsed_ms (hipEvents). */
rwkv_status rwkv_decode_greedy(rwkv_engine *e, int32_t n_slots, const uint32_t *first_tokens,
                               int32_t n_steps, uint32_t *out_tokens, float *elapsed_ms);

/* kernel microbench: one [rows x K] GEMM problem in weight format `fmt` (0 fp16, 1 int8, 2 nf4) against T
 * activation rows; `nmat` distinct weight copies are rotated so re-reads miss the Infinity Cache.
 * ksw = 0 lets the planner choose the in-block K split.  Returns microseconds per launch. */
rwkv_status rwkv_bench_gemm(int32_t rows, int32_t K, int32_t fmt, int32_t T, int32_t hilo, int32_t ksw,
                            int32_t nmat, int32_t iters, float *us_per_launch, float *lds_kib);

#ifdef __cplusplus
}
#endif
#endif /* RWKV_ABI_H */
