// rwkv_kernels.h — host-visible launch descriptors for the gfx950 kernels in rwkv_kernels.hip.
// Pure POD + launch prototypes; no HIP types except hipStream_t.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rwkv {

enum WFmt : int { W_F16 = 0, W_INT8 = 1, W_NF4 = 2 };
enum Act : int { ACT_NONE = 0, ACT_TANH = 1, ACT_SIGMOID = 2, ACT_RELU2 = 3, ACT_SILU = 4, ACT_DECAY7 = 5 };
enum Post : int { POST_NONE = 0, POST_MUL = 1, POST_MIX = 2 };

// Experiment switches (A/B runs and the parity tests of the non-default paths).  Read from the environment ONCE PER ENGINE, when it
// is created, and installed for the thread that drives the engine (`use_knobs`): every step of an engine, and every graph it
// captured, sees the same choices whatever the environment does later.  Defaults are the product.  DESIGN.md 4.1.
struct Knobs {
    int no_ln_fuse = 0;              // RWKV_NO_LN_FUSE: the ln_shift row kernel instead of the LayerNorm prologue of the V6 mix on single-token steps
    int no_v6_fuse = 0;              // RWKV_NO_V6_FUSE: the V6 token-shift LoRA as two GEMM launches instead of v6_mix_kernel
    int no_tile = 0, tile_shape = -1;   // RWKV_NO_TILE: the decode GEMM for every step; RWKV_TILE_SHAPE=0..12: force a prefill tile shape (parity of every shape)
    int no_dense = 0;                // RWKV_NO_DENSE: general row metadata on dense decode steps
    int tile_ksplit = 1;             // RWKV_TILE_KSPLIT=0: no K copies of linear prefill launches (the race screen compares tile shapes BIT for bit, which needs one summation order)
    int promote = -1;                // RWKV_PROMOTE (dev override; -1 = unset: the precision mode decides): bit mask of GEMM launch classes that read hi + lo
                                     // operands in the fp16 modes (1 att r/k/v(/g) + first-stage LoRAs, 2 V7 second-stage LoRAs, 4 Wo, 8 Fk / Fr, 16 Fv,
                                     // 32 head); rwkv_engine.cpp OpdClass
    int v6_ksp_max = 16;             // RWKV_V6_KSP_MAX=1: the V6 token-shift LoRA's first stage never sliced over K (steps of 192 .. 511 rows; launch_v6_mix)
    int v6_ksp_blocks = 160, v6_ksp_min_t = 192;   // (dev) RWKV_V6_KSP_BLOCKS / RWKV_V6_KSP_MIN_T: blocks the slicing aims for, smallest step it applies to
    static Knobs from_env();
};
const Knobs &knobs();                // the calling thread's current set
void use_knobs(const Knobs &k);

constexpr int TILE_ROWS = 16;        // output rows per strip (MFMA 16x16x32 M)
constexpr int KSTEP = 32;            // K per MFMA
constexpr int GEMM_MAX_WAVES = 10;    // 640-thread blocks: <= 168 VGPRs per lane (KSW = 8 variants)
constexpr int GEMM_MAX_WAVES_K16 = 8; // four-tile (NT = 4) and two-tile hi + lo variants: 512-thread blocks, 2 waves per SIMD -> 256 VGPRs per lane
int gemm_variant_max_waves(int NT, int KSW, bool hilo = false);
constexpr int GEMM_MAXP = 8;
constexpr int INT8_BLOCK = 128;
constexpr int NF4_BLOCK = 64;

// A weight matrix W[rows][K] resident in HBM in the pre-tiled MFMA-A layout:
//   fp16: tile = 16 rows x 32 k  = 1 KiB ; lane l holds row (l&15), k = kt*32 + (l>>4)*8 + [0,8)
//   int8: tile = 16 rows x 64 k  = 1 KiB ; lane bytes [0,8) -> k-step 0, [8,16) -> k-step 1
//   nf4 : tile = 16 rows x 128 k = 1 KiB ; lane byte m: k-step m/4, e = (m%4)*2 + {lo,hi nibble}
// tiles of one strip are contiguous along K: tile index = strip*KT + kt.
struct DMat {
    const void *data = nullptr;     // tiled payload
    const void *scales = nullptr;   // int8: half2{a,b} [strip][K/128][16]; nf4: half [strip][K/64][16]
    int fmt = W_F16;
    int rows = 0;                   // multiple of 16
    int K = 0;                      // multiple of 32 (fp16) / 128 (int8, nf4)
    uint64_t bytes = 0;             // payload + scales, for roofline accounting
};

struct RowMeta {                    // device arrays, one entry per row of this step
    const int *token;               // token id
    const int *slot;                // state slot
    const int *prev;                // previous row of the same slot in this step, or -1 (-> state)
    const int *last;                // for first rows: last row of the slot in this step; else -1
    int dense;                      // 1: dense decode step — row t belongs to slot t, one row per slot (prev = -1, last = t):
                                    //    the kernels derive slot / prev / last from the row index instead of loading them
};

// LayerNorm + token-shift PROLOGUE of a GEMM-like kernel (tiny decode steps, T <= LNP_MAX_T): every block redoes the
// row work of `ln_shift_kernel` for all T rows in LDS and builds its MFMA B fragments from there, which removes one
// launch (~5 us of pure latency) per LayerNorm.  Block 0 also publishes x_out (the residual stream) and xx_out (the
// normalised rows); the token-shift STATE is committed by the launch that follows (ShiftCommit), because other
// blocks of this launch still read the old state.
constexpr int LNP_MAX_T = 1;        // single-token decode steps (two rows at once do not fit the register file beside the weights)
constexpr int LNP_MAX_NP = 5;       // partial-sum slabs of the producing GEMM
struct LnProArgs {
    const float *x_in;              // null: no prologue in this launch
    float *x_out;
    const float *P;
    int np;
    long pstride;
    const float *lnw, *lnb;
    const float *sx;                // token-shift state of this layer (read only here)
    long sx_slot_stride;
    RowMeta rm;
    int mode;                       // 0: V5  op = xx*mu + prev*(1-mu);  1: V6/V7  op = xx + (prev-xx)*mu
    float *xx_out;                  // [T][C]
    int C;
};
struct ShiftCommit {                // sx[slot[t]] = src[last[t]] for rows with last[t] >= 0; run by one extra block
    const float *src;               // null: nothing to commit
    float *sx;
    long sx_slot_stride;
    RowMeta rm;
    int T, C;
};

struct GemmProb {
    const void *W;
    const void *S;
    const _Float16 *xhi;            // activation operand [T][ldx] (f16, hi part)
    const _Float16 *xlo;            // lo part (may be null when !HILO)
    int fmt, rows, K, ldx;
    int spb;                        // strips (of 16 output rows) per block
    int nw;                         // waves covering the block's K range (KW = 512 or 256 k each)
    int ksb;                        // blocks splitting K (partials written, linear epilogue only)
    int Kb, nslice;                 // K / ksb and ceil(Kb / (KSW * 32)) of the launch's variant: computed by the host, not by every wave
    int nblk_strip;                 // blocks per K-slice = ceil(strips / spb)
    int block_begin;                // first block of this problem in the launch
    // epilogue:  v = act(acc + bias[row]);  POST_MUL: v *= m0[t][row];  POST_MIX: v = m0 + m1 * v
    int act, post;
    const float *bias;
    const float *m0, *m1;
    int ldm;
    float *out_f32;                 // [T][ldo] (+ kb*partial_stride when ksb>1)
    int ldo;
    long partial_stride;
    _Float16 *out_hi, *out_lo;      // operand output [T][ldh]
    int ldh;
};

struct GemmLaunch {
    GemmProb p[GEMM_MAXP];
    int nprob;
    int T;                          // activation rows
    int threads;                    // block size = 64 * max nw over the problems
    int lds_items;                  // max over problems of spb*nw (LDS = items * NT KiB)
    int single_shot;                // every problem's rounds per wave fit in registers: issue all loads up-front
    int tail;                       // some fp16 problem has a K range that is not a multiple of 256: predicated variant
    int total_blocks;
    int xcd_map;                    // prefill tile GEMM: XCD-banded tile numbering (rwkv_kernels.hip tg_body)
    ShiftCommit commit;             // grid = total_blocks + 1 when commit.src is set
};

void gemm_variant(int T, bool hilo, int &NT, int &KSW);   // tile variant used for T rows
size_t lnp_lds_bytes(int T, int C, bool hilo);                       // extra dynamic LDS of an LN-prologue launch
void launch_gemm(const GemmLaunch &L, bool hilo, hipStream_t s);
bool smallk_supported(const GemmLaunch &L);              // every problem: fp16, K <= 320, fp32 output, no post-op (V7's second LoRA stage)
void launch_smallk(const GemmLaunch &L, bool hilo, hipStream_t s);   // output-stationary: one wave per (problem, strip), all rows of the step
int gemm_max_rounds(int fmt, int NT, bool hilo);
// prefill path (T >= GEMM_TILE_MIN_T): LDS-tiled MFMA GEMM, no K split; uses p[].block_begin and total_blocks only
constexpr int GEMM_TILE_MIN_T = 193;                     // measured crossover (V6-3B Int8): up to 192 rows the decode kernel's 64-row passes win or tie
constexpr int GEMM_TILE_SHAPES = 13;                      // 256x128, 128x128, 64x128, 64x64 (rows x tokens, 128-k chunks); 64x64 and 128x128 with 256-k chunks
int gemm_tile_blocks(int shape, int rows, int T);
constexpr int GEMM_TILE3 = 10;                            // the pipelined 128x128 kernel (non-hi/lo operands, K % 128 == 0)
constexpr int GEMM_TILE3_64 = 11;                         // the same pipeline on 128 rows x 64 tokens (steps of a few hundred rows)
constexpr int GEMM_TILE4_HILO = 12;                       // the software-pipelined kernel for hi + lo operands on 128 rows x 64 tokens (round 6; K % 128 == 0)
inline bool gemm_tile_pipelined(int shape) { return shape >= GEMM_TILE3 && shape <= GEMM_TILE4_HILO; }
bool gemm_tile3_supported(bool hilo, int K);
bool gemm_tile4_supported(bool hilo, int K);
inline bool gemm_tile_shape_supported(int shape, bool hilo, int K) {
    if (shape == 5 && hilo) return false;                // 128 tokens x 256-k chunks, double-buffered, hi + lo: 256 KiB of LDS
    return shape == GEMM_TILE4_HILO ? gemm_tile4_supported(hilo, K) : (shape >= GEMM_TILE3 ? gemm_tile3_supported(hilo, K) : true);
}
void launch_gemm_tile45(const GemmLaunch &L, int kind, int ntl, bool hilo, hipStream_t s);   // rwkv_kernels.hip part 5
void launch_gemm_tile(const GemmLaunch &L, int shape, bool hilo, hipStream_t s);                             // rounds of 256 k a wave can hold at once (single-shot)

// V6 fused time-mix LoRA (tanh(W1 z) -> W2 -> lerp), decode-shaped steps only
struct V6MixArgs {
    const void *W1;                 // tiled fp16 [5*Dm x C]
    const void *W2[5];              // tiled fp16 [C x Dm] each, order (w,k,v,r,g)
    const _Float16 *zhi, *zlo;      // operand z = xx + dx*mu_x  [T][ldz]
    int ldz;
    const float *xx, *dx;           // fp32 [T][C]
    const float *mu[5];
    _Float16 *ohi[5], *olo[5];      // outputs: the five GEMM operands [T][ldh]
    int ldh, T, C, Dm;
    LnProArgs lnp;                  // lnp.x_in set: LayerNorm + shift computed in the kernel (z, xx, dx unused)
    const float *mu_x;              // with lnp: z = xx + dx * mu_x
    _Float16 *mg_hi = nullptr, *mg_lo = nullptr;   // scratch for the two-launch form (>= 512 rows): m_c tiles [5][T/32][32][Dm]
    float *mp = nullptr;                           // fp32 partials of phase 1 sliced over K (192 .. 511 rows): [ksp][5][T][Dm]
    int ksp = 1, ksp_min_t = 192, ksp_blocks = 160, ksp_max = 16;   // slices (set by launch_v6_mix); rule parameters (engine knobs)
};
bool v6_mix_supported(int T, int C, int Dm);
bool v6_mix_wide_supported(int T, int C, int Dm);                   // steps with more than 32 rows: block = (mix, 32-token tile), all strips
bool v6_mix_ln_supported(int T, int C, int Dm, bool hilo, int np);   // the LayerNorm-prologue form (lnp set)
int v6_mix_split(const V6MixArgs &a, bool hilo);      // 0: one launch; n >= 1: phase 1 (in n K slices) + v6_mix_apply_kernel
void launch_v6_mix(const V6MixArgs &a, bool hilo, hipStream_t s);


struct LnShiftArgs {
    const float *x_in;
    float *x_out;                   // x_in + sum of partials (ping-pong residual stream)
    const float *P;                 // partials [np][pstride]
    int np;
    long pstride;
    const float *lnw, *lnb;
    float *sx;                      // token-shift state for this layer: sx + slot*sx_slot_stride
    long sx_slot_stride;
    RowMeta rm;
    int mode;                       // 0: V5  op = xx*mu + prev*(1-mu);  1: V6/V7 op = xx + (prev-xx)*mu
    int nmix;
    const float *mu[6];
    _Float16 *ohi[6];
    _Float16 *olo[6];
    int ldh;
    float *xx_out, *dx_out;         // optional fp32 copies (V6 time-mix LoRA epilogue needs them)
    int C;
    int xcd_rows = 0;               // set by launch_ln_shift on prefill-shaped steps: an XCD's blocks own a contiguous run of rows
};
void launch_ln_shift(const LnShiftArgs &a, int T, hipStream_t s);

struct EmbedArgs {
    const _Float16 *emb;            // raw [V][C]
    const float *lnw, *lnb;         // ln0
    const int *token;
    float *x;
    int C, V;
};
void launch_embed(const EmbedArgs &a, int T, hipStream_t s);

struct LnOutArgs {
    const float *x_in;
    const float *P;
    int np;
    long pstride;
    const float *lnw, *lnb;
    const int *out_rows;            // rows to emit (compacted)
    _Float16 *ohi, *olo;
    int ldh, C;
};
void launch_ln_out(const LnOutArgs &a, int n_out, hipStream_t s);

struct WkvArgs {
    int version;                    // 5, 6, 7
    int H, C;
    int n_seq;                      // active slots in this step
    const int *seq_slot;            // [n_seq]
    const int *seq_begin;           // [n_seq] first row
    const int *seq_len;             // [n_seq]
    int dense;                      // 1: sequence i = slot i = row i, one row each (seq_* are not read)
    float *state;                   // internal WKV state of this layer: state + slot*slot_stride + h*4096
    long slot_stride;
    const float *r, *k, *v;         // [T][C]
    const float *g;                 // [T][C] gate (v5/v6: silu applied; v7: g)
    // v5: wdec = precomputed exp(-exp(time_decay)) [C]; v6: time_decay [C] + td [T][Dd] + D2 fp16 [C][Dd]
    const float *wdec_or_decay;
    const float *u;                 // time_first [C] (v5/v6)
    const float *td;
    const _Float16 *D2;
    int Dd;
    // v7
    const float *w7, *a7, *vg7;     // [T][C]: decay, a, value-gate
    const float *k_k, *k_a, *r_k;   // [C]
    float *v_first;                 // [T][C]; layer 0 writes, others read
    int layer;
    const float *lnx_w, *lnx_b;
    _Float16 *yhi, *ylo;
    int ldh;
    int max_rows;                   // most rows any sequence has in this step (0: unknown): <= 8 selects the short-chunk form of wkv_chunk_kernel
};
void launch_wkv(const WkvArgs &a, bool multi_row, hipStream_t s);   // multi_row: some sequence has > 1 row in this step

// ---- state slab <-> internal layout ---------------------------------------------------
struct StatePackArgs {
    float *slab;                    // public [L][N+2][C]
    float *sxa, *sxf;               // internal [L][C] of this slot
    float *wkv;                     // internal [L][H][64][64] of this slot (T[p=value][q=key])
    int L, C, H;
    int transposed;                 // 1 for v5/v6 (public S[i=key][j=value]), 0 for v7
    int to_slab;                    // 1: internal -> slab, 0: slab -> internal
    int layer_only;                 // >=0: only this layer's WKV rows, slab points at [N][C]
};
void launch_state_pack(const StatePackArgs &a, hipStream_t s);

void launch_softmax(const float *in, float *out, int n_rows, int V, hipStream_t s);
// on-device sampling front-end (f-1): V <= 65536, top_k <= 256
struct SampleRow { float top_p; int top_k; float temperature; float uniform; int kind; float tau; };   // kind 0 nucleus, 1 typical, 2 mirostat (tau = max_surprise)
void launch_logit_adjust(float *logits, int V, const int *rows, const int *toks, const float *vals, int n, hipStream_t s);
void launch_logit_mask(float *logits, int V, const int *rows, const unsigned char *allow, int n, hipStream_t s);   // V % 4 == 0
void launch_nucleus(const float *logits, int n_rows, int V, const SampleRow *sp, bool any_nucleus_typical, bool any_mirostat,
                    int *out_tok, float *out_prob, hipStream_t s);
// two-stage arg-max; scratch_v / scratch_i hold n_rows*32 partial (value, index) pairs
void launch_argmax(const float *logits, int n_rows, int V, int *out_tok, float *scratch_v, int *scratch_i, hipStream_t s);

// ---- load-time: raw fp16 [rows][K] -> tiled / quantised --------------------------------
void launch_tile_f16(const _Float16 *raw, int rows_valid, int rows, int K, void *out, hipStream_t s);
void launch_quant_int8(const _Float16 *raw, int rows, int K, void *out, void *scales, hipStream_t s);
void launch_quant_nf4(const _Float16 *raw, int rows, int K, void *out, void *scales, hipStream_t s);
void launch_f16_to_f32(const _Float16 *in, float *out, long n, int op, hipStream_t s); // op 0: copy, 1: exp(-exp(x))
void launch_vec_blend(float *v, const _Float16 *l, long n, float alpha, hipStream_t s);   // v = alpha * l + (1 - alpha) * v   (LoRA file holds a whole vector tensor)
void launch_vec_op(float *v, long n, int op, hipStream_t s);                              // the load-time transform, in place
// W[rows][K] (fp16 raw) += alpha * B[rows][r] * A^T  (A stored [K][r]) — LoRA blend, fp32 math
void launch_lora_blend(_Float16 *W, const _Float16 *B, const _Float16 *A, int rows, int K, int r, float alpha, hipStream_t s);

}  // namespace rwkv
