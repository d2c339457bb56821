// rwkv_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels for the RWKV V5/V6/V7 forward pass.
//
// Reference arithmetic: the web-rwkv calls behind `runtime.infer` (crates/ai00-core/src/run.rs:1143),
// restated in SURVEY.md Appendix A.  Kernel inventory:
//   gemm_kernel      K4-K7,K12-K15  skinny MFMA GEMM over pre-tiled fp16/int8/nf4 weights, fused epilogues
//   ln_shift_kernel  K2,K3,K14      residual-partials add + LayerNorm + token shift + operand emit
//   embed_kernel     K1             emb gather + ln0
//   ln_out_kernel    K2             final LayerNorm on the rows whose logits are requested
//   wkv_kernel       K8-K11,K13     WKV recurrence (+v6 decay LoRA stage 2, +v7 kappa/a/v-mix), GroupNorm, gate
//   state_pack       K17            public slab [C,N+2,L,1] <-> internal state layout
//   softmax / argmax K16            batched over the vocabulary
//   tile/quant       K18,K19        load-time: raw fp16 -> tiled fp16 / int8 / nf4 ; LoRA blend
//
// gfx950 only: wave = 64 lanes, v_mfma_f32_16x16x32_f16, no portability shims.
#include "rwkv_kernels.h"
#include <type_traits>
#include <cstdlib>
#include <algorithm>


// Build parts: the product build compiles this file once per part (-DRWKV_PART=k, k = 0..4, in parallel: one pass takes
// ~3 minutes, the parts ~1); without RWKV_PART everything is one translation unit (trace builds of scripts/build_variant.py,
// whose `__device__` probe buffers cannot span translation units).
#ifdef RWKV_PART
#define RWKV_PART_ON(k) (RWKV_PART == (k))
#else
#define RWKV_PART_ON(k) 1
#endif
#if defined(RWKV_TRACE) && defined(RWKV_PART)
#error "trace builds are single-part"
#endif

namespace rwkv {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float apply_act(int act, float v) {
    switch (act) {
        case ACT_TANH: return tanhf(v);
        case ACT_SIGMOID: return sigmoidf_(v);
        case ACT_RELU2: { float r = fmaxf(v, 0.0f); return r * r; }
        case ACT_SILU: return v * sigmoidf_(v);
        case ACT_DECAY7: return expf(-0.606531f * sigmoidf_(v));
        default: return v;
    }
}

// the switch outside, four values inside: one pass over the branches per output fragment instead of one per element (same arithmetic)
__device__ __forceinline__ void apply_act4(int act, float (&v)[4]) {
    switch (act) {
        case ACT_NONE: break;
        case ACT_TANH: for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]); break;
        case ACT_SIGMOID: for (int r = 0; r < 4; ++r) v[r] = sigmoidf_(v[r]); break;
        case ACT_RELU2: for (int r = 0; r < 4; ++r) { const float q = fmaxf(v[r], 0.0f); v[r] = q * q; } break;
        case ACT_SILU: for (int r = 0; r < 4; ++r) v[r] = v[r] * sigmoidf_(v[r]); break;
        case ACT_DECAY7: for (int r = 0; r < 4; ++r) v[r] = expf(-0.606531f * sigmoidf_(v[r])); break;
        default: break;
    }
}
// fp32 -> (hi, lo) f16 pair: hi = rn(v) saturated, lo = rn(v - hi).  hi+lo carries ~22 mantissa bits.
__device__ __forceinline__ void split_hilo(float v, _Float16 &hi, _Float16 &lo) {
    float c = fminf(fmaxf(v, -65504.0f), 65504.0f);
    hi = (_Float16)c;
    lo = (_Float16)(c - (float)hi);
}

// Activation operands (the f16 hi/lo X of every GEMM) live in HBM in MFMA B-FRAGMENT order, not row-major:
// 1 KiB tiles of (16 tokens x 32 k); inside a tile lane l = ((k>>3)&3)*16 + (t&15) owns 8 consecutive k (16 B), i.e.
// exactly the register image of `v_mfma_f32_16x16x32_f16`'s B operand.  A wave's fragment load is then ONE contiguous
// 1 KiB read (8 full cache lines) instead of 16 scattered 64 B segments — measured 2.5-3 us per GEMM at T = 32
// (scripts/trace_gemm.py).  `ld` (k per token row) is a multiple of 32; token capacity is a multiple of 16.
__device__ __forceinline__ long opd_off(int t, int k, int ld) {
    return ((long)(t >> 4) * (ld >> 5) + (k >> 5)) * 512 + ((((k >> 3) & 3) << 4) + (t & 15)) * 8 + (k & 7);
}

__device__ __forceinline__ void store_operand4(_Float16 *__restrict__ hi, _Float16 *__restrict__ lo, long off, float4 o) {
    f16x4 h, l;
    _Float16 a, b;
    split_hilo(o.x, a, b); h[0] = a; l[0] = b;
    split_hilo(o.y, a, b); h[1] = a; l[1] = b;
    split_hilo(o.z, a, b); h[2] = a; l[2] = b;
    split_hilo(o.w, a, b); h[3] = a; l[3] = b;
    *(f16x4 *)(hi + off) = h;
    if (lo) *(f16x4 *)(lo + off) = l;
}

// =====================================================================================
// Activation accessors.  Everything a kernel of the step hands to a later kernel (residual rows, fp32 projections, f16
// operands) is addressed through a buffer descriptor (4 SGPRs per array) + a 32-bit byte offset per lane: half the address
// VGPRs of 64-bit global pointers in kernels whose register budget is spent on weight tiles and X fragments, and one place
// to set the cache policy of hand-over data.  Policy 0 = default (L1 + L2).  A write-through / L1-bypassing policy (sc1 on
// both sides) with in-kernel completion counters was built to overlap consecutive launches on two streams and measured:
// the polling of a few hundred waiting workgroups on one counter costs more than the launch boundary it replaces
// (scripts/chain_bench.hip: 10.8-30 us per dependent phase against 6.2 us for plain graph launches), so launches stay
// stream-ordered.  (Round 6, stream-ordered launches as they are: write-through stores — sc0 sc1, nothing left for the end-of-kernel L2 write-back —
// cost +1 % per decode step at 32 slots and +2 % at one; non-temporal stores +7 % / +1 %: profiles/r6_exp_activation_store_policy.log.  Policy 0 stays.)
// =====================================================================================
typedef __amdgpu_buffer_rsrc_t act_t;                              // buffer descriptor of one activation array (4 SGPRs)
constexpr int ACT_SC1 = 0;                                        // cache policy of activation accesses (0: default; 16 would be sc1)
__device__ __forceinline__ act_t act_buf(const void *p) {         // p must be wave-uniform (kernel argument arithmetic)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, -1, 0x00020000);
}
__device__ __forceinline__ float4 act_ld4(act_t b, long fidx) {   // 4 floats at float index fidx (16-byte aligned)
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(b, (unsigned)(fidx * 4), 0, ACT_SC1));
}
__device__ __forceinline__ float act_ld1(act_t b, long fidx) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b, (unsigned)(fidx * 4), 0, ACT_SC1));
}
__device__ __forceinline__ f16x8 act_ldh8(act_t b, long hidx) {   // 8 halfs at half index hidx (16-byte aligned)
    return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(b, (unsigned)(hidx * 2), 0, ACT_SC1));
}
__device__ __forceinline__ void act_st4(act_t b, long fidx, float4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), b, (unsigned)(fidx * 4), 0, ACT_SC1);
}
__device__ __forceinline__ void act_st1(act_t b, long fidx, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(u32, v), b, (unsigned)(fidx * 4), 0, ACT_SC1);
}
__device__ __forceinline__ void act_sth4(act_t b, long hidx, f16x4 v) {   // 4 halfs (8 bytes)
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), b, (unsigned)(hidx * 2), 0, ACT_SC1);
}
// operand emit (hi[, lo] f16 pair of four fp32 values) into an activation operand
__device__ __forceinline__ void act_store_operand4(act_t hi, act_t lo, bool has_lo, long off, float4 o) {
    f16x4 h, l;
    _Float16 a, b;
    split_hilo(o.x, a, b); h[0] = a; l[0] = b;
    split_hilo(o.y, a, b); h[1] = a; l[1] = b;
    split_hilo(o.z, a, b); h[2] = a; l[2] = b;
    split_hilo(o.w, a, b); h[3] = a; l[3] = b;
    act_sth4(hi, off, h);
    if (has_lo) act_sth4(lo, off, l);
}

// Cross-lane sums on the DPP path (v_add_f32 with a dpp source modifier, a few cycles each) instead of `__shfl_xor`,
// which hipcc lowers to ds_bpermute_b32 — an LDS-crossbar round trip per step, and these reductions sit on the
// per-token critical path of the WKV recurrence and of every LayerNorm.  All lanes of the wave must be active.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum over the 16 lanes of a DPP row (lanes 16r .. 16r+15); every lane of the row gets the result
__device__ __forceinline__ float row_sum16(float v) {
    v += dpp_f32<0xB1>(v);                                   // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);                                   // quad_perm [2,3,0,1]
    v += dpp_f32<0x124>(v);                                  // row_ror:4
    v += dpp_f32<0x128>(v);                                  // row_ror:8
    return v;
}
// sum over the 4 lanes of a quad; every lane of the quad gets the result
__device__ __forceinline__ float quad_sum(float v) {
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = row_sum16(v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)) +
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32)) +
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48)));
}

// block-wide sum over 256 threads; `red` is >= 4 floats of LDS.  All threads get the result.
__device__ __forceinline__ float block_sum256(float v, float *red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// =====================================================================================
// GEMM: out[t][row] = epi( sum_k W[row][k] * X[t][k] ),  W pre-tiled, X = f16 (hi[,lo]) operand
//
// K-stationary skinny GEMM.  A block owns `spb` strips of 16 output rows and one K range; wave w of
// the block owns k in [kbeg + w*KW, +KW), KW = KSW*32.  Each wave loads its slice of X ONCE, straight
// from L2 into MFMA B-fragment registers (rows beyond T clamp to the last row, so the traffic is T*64 B
// per k-step), then streams the 1 KiB weight tiles of strip after strip from HBM (non-temporal; the
// next strip's tiles are in flight while the current strip is multiplied).  The per-wave partial
// accumulators are parked in LDS and reduced after ONE barrier; the reducing wave applies the epilogue.
// No X staging, no chunk loop: every load of the kernel is issued within the first few hundred cycles.
// =====================================================================================
template <int FMT> struct Fmt;
template <> struct Fmt<W_F16> { static constexpr int TK = 32, KS = 1, SH = 5; };
template <> struct Fmt<W_INT8> { static constexpr int TK = 64, KS = 2, SH = 6; };
template <> struct Fmt<W_NF4> { static constexpr int TK = 128, KS = 4, SH = 7; };

#ifdef RWKV_TRACE      // dev-only timeline probes (scripts/trace_gemm.py); never defined in the product build
__device__ unsigned long long g_trace[4096 * 8];
#define TRACE_PT(i) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 256) g_trace[((blockIdx.x * 16 + (threadIdx.x >> 6)) << 3) + (i)] = wall_clock64(); } while (0)
extern "C" int rwkv_debug_trace(unsigned long long *out, int n) {
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), (size_t)n * 8, 0, hipMemcpyDeviceToHost);
    static unsigned long long zeros[4096 * 8];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace), zeros, sizeof(zeros), 0, hipMemcpyHostToDevice);
    return rc;
}
__device__ unsigned long long g_trace2[4 * 2048 * 8];   // [kernel id][linear block][probe], wave 0 only
#define TRACE_K(kid, i) do { const unsigned lb_ = blockIdx.x + gridDim.x * blockIdx.y; if (threadIdx.x == 0 && lb_ < 2048) g_trace2[(((kid) * 2048 + lb_) << 3) + (i)] = wall_clock64(); } while (0)
extern "C" int rwkv_debug_trace2(unsigned long long *out) {
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace2), sizeof(g_trace2), 0, hipMemcpyDeviceToHost);
    return rc;
}
#else
#define TRACE_PT(i) do { } while (0)
#define TRACE_K(kid, i) do { } while (0)
#endif

constexpr int RS = 8;                                   // k-steps (of 32) per register round = 256 k
template <int FMT> struct WRound {                      // one strip's tiles for 256 k of this wave's K slice
    u32x4 q[RS / Fmt<FMT>::KS];
    uint2 s;                                            // quant scales of the 256-k group (unused for fp16)
};

__device__ __forceinline__ f16x2 as_h2(u32 v) { return __builtin_bit_cast(f16x2, v); }
__device__ __forceinline__ u32 as_u32(f16x2 v) { return __builtin_bit_cast(u32, v); }

// TAIL = false: the caller guarantees the whole 256-k round lies inside [.., kend) -> branch-free
template <int FMT, bool TAIL>
__device__ __forceinline__ void load_round(WRound<FMT> &w, const GemmProb &P, int strip, int k0, int kend, bool ok, int lane) {
    constexpr int NTILE = RS / Fmt<FMT>::KS, TK = Fmt<FMT>::TK, SH = Fmt<FMT>::SH;
    const int KT = P.K >> SH;
    const u32x4 *base = (const u32x4 *)P.W + ((long)strip * KT + (k0 >> SH)) * 64 + lane;
#pragma unroll
    for (int j = 0; j < NTILE; ++j) {
        if (!TAIL || (ok && k0 + j * TK < kend)) w.q[j] = __builtin_nontemporal_load(base + j * 64);
        else w.q[j] = (u32x4){0u, 0u, 0u, 0u};
    }
    if constexpr (FMT != W_F16) {
        const int NG = P.K >> 8;
        const uint2 *sb = (const uint2 *)P.S + ((long)strip * NG + (k0 >> 8)) * 16 + (lane & 15);
        w.s = (!TAIL || (ok && k0 < kend)) ? *sb : make_uint2(0, 0);
    } else {
        w.s = make_uint2(0, 0);
    }
}

// int8: two bytes of `d` (selected by `sel`) -> half2 of a*q+b, one rounding (v_pk_fma_f16)
__device__ __forceinline__ u32 dq8(u32 d, u32 sel, f16x2 a2, f16x2 b2) {
    u32 p = __builtin_amdgcn_perm(0x64646464u, d, sel);           // bytes -> 0x6400|q == 1024+q exactly
    f16x2 h = as_h2(p) - (f16x2){(_Float16)1024.0f, (_Float16)1024.0f};
    return as_u32(__builtin_elementwise_fma(h, a2, b2));
}

// NF4 code points rounded to fp16 (oracle: NF4_TABLE_F16), as byte tables for v_perm lookups.
static __device__ __constant__ unsigned short nf4_f16_bits[16] = {
    0xBC00, 0xB992, 0xB833, 0xB652, 0xB48D, 0xB1EA, 0xADD4, 0x0000,
    0x2D18, 0x3126, 0x33E0, 0x3568, 0x370D, 0x3880, 0x39C9, 0x3C00};

struct Nf4Lut { u32 tl[4], th[4]; };
__device__ __forceinline__ Nf4Lut make_nf4_lut() {
    Nf4Lut t;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        u32 lo = 0, hi = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            u32 v = nf4_f16_bits[d * 4 + b];
            lo |= (v & 0xFF) << (8 * b);
            hi |= (v >> 8) << (8 * b);
        }
        t.tl[d] = lo; t.th[d] = hi;
    }
    return t;
}
// 4 nibble indices (one per byte of n) -> 4 fp16 code points as two half2 words
__device__ __forceinline__ void nf4_lookup4(u32 n, const Nf4Lut &t, u32 &h01, u32 &h23) {
    u32 sel = n & 0x07070707u;
    u32 mask = ((n >> 3) & 0x01010101u) * 0xFFu;
    u32 la = __builtin_amdgcn_perm(t.tl[1], t.tl[0], sel);
    u32 lb = __builtin_amdgcn_perm(t.tl[3], t.tl[2], sel);
    u32 ha = __builtin_amdgcn_perm(t.th[1], t.th[0], sel);
    u32 hb = __builtin_amdgcn_perm(t.th[3], t.th[2], sel);
    u32 L = (la & ~mask) | (lb & mask);
    u32 Hh = (ha & ~mask) | (hb & mask);
    h01 = __builtin_amdgcn_perm(Hh, L, 0x05010400u);
    h23 = __builtin_amdgcn_perm(Hh, L, 0x07030602u);
}

// MFMA A fragment (16 rows x 32 k) of k-step `ks` (0..RS) of a round
template <int FMT>
__device__ __forceinline__ f16x8 frag(const WRound<FMT> &w, int ks, const Nf4Lut &lut) {
    if constexpr (FMT == W_F16) {
        return __builtin_bit_cast(f16x8, w.q[ks]);
    } else if constexpr (FMT == W_INT8) {
        const u32x4 q = w.q[ks >> 1];
        const u32 d0 = (ks & 1) ? q.z : q.x, d1 = (ks & 1) ? q.w : q.y;
        const u32 ab = (ks >> 2) ? w.s.y : w.s.x;                // 128-block inside the 256-group
        const f16x2 abh = as_h2(ab);
        const f16x2 a2 = {abh[0], abh[0]}, b2 = {abh[1], abh[1]};
        u32x4 r;
        r.x = dq8(d0, 0x04010400u, a2, b2);
        r.y = dq8(d0, 0x04030402u, a2, b2);
        r.z = dq8(d1, 0x04010400u, a2, b2);
        r.w = dq8(d1, 0x04030402u, a2, b2);
        return __builtin_bit_cast(f16x8, r);
    } else {
        const u32x4 q = w.q[ks >> 2];
        const int wsel = ks & 3;
        const u32 d = wsel == 0 ? q.x : wsel == 1 ? q.y : wsel == 2 ? q.z : q.w;
        const u32 sw = (ks >> 2) ? w.s.y : w.s.x;                // 64-blocks 2*(ks>>2) + ((ks>>1)&1)
        const f16x2 sh = as_h2(sw);
        const _Float16 am = ((ks >> 1) & 1) ? sh[1] : sh[0];
        const f16x2 am2 = {am, am};
        u32x4 r;
        u32 a, b;
        nf4_lookup4(d & 0x0F0F0F0Fu, lut, a, b);
        r.x = as_u32(as_h2(a) * am2);
        r.y = as_u32(as_h2(b) * am2);
        nf4_lookup4((d >> 4) & 0x0F0F0F0Fu, lut, a, b);
        r.z = as_u32(as_h2(a) * am2);
        r.w = as_u32(as_h2(b) * am2);
        return __builtin_bit_cast(f16x8, r);
    }
}

// =====================================================================================
// LayerNorm + token-shift prologue (LnProArgs): the row work of ln_shift_kernel redone by EVERY block of a GEMM-like
// launch for all T <= LNP_MAX_T rows.  Leaves in LDS (row stride C + LNP_PAD floats):
//   xx_l[t][c]  normalised rows,  pv_l[t][c]  the slots' shift states,
//   op_l        the calling block's GEMM operand mix(xx, prev, mu) as f16 (hi[, lo]) in MFMA B-fragment order.
// Same arithmetic as ln_shift_kernel (slab sum in slab order, two-pass LayerNorm); the cross-wave sums run over
// blockDim/64 waves instead of 4, so results may differ from the unfused path in the last bit.
// Host guarantees C <= 8 * blockDim.x, np <= LNP_MAX_NP, T <= LNP_MAX_T.  All threads of the block must call it.
// =====================================================================================
constexpr int LNP_PAD = 4;
__device__ __forceinline__ float4 lnp_ld4(const float *p) { return *(const float4 *)p; }
__device__ __forceinline__ float lnp_mix1(int mode, float x, float p, float m) {
    return mode == 0 ? x * m + p * (1.0f - m) : x + (p - x) * m;
}
constexpr int LNP_PTN = 2;
struct LnCarry { int prev[LNP_MAX_T]; };
// part 1: every global load of the prologue; residual sums and shift states land in LDS, per-wave partial sums in `red`
__device__ __forceinline__ void ln_prologue_load(const LnProArgs &a, int T, float *xx_l, float *pv_l, float *red,
                                                 bool publish, LnCarry &k) {
    constexpr int PTN = LNP_PTN;
    const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const int C = a.C, ldl = C + LNP_PAD;
    int (&prev)[LNP_MAX_T] = k.prev;
    const act_t bx = act_buf(a.x_in), bP = act_buf(a.P), bsx = act_buf(a.sx), bxo = act_buf(a.x_out);
    // ---- pass 1, row by row (registers: one row's slabs in flight): residual sum -> LDS, state -> LDS, partial sums
#pragma unroll
    for (int t = 0; t < LNP_MAX_T; ++t) {
        if (t < T) {
            const int slot = a.rm.dense ? t : a.rm.slot[t];
            prev[t] = a.rm.dense ? -1 : a.rm.prev[t];
            float4 v[PTN], sxv[PTN], pp[LNP_MAX_NP][PTN];
#pragma unroll
            for (int i = 0; i < PTN; ++i) {
                const int c = (tid + i * nth) * 4;
                if (c < C) {
                    v[i] = act_ld4(bx, (long)t * C + c);
#pragma unroll
                    for (int j = 0; j < LNP_MAX_NP; ++j)       // branch-free: slabs beyond np re-read slab 0 and are dropped below
                        pp[j][i] = act_ld4(bP, (j < a.np ? j : 0) * a.pstride + (long)t * C + c);
                    sxv[i] = act_ld4(bsx, (long)slot * a.sx_slot_stride + c);
                }
            }
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < PTN; ++i) {
                const int c = (tid + i * nth) * 4;
                if (c < C) {
#pragma unroll
                    for (int j = 0; j < LNP_MAX_NP; ++j) {
                        const bool on = j < a.np;              // select, not multiply: an unused slab may hold anything
                        v[i].x += on ? pp[j][i].x : 0.f; v[i].y += on ? pp[j][i].y : 0.f;
                        v[i].z += on ? pp[j][i].z : 0.f; v[i].w += on ? pp[j][i].w : 0.f;
                    }
                    if (publish) act_st4(bxo, (long)t * C + c, v[i]);
                    *(float4 *)(xx_l + t * ldl + c) = v[i];
                    *(float4 *)(pv_l + t * ldl + c) = sxv[i];
                    sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
                }
            }
            sum = wave_sum(sum);
            if (lane == 0) red[t * 16 + wave] = sum;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
// part 2: reductions, normalisation, operand emit (LDS only, four barriers)
template <bool HILO>
__device__ __forceinline__ void ln_prologue_finish(const LnProArgs &a, int T, const float *mu, float *xx_l, float *pv_l, _Float16 *op_l,
                                                   float *red, bool publish, const LnCarry &k) {
    constexpr int PTN = LNP_PTN;
    const int tid = threadIdx.x, nth = blockDim.x, nwv = nth >> 6, lane = tid & 63, wave = tid >> 6;
    const int C = a.C, ldl = C + LNP_PAD;
    const int (&prev)[LNP_MAX_T] = k.prev;
    const act_t bxxo = act_buf(a.xx_out);
    float4 wv[PTN], bv[PTN], muv[PTN];                           // L2-hot parameters: in flight across the first barrier
#pragma unroll
    for (int i = 0; i < PTN; ++i) {
        const int c = (tid + i * nth) * 4;
        if (c < C) { wv[i] = lnp_ld4(a.lnw + c); bv[i] = lnp_ld4(a.lnb + c); muv[i] = lnp_ld4(mu + c); }
    }
    __syncthreads();
    // ---- pass 2: centred variance (each thread re-reads its own elements from LDS)
    float mean[LNP_MAX_T];
#pragma unroll
    for (int t = 0; t < LNP_MAX_T; ++t) {
        mean[t] = 0.f;
        if (t < T) {
            float m = 0.f, q = 0.f;
            for (int w2 = 0; w2 < nwv; ++w2) m += red[t * 16 + w2];
            mean[t] = m / (float)C;
#pragma unroll
            for (int i = 0; i < PTN; ++i) {
                const int c = (tid + i * nth) * 4;
                if (c < C) {
                    const float4 v = *(const float4 *)(xx_l + t * ldl + c);
                    const float d0 = v.x - mean[t], d1 = v.y - mean[t], d2 = v.z - mean[t], d3 = v.w - mean[t];
                    q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                }
            }
            q = wave_sum(q);
            if (lane == 0) red[32 + t * 16 + wave] = q;
        }
    }
    __syncthreads();
    // ---- pass 3: normalise in place
#pragma unroll
    for (int t = 0; t < LNP_MAX_T; ++t) {
        if (t < T) {
            float m = 0.f;
            for (int w2 = 0; w2 < nwv; ++w2) m += red[32 + t * 16 + w2];
            const float rstd = 1.0f / sqrtf(m / (float)C + 1e-5f);
#pragma unroll
            for (int i = 0; i < PTN; ++i) {
                const int c = (tid + i * nth) * 4;
                if (c < C) {
                    const float4 v = *(const float4 *)(xx_l + t * ldl + c);
                    float4 o;
                    o.x = (v.x - mean[t]) * rstd * wv[i].x + bv[i].x;
                    o.y = (v.y - mean[t]) * rstd * wv[i].y + bv[i].y;
                    o.z = (v.z - mean[t]) * rstd * wv[i].z + bv[i].z;
                    o.w = (v.w - mean[t]) * rstd * wv[i].w + bv[i].w;
                    *(float4 *)(xx_l + t * ldl + c) = o;
                    if (publish) act_st4(bxxo, (long)t * C + c, o);
                }
            }
        }
    }
    __syncthreads();
    // ---- pass 4: this block's GEMM operand  op = mix(xx, prev, mu)  as f16 in MFMA B-fragment order, T rows per tile:
    //      halfs [(k>>5)*4 + ((k>>3)&3)][t][k&7]  (+ the lo parts behind them).  A row that follows another row of its slot
    //      inside this step shifts from that row, otherwise from the state.
    _Float16 *op_lo = op_l + (size_t)T * C;
#pragma unroll
    for (int t = 0; t < LNP_MAX_T; ++t) {
        if (t < T) {
#pragma unroll
            for (int i = 0; i < PTN; ++i) {
                const int c = (tid + i * nth) * 4;
                if (c < C) {
                    const float4 x = *(const float4 *)(xx_l + t * ldl + c);
                    const float4 pvv = prev[t] >= 0 ? *(const float4 *)(xx_l + prev[t] * ldl + c) : *(const float4 *)(pv_l + t * ldl + c);
                    const float o0 = lnp_mix1(a.mode, x.x, pvv.x, muv[i].x), o1 = lnp_mix1(a.mode, x.y, pvv.y, muv[i].y);
                    const float o2 = lnp_mix1(a.mode, x.z, pvv.z, muv[i].z), o3 = lnp_mix1(a.mode, x.w, pvv.w, muv[i].w);
                    f16x4 hh, ll;
                    _Float16 h, l;
                    split_hilo(o0, h, l); hh[0] = h; ll[0] = l;
                    split_hilo(o1, h, l); hh[1] = h; ll[1] = l;
                    split_hilo(o2, h, l); hh[2] = h; ll[2] = l;
                    split_hilo(o3, h, l); hh[3] = h; ll[3] = l;
                    const int off = ((((c >> 5) << 2) + ((c >> 3) & 3)) * T + t) * 8 + (c & 7);
                    *(f16x4 *)(op_l + off) = hh;
                    if constexpr (HILO) *(f16x4 *)(op_lo + off) = ll;
                }
            }
        }
    }
    __syncthreads();
}
__device__ __forceinline__ size_t lnp_op_off(int T, int k, int tl) { return (size_t)((k >> 3) * T + tl) * 8; }   // k multiple of 8
#if RWKV_PART_ON(0)
static int knob_env(const char *name, int dflt) {
    const char *v = std::getenv(name);
    return (v && *v) ? std::atoi(v) : dflt;
}
Knobs Knobs::from_env() {
    Knobs k;
    k.no_ln_fuse = knob_env("RWKV_NO_LN_FUSE", 0); k.no_v6_fuse = knob_env("RWKV_NO_V6_FUSE", 0);
    k.no_tile = knob_env("RWKV_NO_TILE", 0); k.tile_shape = knob_env("RWKV_TILE_SHAPE", -1);
    k.no_dense = knob_env("RWKV_NO_DENSE", 0);
    k.tile_ksplit = knob_env("RWKV_TILE_KSPLIT", 1);
    k.promote = knob_env("RWKV_PROMOTE", -1);
    k.v6_ksp_max = knob_env("RWKV_V6_KSP_MAX", 16); k.v6_ksp_blocks = knob_env("RWKV_V6_KSP_BLOCKS", 160); k.v6_ksp_min_t = knob_env("RWKV_V6_KSP_MIN_T", 192);
    return k;
}
static thread_local Knobs t_knobs;
const Knobs &knobs() { return t_knobs; }
void use_knobs(const Knobs &k) { t_knobs = k; }

size_t lnp_lds_bytes(int T, int C, bool hilo) { return (size_t)(2 * T * (C + LNP_PAD) + 64) * 4 + (size_t)T * C * 2 * (hilo ? 2 : 1); }
#endif

// token-shift state commit of the previous launch's prologue (one extra block)
__device__ __forceinline__ void shift_commit(const ShiftCommit &c) {
    for (int t = 0; t < c.T; ++t) {
        const int last = c.rm.dense ? t : c.rm.last[t];
        if (last < 0) continue;
        const act_t src = act_buf(c.src);
        float *dst = c.sx + (long)(c.rm.dense ? t : c.rm.slot[t]) * c.sx_slot_stride;
        for (int i = threadIdx.x * 4; i < c.C; i += blockDim.x * 4) *(float4 *)(dst + i) = act_ld4(src, (long)last * c.C + i);
    }
}

#if RWKV_PART_ON(0)
// make a wave-uniform value live in SGPRs from this point on (an empty asm the compiler cannot look through)
__device__ __forceinline__ int pin_s(int v) { v = __builtin_amdgcn_readfirstlane(v); asm volatile("" : "+s"(v)); return v; }
__device__ __forceinline__ const void *pin_p(const void *p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = (unsigned)pin_s((int)(unsigned)v), hi = (unsigned)pin_s((int)(unsigned)(v >> 32));
    return (const void *)(((unsigned long long)hi << 32) | lo);
}
template <int NT, int KSW, bool HILO, bool SHOT, bool TAIL, int FMT>
__device__ __forceinline__ void gemm_body(const GemmLaunch &L, const GemmProb &P, unsigned char *smem) {
    constexpr int KW = KSW * 32, SUB = KSW / RS, RK = RS * 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform: scalar branches/addresses
    const int lb = (int)blockIdx.x - P.block_begin;
    // (a scalar integer division is ~40 SALU instructions in front of the first load: the host hands over Kb and nslice, and only
    // K-split problems divide the block index)
    const int kb = P.ksb == 1 ? 0 : lb / P.nblk_strip, sb = lb - kb * P.nblk_strip;
    const int spb = P.spb, nw = P.nw;                             // nw = waves of this block that own K slices
    const int strip0 = sb * spb;
    const int nstrip = min(spb, (P.rows >> 4) - strip0);
    const int Kb = P.Kb;                                          // K / ksb
    const int kbeg = kb * Kb, kend = kbeg + Kb;
    const int nslice = P.nslice;                                  // ceil(Kb / KW) >= nw; a wave takes slices wave, wave+nw, ...
    f32x4 *red = (f32x4 *)smem;                                   // [spb][nw][NT][64 lanes]

    Nf4Lut lut;
    if constexpr (FMT == W_NF4) lut = make_nf4_lut();

    // one 256-k round of MFMAs: acc += W(round) * X(sub).  Two accumulators per n-tile (even / odd k-steps) halve the
    // MFMA read-after-write stalls of the otherwise serial chain.
    auto mma_round = [&](const WRound<FMT> &w, f32x4 (&acc)[NT], f32x4 (&acc2)[NT], const f16x8 (&xb)[NT][KSW],
                         const f16x8 (*xl)[HILO ? KSW : 1], int sub, int k0) {
#pragma unroll
        for (int ks = 0; ks < RS; ++ks) {
            if (!TAIL || k0 + (sub * RS + ks) * 32 < kend) {
                const f16x8 a = frag<FMT>(w, ks, lut);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    f32x4 &d = (NT == 1 && (ks & 1)) ? acc2[nt] : acc[nt];   // NT = 2 already has two independent chains
                    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xb[nt][sub * RS + ks], d, 0, 0, 0);
                    if constexpr (HILO) d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xl[nt][sub * RS + ks], d, 0, 0, 0);
                }
            }
        }
    };

    // activation arrays of this problem (write-through / L1-bypassing accessors, see "Step-local activations")
    const act_t bxh = act_buf(P.xhi), bxl = act_buf(P.xlo);
    const act_t bo32 = act_buf(P.out_f32 ? P.out_f32 + (long)kb * P.partial_stride : nullptr);
    const act_t boh = act_buf(P.out_hi), bol = act_buf(P.out_lo), bm0 = act_buf(P.m0), bm1 = act_buf(P.m1);

    // rounds a wave holds in registers at once (gemm_max_rounds)
    constexpr int MAXR = (NT == 4 || (NT == 2 && HILO)) ? 2 : (FMT == W_F16 ? 2 : ((NT == 2 || HILO) ? 3 : 4));
    WRound<FMT> cur, nxt, w[SHOT ? MAXR : 1];
    // streamed (not single-shot) quantised weights: a ring of RD rounds in flight per wave.  With only cur/nxt
    // (one round ahead) a wave's K slice is a serial chain of memory latencies — 6 rounds x ~0.9 us at T = 1
    // (scripts/trace_gemm.py); fp16 rounds are twice the registers and stay at one round ahead.
    constexpr int RD = (!SHOT && FMT != W_F16) ? ((NT >= 2 || HILO) ? 2 : 4) : 1;
    WRound<FMT> ring[RD];
    // weights of one K slice of this wave: everything (single shot), the first RD rounds (ring) or the first round
    auto issue_w = [&](int k0, int nsub, int nround, bool ringed) {
        if constexpr (SHOT) {
            // single shot: every weight tile of this wave is in flight before the first MFMA (host: spb * SUB <= MAXR)
#pragma unroll
            for (int r = 0; r < MAXR; ++r) {
                const int s = r / SUB, sub = r % SUB;
                if (s < nstrip && sub < nsub) load_round<FMT, TAIL>(w[r], P, strip0 + s, k0 + sub * RK, kend, true, lane);
            }
        } else if (ringed) {
#pragma unroll
            for (int j = 0; j < RD; ++j)
                if (j < nround) load_round<FMT, TAIL>(ring[j], P, strip0 + j / SUB, k0 + (j % SUB) * RK, kend, true, lane);
        } else {
            load_round<FMT, TAIL>(cur, P, strip0, k0, kend, true, lane);
        }
    };

    TRACE_PT(0);
    for (int t0 = 0; t0 < L.T; t0 += NT * 16) {
        for (int sl = wave; sl < nslice && wave < nw; sl += nw) {
            const int k0 = kbeg + sl * KW;
            // rounds of this slice that lie inside the K range (the last slice may be short)
            const int nsub = TAIL ? SUB : min(SUB, (kend - k0) / RK);
            const int nround = nstrip * nsub;
            TRACE_PT(6);
            // X slice of this wave -> B fragments: one contiguous 1 KiB tile per (n-tile, k-step), see opd_off
            f16x8 xb[NT][KSW], xl[HILO ? NT : 1][HILO ? KSW : 1];
            const bool ringed = RD > 1 && nsub == SUB;
            // X first, weights after: the X fragments are L2 hits and complete first, so (in-order vmcnt) the MFMAs of round
            // r only wait for rounds <= r while later rounds are still streaming in from HBM.  (Weights-first was measured
            // and is slower: issuing is throttled by the CU's memory pipeline either way, and X then lands last.)
            auto load_x = [&]() {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    // token tile of this n-tile (clamped: results of tiles / lanes beyond T are never stored); lanes beyond the
                    // step's last token re-read its 16 B so a T = 1 step still moves 64 B per k-step, not 1 KiB
                    const int tile = min((t0 >> 4) + nt, (L.T - 1) >> 4);
                    const int tl = min(lane & 15, L.T - 1 - tile * 16);
                    const long xo = ((long)tile * (P.ldx >> 5) + (k0 >> 5)) * 512 + ((lane >> 4) * 16 + tl) * 8;
#pragma unroll
                    for (int sub = 0; sub < SUB; ++sub) {
                        if (TAIL || sub < nsub) {                      // one uniform branch per 256-k round
#pragma unroll
                            for (int k8 = 0; k8 < RS; ++k8) {
                                const int ks = sub * RS + k8;
                                const bool in = !TAIL || (k0 + ks * 32 < kend);
                                xb[nt][ks] = in ? act_ldh8(bxh, xo + ks * 512) : (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
                                if constexpr (HILO) xl[nt][ks] = in ? act_ldh8(bxl, xo + ks * 512) : (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
                            }
                        } else {
#pragma unroll
                            for (int k8 = 0; k8 < RS; ++k8) {
                                xb[nt][sub * RS + k8] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
                                if constexpr (HILO) xl[nt][sub * RS + k8] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
                            }
                        }
                    }
                }
            };
            load_x();
            TRACE_PT(7);
            issue_w(k0, nsub, nround, ringed);
#ifdef RWKV_TRACE
            TRACE_PT(1);
#if RWKV_TRACE >= 2
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            TRACE_PT(5);
#endif
#endif
            f32x4 acc[NT], acc2[NT];
            auto park = [&](int s) {                               // partial sums of strip s -> LDS slot of this wave
                f32x4 *slot = red + ((s * nw + wave) * NT) * 64 + lane;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const f32x4 v = acc[nt] + acc2[nt];
                    if (sl == wave) slot[nt * 64] = v; else slot[nt * 64] += v;
                }
            };
            if constexpr (SHOT) {
#pragma unroll
                for (int r = 0; r < MAXR; ++r) {
                    const int s = r / SUB, sub = r % SUB;          // compile-time after unrolling
                    if (s < nstrip) {
                        if (sub == 0) {
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) acc[nt] = acc2[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        }
                        if (sub < nsub) mma_round(w[r], acc, acc2, xb, xl, sub, k0);
                        if (sub == SUB - 1) park(s);
                    }
                }
            } else if (ringed) {
                for (int r0 = 0; r0 < nround; r0 += RD) {
#pragma unroll
                    for (int j = 0; j < RD; ++j) {                 // RD % SUB == 0: sub is compile-time, xb indices stay static
                        const int r = r0 + j, s = r / SUB;
                        constexpr int SUBM = SUB - 1;
                        const int sub = j & SUBM;
                        if (r < nround) {
                            if (sub == 0) {
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt) acc[nt] = acc2[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                            }
                            mma_round(ring[j], acc, acc2, xb, xl, sub, k0);
                            if (r + RD < nround)
                                load_round<FMT, TAIL>(ring[j], P, strip0 + (r + RD) / SUB, k0 + sub * RK, kend, true, lane);
                            if (sub == SUB - 1) park(s);
                        }
                    }
                }
            } else {
                for (int s = 0; s < nstrip; ++s) {
#pragma unroll
                    for (int sub = 0; sub < SUB; ++sub) {          // compile-time: xb indices stay static
                        if (sub < nsub) {
                            const bool more_sub = sub + 1 < nsub;
                            if (more_sub || s + 1 < nstrip)
                                load_round<FMT, TAIL>(nxt, P, strip0 + (more_sub ? s : s + 1), k0 + (more_sub ? sub + 1 : 0) * RK, kend, true, lane);
                            if (sub == 0) {
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt) acc[nt] = acc2[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                            }
                            mma_round(cur, acc, acc2, xb, xl, sub, k0);
                            cur = nxt;
                        }
                    }
                    park(s);
                }
            }
            (void)nround;
        }
        TRACE_PT(2);
        // ---- reduce + epilogue: (strip, n-tile) items round-robin over the block's waves.  Everything an item needs that does not
        // depend on the reduction (bias, the POST_MUL / POST_MIX operands) is requested BEFORE the barrier, for the wave's first item,
        // and lands while the wave waits for the block's slowest; the K partials are read with ONE wait (all loads, then the adds, in
        // slice order as before) instead of a load-wait-add round trip per partial.
        // Epilogue parameters, pinned into SGPRs HERE (in front of the barrier, behind the K loop whose registers they must not crowd).
        // The argument struct lives in the kernarg segment, so every field read is a scalar load; left to the compiler they were
        // re-read inside the item loop — a scalar-cache round trip in front of every output row, inside the ~1 us between the barrier
        // and the end of the kernel.
        const int e_act = pin_s(P.act), e_post = pin_s(P.post), e_ldm = pin_s(P.ldm), e_ldo = pin_s(P.ldo), e_ldh = pin_s(P.ldh);
        const float *e_bias = (const float *)pin_p(P.bias);
        const bool e_f32 = pin_s(P.out_f32 != nullptr) != 0, e_hi = pin_s(P.out_hi != nullptr) != 0, e_lo = pin_s(P.out_lo != nullptr) != 0;
        const int nwaves = blockDim.x >> 6, nitem = nstrip * NT;
        float4 pb = make_float4(0.f, 0.f, 0.f, 0.f), pm0 = pb, pm1 = pb;
        auto prefetch = [&](int item) {
            const int s = item / NT, nt = item - s * NT;
            const int row0 = (strip0 + s) * 16 + (lane >> 4) * 4;
            const int t = t0 + nt * 16 + (lane & 15);
            if (t < L.T) {
                if (e_bias) pb = *(const float4 *)(e_bias + row0);
                if (e_post != POST_NONE) pm0 = act_ld4(bm0, (long)t * e_ldm + row0);
                if (e_post == POST_MIX) pm1 = act_ld4(bm1, (long)t * e_ldm + row0);
            }
        };
        if (wave < nitem) prefetch(wave);
        __syncthreads();
        TRACE_PT(3);
        for (int item = wave; item < nitem; item += nwaves) {
            if (item != wave) prefetch(item);
            const int s = item / NT, nt = item - s * NT;
            // K partials in slice order, three LDS reads per wait (ten of them in one array made hipcc keep 40 more registers live
            // across the K loop and spill; one read per wait was nine LDS round trips)
            const f32x4 *rp = red + ((s * nw) * NT + nt) * 64 + lane;
            f32x4 v4 = rp[0];
            for (int w2 = 1; w2 < nw; w2 += 3) {
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                const f32x4 pa = rp[w2 * NT * 64];
                const f32x4 pb2 = w2 + 1 < nw ? rp[(w2 + 1) * NT * 64] : zero;
                const f32x4 pc = w2 + 2 < nw ? rp[(w2 + 2) * NT * 64] : zero;
                v4 += pa; v4 += pb2; v4 += pc;
            }
            const int row0 = (strip0 + s) * 16 + (lane >> 4) * 4;
            const int t = t0 + nt * 16 + (lane & 15);
            if (t < L.T) {
                float v[4] = {v4[0], v4[1], v4[2], v4[3]};
                if (e_bias) { v[0] += pb.x; v[1] += pb.y; v[2] += pb.z; v[3] += pb.w; }
                apply_act4(e_act, v);
                if (e_post == POST_MUL) {
                    v[0] *= pm0.x; v[1] *= pm0.y; v[2] *= pm0.z; v[3] *= pm0.w;
                } else if (e_post == POST_MIX) {
                    v[0] = pm0.x + pm1.x * v[0]; v[1] = pm0.y + pm1.y * v[1]; v[2] = pm0.z + pm1.z * v[2]; v[3] = pm0.w + pm1.w * v[3];
                }
                if (e_f32) act_st4(bo32, (long)t * e_ldo + row0, make_float4(v[0], v[1], v[2], v[3]));
                if (e_hi) act_store_operand4(boh, bol, e_lo, opd_off(t, row0, e_ldh), make_float4(v[0], v[1], v[2], v[3]));
            }
        }
        if (t0 + NT * 16 < L.T) __syncthreads();
    }
    TRACE_PT(4);
}

template <int NT, int KSW, bool HILO, bool SHOT, bool TAIL>
__global__ __launch_bounds__(((NT == 4 || (NT == 2 && HILO)) ? GEMM_MAX_WAVES_K16 : GEMM_MAX_WAVES) * 64) void gemm_kernel(const GemmLaunch L) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if ((int)blockIdx.x >= L.total_blocks) {                      // the extra block of a launch that carries a commit
        shift_commit(L.commit);
        return;
    }
    int pi = 0;
    for (int i = 1; i < L.nprob; ++i)
        if ((int)blockIdx.x >= L.p[i].block_begin) pi = i;
    const GemmProb &P = L.p[pi];
    // (One kernel carries the three weight formats: building it with a single format compiled in changes no launch by more than noise —
    // profiles/r5_exp_single_format_kernel.log — so code size is not where a launch's fixed cost goes.)
    if (P.fmt == W_F16) gemm_body<NT, KSW, HILO, SHOT, TAIL, W_F16>(L, P, smem);
    else if (P.fmt == W_INT8) gemm_body<NT, KSW, HILO, SHOT, false, W_INT8>(L, P, smem);   // quantised K is a multiple of 256
    else gemm_body<NT, KSW, HILO, SHOT, false, W_NF4>(L, P, smem);
}

int gemm_variant_max_waves(int NT, int, bool hilo) { return (NT == 4 || (NT == 2 && hilo)) ? GEMM_MAX_WAVES_K16 : GEMM_MAX_WAVES; }

void gemm_variant(int T, bool hilo, int &NT, int &KSW) {
    // Every variant runs 256-k waves (ten per block at K = 2560): a wave's loads return in order and what a CU can pull from HBM grows with
    // its waves, not with the loads each keeps in flight (profiles/r3_exp_stream_waves_x_loads.log: 27 MB over 256 workgroups: 5 waves
    // 8.6-10.8 us, 10 waves 6.9-8.0, 16 waves 6.8-7.1 whatever the depth).  (The 512-k form and its LayerNorm-prologue launch lost that
    // A/B in round 3 and were removed in round 5.)
    KSW = 8;
    // hi + lo operands (Precision::Fp32, or a promoted launch): 17+ rows run two token tiles per pass in 512-thread blocks (128 X registers, the
    // register shape of the four-tile variant) — one pass over the weights for up to 32 rows instead of one per 16
    if (hilo) NT = T <= 16 ? 1 : 2;
    else if (T <= 16) NT = 1;
    else if (T <= 32) NT = 2;
    else NT = 4;                                                  // 33..64 rows in ONE pass over the weights (128 X registers)
}

void launch_gemm(const GemmLaunch &L, bool hilo, hipStream_t s) {
    int NT, KSW;
    gemm_variant(L.T, hilo, NT, KSW);
    const size_t lds = (size_t)L.lds_items * NT * 64 * 16;
    dim3 grid(L.total_blocks + (L.commit.src ? 1 : 0)), block(L.threads);
    static bool attr_done[16] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
#define GEMM_V3(X, sh, tl) X(1, 8, true, sh, tl) X(2, 8, true, sh, tl) X(2, 8, false, sh, tl) X(1, 8, false, sh, tl) X(4, 8, false, sh, tl)
#define GEMM_VARIANTS(X) GEMM_V3(X, true, true) GEMM_V3(X, true, false) GEMM_V3(X, false, true) GEMM_V3(X, false, false)
    if (!attr_done[dev & 15]) {                               // allow > 64 KiB dynamic LDS (gfx950: 160 KiB / CU)
        const int cap = 160 * 1024;
#define SET_ATTR(a, b, c, d, e) (void)hipFuncSetAttribute((const void *)gemm_kernel<a, b, c, d, e>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        GEMM_VARIANTS(SET_ATTR)
#undef SET_ATTR
        attr_done[dev & 15] = true;
    }
    const bool shot = L.single_shot != 0, tail = L.tail != 0;
#define LAUNCH(a, b, c, d, e) if (NT == a && KSW == b && hilo == c && shot == d && tail == e) hipLaunchKernelGGL((gemm_kernel<a, b, c, d, e>), grid, block, lds, s, L);
    GEMM_VARIANTS(LAUNCH)
#undef LAUNCH
#undef GEMM_VARIANTS
#undef GEMM_V3
}

int gemm_max_rounds(int fmt, int NT, bool hilo) { return (NT == 4 || (NT == 2 && hilo)) ? 2 : (fmt == W_F16 ? 2 : ((NT == 2 || hilo) ? 3 : 4)); }   // X registers vs the VGPR budget
#endif  // part 0: decode GEMM


#if RWKV_PART_ON(1)
// =====================================================================================
// V6 data-dependent token-shift, fused (decode, T <= 32):  x_c = xx + dx * (mu_c + W2_c * tanh(W1_c * z)),
// c in (w,k,v,r,g) — SURVEY A.4.  One launch instead of two dependent GEMMs: a block owns one mix c and 8 strips
// of W2_c; its 8 waves first split K to compute the block's OWN copy of m_c = tanh(W1_c z) ([T][Dm], W1_c is
// Dm x C fp16 = 164 KiB from L2, recomputed by the C/128 blocks of that c), park the partials in LDS, reduce,
// apply tanh and keep m_c in LDS as the f16 (hi, lo) operand; then wave w multiplies strip w of W2_c (K = Dm)
// with it and applies the lerp epilogue, emitting the five GEMM operands.
// =====================================================================================
// WIDE (steps with more than 32 rows, prefill included): grid (1, 5, ceil(T/32)) — a block owns one mix c and one 32-token
// tile, computes m_c for its tokens once and then walks ALL strips of W2_c (8 waves, strip = wave, wave+8, ...).  Per block:
// z tile 160 KB + W1_c 164 KB + W2_c 164 KB + the xx/dx tile, against 5 x C/16 x T/16 tiles of work: one launch of ~15 us
// for a 512-row step instead of the two tile-GEMM launches (W1: 24 workgroups, 25 us; W2 with K = 32: 46 us) it replaces.
// SPLIT (steps of >= 512 rows, a multiple of 32): the wide form taken apart at its barrier.  As one launch every (mix, token tile)
// block pulls the tile's xx / dx rows (655 KB of its 1.3 MB) and a launch of 320 such blocks on 256 CUs is two rounds.  P1ONLY stops
// after phase 1 and leaves m_c in HBM (1.3 MB per step); v6_mix_apply_kernel then runs phase 2 per (8 strips, token tile) for all five
// mixes — xx / dx are read once instead of five times, 1280 small blocks balance.
template <int NT, bool HILO, int DS, bool LNP, bool WIDE = false, bool P1ONLY = false>
__global__ __launch_bounds__(512) void v6_mix_kernel(const V6MixArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = blockIdx.y;                                     // mix index
    const int sg = blockIdx.x;                                    // strip group (8 strips of W2_c)
    const int C = a.C, Dm = a.Dm, T = a.T;
    const int tz = blockIdx.z;
    const int t0 = tz * NT * 16;                                  // first token of this block's tile (decode form: 17..32 rows run two NT = 1 tiles)
    // DS = Dm/16 strips of W1_c (2 or 4)
    const int KT1 = C >> 5;                                       // k-tiles of W1 per strip
    // P1ONLY with a.ksp > 1 (round 6, steps of a few hundred rows): blockIdx.x = K slice — the (mix, token tile) pairs of such a step are fewer
    // than the chip's CUs, and as whole-K blocks each of them pulls W1_c and its z tile (328 KB) through one CU; sliced over K, a.ksp blocks
    // share that, write fp32 partials (a.mp) and v6_mix_apply_kernel sums them in slice order before the tanh
    const int ksp = P1ONLY ? a.ksp : 1, kq = P1ONLY ? (int)blockIdx.x : 0;
    const int kst = (KT1 >> 3) / ksp;                             // k-steps per wave in phase 1 (C/8/32 over the K slices)
    const int kw0 = (kq * 8 + wave) * kst;                        // first k-tile of this wave
    f32x4 *red = (f32x4 *)smem;                                   // [8 waves][DS][NT][64]
    const int mstride = Dm + 8;                                   // halfs per token row of m_c
    _Float16 *m_hi = (_Float16 *)(smem + (size_t)8 * 4 * NT * 64 * 16);
    _Float16 *m_lo = m_hi + NT * 16 * mstride;

    TRACE_K(0, 0);
    // phase-2 operands of this wave (W2 strip, mu, xx, dx): independent of phase 1, so fetched first — their L2/MALL
    // latency hides behind phase 1 instead of following its barrier
    const int strip = sg * 8 + wave;                              // >= C/16: idle in phase 2
    const int strip_c = min(strip, (C >> 4) - 1);
    const int row0 = strip_c * 16 + (lane >> 4) * 4;
    u32x4 w2t[DS / 2];
    float4 xxv[NT], dxv[NT];
    float4 mu = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (!WIDE) {
#pragma unroll
        for (int ks = 0; ks < DS / 2; ++ks) w2t[ks] = ((const u32x4 *)a.W2[c])[((long)strip_c * (DS / 2) + ks) * 64 + lane];
        mu = *(const float4 *)(a.mu[c] + row0);
    }
    const act_t bxx = act_buf(a.xx), bdx = act_buf(a.dx), bzh = act_buf(a.zhi), bzl = act_buf(a.zlo);
    const act_t boh = act_buf(a.ohi[c]), bol = act_buf(a.olo[c]);
    // LNP (single-token steps): LayerNorm + token shift are redone here by every block (ln_prologue_*): z arrives in LDS in
    // fragment order, xx and dx for the epilogue come from the prologue's LDS rows
    float *xx_l = (float *)(m_lo + NT * 16 * mstride);
    const int ldl = C + LNP_PAD;
    float *pv_l = xx_l + LNP_MAX_T * ldl, *lred = pv_l + LNP_MAX_T * ldl;
    _Float16 *z_l = (_Float16 *)(lred + 64);
    const u32x4 *w1 = (const u32x4 *)a.W1;
    constexpr int KB = DS == 2 ? ((HILO && NT == 2) ? 5 : 10) : 4;                                          // k-steps per batch: all loads of a batch in flight at once
    u32x4 wt[KB][DS];
    auto load_wt = [&](int k0) {                                  // this wave's W1_c tiles of k-steps k0 .. k0+KB
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            if (k0 + j < kst) {                                    // wave-uniform
                const int kt = kw0 + k0 + j;
#pragma unroll
                for (int d = 0; d < DS; ++d) wt[j][d] = w1[((long)(c * DS + d) * KT1 + kt) * 64 + lane];
            }
        }
    };
    if constexpr (LNP) {
        LnCarry carry;
        const bool pub = blockIdx.x == 0 && blockIdx.y == 0;
        // W1 does not depend on the row: its first batch (the whole slice of this wave at C = 2560) is requested before the
        // prologue, so the 164 KiB a block pulls from L2 overlap the row loads and the four barriers of the LayerNorm instead
        // of following them (phase 1 of a single-token step: 5.2 -> ~3 us in the in-kernel timeline)
        load_wt(0);
        ln_prologue_load(a.lnp, T, xx_l, pv_l, lred, pub, carry);
        ln_prologue_finish<HILO>(a.lnp, T, a.mu_x, xx_l, pv_l, z_l, lred, pub, carry);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int t = min(nt * 16 + (lane & 15), T - 1);
            const float4 x = *(const float4 *)(xx_l + t * ldl + row0);
            const float4 pr = carry.prev[t < LNP_MAX_T ? t : 0] >= 0 ? *(const float4 *)(xx_l + carry.prev[t < LNP_MAX_T ? t : 0] * ldl + row0)
                                                                      : *(const float4 *)(pv_l + t * ldl + row0);
            xxv[nt] = x;
            dxv[nt] = make_float4(pr.x - x.x, pr.y - x.y, pr.z - x.z, pr.w - x.w);
        }
    } else if constexpr (!WIDE) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            int t = t0 + nt * 16 + (lane & 15);
            t = t < T ? t : T - 1;
            xxv[nt] = act_ld4(bxx, (long)t * C + row0);
            dxv[nt] = act_ld4(bdx, (long)t * C + row0);
        }
    }
    // ---- phase 1: partial m_c over this wave's K slice
    f32x4 acc[DS][NT];
#pragma unroll
    for (int d = 0; d < DS; ++d)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[d][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < kst; k0 += KB) {
        f16x8 zb[KB][NT], zl[HILO ? KB : 1][HILO ? NT : 1];
        if (!LNP || k0 > 0) load_wt(k0);
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            if (k0 + j < kst) {                                    // wave-uniform
                const int kt = kw0 + k0 + j;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int tile = min((t0 >> 4) + nt, (T - 1) >> 4);
                    const int tl = min(lane & 15, T - 1 - tile * 16);
                    if constexpr (LNP) {
                        zb[j][nt] = *(const f16x8 *)(z_l + lnp_op_off(T, kt * 32 + (lane >> 4) * 8, tl));
                        if constexpr (HILO) zl[j][nt] = *(const f16x8 *)(z_l + (size_t)T * C + lnp_op_off(T, kt * 32 + (lane >> 4) * 8, tl));
                    } else {
                        const long zo = ((long)tile * (a.ldz >> 5) + kt) * 512 + ((lane >> 4) * 16 + tl) * 8;
                        zb[j][nt] = act_ldh8(bzh, zo);
                        if constexpr (HILO) zl[j][nt] = act_ldh8(bzl, zo);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            if (k0 + j < kst) {
#pragma unroll
                for (int d = 0; d < DS; ++d) {
                    const f16x8 af = __builtin_bit_cast(f16x8, wt[j][d]);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        acc[d][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, zb[j][nt], acc[d][nt], 0, 0, 0);
                        if constexpr (HILO) acc[d][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, zl[j][nt], acc[d][nt], 0, 0, 0);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int d = 0; d < DS; ++d)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) red[((wave * 4 + d) * NT + nt) * 64 + lane] = acc[d][nt];
    TRACE_K(0, 1);
    __syncthreads();
    TRACE_K(0, 2);
    // reduce the 8 K-partials, tanh, park m_c[t][d] as f16 hi/lo
    for (int item = wave; item < DS * NT; item += 8) {
        const int d = item / NT, nt = item - d * NT;
        f32x4 v = red[((0 * 4 + d) * NT + nt) * 64 + lane];
        for (int w2 = 1; w2 < 8; ++w2) v += red[((w2 * 4 + d) * NT + nt) * 64 + lane];
        const int t = nt * 16 + (lane & 15);
        if constexpr (P1ONLY) {
            if (ksp > 1) {                                         // this K slice's partial, fp32, [slice][mix][token][Dm]
                const long po = (((long)kq * 5 + c) * ((long)gridDim.z * NT * 16) + t0 + t) * Dm + d * 16 + (lane >> 4) * 4;
                *(f32x4 *)(a.mp + po) = v;
                continue;
            }
        }
        f16x4 hh, ll;
#pragma unroll
        for (int r = 0; r < 4; ++r) { _Float16 x, y; split_hilo(tanhf(v[r]), x, y); hh[r] = x; ll[r] = y; }
        if constexpr (P1ONLY) {
            const long mo = (((long)c * gridDim.z + tz) * (NT * 16) + t) * Dm + d * 16 + (lane >> 4) * 4;
            *(f16x4 *)(a.mg_hi + mo) = hh;
            if constexpr (HILO) *(f16x4 *)(a.mg_lo + mo) = ll;
        } else {
            *(f16x4 *)(m_hi + t * mstride + d * 16 + (lane >> 4) * 4) = hh;
            if constexpr (HILO) *(f16x4 *)(m_lo + t * mstride + d * 16 + (lane >> 4) * 4) = ll;
        }
    }
    if constexpr (P1ONLY) return;
    __syncthreads();
    TRACE_K(0, 3);
    // ---- phase 2: a strip of W2_c times m_c, lerp epilogue.  Decode form: strip sg*8 + wave, operands prefetched before
    //      phase 1.  WIDE: every strip of W2_c, round-robin over the block's 8 waves, operands fetched per strip.
    auto do_strip = [&](int sidx, const u32x4 (&w2)[DS / 2], const float4 &muv, const float4 (&xxs)[NT], const float4 (&dxs)[NT]) {
        const int r0 = sidx * 16 + (lane >> 4) * 4;
        f32x4 o[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) o[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < DS / 2; ++ks) {
            const f16x8 af = __builtin_bit_cast(f16x8, w2[ks]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int off = (nt * 16 + (lane & 15)) * mstride + ks * 32 + (lane >> 4) * 8;
                o[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, *(const f16x8 *)(m_hi + off), o[nt], 0, 0, 0);
                if constexpr (HILO) o[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, *(const f16x8 *)(m_lo + off), o[nt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int t = t0 + nt * 16 + (lane & 15);
            if (t < T) {
                const float4 xx = xxs[nt], dx = dxs[nt];
                float4 r;
                r.x = xx.x + dx.x * (muv.x + o[nt][0]);
                r.y = xx.y + dx.y * (muv.y + o[nt][1]);
                r.z = xx.z + dx.z * (muv.z + o[nt][2]);
                r.w = xx.w + dx.w * (muv.w + o[nt][3]);
                act_store_operand4(boh, bol, a.olo[c] != nullptr, opd_off(t, r0, a.ldh), r);
            }
        }
    };
    if constexpr (WIDE) {
        // strips sg*8 + wave, then + 8*gridDim.x: the launch spreads a mix's strips over gridDim.x blocks when the step has few
        // token tiles (each of them recomputes m_c for its tile); the next strip's operands are requested before the current
        // strip is multiplied, so a strip costs its MFMAs and stores, not an L2 round trip
        struct StripIn { u32x4 w2[DS / 2]; float4 muv, xxs[NT], dxs[NT]; };
        const int sstep = 8 * (int)gridDim.x, nst = C >> 4;
        auto fetch = [&](int sidx, StripIn &in) {
            const int sc = min(sidx, nst - 1);
            const int r0 = sc * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int ks = 0; ks < DS / 2; ++ks) in.w2[ks] = ((const u32x4 *)a.W2[c])[((long)sc * (DS / 2) + ks) * 64 + lane];
            in.muv = *(const float4 *)(a.mu[c] + r0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int t = min(t0 + nt * 16 + (lane & 15), T - 1);
                in.xxs[nt] = act_ld4(bxx, (long)t * C + r0);
                in.dxs[nt] = act_ld4(bdx, (long)t * C + r0);
            }
        };
        StripIn cur, nxt;
        int sidx = sg * 8 + wave;
        if (sidx < nst) fetch(sidx, cur);
        for (; sidx < nst; sidx += sstep) {
            if (sidx + sstep < nst) fetch(sidx + sstep, nxt);
            do_strip(sidx, cur.w2, cur.muv, cur.xxs, cur.dxs);
            cur = nxt;
        }
    } else {
        if (strip < (C >> 4)) do_strip(strip, w2t, mu, xxv, dxv);
    }
    TRACE_K(0, 4);
}

// Phase 2 of the split form: block = (8 strips of the C rows, one 32-token tile), wave = one strip, all five mixes.
template <bool HILO, int DS, int NT = 2>
__global__ __launch_bounds__(512) void v6_mix_apply_kernel(const V6MixArgs a) {
    constexpr int Dm = DS * 16, mstride = Dm + 8;
    __shared__ __attribute__((aligned(16))) _Float16 m_hi[5 * NT * 16 * mstride], m_lo[HILO ? 5 * NT * 16 * mstride : 8];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = a.C, T = a.T, tz = blockIdx.y, t0 = tz * NT * 16, ntile = gridDim.y;
    const int strip = min((int)blockIdx.x * 8 + wave, (C >> 4) - 1);      // clamped: the last group's spare waves redo the last strip
    const int r0 = strip * 16 + (lane >> 4) * 4;
    const act_t bxx = act_buf(a.xx), bdx = act_buf(a.dx);
    // everything this wave needs, issued before the m tiles are staged
    float4 xxv[NT], dxv[NT], muv[5];
    u32x4 w2[5][DS / 2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int t = min(t0 + nt * 16 + (lane & 15), T - 1);
        xxv[nt] = act_ld4(bxx, (long)t * C + r0);
        dxv[nt] = act_ld4(bdx, (long)t * C + r0);
    }
#pragma unroll
    for (int c = 0; c < 5; ++c) {
#pragma unroll
        for (int ks = 0; ks < DS / 2; ++ks) w2[c][ks] = ((const u32x4 *)a.W2[c])[((long)strip * (DS / 2) + ks) * 64 + lane];
        muv[c] = *(const float4 *)(a.mu[c] + r0);
    }
    // m_c tiles of this token tile: 5 x 32 x Dm halves, 16-byte pieces
    constexpr int PPR = Dm / 8, NP = 5 * NT * 16 * PPR;                    // pieces per token row, pieces in all
    const int ksp = a.ksp;
    if (ksp > 1) {
        // phase 1 ran sliced over K: the slices' fp32 partials are summed in slice order, then tanh.  Five slices' loads are in flight at once
        // (a rolled loop over the slices is one L2 round trip per slice: 10.5 us for this kernel at 256 rows against 5.2 for phase 1)
        const long Tp = (long)ntile * NT * 16, qstride = 5 * Tp * Dm;
        for (int i = tid; i < NP; i += 512) {
            const int c = i / (NT * 16 * PPR), rem = i - c * (NT * 16 * PPR), t = rem / PPR, d8 = rem - t * PPR;
            const float *pp = a.mp + ((long)c * Tp + t0 + t) * Dm + d8 * 8;
            f32x4 s0 = (f32x4){0.f, 0.f, 0.f, 0.f}, s1 = s0;
            for (int q0 = 0; q0 < ksp; q0 += 5) {
                f32x4 v0[5], v1[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const float *pq = pp + (long)min(q0 + j, ksp - 1) * qstride;          // clamped, not predicated: straight-line loads
                    v0[j] = *(const f32x4 *)pq; v1[j] = *(const f32x4 *)(pq + 4);
                }
#pragma unroll
                for (int j = 0; j < 5; ++j) if (q0 + j < ksp) { s0 += v0[j]; s1 += v1[j]; }
            }
            f16x8 hh, ll;
#pragma unroll
            for (int e = 0; e < 8; ++e) { _Float16 x, y; split_hilo(tanhf(e < 4 ? s0[e] : s1[e - 4]), x, y); hh[e] = x; ll[e] = y; }
            *(f16x8 *)(m_hi + (c * NT * 16 + t) * mstride + d8 * 8) = hh;
            if constexpr (HILO) *(f16x8 *)(m_lo + (c * NT * 16 + t) * mstride + d8 * 8) = ll;
        }
    } else {
        for (int i = tid; i < NP; i += 512) {
            const int c = i / (NT * 16 * PPR), rem = i - c * (NT * 16 * PPR), t = rem / PPR, d8 = rem - t * PPR;
            const long go = (((long)c * ntile + tz) * (NT * 16) + t) * Dm + d8 * 8;
            *(u32x4 *)(m_hi + (c * NT * 16 + t) * mstride + d8 * 8) = *(const u32x4 *)(a.mg_hi + go);
            if constexpr (HILO) *(u32x4 *)(m_lo + (c * NT * 16 + t) * mstride + d8 * 8) = *(const u32x4 *)(a.mg_lo + go);
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        const act_t boh = act_buf(a.ohi[c]), bol = act_buf(a.olo[c]);
        f32x4 o[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) o[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < DS / 2; ++ks) {
            const f16x8 af = __builtin_bit_cast(f16x8, w2[c][ks]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int off = (c * NT * 16 + nt * 16 + (lane & 15)) * mstride + ks * 32 + (lane >> 4) * 8;
                o[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, *(const f16x8 *)(m_hi + off), o[nt], 0, 0, 0);
                if constexpr (HILO) o[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, *(const f16x8 *)(m_lo + off), o[nt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int t = t0 + nt * 16 + (lane & 15);
            if (t < T) {
                const float4 xx = xxv[nt], dx = dxv[nt];
                float4 r;
                r.x = xx.x + dx.x * (muv[c].x + o[nt][0]);
                r.y = xx.y + dx.y * (muv[c].y + o[nt][1]);
                r.z = xx.z + dx.z * (muv[c].z + o[nt][2]);
                r.w = xx.w + dx.w * (muv[c].w + o[nt][3]);
                act_store_operand4(boh, bol, a.olo[c] != nullptr, opd_off(t, r0, a.ldh), r);
            }
        }
    }
}

bool v6_mix_supported(int T, int C, int Dm) { return T <= 32 && C % 256 == 0 && (Dm == 32 || Dm == 64); }
bool v6_mix_wide_supported(int T, int C, int Dm) { return T > 32 && C % 256 == 0 && (Dm == 32 || Dm == 64); }

// Two launches (phase 1 + v6_mix_apply_kernel) for this step?  0: one launch; n >= 1: two, phase 1 in n K slices.
//  * >= 512 rows (a multiple of 32): one slice (round 3);
//  * 192 .. 511 rows (round 6): phase 1 sliced over K — the smallest slice count (a divisor of the k-steps per wave) that gives the launch >= 160
//    blocks; partials in a.mp.  V6-3B at 256 rows: one launch of 240 blocks 16.4 us -> 200 slice blocks 5.2 us + 160 apply blocks 9.0 us (rocprofv3,
//    profiles/r6_exp_v6mix_kslices.log); at 512 rows even, at 128 rows slower: the rule's range.
int v6_mix_split(const V6MixArgs &a, bool hilo) {
    constexpr int V6_SPLIT_MIN_T = 512;
    if (a.T <= 32 || a.T % 32) return 0;
    const int ntile = a.T / 32;
    if (a.mp && a.T >= a.ksp_min_t && a.T < V6_SPLIT_MIN_T) {
        const int per_wave = (a.C >> 5) >> 3;
        int best = 1;
        for (int q = 1; q <= per_wave && q <= a.ksp_max; ++q) {
            if (per_wave % q) continue;
            best = q;
            if (5 * ntile * q >= a.ksp_blocks) break;
        }
        if (best > 1) return best;
    }
    return (a.T >= V6_SPLIT_MIN_T && a.mg_hi && (!hilo || a.mg_lo)) ? 1 : 0;
}

void launch_v6_mix(const V6MixArgs &a, bool hilo, hipStream_t s) {
    // (the wide form — 40 blocks that each walk every strip — for 17..32 rows too: 2.250 -> 2.36 ms per 32-slot step,
    // profiles/r3_exp_ab_v6mix_wide_for_32_rows.log; 100 blocks of 8 strips is the measured optimum between that and the 200-block split)
    const bool wide = a.T > 32;                                // v6_mix_wide_supported: one block per (mix, 32-token tile)
    // (17..32 rows as two NT = 1 token tiles — 200 workgroups that each pull W1_c + half of z — was measured and dropped: every
    // workgroup still pulls all of W1_c, so the launch's L2 traffic grows by half: 2.250 -> 2.267 ms per 32-slot step,
    // profiles/r3_exp_ab_v6mix_ksw8.log.)
    const int NT = (a.T <= 16 && !wide) ? 1 : 2;
    const bool lnp = a.lnp.x_in != nullptr;                    // host: T <= LNP_MAX_T, !hilo, C <= 4096 (v6_mix_ln_supported)
    const size_t lds = (size_t)8 * 4 * NT * 64 * 16 + (size_t)2 * NT * 16 * (a.Dm + 8) * 2 + (lnp ? lnp_lds_bytes(LNP_MAX_T, a.C, hilo) : 0);
    dim3 grid((a.C / 16 + 7) / 8, 5, 1), block(512);
    if (wide) {
        const int ntile = (a.T + 31) / 32;
        grid = dim3(std::max(1, std::min(8, 256 / (5 * ntile))), 5, ntile);   // fill the chip when the step has few token tiles
        V6MixArgs b = a;
        b.ksp = std::max(1, v6_mix_split(a, hilo));
        if (v6_mix_split(a, hilo) > 0) {
            const V6MixArgs &a = b;                             // (shadows: the launches below take the slice count)
            grid = dim3(a.ksp, 5, ntile);
            const dim3 g2((a.C / 16 + 7) / 8, ntile);
            // the apply launch on 16-token tiles when 32-token tiles leave it fewer blocks than the chip has CUs (256-row steps: 160 -> 320 blocks,
            // 9.07 -> 8.57 us; profiles/r6_exp_v6mix_kslices.log)
            const bool nt1 = (long)g2.x * g2.y < 256;
            const dim3 g1((a.C / 16 + 7) / 8, (a.T + 15) / 16);
#define V6P1(h, ds) hipLaunchKernelGGL((v6_mix_kernel<2, h, ds, false, true, true>), grid, block, lds, s, a)
#define V6AP(h, ds) do { if (nt1) hipLaunchKernelGGL((v6_mix_apply_kernel<h, ds, 1>), g1, block, 0, s, a); else hipLaunchKernelGGL((v6_mix_apply_kernel<h, ds, 2>), g2, block, 0, s, a); } while (0)
            if (a.Dm == 32) { if (hilo) { V6P1(true, 2); V6AP(true, 2); } else { V6P1(false, 2); V6AP(false, 2); } }
            else { if (hilo) { V6P1(true, 4); V6AP(true, 4); } else { V6P1(false, 4); V6AP(false, 4); } }
#undef V6P1
#undef V6AP
            return;
        }
        if (a.Dm == 32) { if (hilo) hipLaunchKernelGGL((v6_mix_kernel<2, true, 2, false, true>), grid, block, lds, s, a); else hipLaunchKernelGGL((v6_mix_kernel<2, false, 2, false, true>), grid, block, lds, s, a); }
        else { if (hilo) hipLaunchKernelGGL((v6_mix_kernel<2, true, 4, false, true>), grid, block, lds, s, a); else hipLaunchKernelGGL((v6_mix_kernel<2, false, 4, false, true>), grid, block, lds, s, a); }
        return;
    }
    static bool attr_done[16] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_done[dev & 15]) {
        (void)hipFuncSetAttribute((const void *)v6_mix_kernel<1, false, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)v6_mix_kernel<1, false, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done[dev & 15] = true;
    }
#define V6L(nt, h, ds, ln) hipLaunchKernelGGL((v6_mix_kernel<nt, h, ds, ln>), grid, block, lds, s, a)
    const int DS = a.Dm >> 4;
    if (lnp) { if (DS == 2) V6L(1, false, 2, true); else V6L(1, false, 4, true); }
    else if (hilo) { if (NT == 1) { if (DS == 2) V6L(1, true, 2, false); else V6L(1, true, 4, false); } else { if (DS == 2) V6L(2, true, 2, false); else V6L(2, true, 4, false); } }
    else      { if (NT == 1) { if (DS == 2) V6L(1, false, 2, false); else V6L(1, false, 4, false); } else { if (DS == 2) V6L(2, false, 2, false); else V6L(2, false, 4, false); } }
#undef V6L
}
bool v6_mix_ln_supported(int T, int C, int Dm, bool hilo, int np) {
    return v6_mix_supported(T, C, Dm) && T <= LNP_MAX_T && !hilo && np <= LNP_MAX_NP && C <= 8 * 512;
}

// =====================================================================================
// Small-K GEMM, output-stationary (V7's second LoRA stage: four [C x D] matrices, D = 64 .. 320, against the first stage's [T x D]
// outputs; K14).  One wave owns one strip (16 output rows) of one problem for ALL rows of the step: its <= KMAX / 32 weight tiles are
// loaded once and stay in registers, then it walks the token tiles — the next tile's X fragments (one contiguous 1 KiB per k-step) are
// requested before the current tile is multiplied — and stores bias + activation straight from the accumulators.  No K split across
// waves, so no LDS, no barrier, no reduce; the launch is a load -> MFMA -> store chain of ~10 instructions per token tile.  Replaces
// the generic kernels on these launches (which split a K of 96 over waves sized for K = 2560: 6.4 us at 32 rows, 10.8 us at 256).
// =====================================================================================
constexpr int SK_KMAX = 320, SK_TG = 4;
template <bool HILO>
__global__ __launch_bounds__(256) void smallk_kernel(const GemmLaunch L) {
    const int lane = threadIdx.x & 63;
    const int item = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);            // (problem, strip), problems back to back
    if (item >= L.total_blocks) return;
    int pi = 0;
    for (int i = 1; i < L.nprob; ++i) if (item >= L.p[i].block_begin) pi = i;
    const GemmProb &P = L.p[pi];
    const int strip = item - P.block_begin, KT = P.K >> 5;
    constexpr int MK = SK_KMAX / 32;
    u32x4 w[MK];
    const u32x4 *wb = (const u32x4 *)P.W + (long)strip * KT * 64 + lane;
#pragma unroll
    for (int j = 0; j < MK; ++j) if (j < KT) w[j] = wb[j * 64];
    const act_t bxh = act_buf(P.xhi), bxl = act_buf(P.xlo), bo = act_buf(P.out_f32);
    const int row0 = strip * 16 + (lane >> 4) * 4;
    const float4 bias = P.bias ? *(const float4 *)(P.bias + row0) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int ntile = (L.T + 15) >> 4, act = P.act, ldo = P.ldo;
    const long tstride = (long)(P.ldx >> 5) * 512;                              // halves between token tiles of the operand
    struct XT { f16x8 h[MK], l[HILO ? MK : 1]; };
    auto load_x = [&](int tile, XT &x) {
        const int tl = min(lane & 15, L.T - 1 - tile * 16);                      // lanes beyond the last row re-read it (never stored)
        const long xo = tile * tstride + ((lane >> 4) * 16 + tl) * 8;
#pragma unroll
        for (int j = 0; j < MK; ++j) {
            if (j < KT) {
                x.h[j] = act_ldh8(bxh, xo + j * 512);
                if constexpr (HILO) x.l[j] = act_ldh8(bxl, xo + j * 512);
            }
        }
    };
    auto mul_store = [&](int tile, const XT &x) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = acc;
#pragma unroll
        for (int j = 0; j < MK; ++j) {
            if (j < KT) {
                const f16x8 af = __builtin_bit_cast(f16x8, w[j]);
                f32x4 &d = (j & 1) ? acc2 : acc;
                d = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, x.h[j], d, 0, 0, 0);
                if constexpr (HILO) d = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, x.l[j], d, 0, 0, 0);
            }
        }
        const int t = tile * 16 + (lane & 15);
        if (t < L.T) {
            float v[4] = {acc[0] + acc2[0] + bias.x, acc[1] + acc2[1] + bias.y, acc[2] + acc2[2] + bias.z, acc[3] + acc2[3] + bias.w};
            apply_act4(act, v);
            act_st4(bo, (long)t * ldo + row0, make_float4(v[0], v[1], v[2], v[3]));
        }
    };
    // two named buffers, the tile loop unrolled by two: a buffer picked by a run-time index would live in scratch memory
    // (prefill-shaped steps: blockIdx.y walks groups of SK_TG token tiles, so a wave's chain stays four tiles long whatever the step)
    const int tbeg = (int)blockIdx.y * SK_TG, tend = min(ntile, tbeg + SK_TG);
    XT xa, xb;
    load_x(tbeg, xa);
    for (int tile = tbeg; tile < tend; tile += 2) {
        if (tile + 1 < tend) load_x(tile + 1, xb);
        mul_store(tile, xa);
        if (tile + 2 < tend) load_x(tile + 2, xa);
        if (tile + 1 < tend) mul_store(tile + 1, xb);
    }
}
// every problem: fp16 weights, K <= SK_KMAX, fp32 output only, no post-op, no K split
bool smallk_supported(const GemmLaunch &L) {
    if (L.nprob < 1) return false;
    for (int i = 0; i < L.nprob; ++i) {
        const GemmProb &g = L.p[i];
        if (g.fmt != W_F16 || g.K > SK_KMAX || g.K % 32 || g.rows % 16 || !g.out_f32 || g.out_hi || g.post != POST_NONE || g.ksb != 1) return false;
    }
    return true;
}
// L.p[i].block_begin = first (problem, strip) item of problem i, L.total_blocks = items in all
void launch_smallk(const GemmLaunch &L, bool hilo, hipStream_t s) {
    const dim3 grid((L.total_blocks + 3) / 4, ((L.T + 15) / 16 + SK_TG - 1) / SK_TG), block(256);
    if (hilo) hipLaunchKernelGGL(smallk_kernel<true>, grid, block, 0, s, L);
    else hipLaunchKernelGGL(smallk_kernel<false>, grid, block, 0, s, L);
}
#endif  // part 1: fused V6 mix

#if RWKV_PART_ON(2) || RWKV_PART_ON(5)      // helpers shared by the tile kernels of part 2 (64x64 shapes, gemm_tile3) and part 5 (tile4 / tile5)
// =====================================================================================
// Prefill GEMM (T >= GEMM_TILE_MIN_T = 193): LDS-tiled MFMA GEMM over the same pre-tiled weights.
// Block = 8 waves; wave w owns strips {2w, 2w+1} of the block's 16 strips (256 output rows) and all 8 n-tiles of
// the block's 128-token tile (64 accumulator registers).  K is walked in chunks of 128: the X chunk
// [128 tokens][128 k] is staged through LDS (register-staged, double-buffered, one barrier per chunk) and shared
// by all waves; the wave's weight tiles go HBM/L2 -> registers (prefetched one chunk ahead) and every
// dequantised A fragment feeds 8 MFMAs.  Bound: MFMA (2*rows*K*T flops), weights re-read T/128 times from L2/MALL.
// =====================================================================================
// KC = k per chunk (128 or 256)
template <int FMT, int SPW, int KC> struct TRound { u32x4 q[SPW][(KC / 32) / Fmt<FMT>::KS]; uint2 s[SPW]; };

// FULL = true: the whole chunk lies inside K -> no predicates (strips beyond the matrix are clamped to the last
// strip: their results are never stored), so the compiler can count the loads in flight (s_waitcnt vmcnt(N)).
template <int FMT, int SPW, int KC, bool FULL>
__device__ __forceinline__ void tg_load(TRound<FMT, SPW, KC> &w, const GemmProb &P, int strip, int nstrips, int k0, int lane) {
    constexpr int NTILE = (KC / 32) / Fmt<FMT>::KS, TK = Fmt<FMT>::TK, SH = Fmt<FMT>::SH;
    const int KT = P.K >> SH;
#pragma unroll
    for (int h = 0; h < SPW; ++h) {
        const int sidx = min(strip + h, nstrips - 1);
        const u32x4 *base = (const u32x4 *)P.W + ((long)sidx * KT + (k0 >> SH)) * 64 + lane;
#pragma unroll
        for (int j = 0; j < NTILE; ++j) {
            if (FULL || k0 + j * TK < P.K) w.q[h][j] = base[j * 64];
            else w.q[h][j] = (u32x4){0u, 0u, 0u, 0u};
        }
        if constexpr (FMT != W_F16) {
            const int NG = P.K >> 8;                              // quantised K is a multiple of 256: one scale word per chunk
            w.s[h] = (FULL || k0 < P.K) ? *((const uint2 *)P.S + ((long)sidx * NG + (k0 >> 8)) * 16 + (lane & 15)) : make_uint2(0, 0);
        } else {
            w.s[h] = make_uint2(0, 0);
        }
    }
}

// A fragment of k-step ks (0..KC/32) of the chunk starting at k0
template <int FMT, int SPW, int KC>
__device__ __forceinline__ f16x8 tg_frag(const TRound<FMT, SPW, KC> &w, int h, int ks, int k0, const Nf4Lut &lut) {
    const int kk = k0 + ks * 32;                                  // absolute k of this k-step
    if constexpr (FMT == W_F16) {
        return __builtin_bit_cast(f16x8, w.q[h][ks]);
    } else if constexpr (FMT == W_INT8) {
        const u32x4 q = w.q[h][ks >> 1];
        const u32 d0 = (ks & 1) ? q.z : q.x, d1 = (ks & 1) ? q.w : q.y;
        const u32 ab = ((kk >> 7) & 1) ? w.s[h].y : w.s[h].x;    // 128-block inside the 256-group
        const f16x2 abh = as_h2(ab);
        const f16x2 a2 = {abh[0], abh[0]}, b2 = {abh[1], abh[1]};
        u32x4 r;
        r.x = dq8(d0, 0x04010400u, a2, b2);
        r.y = dq8(d0, 0x04030402u, a2, b2);
        r.z = dq8(d1, 0x04010400u, a2, b2);
        r.w = dq8(d1, 0x04030402u, a2, b2);
        return __builtin_bit_cast(f16x8, r);
    } else {
        const u32x4 q = w.q[h][ks >> 2];
        const int wsel = ks & 3;
        const u32 d = wsel == 0 ? q.x : wsel == 1 ? q.y : wsel == 2 ? q.z : q.w;
        const u32 sw = ((kk >> 7) & 1) ? w.s[h].y : w.s[h].x;    // 64-blocks 2*((kk>>7)&1) + ((kk>>6)&1)
        const f16x2 sh = as_h2(sw);
        const _Float16 am = ((kk >> 6) & 1) ? sh[1] : sh[0];
        const f16x2 am2 = {am, am};
        u32x4 r;
        u32 a, b;
        nf4_lookup4(d & 0x0F0F0F0Fu, lut, a, b);
        r.x = as_u32(as_h2(a) * am2);
        r.y = as_u32(as_h2(b) * am2);
        nf4_lookup4((d >> 4) & 0x0F0F0F0Fu, lut, a, b);
        r.z = as_u32(as_h2(a) * am2);
        r.w = as_u32(as_h2(b) * am2);
        return __builtin_bit_cast(f16x8, r);
    }
}

// Tile shape: WAVES waves x SPW strips per wave (rows = WAVES*SPW*16) x NTL n-tiles (tokens = NTL*16)
// 16 bytes per lane straight from global memory into LDS (no VGPR, no ds_write): lane l lands at lds + 16*l, so a
// B-fragment tile of the tiled operand (opd_off) arrives in LDS already in fragment order.  `lds` is wave-uniform.
__device__ __forceinline__ void glds16(const _Float16 *g, _Float16 *lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(uintptr_t)g,
                                     (__attribute__((address_space(3))) void *)(uintptr_t)(uint32_t)(uintptr_t)lds, 16, 0, 0);
}

// Epilogue of the tile kernels: acc[h][nt][r] = row (strip+h)*16 + (lane>>4)*4 + r, token t0 + nt*16 + (lane&15).
// The problem's parameters are read once, the activation switch is taken once per (strip, token tile) fragment and the bias / POST operands
// come as one 16-byte load each: the per-element form (a bias test, a six-way switch and three scalar loads per output, inlined SPW x NTL x 4
// times) was ~10 k instructions of straight-line code per weight format at the end of every block — 2/3 of the pipelined kernel's 536 KB.
template <int SPW, int NTL>
__device__ __forceinline__ void tg_epilogue(const GemmLaunch &L, const GemmProb &P, f32x4 (&acc)[SPW][NTL], int strip, int nstrips, int t0, int lane) {
    const int act = P.act, post = P.post, ldm = P.ldm, ldo = P.ldo, ldh = P.ldh;
    const float *bias = P.bias, *m0 = P.m0, *m1 = P.m1;
    float *out32 = P.out_f32;
    _Float16 *ohi = P.out_hi, *olo = P.out_lo;
#pragma unroll
    for (int h = 0; h < SPW; ++h) {
        if (strip + h < nstrips) {
            const int row0 = (strip + h) * 16 + (lane >> 4) * 4;
            float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bias) b4 = *(const float4 *)(bias + row0);
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) {
                const int t = t0 + nt * 16 + (lane & 15);
                if (t < L.T) {
                    float v[4] = {acc[h][nt][0], acc[h][nt][1], acc[h][nt][2], acc[h][nt][3]};
                    if (bias) { v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w; }
                    apply_act4(act, v);
                    if (post == POST_MUL) {
                        const float4 m = *(const float4 *)(m0 + (long)t * ldm + row0);
                        v[0] *= m.x; v[1] *= m.y; v[2] *= m.z; v[3] *= m.w;
                    } else if (post == POST_MIX) {
                        const float4 m = *(const float4 *)(m0 + (long)t * ldm + row0), n = *(const float4 *)(m1 + (long)t * ldm + row0);
                        v[0] = m.x + n.x * v[0]; v[1] = m.y + n.y * v[1]; v[2] = m.z + n.z * v[2]; v[3] = m.w + n.w * v[3];
                    }
                    if (out32) *(float4 *)(out32 + (long)t * ldo + row0) = make_float4(v[0], v[1], v[2], v[3]);
                    if (ohi) {
                        f16x4 hh, ll;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { _Float16 a, b; split_hilo(v[r], a, b); hh[r] = a; ll[r] = b; }
                        const long oo = opd_off(t, row0, ldh);
                        *(f16x4 *)(ohi + oo) = hh;
                        if (olo) *(f16x4 *)(olo + oo) = ll;
                    }
                }
            }
        }
    }
}

#endif
#if RWKV_PART_ON(2)
template <bool HILO, int WAVES, int SPW, int NTL, int KC, int FMT, bool GLDS>
__device__ __forceinline__ void tg_body(const GemmLaunch &L, const GemmProb &P, unsigned char *smem) {
    constexpr int BT = NTL * 16, THREADS = WAVES * 64, STRIPS = WAVES * SPW, TG_KC = KC;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nstrips = P.rows >> 4;
    const int ntt = (L.T + BT - 1) / BT;                          // token tiles
    // Tile of this block.  Workgroups go round-robin to the 8 XCDs (block b -> XCD b % 8, observed placement; only speed
    // depends on it) and every XCD has its own 4 MiB L2.  With the plain row-major numbering each XCD touches every weight
    // strip AND every token tile; instead the tiles are numbered so that the blocks an XCD receives, in its dispatch order,
    // walk one contiguous band of the row-major tile sequence: the XCD streams 1/8 of the weight strips once and re-reads
    // them for its token tiles out of its own L2.  (A bijection for any grid; L.xcd_map = 0 restores row-major.)
    int lb = (int)blockIdx.x - P.block_begin;
    const int nb = ((nstrips + STRIPS - 1) / STRIPS) * ntt;       // tiles of this problem
    // K split (P.ksb > 1, linear epilogues): copy kb of the tile grid walks the chunks [kb nchunk / ksb, (kb+1) nchunk / ksb) and
    // writes partial slab kb — small steps (a few hundred rows) leave Wo / Fv with fewer tiles than the chip has CUs
    const int kb = lb / nb;
    lb -= kb * nb;
    if (L.xcd_map) {
        const int k = lb & 7, j = lb >> 3;                        // XCD class relative to the problem's first block, rank in it
        int start = 0;
        for (int m = 0; m < k; ++m) start += (nb - m + 7) >> 3;   // tiles owned by the classes before this one
        lb = start + j;
    }
    // the banded sequence is row-tile major (token tile fastest: an XCD streams its eighth of the weights once and re-reads them from its L2 for every
    // token tile) or, L.xcd_map == 2, token-tile major (row tile fastest: an XCD keeps ITS token tiles' operand rows in its L2 and streams the weights
    // past them): measured better only for hi + lo operands at 2048 rows (rwkv_engine.cpp)
    const int nrb_ = nb / ntt;
    const int rb = L.xcd_map == 2 ? lb % nrb_ : lb / ntt, tt = L.xcd_map == 2 ? lb / nrb_ : lb - rb * ntt;
    const int strip = rb * STRIPS + wave * SPW;
    const int t0 = tt * BT;
    const int K = P.K;
    const int nchunk_all = (K + TG_KC - 1) / TG_KC;
    const int c0 = (int)((long)kb * nchunk_all / P.ksb), nchunk = (int)((long)(kb + 1) * nchunk_all / P.ksb);   // this copy: chunks [c0, nchunk)
    constexpr int PART = NTL * (KC / 32) * 512;                   // halfs per (buffer, hi|lo): [token tile][k-tile][lane][8], fragment order
    _Float16 *xs = (_Float16 *)smem;                              // [buf][hi|lo][token tile][k-tile][lane][8]
    constexpr int XP = BT * TG_KC / 8 / THREADS;                  // 16-byte pieces per thread per part
    Nf4Lut lut;
    if constexpr (FMT == W_NF4) lut = make_nf4_lut();

    f32x4 acc[SPW][NTL];
#pragma unroll
    for (int h = 0; h < SPW; ++h)
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) acc[h][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // X staging registers (one chunk ahead of the LDS buffer being multiplied).  The operand is B-tiled in HBM
    // (opd_off): piece p of a chunk = (tile p/64, lane p%64), so a wave reads one contiguous 1 KiB tile per step and
    // writes it to LDS in the same order (stage_store).  Token tiles beyond the step
    // are clamped to its last tile (their results are never stored); FULL chunks carry no predicates at all.
    struct XRegs { uint4 h[XP], l[HILO ? XP : 1]; };
    constexpr int KTC = TG_KC / 32;                               // k-tiles per chunk
    const int last_tile = (L.T - 1) >> 4;
    auto stage_load = [&](XRegs &x, int c, auto full) {
        constexpr bool FULL = decltype(full)::value;
        const int k0 = c * TG_KC;
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int p = tid + i * THREADS;
            const int tile = p >> 6, j = p & 63;
            const int ttile = min((t0 >> 4) + tile / KTC, last_tile), kt = tile % KTC;
            const int k = k0 + kt * 32;
            if (FULL || k < K) {
                const long xo = ((long)ttile * (P.ldx >> 5) + (k >> 5)) * 512 + j * 8;
                x.h[i] = *(const uint4 *)(P.xhi + xo);
                if constexpr (HILO) x.l[i] = *(const uint4 *)(P.xlo + xo);
            } else {
                x.h[i] = make_uint4(0, 0, 0, 0);
                if constexpr (HILO) x.l[i] = make_uint4(0, 0, 0, 0);
            }
        }
    };
    // the LDS image keeps the operand's own fragment order (piece p = tile p/64, lane p%64 lands at 16 p): contiguous
    // ds_write_b128 per wave, and the MFMA loop reads a fragment with one contiguous conflict-free ds_read_b128 (a padded
    // row-major image cost 40 % extra LDS cycles in bank conflicts under gfx950's b128 lane grouping)
    auto stage_store = [&](const XRegs &x, int buf) {
        _Float16 *bh = xs + (HILO ? buf * 2 : buf) * PART;
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int p = tid + i * THREADS;
            *(uint4 *)(bh + p * 8) = x.h[i];
            if constexpr (HILO) *(uint4 *)(bh + PART + p * 8) = x.l[i];
        }
    };
    // GLDS staging: tile i = (token tile i / KTC, k-tile i % KTC) of the chunk is one 1 KiB direct-to-LDS load, issued by
    // wave i % WAVES; the LDS image [tile][lane][8] IS the B-fragment order, so the MFMA loop reads it with conflict-free
    // contiguous ds_read_b128 and no staging registers or ds_write pass exist.
    auto glds_issue = [&](int c, int buf, auto full) {
        constexpr bool FULL = decltype(full)::value;
        const int k0 = c * TG_KC;
        _Float16 *bh = xs + (HILO ? buf * 2 : buf) * PART;
#pragma unroll
        for (int j = 0; j < (NTL * KTC + WAVES - 1) / WAVES; ++j) {
            const int i = wave + j * WAVES;
            if (i < NTL * KTC) {
                const int tt = i / KTC, kt = i % KTC, k = k0 + kt * 32;
                if (FULL || k < K) {
                    const int ttile = min((t0 >> 4) + tt, last_tile);
                    const long xo = ((long)ttile * (P.ldx >> 5) + (k >> 5)) * 512 + lane * 8;
                    glds16(P.xhi + xo, bh + i * 512);
                    if constexpr (HILO) glds16(P.xlo + xo, bh + PART + i * 512);
                }
            }
        }
    };
    auto mma_chunk = [&](const TRound<FMT, SPW, KC> &w, int c, auto full) {
        constexpr bool FULL = decltype(full)::value;
        const int k0 = c * TG_KC;
        const _Float16 *bh = xs + (HILO ? ((c - c0) & 1) * 2 : ((c - c0) & 1)) * PART;
#pragma unroll
        for (int ks = 0; ks < KC / 32; ++ks) {
            if (FULL || k0 + ks * 32 < K) {
                f16x8 xv[NTL], xc[HILO ? NTL : 1];                  // all B fragments of the k-step in flight at once
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt) {
                    const int off = ((nt * KTC + ks) * 64 + lane) * 8;
                    xv[nt] = *(const f16x8 *)(bh + off);
                    if constexpr (HILO) xc[nt] = *(const f16x8 *)(bh + PART + off);
                }
                f16x8 af[SPW];
#pragma unroll
                for (int h = 0; h < SPW; ++h) af[h] = tg_frag<FMT, SPW, KC>(w, h, ks, k0, lut);
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt) {
#pragma unroll
                    for (int h = 0; h < SPW; ++h) {
                        acc[h][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[h], xv[nt], acc[h][nt], 0, 0, 0);
                        if constexpr (HILO) acc[h][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[h], xc[nt], acc[h][nt], 0, 0, 0);
                    }
                }
            }
        }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;

    // software pipeline: while chunk c is multiplied, the weights and the X rows of chunk c+1 are in flight to
    // registers (unpredicated in the steady state so the compiler can count them); they go to the other LDS buffer
    // after the MFMAs; one barrier per chunk.  (A distance-2 variant was measured slower: the register-set rotation
    // forces the waits anyway and the extra sets spill.)
    const int nfull = K / TG_KC;                                   // chunks entirely inside K
    TRound<FMT, SPW, KC> cur, nxt;
    if constexpr (GLDS) {
        // chunk c+1's X tiles are issued into the other buffer at the top of iteration c (every wave passed the barrier
        // that ended iteration c-1, so nobody still reads it); the barrier at the bottom (with the vmcnt(0) the compiler
        // puts in front of it) publishes them
        if (nfull > c0) { tg_load<FMT, SPW, KC, true>(cur, P, strip, nstrips, c0 * TG_KC, lane); glds_issue(c0, 0, T_{}); }
        else { tg_load<FMT, SPW, KC, false>(cur, P, strip, nstrips, c0 * TG_KC, lane); glds_issue(c0, 0, F_{}); }
        __syncthreads();
        for (int c = c0; c < nchunk; ++c) {
            if (c + 1 < nfull && c + 1 < nchunk) {
                tg_load<FMT, SPW, KC, true>(nxt, P, strip, nstrips, (c + 1) * TG_KC, lane);
                glds_issue(c + 1, (c + 1 - c0) & 1, T_{});
                mma_chunk(cur, c, T_{});
                __syncthreads();
                cur = nxt;
            } else {
                if (c + 1 < nchunk) {
                    tg_load<FMT, SPW, KC, false>(nxt, P, strip, nstrips, (c + 1) * TG_KC, lane);
                    glds_issue(c + 1, (c + 1 - c0) & 1, F_{});
                }
                if (c < nfull) mma_chunk(cur, c, T_{}); else mma_chunk(cur, c, F_{});
                if (c + 1 < nchunk) {
                    __syncthreads();
                    cur = nxt;
                }
            }
        }
    } else {
    XRegs xa;
    if (nfull > c0) { tg_load<FMT, SPW, KC, true>(cur, P, strip, nstrips, c0 * TG_KC, lane); stage_load(xa, c0, T_{}); }
    else { tg_load<FMT, SPW, KC, false>(cur, P, strip, nstrips, c0 * TG_KC, lane); stage_load(xa, c0, F_{}); }
    stage_store(xa, 0);
    __syncthreads();
    for (int c = c0; c < nchunk; ++c) {
        if (c + 1 < nfull && c + 1 < nchunk) {                     // steady state: everything unpredicated
            tg_load<FMT, SPW, KC, true>(nxt, P, strip, nstrips, (c + 1) * TG_KC, lane);
            stage_load(xa, c + 1, T_{});
            mma_chunk(cur, c, T_{});
            stage_store(xa, (c + 1 - c0) & 1);
            __syncthreads();
            cur = nxt;
        } else {
            if (c + 1 < nchunk) {
                tg_load<FMT, SPW, KC, false>(nxt, P, strip, nstrips, (c + 1) * TG_KC, lane);
                stage_load(xa, c + 1, F_{});
            }
            if (c < nfull) mma_chunk(cur, c, T_{}); else mma_chunk(cur, c, F_{});
            if (c + 1 < nchunk) {
                stage_store(xa, (c + 1 - c0) & 1);
                __syncthreads();
                cur = nxt;
            }
        }
    }
    }
    if (P.ksb > 1) {                                              // partial slab kb (host: out_f32 only, linear epilogue)
        GemmProb Q = P;
        Q.out_f32 = P.out_f32 + (long)kb * P.partial_stride;
        tg_epilogue<SPW, NTL>(L, Q, acc, strip, nstrips, t0, lane);
    } else {
        tg_epilogue<SPW, NTL>(L, P, acc, strip, nstrips, t0, lane);
    }
}

template <bool HILO, int WAVES, int SPW, int NTL, int KC, bool GLDS>
__global__ __launch_bounds__(WAVES * 64) void gemm_tile_kernel(const GemmLaunch L) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int pi = 0;
    for (int i = 1; i < L.nprob; ++i)
        if ((int)blockIdx.x >= L.p[i].block_begin) pi = i;
    const GemmProb &P = L.p[pi];
    if (P.fmt == W_F16) tg_body<HILO, WAVES, SPW, NTL, KC, W_F16, GLDS>(L, P, smem);
    else if (P.fmt == W_INT8) tg_body<HILO, WAVES, SPW, NTL, KC, W_INT8, GLDS>(L, P, smem);
    else tg_body<HILO, WAVES, SPW, NTL, KC, W_NF4, GLDS>(L, P, smem);
}

#endif
#if RWKV_PART_ON(2) || RWKV_PART_ON(5)      // pipelined-kernel helpers: inline-asm loads / LDS-DMA / counted waits, weight register sets
// =====================================================================================
// (Round 3 built the 256-row form of this kernel for the quantised formats — four strips per wave, 128 accumulators, one block per CU,
// a third fewer bytes through the CU's L2 port per flop — and measured it: 395-563 TFLOP/s against 658-861 for this one on the same
// matrices, V6-3B Int8 prefill 72.0 -> 59.0 k tok/s; with one wave per SIMD nothing covers the dequantisation and the LDS reads between
// MFMAs.  profiles/r3_exp_tile_256x128_pipelined.log; the code was removed.)
// Pipelined tile kernel (shape 10): 128 rows x 128 tokens per block, 4 waves, wave w owns strips {2w, 2w+1} x all 8 token tiles
// (64 accumulator registers).  Per k-step the block moves (128 + 128) x 64 B through the CU's L2 port for 64 MFMAs — half
// of what the 64x64 shapes move per MFMA, which is what bounds them (the port sustains ~56 B/clk, they need 128 B/clk at full
// MFMA rate) — and everything is prefetched further ahead than one chunk:
//   * X: ring of T3_NB = 4 LDS stages of [8 token tiles][2 k-steps] fragment tiles (64 k, 16 KiB), filled by global_load_lds
//     three stages ahead.  A stage is published by a COUNTED s_waitcnt vmcnt(N) + raw s_barrier at the end of the stage before
//     it is read (N = the VMEM instructions issued after that stage's DMAs: the two younger stages' 4 + 4 DMAs and one weight
//     group), so neither the younger DMAs nor the weight prefetch are drained at a barrier — __syncthreads() would wait vmcnt(0).
//     It is restaged one barrier after its last ds_read (lgkmcnt(0) in front of that barrier).
//   * weights: HBM/L2 -> registers in groups of 128 k (two stages), three register sets, loaded two groups (4 stages) ahead.
// Host guarantees K % 128 == 0 and a non-hi/lo operand.  vmcnt is in-order, so a count that is too SMALL only waits longer; the
// weight-group constant below is therefore the number of 16-byte tile loads (scale loads not counted).
// =====================================================================================
constexpr int T3_NB = 4;
// Every VMEM instruction of the K loop is inline asm: hipcc then keeps no vmcnt bookkeeping for the loop (its own bookkeeping
// degrades to vmcnt(0) as soon as a load sits behind a branch or an LDS-DMA is pending), and the counted waits below are exact.
// A register written by such a load is used only after (1) a counted wait that retires the load and (2) t3_arrived(), an empty
// asm that makes the compiler treat the register as produced at that point, so no use can be scheduled ahead of the wait.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// Addresses are SGPR base + 32-bit VGPR byte offset (+ immediate): the lane part (strip, lane) is computed once per block, the part that
// moves along K is wave-uniform and lives in scalar registers — no 64-bit vector add per load, and nothing of the problem record is
// re-read inside the loop.  (Round 5, s_memtime stamps: with per-lane 64-bit pointers rebuilt from the GemmProb fields at every weight
// group — two scalar loads and a wait for them, ~25 scalar and 6 vector address instructions — issuing a stage's loads took 300 cycles
// of the 1,200 a stage of a block alone on its CU lasts, 600 of 1,780 with two blocks on the CU: profiles/r5_exp_tile3_stage_trace.log.)
template <int OFF> __device__ __forceinline__ void t3_ld16(u32x4 &d, unsigned voff, const void *sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(d) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
__device__ __forceinline__ void t3_ld8(u32x2 &d, unsigned voff, const void *sbase) {
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(d) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void t3_dma16(unsigned voff, const void *sbase, unsigned lds_byte) {      // lane l -> LDS byte lds_byte + 16 l; M0 preserved
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
}
template <int N> __device__ __forceinline__ void t3_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}
// ds_read_b128 whose completion the compiler does not track (see tg3_body's stage): paired with t3_lgkm_wait<N>, which retires
// everything but the N youngest LGKM operations and hands the register back to the compiler
template <int OFF> __device__ __forceinline__ void t3_lds16(f16x8 &d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int N> __device__ __forceinline__ void t3_lgkm_wait(f16x8 &x) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x) : "n"(N)); }
// the NTW fragment reads of k-step Q of a stage ([token tile][k-step][lane][8] halfs: tile nt, k-step q at byte (2 nt + q) * 1024)
template <int NTW, int Q, int NT = 0> __device__ __forceinline__ void t3_reads(f16x8 (&x)[NTW], unsigned xa) {
    if constexpr (NT < NTW) {
        t3_lds16<(NT * 2 + Q) * 1024>(x[NT], xa);
        t3_reads<NTW, Q, NT + 1>(x, xa);
    }
}
// first k-step: the read of the second k-step's tile nt goes out in front of the MFMA pair on tile nt — NTW reads in flight behind it
template <int NTW, int SPW, int NT = 0>
__device__ __forceinline__ void t3_head(f32x4 (&acc)[SPW][NTW], const f16x8 (&a)[SPW], f16x8 (&x0)[NTW], f16x8 (&x1)[NTW], unsigned xa) {
    if constexpr (NT < NTW) {
        t3_lds16<(NT * 2 + 1) * 1024>(x1[NT], xa);
        t3_lgkm_wait<NTW>(x0[NT]);
#pragma unroll
        for (int h = 0; h < SPW; ++h) acc[h][NT] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[h], x0[NT], acc[h][NT], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        t3_head<NTW, SPW, NT + 1>(acc, a, x0, x1, xa);
    }
}
// second k-step: tile nt's fragment has NTW - 1 - nt younger reads behind it
template <int NTW, int SPW, int NT = 0>
__device__ __forceinline__ void t3_tail(f32x4 (&acc)[SPW][NTW], const f16x8 (&a)[SPW], f16x8 (&x)[NTW]) {
    if constexpr (NT < NTW) {
        t3_lgkm_wait<NTW - 1 - NT>(x[NT]);
#pragma unroll
        for (int h = 0; h < SPW; ++h) acc[h][NT] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[h], x[NT], acc[h][NT], 0, 0, 0);
        t3_tail<NTW, SPW, NT + 1>(acc, a, x);
    }
}
// weight group of 128 k for SPW strips: tile loads + (quantised) one scale word per strip; NLOAD = VMEM instructions issued
template <int FMT, int SPW> struct T3Set {
    static constexpr int NQ = 4 / Fmt<FMT>::KS, NLOAD = SPW * NQ + (FMT == W_F16 ? 0 : SPW);
    u32x4 q[SPW][NQ];
    u32x2 s[SPW];
};
// the lane's byte offsets into the weight tiles / scale words of its SPW strips (K-independent), and the group load: wg / sg = the matrix'
// and scale bases advanced to the group's k (wave-uniform)
template <int SPW> struct T3Off { unsigned w[SPW], s[SPW]; };
template <int FMT, int SPW, int J = 0>
__device__ __forceinline__ void t3_load_tiles(T3Set<FMT, SPW> &w, int h, unsigned voff, const void *wg) {
    if constexpr (J < T3Set<FMT, SPW>::NQ) {
        t3_ld16<J * 1024>(w.q[h][J], voff, wg);
        t3_load_tiles<FMT, SPW, J + 1>(w, h, voff, wg);
    }
}
template <int FMT, int SPW>
__device__ __forceinline__ void t3_load(T3Set<FMT, SPW> &w, const T3Off<SPW> &o, const void *wg, const void *sg) {
#pragma unroll
    for (int h = 0; h < SPW; ++h) {
        t3_load_tiles<FMT, SPW>(w, h, o.w[h], wg);
        if constexpr (FMT != W_F16) t3_ld8(w.s[h], o.s[h], sg);
    }
}
template <int FMT, int SPW> __device__ __forceinline__ void t3_arrived(T3Set<FMT, SPW> &w) {
#pragma unroll
    for (int h = 0; h < SPW; ++h) {
#pragma unroll
        for (int j = 0; j < T3Set<FMT, SPW>::NQ; ++j) asm volatile("" : "+v"(w.q[h][j]));
        if constexpr (FMT != W_F16) asm volatile("" : "+v"(w.s[h]));
    }
}
// A fragment of k-step ks (0..3) of the group at k0: same arithmetic as tg_frag, on a T3Set
template <int FMT, int SPW>
__device__ __forceinline__ f16x8 t3_frag(const T3Set<FMT, SPW> &w, int h, int ks, int k0, const Nf4Lut &lut) {
    TRound<FMT, 1, 128> r;
#pragma unroll
    for (int j = 0; j < T3Set<FMT, SPW>::NQ; ++j) r.q[0][j] = w.q[h][j];
    r.s[0] = make_uint2(w.s[h].x, w.s[h].y);
    return tg_frag<FMT, 1, 128>(r, 0, ks, k0, lut);
}

#endif
#if RWKV_PART_ON(2)
#ifdef RWKV_T3_TRACE
// dev build (scripts/build_variant.py ... -DRWKV_T3_TRACE): shader-clock stamps at the points of a stage where the LGKM counter is drained
// anyway (s_memtime answers through it), summed per wave and printed by a few waves — where a stage's cycles go
__device__ unsigned g_t3_printed = 0;
#define T3_STAMP(var) do { __builtin_amdgcn_sched_barrier(0); var = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define T3_STAMP(var) do { } while (0)
#endif
// NTL = token tiles per block: 8 = the 128 x 128 tile; 4 = 128 rows x 64 tokens for steps of a few hundred rows (round 4: at 256 rows
// the 128-token tile leaves 160 blocks for a 10304-row launch; this one 324, each moving half the operand).
// What a stage costs (round 5, s_memtime stamps, profiles/r5_exp_tile3_stage_trace.log): a wave alone on its SIMD pays the SUM of what its
// instructions cost to issue — 16 MFMAs 270 cycles, two LDS-DMAs and the weight loads 230, ~40 dequantisation VALU + eight fragment reads +
// waits 350, the barrier 100: 1,200 per stage — and two waves on a SIMD overlap them (1,780 for a stage of each).  At 256 rows a 10304-row
// launch is 324 tiles on 256 CUs and takes as long as a CU that got two.  Built on that and measured (profiles/r5_exp_tile3_192x64.log):
// 192 x 64 tiles, one per CU, three strips per wave (33.3 us against 33.6: a block's time follows its MFMA count), and the same tile in
// eight waves over two K halves (29.9 against 32.6 isolated, nothing in the model: 49.1 -> 48.8 k tok/s).  Both removed.
// SPW = strips per wave (2 in every shape the engine uses).
template <int FMT, int NTL, int SPW = 2>
__device__ __forceinline__ void tg3_body(const GemmLaunch &L, const GemmProb &P, unsigned char *smem) {
    constexpr int BT = NTL * 16, WAVES = 4, STRIPS = WAVES * SPW, NA = T3Set<FMT, SPW>::NLOAD, NTW = NTL;
    constexpr int DPW = 2 * NTL / WAVES;                          // X tiles (DMAs) per wave per stage: 2 NTL tiles over the block's four waves
    constexpr int STAGE_HALFS = NTL * 2 * 512;                    // [token tile][k-step][lane][8]
    using Set = T3Set<FMT, SPW>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nstrips = P.rows >> 4;
    const int ntt = (L.T + BT - 1) / BT;
    int lb = (int)blockIdx.x - P.block_begin;
    // K split (linear epilogues only, P.ksb > 1): the launch's tiles are replicated ksb times, copy kb walks the 128-k groups
    // [kb G / ksb, (kb+1) G / ksb) of the G = K / 128 and writes partial slab kb, which the next row kernel sums — a launch of 320
    // tiles (Wo, Fv of the 3 B model at 2048 rows) leaves 192 of the kernel's 512 slots empty and costs a whole round anyway; as
    // 3 x 320 it fills 94 % of two rounds that are a third as long.
    const int nb = ((nstrips + STRIPS - 1) / STRIPS) * ntt;       // tiles of one copy
    const int kb = lb / nb;
    lb -= kb * nb;
    if (L.xcd_map) {                                              // XCD-banded tile numbering, as in tg_body
        const int k = lb & 7, j = lb >> 3;
        int start = 0;
        for (int m = 0; m < k; ++m) start += (nb - m + 7) >> 3;
        lb = start + j;
    }
    // the banded sequence is row-tile major (token tile fastest: an XCD streams its eighth of the weights once and re-reads them from its L2 for every
    // token tile) or, L.xcd_map == 2, token-tile major (row tile fastest: an XCD keeps ITS token tiles' operand rows in its L2 and streams the weights
    // past them): measured better only for hi + lo operands at 2048 rows (rwkv_engine.cpp)
    const int nrb_ = nb / ntt;
    const int rb = L.xcd_map == 2 ? lb % nrb_ : lb / ntt, tt = L.xcd_map == 2 ? lb / nrb_ : lb - rb * ntt;
    const int strip = rb * STRIPS + wave * SPW;
    const int t0 = tt * BT;
    const int G = P.K >> 7, g0 = (int)((long)kb * G / P.ksb), g1 = (int)((long)(kb + 1) * G / P.ksb);
    const int kofs = g0 * 128;                                    // first k of this copy
    const int nsc = g1 - g0, nst = nsc * 2;
    const _Float16 *xs = (const _Float16 *)smem;                  // [T3_NB][token tile 0..7][k-step 0..1][lane][8]
    const unsigned xs_byte = (unsigned)(uintptr_t)smem;           // LDS byte address of the ring (dynamic LDS starts at 0 here, kept general)
    const int last_tile = (L.T - 1) >> 4;
    Nf4Lut lut;
    if constexpr (FMT == W_NF4) lut = make_nf4_lut();

    f32x4 acc[SPW][NTW];
#pragma unroll
    for (int h = 0; h < SPW; ++h)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[h][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // the DPW X tiles of a stage this wave fetches: i = m*WAVES + wave -> (token tile i >> 1, k-step i & 1); tiles past the step
    // are clamped to its last tile (their columns are never stored)
    // (byte offsets of the lane inside the operand, 32-bit: the operand is T x K halfs; the stage's k moves in the scalar base)
    unsigned xoff[DPW];
#pragma unroll
    for (int m = 0; m < DPW; ++m) {
        const int i = m * WAVES + wave;
        const int ttile = min((t0 >> 4) + (i >> 1), last_tile);
        xoff[m] = (unsigned)(((ttile * (P.ldx >> 5) + (i & 1)) * 512 + lane * 8) * 2);
    }
    const char *const xbase = (const char *)(P.xhi + (long)(kofs >> 5) * 512);
    auto dma = [&](int s) {
        const unsigned dst = xs_byte + (unsigned)(((s & (T3_NB - 1)) * STAGE_HALFS + wave * 512) * 2);
        const char *xs_g = xbase + (long)s * 2048;                  // stage s: 64 k = two k-steps of 1 KiB tiles
#pragma unroll
        for (int m = 0; m < DPW; ++m) t3_dma16(xoff[m], xs_g, dst + m * WAVES * 1024);
    };
    // the weights: lane offsets once, the group's k in the scalar bases
    T3Off<SPW> woff;
    {
        constexpr int SH = Fmt<FMT>::SH;
        const int KT = P.K >> SH, KG = P.K >> 8;
#pragma unroll
        for (int h = 0; h < SPW; ++h) {
            const int sidx = min(strip + h, nstrips - 1);
            woff.w[h] = (unsigned)(sidx * KT * 64 + lane) * 16u;
            woff.s[h] = (unsigned)(sidx * KG * 16 + (lane & 15)) * 8u;
        }
    }
    const char *const wbase = (const char *)P.W, *const sbase = (const char *)P.S;
    auto wload = [&](Set &w, int k0) {                               // weight group at absolute k0 (a multiple of 128)
        t3_load<FMT, SPW>(w, woff, wbase + (long)(k0 >> Fmt<FMT>::SH) * 1024, sbase + (long)(k0 >> 8) * 128);
    };
#ifdef RWKV_T3_TRACE
    unsigned long long tr_t0 = 0, tr_t1 = 0, tr_t2 = 0, tr_t3 = 0, tr_issue = 0, tr_stage = 0, tr_wait = 0, tr_begin = 0, tr_head = 0, tr_mid = 0;
#endif
    auto stage = [&](const Set &w, int s, auto half) {
        constexpr int H = decltype(half)::value;
        const int k0 = kofs + (s >> 1) * 128;
        // Fragment reads as inline asm with counted lgkmcnt waits (LDS returns in order; a scalar load the compiler may have in
        // flight only makes a counted wait stricter): the stage's first NTW reads go out together, the dequantisation of the first
        // k-step runs under them, and the read of the second k-step's tile nt is issued in front of the MFMA pair on the first
        // k-step's tile nt — NTW reads are always in flight behind the one being waited for.  (Left to itself hipcc sinks every
        // ds_read to its use and waits lgkmcnt(0) per MFMA pair.)
        const unsigned xa = xs_byte + (unsigned)(((s & (T3_NB - 1)) * STAGE_HALFS + lane * 8) * 2);
        f16x8 x0[NTW], x1[NTW];
        t3_reads<NTW, 0>(x0, xa);
        __builtin_amdgcn_sched_barrier(0);
        f16x8 a0[SPW], a1[SPW];
#pragma unroll
        for (int h = 0; h < SPW; ++h) a0[h] = t3_frag<FMT, SPW>(w, h, H * 2, k0, lut);
        __builtin_amdgcn_sched_barrier(0);
        t3_head<NTW, SPW>(acc, a0, x0, x1, xa);
#if defined(RWKV_T3_TRACE) && RWKV_T3_TRACE >= 2
        { unsigned long long m; T3_STAMP(m); tr_head += m - tr_t1; tr_mid = m; }       // (drains the second k-step's reads: a perturbation of its own)
#endif
#pragma unroll
        for (int h = 0; h < SPW; ++h) a1[h] = t3_frag<FMT, SPW>(w, h, H * 2 + 1, k0, lut);
        __builtin_amdgcn_sched_barrier(0);
        t3_tail<NTW, SPW>(acc, a1, x1);
    };
    // End of stage s: stage s+1 must be complete in LDS for every wave, and (s odd) the weight group of the next two stages in
    // this wave's registers.  Steady state (s + 4 < nst): younger than both are exactly DMA(s+2), DMA(s+3) and ONE weight group
    // (issued at the start of whichever of s-1, s is even) = 2 DPW + NA instructions; the last four stages drain instead.
    auto publish = [&](int s) {
        if (s + 4 < nst) t3_wait_barrier<2 * DPW + NA>();
        else if (s + 1 < nst) t3_wait_barrier<0>();
    };
#ifdef RWKV_T3_TRACE
    T3_STAMP(tr_begin);
#endif
    auto super = [&](Set &cur, Set &refill, int sc) {
        int s = 2 * sc;
        T3_STAMP(tr_t0);
        t3_arrived<FMT, SPW>(cur);
        if (sc + 2 < nsc) wload(refill, kofs + (sc + 2) * 128);
        if (s + 3 < nst) dma(s + 3);
        T3_STAMP(tr_t1);
        stage(cur, s, std::integral_constant<int, 0>{});
        T3_STAMP(tr_t2);
        publish(s);
        T3_STAMP(tr_t3);
#ifdef RWKV_T3_TRACE
        tr_issue += tr_t1 - tr_t0; tr_stage += tr_t2 - tr_t1; tr_wait += tr_t3 - tr_t2;
#endif
        s += 1;
        T3_STAMP(tr_t0);
        if (s + 3 < nst) dma(s + 3);
        T3_STAMP(tr_t1);
        stage(cur, s, std::integral_constant<int, 1>{});
        T3_STAMP(tr_t2);
        publish(s);
        T3_STAMP(tr_t3);
#ifdef RWKV_T3_TRACE
        tr_issue += tr_t1 - tr_t0; tr_stage += tr_t2 - tr_t1; tr_wait += tr_t3 - tr_t2;
#endif
    };

    Set a0, a1, a2;
    wload(a0, kofs);
    wload(a1, kofs + (nsc > 1 ? 128 : 0));
    wload(a2, kofs);              // placeholder contents (every register defined); refilled at sc = 0
    dma(0);
    if (nst > 1) dma(1);
    if (nst > 2) dma(2);
    if (nst > 2) t3_wait_barrier<2 * DPW>(); else t3_wait_barrier<0>();   // DMA(0) and the three weight groups have landed
    for (int sc = 0; sc < nsc; sc += 3) {
        super(a0, a2, sc);
        if (sc + 1 < nsc) super(a1, a0, sc + 1);
        if (sc + 2 < nsc) super(a2, a1, sc + 2);
    }
#ifdef RWKV_T3_TRACE
    {
        unsigned long long tr_end;
        T3_STAMP(tr_end);
        if (lane == 0 && ((int)blockIdx.x == 0 || (int)blockIdx.x == 100 || (int)blockIdx.x == L.total_blocks - 1) && atomicAdd(&g_t3_printed, 1u) < 24u)
            printf("t3trace blk %d/%d wave %d fmt %d spw %d nst %d | loop %llu = issue %llu + stage %llu + wait/barrier %llu (cycles; per stage %llu; first k-step of the stages %llu)\n",
                   (int)blockIdx.x, L.total_blocks, wave, FMT, SPW, nst, tr_end - tr_begin, tr_issue, tr_stage, tr_wait, (tr_end - tr_begin) / (nst > 0 ? nst : 1), tr_head);
    }
#endif
    if (P.ksb > 1) {                                              // partial slab kb (host: out_f32 only, linear epilogue)
        GemmProb Q = P;
        Q.out_f32 = P.out_f32 + (long)kb * P.partial_stride;
        tg_epilogue<SPW, NTW>(L, Q, acc, strip, nstrips, t0, lane);
    } else {
        tg_epilogue<SPW, NTW>(L, P, acc, strip, nstrips, t0, lane);
    }
}

template <int NTL>
__global__ __launch_bounds__(256, 2) void gemm_tile3_kernel(const GemmLaunch L) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int pi = 0;
    for (int i = 1; i < L.nprob; ++i)
        if ((int)blockIdx.x >= L.p[i].block_begin) pi = i;
    const GemmProb &P = L.p[pi];
    if (P.fmt == W_F16) tg3_body<W_F16, NTL>(L, P, smem);
    else if (P.fmt == W_INT8) tg3_body<W_INT8, NTL>(L, P, smem);
    else tg3_body<W_NF4, NTL>(L, P, smem);
}
bool gemm_tile3_supported(bool hilo, int K) { return !hilo && K % 128 == 0; }

#endif
#if RWKV_PART_ON(5)
// =====================================================================================
// Software-pipelined tile kernel for hi + lo operands (shape 12; round 6): the prefill GEMM of Precision::Fp32 and of the launch classes
// Precision::Fp16 promotes (rwkv_engine.cpp OpdClass).  Same tile, ring and weight path as gemm_tile3_kernel<4> — 128 rows x 64 tokens, 4 waves,
// wave w owns strips {2w, 2w+1} x all token tiles, X in a ring of four LDS stages filled by global_load_lds, weights in three register sets
// loaded two 128-k groups ahead — with a stage that holds the hi AND the lo tiles of its tokens: every A fragment feeds both (the
// dequantisation is paid once, the weights are fetched once), which is why a hi + lo launch costs 1.35 x a plain one here and 1.85 x on the
// 64x64 shape (V6-3B r/k/v/g Int8 at 256 rows: 46.8 us against 76.5; at 2048 rows 262 against 547; profiles/r6_exp_tile4.log).
// The K loop is a software pipeline over K-STEPS (32 k) instead of a sequence of phases per stage: every MFMA of k-step q is followed by one
// slice of the work for k-step q + 1 and for the stages ahead —
//   * the ds_read_b128 of one B fragment of k-step q + 1 (slots 0 .. NBT-1), into the other of two fragment register sets;
//   * its share of the dequantisation of the A fragments of k-step q + 1 (`t4_unit`), into the other of two A register sets;
//   * in odd k-steps, behind the reads: the LDS-DMAs of stage s + 4 (the slot of stage s, which every wave finished reading before the
//     barrier in the middle of stage s) and, every second stage, the loads of weight group g + 3 into the set group g just left.
// `__builtin_amdgcn_sched_barrier(0)` after every slot and volatile-asm pins on the VALU results fix the interleave (checked in the ISA).
// One barrier per stage, between its two k-steps:   s_waitcnt vmcnt(2 DPW + NA) lgkmcnt(0); s_barrier   publishes stage s + 1 (whose first
// fragments are read during the stage's second k-step) — the VMEM instructions issued after DMA(s + 1) are DMA(s + 2), DMA(s + 3) and
// exactly one weight group; the last four stages drain with vmcnt(0).
// An accumulator receives, per k-step, its hi product and then its lo product, k-steps in order — the order of every other tile kernel's
// hi + lo form: results are BIT-identical to the 64x64 shape's (tests/test_gpu_bench_paths.py race screen).
// What the pipelining itself is worth was measured on the plain (non-hi/lo) form of this body against gemm_tile3 (profiles/r6_exp_tile4.log):
// nothing — 33.6 us against 34.7 (r/k/v/g Int8, 256 rows), 170 against 167 at 2048 rows.  Ablations of the same build: without the MFMAs the
// loop still takes 29.1 us, without any VMEM instruction inside the loop 25.9 (fp16 weights: 21.8).  A stage is not the sum of a wave's issue
// slots (round 5's reading) but the VMEM instructions' issue time — each LDS-DMA / tile load holds the in-order wave for 60-185 cycles
// (MI355X_MICROARCH.md), MFMAs behind it included — plus the LDS reads and waits; re-ordering the same instructions inside the same wave moves
// nothing.  The plain form was therefore dropped again (gemm_tile3 stays); what would move it is taking the VMEM instructions out of the
// MFMA waves (loader waves + weights through LDS).
// =====================================================================================
template <class F, int... I> __device__ __forceinline__ void t4_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void t4_for(F &&f) { if constexpr (N > 0) t4_for_impl(f, std::make_integer_sequence<int, N>{}); }

// per-(strip, weight group) dequantisation constants, computed when the pipeline moves on to the group's register set
template <int FMT> struct T4Scale { u32 v[2][2]; };
template <int FMT>
__device__ __forceinline__ void t4_scales(T4Scale<FMT> &sc, const T3Set<FMT, 2> &w, int gabs /* absolute index of the 128-k group */) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if constexpr (FMT == W_INT8) {                              // {a, b} of the 128-block: group parity selects the word of the 256-k scale pair
            const f16x2 abh = as_h2((gabs & 1) ? w.s[h].y : w.s[h].x);
            sc.v[h][0] = as_u32((f16x2){abh[0], abh[0]});
            sc.v[h][1] = as_u32((f16x2){abh[1], abh[1]});
        } else if constexpr (FMT == W_NF4) {                        // absmax of the group's two 64-blocks
            const f16x2 sh = as_h2((gabs & 1) ? w.s[h].y : w.s[h].x);
            sc.v[h][0] = as_u32((f16x2){sh[0], sh[0]});
            sc.v[h][1] = as_u32((f16x2){sh[1], sh[1]});
        } else {
            sc.v[h][0] = sc.v[h][1] = 0;
        }
    }
}
// One dequantisation unit = two of the four registers of an A fragment (UNITS = 2 per fragment).  Same arithmetic as tg_frag, bit for bit.
// Every result is passed through a volatile asm: a pure VALU chain has no position of its own — the DAG scheduler sinks it to its use, i.e. into
// the NEXT k-step in front of the MFMA that reads it, which is the phase structure this kernel exists to avoid.  Volatile asm statements keep
// their order, so the unit stays between the ds_read / wait statements of its slot.  Int8 is written out as asm: the two registers' chains
// (v_perm byte -> 0x6400|q, v_pk_add -1024, v_pk_fma a q + b: the instructions hipcc emits for dq8) interleaved, which also removes the wait
// state the packed ops need between a write and its dependent read (hipcc pads each chain with s_nop when it schedules them one after the other).
template <int FMT> struct T4Dq { static constexpr int UNITS = 2; };
__device__ __forceinline__ u32 t4_pin(u32 v) { asm volatile("" : "+v"(v)); return v; }
template <int FMT, int KS, int U>
__device__ __forceinline__ void t4_unit(u32x4 &dst, const T3Set<FMT, 2> &w, const T4Scale<FMT> &sc, const Nf4Lut &lut, int h) {
    if constexpr (FMT == W_F16) {
        dst[U * 2] = t4_pin(w.q[h][KS][U * 2]);
        dst[U * 2 + 1] = t4_pin(w.q[h][KS][U * 2 + 1]);
    } else if constexpr (FMT == W_INT8) {
        const u32x4 q = w.q[h][KS >> 1];
        const u32 d = U == 0 ? ((KS & 1) ? q.z : q.x) : ((KS & 1) ? q.w : q.y);
        u32 r0, r1;
        asm volatile("v_perm_b32 %0, %3, %2, %4\n\t"
                     "v_perm_b32 %1, %3, %2, %5\n\t"
                     "v_pk_add_f16 %0, %0, %6 op_sel_hi:[1,0]\n\t"
                     "v_pk_add_f16 %1, %1, %6 op_sel_hi:[1,0]\n\t"
                     "v_pk_fma_f16 %0, %0, %7, %8\n\t"
                     "v_pk_fma_f16 %1, %1, %7, %8\n\t"
                     "s_nop 0"
                     : "=&v"(r0), "=&v"(r1)
                     : "v"(d), "s"(0x64646464u), "v"(0x04010400u), "v"(0x04030402u), "s"(0xE400u), "v"(sc.v[h][0]), "v"(sc.v[h][1]));
        dst[U * 2] = r0;
        dst[U * 2 + 1] = r1;
    } else {
        const u32x4 q = w.q[h][0];
        const u32 d = KS == 0 ? q.x : KS == 1 ? q.y : KS == 2 ? q.z : q.w;
        const f16x2 am2 = as_h2(sc.v[h][(KS >> 1) & 1]);
        u32 a, b;
        nf4_lookup4((U ? (d >> 4) : d) & 0x0F0F0F0Fu, lut, a, b);
        dst[U * 2] = t4_pin(as_u32(as_h2(a) * am2));
        dst[U * 2 + 1] = t4_pin(as_u32(as_h2(b) * am2));
    }
}
// VMEM item E (0 .. NA-1) of a weight group load: per strip NQ tile loads, then (quantised) the scale word
template <int FMT, int E>
__device__ __forceinline__ void t4_wload_item(T3Set<FMT, 2> &w, const T3Off<2> &o, const void *wg, const void *sg) {
    constexpr int NQ = T3Set<FMT, 2>::NQ, PER = NQ + (FMT == W_F16 ? 0 : 1), h = E / PER, r = E % PER;
    if constexpr (r < NQ) t3_ld16<r * 1024>(w.q[h][r], o.w[h], wg);
    else t3_ld8(w.s[h], o.s[h], sg);
}

template <int FMT, int NTL, bool HILO>
__device__ __forceinline__ void tg4_body(const GemmLaunch &L, const GemmProb &P, unsigned char *smem) {
    constexpr int SPW = 2, WAVES = 4, STRIPS = WAVES * SPW, BT = NTL * 16;
    constexpr int HB = HILO ? 2 : 1, NBT = NTL * HB;              // B fragments per k-step: [hi | lo][token tile]
    constexpr int M = SPW * NBT;                                  // MFMAs per k-step per wave
    constexpr int STAGE_TILES = NBT * 2, STAGE_BYTES = STAGE_TILES * 1024, DPW = STAGE_TILES / WAVES;
    constexpr int NA = T3Set<FMT, SPW>::NLOAD, UNITS = T4Dq<FMT>::UNITS, UT = SPW * UNITS;
    static_assert(!HILO || NTL == 4, "hi + lo operands: 128 x 64 tiles only");
    static_assert(M % UT == 0 || UT % M == 0, "dequantisation units spread evenly over the MFMA slots");
    using Set = T3Set<FMT, SPW>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nstrips = P.rows >> 4;
    const int ntt = (L.T + BT - 1) / BT;
    int lb = (int)blockIdx.x - P.block_begin;
    const int nb = ((nstrips + STRIPS - 1) / STRIPS) * ntt;       // tiles of one K copy (P.ksb > 1: linear epilogues, as in tg3_body)
    const int kb = lb / nb;
    lb -= kb * nb;
    if (L.xcd_map) {                                              // XCD-banded tile numbering, as in tg_body
        const int k = lb & 7, j = lb >> 3;
        int start = 0;
        for (int m = 0; m < k; ++m) start += (nb - m + 7) >> 3;
        lb = start + j;
    }
    // the banded sequence is row-tile major (token tile fastest: an XCD streams its eighth of the weights once and re-reads them from its L2 for every
    // token tile) or, L.xcd_map == 2, token-tile major (row tile fastest: an XCD keeps ITS token tiles' operand rows in its L2 and streams the weights
    // past them): measured better only for hi + lo operands at 2048 rows (rwkv_engine.cpp)
    const int nrb_ = nb / ntt;
    const int rb = L.xcd_map == 2 ? lb % nrb_ : lb / ntt, tt = L.xcd_map == 2 ? lb / nrb_ : lb - rb * ntt;
    const int strip = rb * STRIPS + wave * SPW;
    const int t0 = tt * BT;
    const int G = P.K >> 7, g0 = (int)((long)kb * G / P.ksb), g1 = (int)((long)(kb + 1) * G / P.ksb);
    const int kofs = g0 * 128;                                    // first k of this copy
    const int nsc = g1 - g0, nst = nsc * 2;                       // weight groups (128 k) and stages (64 k) of this copy
    const unsigned xs_byte = (unsigned)(uintptr_t)smem;
    const int last_tile = (L.T - 1) >> 4;
    Nf4Lut lut;
    if constexpr (FMT == W_NF4) lut = make_nf4_lut();

    f32x4 acc[SPW][NTL];
#pragma unroll
    for (int h = 0; h < SPW; ++h)
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) acc[h][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // The DPW tiles of a stage this wave fetches: stage tile i = m * WAVES + wave = fragment b = i >> 1 (= part * NTL + token tile) of k-step
    // i & 1, landing at byte i * 1024 of the stage.  b >> log2(NTL) (hi / lo) depends on m only: (m * 2 + (wave >> 1)) / NTL with wave >> 1 < 2.
    unsigned xoff[DPW];
#pragma unroll
    for (int m = 0; m < DPW; ++m) {
        const int i = m * WAVES + wave, nt = (i >> 1) % NTL;
        const int ttile = min((t0 >> 4) + nt, last_tile);
        xoff[m] = (unsigned)(((ttile * (P.ldx >> 5) + (i & 1)) * 512 + lane * 8) * 2);
    }
    const char *const xbase_hi = (const char *)(P.xhi + (long)(kofs >> 5) * 512);
    const char *const xbase_lo = HILO ? (const char *)(P.xlo + (long)(kofs >> 5) * 512) : xbase_hi;
    auto dma_item = [&](int s, auto mc) {                            // DMA m of stage s (64 k = two k-steps of 1 KiB tiles per token tile)
        constexpr int m = decltype(mc)::value;
        constexpr bool lo = HILO && (m * 2) / NTL == 1;
        const unsigned dst = xs_byte + (unsigned)((s & (T3_NB - 1)) * STAGE_BYTES + (m * WAVES + wave) * 1024);
        t3_dma16(xoff[m], (lo ? xbase_lo : xbase_hi) + (long)s * 2048, dst);
    };
    T3Off<SPW> woff;
    {
        constexpr int SH = Fmt<FMT>::SH;
        const int KT = P.K >> SH, KG = P.K >> 8;
#pragma unroll
        for (int h = 0; h < SPW; ++h) {
            const int sidx = min(strip + h, nstrips - 1);
            woff.w[h] = (unsigned)(sidx * KT * 64 + lane) * 16u;
            woff.s[h] = (unsigned)(sidx * KG * 16 + (lane & 15)) * 8u;
        }
    }
    const char *const wbase = (const char *)P.W + (long)(kofs >> Fmt<FMT>::SH) * 1024, *const sbase = (const char *)P.S;
    auto wptr = [&](int g) { return wbase + (long)g * (128 >> Fmt<FMT>::SH) * 1024; };               // weight tiles of local group g
    auto sptr = [&](int g) { return sbase + (long)(((kofs >> 7) + g) >> 1) * 128; };                     // its 256-k scale words
    const unsigned rd0 = xs_byte + (unsigned)(lane * 16);            // fragment read address inside a stage

    // ---- one k-step: MFMAs on (Ac, Bc), preparation of (An, Bn) for the next one.  KSN = k-step of the NEXT inside its weight group `wn`,
    // QN = its k-step inside its stage (slot address rdn); VK = VMEM work of this k-step: 0 none, 1 DMA(sd), 2 weight group gw -> wl, then DMA(sd).
    auto kstep = [&](const u32x4 (&Ac)[SPW], u32x4 (&An)[SPW], f16x8 (&Bc)[NBT], f16x8 (&Bn)[NBT], const Set &wn, const T4Scale<FMT> &sc, unsigned rdn,
                     auto ksn_c, auto qn_c, auto vk_c, Set &wl, int gw, int sd) {
        constexpr int KSN = decltype(ksn_c)::value, QN = decltype(qn_c)::value, VK = decltype(vk_c)::value;
        constexpr int NITEM = VK == 0 ? 0 : (VK == 1 ? DPW : NA + DPW), VSLOTS = M - NBT;
        const bool have_w = VK == 2 && gw < nsc, have_d = VK != 0 && sd < nst;
        const void *wg = nullptr, *sg = nullptr;
        if constexpr (VK == 2) { wg = wptr(gw); sg = sptr(gw); }
        t4_for<M>([&](auto ic) {
            constexpr int i = decltype(ic)::value, b = i / SPW, h = i % SPW;
            if constexpr (h == 0) t3_lgkm_wait<(NBT - 1 - b) + (i < NBT ? i : NBT)>(Bc[b]);
            acc[h][b % NTL] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, Ac[h]), Bc[b], acc[h][b % NTL], 0, 0, 0);
            if constexpr (i < NBT) t3_lds16<(i * 2 + QN) * 1024>(Bn[i], rdn);
            if constexpr (UT >= M) {
                t4_for<UT / M>([&](auto jc) { constexpr int j = i * (UT / M) + decltype(jc)::value; t4_unit<FMT, KSN, j % UNITS>(An[j / UNITS], wn, sc, lut, j / UNITS); });
            } else if constexpr (i % (M / UT) == 0) {
                constexpr int j = i / (M / UT);
                t4_unit<FMT, KSN, j % UNITS>(An[j / UNITS], wn, sc, lut, j / UNITS);
            }
            if constexpr (NITEM > 0 && i >= NBT) {                   // VMEM items e in [lo, hi) of this slot, behind the reads
                constexpr int sl = i - NBT, lo = sl * NITEM / VSLOTS, hi = (sl + 1) * NITEM / VSLOTS;
                t4_for<hi - lo>([&](auto ec) {
                    constexpr int e = lo + decltype(ec)::value;
                    if constexpr (VK == 2 && e < NA) { if (have_w) t4_wload_item<FMT, e>(wl, woff, wg, sg); }
                    else { if (have_d) dma_item(sd, std::integral_constant<int, e - (VK == 2 ? NA : 0)>{}); }
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    // the barrier in the middle of stage s: stage s + 1 complete in LDS for every wave, this wave's reads of stage s retired
    auto publish = [&](int s) {
        if (s + 4 < nst) t3_wait_barrier<2 * DPW + NA>(); else t3_wait_barrier<0>();
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    u32x4 A0[SPW], A1[SPW];
    f16x8 B0[NBT], B1[NBT];
    T4Scale<FMT> sc_cur, sc_nxt;
    auto slot_rd = [&](int s) { return rd0 + (unsigned)((s & (T3_NB - 1)) * STAGE_BYTES); };
    // ---- one weight group = four k-steps = stages 2g, 2g + 1.  `cur` holds group g (its scales in sc_cur), `nxt` group g + 1; `cur` is
    // refilled with group g + 3 during the last k-step (its last dequantisation ran during the third).
    auto group = [&](Set &cur, Set &nxt, int g) {
        const int s = 2 * g;
        kstep(A0, A1, B0, B1, cur, sc_cur, slot_rd(s), I1{}, I1{}, I0{}, cur, 0, 0);                   // k-step 4g:     next = k-step 1 of stage s
        publish(s);
        kstep(A1, A0, B1, B0, cur, sc_cur, slot_rd(s + 1), I2{}, I0{}, I1{}, cur, 0, s + 4);          // k-step 4g + 1: next = k-step 0 of stage s + 1; DMA(s + 4)
        kstep(A0, A1, B0, B1, cur, sc_cur, slot_rd(s + 1), I3{}, I1{}, I0{}, cur, 0, 0);               // k-step 4g + 2
        publish(s + 1);
        t3_arrived<FMT, SPW>(nxt);                                                                      // group g + 1 landed (retired by the wait above)
        t4_scales<FMT>(sc_nxt, nxt, (kofs >> 7) + g + 1);
        kstep(A1, A0, B1, B0, nxt, sc_nxt, slot_rd(s + 2), I0{}, I0{}, I2{}, cur, g + 3, s + 5);      // k-step 4g + 3: next = first of group g + 1; W(g + 3), DMA(s + 5)
        sc_cur = sc_nxt;
    };

    Set a0, a1, a2;
    t3_load<FMT, SPW>(a0, woff, wptr(0), sptr(0));
    const int g_second = min(1, nsc - 1);                                      // (s_min: a `nsc > 1 ? 1 : 0` select inside the 64-bit address goes to the VALU)
    t3_load<FMT, SPW>(a1, woff, wptr(g_second), sptr(g_second));
    t4_for<DPW>([&](auto mc) { dma_item(0, mc); });
    if (nst > 1) t4_for<DPW>([&](auto mc) { dma_item(1, mc); });
    if (nst > 2) t4_for<DPW>([&](auto mc) { dma_item(2, mc); });
    if (nst > 2) t3_wait_barrier<2 * DPW>(); else t3_wait_barrier<0>();      // W(0), W(1) and DMA(0) have landed
    t3_arrived<FMT, SPW>(a0);
    t3_arrived<FMT, SPW>(a1);
    a2 = a0;                                                                  // every register defined; refilled with group 2 below
    t4_scales<FMT>(sc_cur, a0, kofs >> 7);
    // "k-step -1": fragments of k-step 0, W(2), DMA(3)
    t4_for<NBT>([&](auto bc) { t3_lds16<decltype(bc)::value * 2 * 1024>(B0[decltype(bc)::value], slot_rd(0)); });
    t4_for<UT>([&](auto jc) { constexpr int j = decltype(jc)::value; t4_unit<FMT, 0, j % UNITS>(A0[j / UNITS], a0, sc_cur, lut, j / UNITS); });
    if (2 < nsc) t3_load<FMT, SPW>(a2, woff, wptr(2), sptr(2));
    if (3 < nst) t4_for<DPW>([&](auto mc) { dma_item(3, mc); });
    __builtin_amdgcn_sched_barrier(0);
    for (int g = 0; g < nsc; g += 3) {
        group(a0, a1, g);
        if (g + 1 < nsc) group(a1, a2, g + 1);
        if (g + 2 < nsc) group(a2, a0, g + 2);
    }
    if (P.ksb > 1) {                                              // partial slab kb (host: out_f32 only, linear epilogue)
        GemmProb Q = P;
        Q.out_f32 = P.out_f32 + (long)kb * P.partial_stride;
        tg_epilogue<SPW, NTL>(L, Q, acc, strip, nstrips, t0, lane);
    } else {
        tg_epilogue<SPW, NTL>(L, P, acc, strip, nstrips, t0, lane);
    }
}

__global__ __launch_bounds__(256, 2) void gemm_tile4_kernel(const GemmLaunch L) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int pi = 0;
    for (int i = 1; i < L.nprob; ++i)
        if ((int)blockIdx.x >= L.p[i].block_begin) pi = i;
    const GemmProb &P = L.p[pi];
    if (P.fmt == W_F16) tg4_body<W_F16, 4, true>(L, P, smem);
    else if (P.fmt == W_INT8) tg4_body<W_INT8, 4, true>(L, P, smem);
    else tg4_body<W_NF4, 4, true>(L, P, smem);
}
bool gemm_tile4_supported(bool hilo, int K) { return hilo && K % 128 == 0; }


// (Round 6 also built STREAM-K over this kernel and gemm_tile3: a grid of 512 resident blocks, the launch's (tile, 128-k group) units dealt evenly per
// XCD band, a block walking its run from the end, partial accumulators handed to the block that reaches the tile's last group through write-through
// stores and flags — correct on 24 configurations (same sums in another order, no timeouts) and no faster: r/k/v/g Int8 at 256 rows 34.1 -> 34.3 us
// (324 tiles on 512 slots), at 2048 rows 150 -> 166 (1,296 tiles: the dispatcher's own refilling beats a persistent grid); Fv without K copies
// 65 -> 40 at 256 rows, where its K copies already reach 30.  A launch's time is not the slowest CU's: profiles/r6_exp_stream_k.log; code at 0a1ea2f.
// What that experiment left behind: scripts/check_sgpr_hazard.py — an inline-asm VMEM instruction whose scalar base reaches it through a v_readlane
// (a spilled SGPR) needs five wait states hipcc does not insert inside asm statements; the hand-off stores faulted on it.)
// (Round 6 also built the loader / consumer form of this kernel — waves 4..7 issue every LDS-DMA, X stages and raw weight tiles through LDS
// rings, waves 0..3 only read LDS, dequantise and multiply, one 8-wave block per CU; parity green and bit-identical — and measured it: a block
// alone on its CU is 25 % faster per stage, but one 8-wave block per CU loses the overlap of two independent 4-wave blocks: r/k/v/g Int8 at 2048
// rows 173.6 us against 142.5, at 256 rows 35.8-37.5 against 34.6.  profiles/r6_exp_tile5_and_xcd_order.log; the code is in the history at
// 3bb646c.)

// part-5 launcher (called by launch_gemm_tile in part 2)
void launch_gemm_tile45(const GemmLaunch &L, int kind, int ntl, bool hilo, hipStream_t s) {
    (void)kind; (void)ntl; (void)hilo;
    static bool attr[16] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr[dev & 15]) {
        (void)hipFuncSetAttribute((const void *)gemm_tile4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr[dev & 15] = true;
    }
    hipLaunchKernelGGL(gemm_tile4_kernel, dim3(L.total_blocks), dim3(256), (size_t)T3_NB * 4 * 2 * 2 * 1024, s, L);
}
#endif  // part 5: software-pipelined hi + lo tile kernel
#if RWKV_PART_ON(2)
// tile shapes, largest first: {waves, strips per wave, n-tiles, k per chunk}
static const int kTileShapes[GEMM_TILE_SHAPES][5] = {{8, 2, 8, 128, 0}, {8, 1, 8, 128, 0}, {4, 1, 8, 128, 0}, {4, 1, 4, 128, 0}, {4, 1, 4, 256, 0}, {8, 1, 8, 256, 0},
                                                     {4, 1, 4, 256, 1}, {4, 2, 4, 128, 1}, {4, 2, 8, 128, 1}, {8, 2, 8, 128, 1}, {4, 2, 8, 128, 2}, {4, 2, 4, 128, 2},
                                                     {4, 2, 4, 128, 3}};
// (Round 6 re-measured the 256 x 128 tile on FOUR waves — 64 rows x 128 tokens per wave, 128 accumulators in AGPRs, one block per CU, a third fewer LDS
// and L2-port bytes per flop — in this file's plain chunked body: 537 TFLOP/s on the 7 B r/k/v/g launch against 488 for the same body on 128 x 128 and
// 860 for the pipelined 128 x 128 kernel; round 3 had measured its pipelined form at 395-563.  profiles/r6_exp_tile_256x128_4waves.log; removed.)
// (128 rows x 64 tokens with 8 waves, 256-k and 128-k chunks — half the operand re-reads of the 64x64 shapes on steps of a few hundred
// rows — was built and measured in round 3: slower on every matrix but one, profiles/r3_exp_tile_128x64.log; removed.)
int gemm_tile_blocks(int shape, int rows, int T) {
    const int strips = kTileShapes[shape][0] * kTileShapes[shape][1], bt = kTileShapes[shape][2] * 16;
    return ((rows / 16 + strips - 1) / strips) * ((T + bt - 1) / bt);
}

void launch_gemm_tile(const GemmLaunch &L, int shape, bool hilo, hipStream_t s) {
    if (kTileShapes[shape][4] >= 3) {                          // part 5: software-pipelined hi + lo kernel
        launch_gemm_tile45(L, kTileShapes[shape][4], kTileShapes[shape][2], hilo, s);
        return;
    }
    if (kTileShapes[shape][4] == 2) {                          // pipelined kernel (caller checked gemm_tile3_supported)
        static bool attr3[16] = {false};
        int dev3 = 0;
        (void)hipGetDevice(&dev3);
        if (!attr3[dev3 & 15]) {
            (void)hipFuncSetAttribute((const void *)gemm_tile3_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)gemm_tile3_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr3[dev3 & 15] = true;
        }
        if (kTileShapes[shape][2] == 8) hipLaunchKernelGGL(gemm_tile3_kernel<8>, dim3(L.total_blocks), dim3(256), (size_t)T3_NB * 8 * 2 * 512 * 2, s, L);
        else hipLaunchKernelGGL(gemm_tile3_kernel<4>, dim3(L.total_blocks), dim3(256), (size_t)T3_NB * 4 * 2 * 512 * 2, s, L);
        return;
    }
    const int bt = kTileShapes[shape][2] * 16, kc = kTileShapes[shape][3];
    const size_t lds = (size_t)2 * (hilo ? 2 : 1) * (bt / 16) * (kc / 32) * 1024;
    static bool attr_done[16] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
#define TG_SH(X, h) X(h, 8, 2, 8, 128, false, 0) X(h, 8, 1, 8, 128, false, 1) X(h, 4, 1, 8, 128, false, 2) X(h, 4, 1, 4, 128, false, 3) X(h, 4, 1, 4, 256, false, 4) \
                    X(h, 8, 1, 8, 256, false, 5) X(h, 4, 1, 4, 256, true, 6) X(h, 4, 2, 4, 128, true, 7) X(h, 4, 2, 8, 128, true, 8) X(h, 8, 2, 8, 128, true, 9)
#define TG_VARIANTS(X) TG_SH(X, true) TG_SH(X, false)
    if (!attr_done[dev & 15]) {
#define SET_ATTR(h, w, p, n, k, g, i) (void)hipFuncSetAttribute((const void *)gemm_tile_kernel<h, w, p, n, k, g>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        TG_VARIANTS(SET_ATTR)
#undef SET_ATTR
        attr_done[dev & 15] = true;
    }
#define LAUNCH(h, w, p, n, k, g, i) if (hilo == h && shape == i) hipLaunchKernelGGL((gemm_tile_kernel<h, w, p, n, k, g>), dim3(L.total_blocks), dim3(w * 64), lds, s, L);
    TG_VARIANTS(LAUNCH)
#undef LAUNCH
#undef TG_VARIANTS
#undef TG_SH
}

#endif  // part 2: prefill tile GEMM

#if RWKV_PART_ON(3)
// =====================================================================================
// Row kernels (one 256-thread block per row)
// =====================================================================================
// PT = float4 groups per thread, NTHR = threads per row block (C <= PT*NTHR*4).  Everything stays in registers (fully unrolled,
// predicated), all global loads of a phase are issued before the first dependent use.  Decode-shaped steps (a few dozen rows =
// a few dozen workgroups on 256 CUs) run ln_shift with 1024 threads per row: the kernel is one latency chain over ~90 KB per
// row, and 16 waves with ~6 loads each get it issued four times as fast as 4 waves with ~21.
#define ROW_FOR(i, c) _Pragma("unroll") for (int i = 0, c = threadIdx.x * 4; i < PT; ++i, c += NTHR * 4) if (c < C)

__device__ __forceinline__ float4 ld4(const float *p) { return *(const float4 *)p; }
__device__ __forceinline__ float4 operator+(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

template <int PT, int NTHR = 256>
__device__ __forceinline__ void row_load_sum(act_t x_in, act_t P, int np, long pstride,
                                             int row, int C, float4 (&v)[PT]) {
    constexpr int MAXNP = 8;                                       // all loads in flight at once, summed in fixed order
    float4 pp[MAXNP][PT];
    ROW_FOR(i, c) v[i] = act_ld4(x_in, (long)row * C + c);
#pragma unroll
    for (int j = 0; j < MAXNP; ++j) {
        if (j < np) { ROW_FOR(i, c) pp[j][i] = act_ld4(P, j * pstride + (long)row * C + c); }
    }
#pragma unroll
    for (int j = 0; j < MAXNP; ++j) {
        if (j < np) { ROW_FOR(i, c) v[i] = v[i] + pp[j][i]; }
    }
}
// block-wide sum without the leading barrier: `buf` (NTHR/64 floats of LDS) must not have been read since the last barrier
// by anything still in flight — callers alternate two buffers.
template <int NTHR>
__device__ __forceinline__ float block_sum_nb(float v, float *buf) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) buf[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = buf[0];
#pragma unroll
    for (int w = 1; w < NTHR / 64; ++w) r += buf[w];
    return r;
}
// two-pass LayerNorm (mean, then centred variance), eps 1e-5 — same order as the oracle's _ln.
// `red` = 2 * NTHR/64 floats of LDS (mean uses the first half, variance the second); wv/bv = the row's weight and bias, loaded
// by the caller together with everything else the kernel needs so that no load waits behind a reduction.
template <int PT, int NTHR = 256>
__device__ __forceinline__ void row_layernorm(float4 (&v)[PT], int C, const float4 (&wv)[PT], const float4 (&bv)[PT], float *red) {
    float s = 0.f;
    ROW_FOR(i, c) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = block_sum_nb<NTHR>(s, red) / (float)C;
    float q = 0.f;
    ROW_FOR(i, c) {
        const float d0 = v[i].x - mean, d1 = v[i].y - mean, d2 = v[i].z - mean, d3 = v[i].w - mean;
        q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    const float var = block_sum_nb<NTHR>(q, red + NTHR / 64) / (float)C;
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    ROW_FOR(i, c) {
        v[i].x = (v[i].x - mean) * rstd * wv[i].x + bv[i].x;
        v[i].y = (v[i].y - mean) * rstd * wv[i].y + bv[i].y;
        v[i].z = (v[i].z - mean) * rstd * wv[i].z + bv[i].z;
        v[i].w = (v[i].w - mean) * rstd * wv[i].w + bv[i].w;
    }
}


// (Round 6: the two LayerNorms of a prefill row — its own and its predecessor's — under two barriers instead of four, both rows' statistics in one
// block reduction, bit-identical: 8.40 -> 8.39 us at 256 rows.  The barriers are not where the kernel's time is; not kept.)
// (Four consecutive rows per block on prefill-shaped steps — a row whose predecessor the block has just normalised takes it from
// registers instead of loading and normalising it again — was built and measured in round 3: slower everywhere, 70.2 -> 69.1 k tok/s at
// 2048 rows and 37.2 -> 35.1 k at 256, where it leaves 64 blocks for 256 CUs; profiles/r3_exp_ln_rows.log.  One row per block stays.)
template <int PT, int NTHR>
__global__ __launch_bounds__(NTHR) void ln_shift_kernel(const LnShiftArgs a) {
    __shared__ float red[2 * NTHR / 64];
    // a.xcd_rows (prefill-shaped steps): block b runs on XCD b mod 8, and row t re-reads row t-1 with all its partial slabs — rows are
    // numbered so that an XCD owns a contiguous run of them and the predecessor comes from that XCD's L2 instead of HBM
    int t = blockIdx.x;
    if (a.xcd_rows) {
        const int nrow = gridDim.x, x = t & 7, slot = t >> 3, base = nrow >> 3, rem = nrow & 7;
        t = x * base + min(x, rem) + slot;
    }
    const int C = a.C;
    TRACE_K(2, 0);
    // every load that does not depend on another load is issued here, parameters first: the kernel is one latency
    // chain (it moves ~100 KB), so each load left behind a reduction costs a full L2/MALL round trip
    float4 wv[PT], bv[PT], mu0[PT], mu1[PT];
    ROW_FOR(i, c) { wv[i] = ld4(a.lnw + c); bv[i] = ld4(a.lnb + c); }
    if (a.nmix > 0) { ROW_FOR(i, c) mu0[i] = ld4(a.mu[0] + c); }
    if (a.nmix > 1) { ROW_FOR(i, c) mu1[i] = ld4(a.mu[1] + c); }
    // one float4 per thread (C <= 4 * NTHR): the mixes beyond the second (V5: four, V7: six) are requested here as well — left in the emit
    // loop each was an L2 round trip behind the two reductions of a kernel that is nothing but a latency chain
    float4 mux[PT == 1 ? 4 : 1];
    if constexpr (PT == 1) {
#pragma unroll
        for (int m = 2; m < 6; ++m) if (m < a.nmix && threadIdx.x * 4 < C) mux[m - 2] = ld4(a.mu[m] + threadIdx.x * 4);
    }
    // dense decode step (row t = slot t, one row per slot): no metadata round trip in front of the state load
    const int slot = a.rm.dense ? t : a.rm.slot[t], prev = a.rm.dense ? -1 : a.rm.prev[t], last = a.rm.dense ? t : a.rm.last[t];
    float *__restrict__ sx = a.sx + (long)slot * a.sx_slot_stride;
    float4 xv[PT], pv[PT];
    if (prev < 0) { ROW_FOR(i, c) pv[i] = ld4(sx + c); }
    const act_t bx = act_buf(a.x_in), bP = act_buf(a.P), bxo = act_buf(a.x_out), bxx = act_buf(a.xx_out), bdx = act_buf(a.dx_out);
    row_load_sum<PT, NTHR>(bx, bP, a.np, a.pstride, t, C, xv);
    if (prev >= 0) row_load_sum<PT, NTHR>(bx, bP, a.np, a.pstride, prev, C, pv);
    TRACE_K(2, 1);
    if (a.x_out) { ROW_FOR(i, c) act_st4(bxo, (long)t * C + c, xv[i]); }
    row_layernorm<PT, NTHR>(xv, C, wv, bv, red);
    TRACE_K(2, 2);
    if (prev >= 0) row_layernorm<PT, NTHR>(pv, C, wv, bv, red);
    if (last >= 0) {            // this block owns the slot's token-shift state write (after its own read above)
        if (last == t) {
            ROW_FOR(i, c) *(float4 *)(sx + c) = xv[i];
        } else {
            float4 lv[PT];
            row_load_sum<PT, NTHR>(bx, bP, a.np, a.pstride, last, C, lv);
            row_layernorm<PT, NTHR>(lv, C, wv, bv, red);
            ROW_FOR(i, c) *(float4 *)(sx + c) = lv[i];
        }
    }
    float4 dxv[PT];
    ROW_FOR(i, c) dxv[i] = make_float4(pv[i].x - xv[i].x, pv[i].y - xv[i].y, pv[i].z - xv[i].z, pv[i].w - xv[i].w);
    if (a.xx_out) { ROW_FOR(i, c) act_st4(bxx, (long)t * C + c, xv[i]); }
    if (a.dx_out) { ROW_FOR(i, c) act_st4(bdx, (long)t * C + c, dxv[i]); }
#pragma unroll
    for (int m = 0; m < 6; ++m) {
        if (m < a.nmix) {
            const act_t oh = act_buf(a.ohi[m]), ol = act_buf(a.olo[m]);
            const bool has_lo = a.olo[m] != nullptr;
            float4 muv[PT];
            if (m == 0) { ROW_FOR(i, c) muv[i] = mu0[i]; }
            else if (m == 1) { ROW_FOR(i, c) muv[i] = mu1[i]; }
            else if constexpr (PT == 1) { muv[0] = mux[m - 2]; }
            else { const float *__restrict__ mu = a.mu[m]; ROW_FOR(i, c) muv[i] = ld4(mu + c); }
            ROW_FOR(i, c) {
                float4 o;
                if (a.mode == 0) {
                    o.x = xv[i].x * muv[i].x + pv[i].x * (1.0f - muv[i].x);
                    o.y = xv[i].y * muv[i].y + pv[i].y * (1.0f - muv[i].y);
                    o.z = xv[i].z * muv[i].z + pv[i].z * (1.0f - muv[i].z);
                    o.w = xv[i].w * muv[i].w + pv[i].w * (1.0f - muv[i].w);
                } else {
                    o.x = xv[i].x + dxv[i].x * muv[i].x;
                    o.y = xv[i].y + dxv[i].y * muv[i].y;
                    o.z = xv[i].z + dxv[i].z * muv[i].z;
                    o.w = xv[i].w + dxv[i].w * muv[i].w;
                }
                act_store_operand4(oh, ol, has_lo, opd_off(t, c, a.ldh), o);
            }
        }
    }
    TRACE_K(2, 3);
}
#define ROW_DISPATCH(KERN, C_, GRID, ...)                                                          \
    do {                                                                                           \
        if ((C_) <= 1024) hipLaunchKernelGGL((KERN<1>), dim3(GRID), dim3(256), 0, s, __VA_ARGS__); \
        else if ((C_) <= 2048) hipLaunchKernelGGL((KERN<2>), dim3(GRID), dim3(256), 0, s, __VA_ARGS__); \
        else if ((C_) <= 4096) hipLaunchKernelGGL((KERN<4>), dim3(GRID), dim3(256), 0, s, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERN<8>), dim3(GRID), dim3(256), 0, s, __VA_ARGS__);             \
    } while (0)

void launch_ln_shift(const LnShiftArgs &a, int T, hipStream_t s) {
    // (Spare workgroups of this launch touching the next GEMM's weights into the Infinity Cache were built and measured in round 3:
    // a 32-slot step went from 2.25 to 3.00 ms — profiles/r3_exp_prefetch_by_row_kernel_workgroups.log — and the code was removed.)
    // Threads per row (measured, profiles/r3_exp_ln_threads.log): 1024 up to 256 rows (the kernel is one latency chain over ~90 KB per row,
    // and 16 waves with ~6 loads each get it issued four times as fast as 4 waves with ~21; still +3 % on a 256-token chunk), 512 above
    // (+0.6 % at 2048 rows; 1024 there is -1.8 %: 2 blocks per CU).  Rows of prefill-shaped steps are numbered XCD-banded (xcd_rows).
    LnShiftArgs a2 = a;
    a2.xcd_rows = T > 64 ? 1 : 0;
    if (T <= 256 && a.C <= 8192) {
        if (a.C <= 4096) hipLaunchKernelGGL((ln_shift_kernel<1, 1024>), dim3(T), dim3(1024), 0, s, a2);
        else hipLaunchKernelGGL((ln_shift_kernel<2, 1024>), dim3(T), dim3(1024), 0, s, a2);
        return;
    }
    if (a.C <= 8192) {
        if (a.C <= 2048) hipLaunchKernelGGL((ln_shift_kernel<1, 512>), dim3(T), dim3(512), 0, s, a2);
        else if (a.C <= 4096) hipLaunchKernelGGL((ln_shift_kernel<2, 512>), dim3(T), dim3(512), 0, s, a2);
        else hipLaunchKernelGGL((ln_shift_kernel<4, 512>), dim3(T), dim3(512), 0, s, a2);
        return;
    }
    hipLaunchKernelGGL((ln_shift_kernel<8, 256>), dim3(T), dim3(256), 0, s, a2);      // wider than 8192 channels: 256 threads x 8 float4
}

template <int PT>
__global__ __launch_bounds__(256) void embed_kernel(const EmbedArgs a) {
    constexpr int NTHR = 256;
    __shared__ float red[8];
    const int t = blockIdx.x, C = a.C;
    float4 wv[PT], bv[PT];
    ROW_FOR(i, c) { wv[i] = ld4(a.lnw + c); bv[i] = ld4(a.lnb + c); }
    int tok = a.token[t];
    tok = tok < 0 ? 0 : (tok >= a.V ? a.V - 1 : tok);
    float4 v[PT];
    ROW_FOR(i, c) {
        const f16x4 e = *(const f16x4 *)(a.emb + (long)tok * C + c);
        v[i] = make_float4((float)e[0], (float)e[1], (float)e[2], (float)e[3]);
    }
    row_layernorm<PT>(v, C, wv, bv, red);
    const act_t bx = act_buf(a.x);
    ROW_FOR(i, c) act_st4(bx, (long)t * C + c, v[i]);
}
void launch_embed(const EmbedArgs &a, int T, hipStream_t s) { ROW_DISPATCH(embed_kernel, a.C, T, a); }

template <int PT>
__global__ __launch_bounds__(256) void ln_out_kernel(const LnOutArgs a) {
    constexpr int NTHR = 256;
    __shared__ float red[8];
    const int o = blockIdx.x, C = a.C;
    float4 wv[PT], bv[PT];
    ROW_FOR(i, c) { wv[i] = ld4(a.lnw + c); bv[i] = ld4(a.lnb + c); }
    const int t = a.out_rows ? a.out_rows[o] : o;               // null: every row is emitted in order (dense decode step)
    float4 v[PT];
    row_load_sum<PT>(act_buf(a.x_in), act_buf(a.P), a.np, a.pstride, t, C, v);
    row_layernorm<PT>(v, C, wv, bv, red);
    const act_t oh = act_buf(a.ohi), ol = act_buf(a.olo);
    ROW_FOR(i, c) act_store_operand4(oh, ol, a.olo != nullptr, opd_off(o, c, a.ldh), v[i]);
}
void launch_ln_out(const LnOutArgs &a, int n_out, hipStream_t s) { ROW_DISPATCH(ln_out_kernel, a.C, n_out, a); }

// =====================================================================================
// WKV recurrence.  One 256-thread block per (active slot, head); the slot's rows of this step are
// consumed sequentially with the 64x64 state held in registers:
//   thread (ig = tid>>4, jg = tid&15) owns T[p = a*16+ig][q = jg*4 .. +4], a = 0..3
// Internal layout T[p = value index][q = key index] for every version (v5/v6: T[j][i] = S_ij).
//   v5/v6: out_p = sum_q r_q (u_q k_q v_p + T_pq) ;  T_pq <- k_q v_p + w_q T_pq
//   v7   : sa_p = sum_q T_pq (-kk_q) ; T_pq <- T_pq w_q + sa_p (kk_q a_q) + v_p k_q ; out_p = sum_q T_pq r_q
// followed by GroupNorm over the head (eps 64e-5), gate, and (v7) the r.k.r_k bonus.
// =====================================================================================
__device__ __forceinline__ float sum16(float v) { return row_sum16(v); }   // the 16 lanes sharing (tid>>4) are one DPP row
// Four row sums at once: lane j of the row ends up with the sum of o[j & 3] over the row's 16 lanes.  The same addition tree as
// four row_sum16 (lane pairs, then pairs of pairs, then the four quads), so the same bits — but after each of the first two levels a
// lane keeps only the half of the values its partner does not, which makes it 5 DPP adds + 6 selects instead of 16 DPP adds.
__device__ __forceinline__ float row_sum16x4(float o0, float o1, float o2, float o3, int j) {
    const bool b0 = j & 1, b1 = j & 2;
    const float p0 = (b0 ? o1 : o0) + dpp_f32<0xB1>(b0 ? o0 : o1);
    const float p1 = (b0 ? o3 : o2) + dpp_f32<0xB1>(b0 ? o2 : o3);
    float q = (b1 ? p1 : p0) + dpp_f32<0x4E>(b1 ? p0 : p1);
    q += dpp_f32<0x124>(q);
    q += dpp_f32<0x128>(q);
    return q;
}

// Decode form (one row per sequence; also correct for several): <= 102 VGPRs, 5 blocks per CU = 1280 (B=32 x 40 heads)
// in one generation.  Steps in which a sequence has several rows use wkv_chunk_kernel below.
__global__ __launch_bounds__(256, 5) void wkv_kernel(const WkvArgs a) {
    __shared__ __attribute__((aligned(16))) float sh_r[64], sh_k[64], sh_v[64], sh_w[64], sh_u[64], sh_kk[64], sh_ka[64];
    __shared__ float sh_out[64];
    const int version = a.version;
    const int seq = blockIdx.x, h = blockIdx.y;
    const int tid = threadIdx.x, ig = tid >> 4, jg = tid & 15;
    TRACE_K(1, 0);
    const int slot = a.dense ? seq : a.seq_slot[seq], row0 = a.dense ? seq : a.seq_begin[seq], nrow = a.dense ? 1 : a.seq_len[seq];
    const int C = a.C, cb = h * 64;
    float *st = a.state + (long)slot * a.slot_stride + (long)h * 4096;

    float4 T[4];
    // the 16 KiB state tile is touched exactly once per step: streamed in and out past L2 (non-temporal).  Left to the
    // write-back L2, the 21 MB of dirty state lines (B = 32) drain at the kernel boundary: 14 -> 10.4 us per launch.
    // (The small activations handed to the next kernel are better off with plain stores — measured both ways.)
#pragma unroll
    for (int aa = 0; aa < 4; ++aa) T[aa] = __builtin_bit_cast(float4, __builtin_nontemporal_load((const f32x4 *)(st + (aa * 16 + ig) * 64 + jg * 4)));
    if (version != 7 && tid < 64) sh_u[tid] = a.u[cb + tid];
    if (version == 5 && tid < 64) sh_w[tid] = a.wdec_or_decay[cb + tid];

    // loop-invariant per-channel parameters of wave 0 (tid < 64)
    float lnw = 0.f, lnb = 0.f, kk_p = 0.f, ka_p = 0.f, rk_p = 0.f;
    if (tid < 64) {
        lnw = a.lnx_w[cb + tid]; lnb = a.lnx_b[cb + tid];
        if (version == 7) { kk_p = a.k_k[cb + tid]; ka_p = a.k_a[cb + tid]; rk_p = a.r_k[cb + tid]; }
    }
    const int ch = tid >> 2, part = tid & 3;                 // v6 decay LoRA: 4 threads per channel
    const int per = a.Dd >> 2;
    const float decay0 = version == 6 ? a.wdec_or_decay[cb + ch] : 0.f;
    const act_t br = act_buf(a.r), bk = act_buf(a.k), bv = act_buf(a.v), bg = act_buf(a.g), btd = act_buf(a.td);
    const act_t ba7 = act_buf(a.a7), bw7 = act_buf(a.w7), bvg7 = act_buf(a.vg7), bvf = act_buf(a.v_first);
    const act_t byh = act_buf(a.yhi), byl = act_buf(a.ylo);

    for (int it = 0; it < nrow; ++it) {
        const int t = row0 + it;
        const long rb = (long)t * C + cb;
        float r = 0.f, k = 0.f, v = 0.f, gt = 0.f, av = 0.f, vg = 0.f, w7 = 0.f, vf = 0.f;
        float dsum = 0.f;
        // ---- issue every global load of this token before the first barrier
        if (tid < 64) {
            r = act_ld1(br, rb + tid); k = act_ld1(bk, rb + tid); v = act_ld1(bv, rb + tid); gt = act_ld1(bg, rb + tid);
            if (version == 7) {
                av = act_ld1(ba7, rb + tid); w7 = act_ld1(bw7, rb + tid);
                if (a.layer != 0) { vg = act_ld1(bvg7, rb + tid); vf = act_ld1(bvf, rb + tid); }
            }
        }
        if (version == 6) {
            // decay LoRA stage 2: d_c = time_decay_c + sum_d D2[c][d] td[d];  w = exp(-exp(d))
            const _Float16 *d2 = a.D2 + (long)(cb + ch) * a.Dd + part * per;
            const long tdo = (long)t * a.Dd + part * per;
            for (int d = 0; d < per; d += 8) {
                const f16x8 wv = *(const f16x8 *)(d2 + d);
                const float4 t0v = act_ld4(btd, tdo + d), t1v = act_ld4(btd, tdo + d + 4);
                dsum += (float)wv[0] * t0v.x + (float)wv[1] * t0v.y + (float)wv[2] * t0v.z + (float)wv[3] * t0v.w +
                        (float)wv[4] * t1v.x + (float)wv[5] * t1v.y + (float)wv[6] * t1v.z + (float)wv[7] * t1v.w;
            }
            dsum = quad_sum(dsum);
        }
        TRACE_K(1, 1);
        __syncthreads();                                   // previous iteration's LDS readers done
        TRACE_K(1, 2);
        if (tid < 64) {
            if (version == 7) {
                float kk = k * kk_p;
                const float ss = wave_sum(kk * kk);        // tid<64 == wave 0: L2 norm over the head
                kk = kk / fmaxf(sqrtf(ss), 1e-12f);
                k = k * (1.0f + (av - 1.0f) * ka_p);
                if (a.layer == 0) act_st1(bvf, rb + tid, v);
                else v = v + (vf - v) * vg;
                sh_kk[tid] = -kk;                          // -kappa
                sh_ka[tid] = kk * av;                      // kappa * a
                sh_w[tid] = w7;
            }
            sh_r[tid] = r; sh_k[tid] = k; sh_v[tid] = v;
        }
        if (version == 6 && part == 0) sh_w[ch] = expf(-expf(decay0 + dsum));
        __syncthreads();
        const float4 rq = *(const float4 *)(sh_r + jg * 4);
        const float4 kq = *(const float4 *)(sh_k + jg * 4);
        const float4 wq = *(const float4 *)(sh_w + jg * 4);
        float outp[4];
        if (version != 7) {
            const float4 uq = *(const float4 *)(sh_u + jg * 4);
#pragma unroll
            for (int aa = 0; aa < 4; ++aa) {
                const float vp = sh_v[aa * 16 + ig];
                float4 &S = T[aa];
                float o;
                float kv;
                kv = kq.x * vp; o = rq.x * (uq.x * kv + S.x); S.x = kv + wq.x * S.x;
                kv = kq.y * vp; o += rq.y * (uq.y * kv + S.y); S.y = kv + wq.y * S.y;
                kv = kq.z * vp; o += rq.z * (uq.z * kv + S.z); S.z = kv + wq.z * S.z;
                kv = kq.w * vp; o += rq.w * (uq.w * kv + S.w); S.w = kv + wq.w * S.w;
                outp[aa] = sum16(o);
            }
        } else {
            const float4 nk = *(const float4 *)(sh_kk + jg * 4);
            const float4 ka = *(const float4 *)(sh_ka + jg * 4);
#pragma unroll
            for (int aa = 0; aa < 4; ++aa) {
                const float vp = sh_v[aa * 16 + ig];
                float4 &S = T[aa];
                float sa = S.x * nk.x + S.y * nk.y + S.z * nk.z + S.w * nk.w;
                sa = sum16(sa);
                S.x = S.x * wq.x + sa * ka.x + vp * kq.x;
                S.y = S.y * wq.y + sa * ka.y + vp * kq.y;
                S.z = S.z * wq.z + sa * ka.z + vp * kq.z;
                S.w = S.w * wq.w + sa * ka.w + vp * kq.w;
                float o = S.x * rq.x + S.y * rq.y + S.z * rq.z + S.w * rq.w;
                outp[aa] = sum16(o);
            }
        }
        if (jg == 0) {
#pragma unroll
            for (int aa = 0; aa < 4; ++aa) sh_out[aa * 16 + ig] = outp[aa];
        }
        TRACE_K(1, 3);
        __syncthreads();
        if (tid < 64) {                                    // wave 0: GroupNorm over the head + gate
            const float o = sh_out[tid];
            const float mean = wave_sum(o) * (1.0f / 64.0f);
            const float d = o - mean;
            const float var = wave_sum(d * d) * (1.0f / 64.0f);
            float y = d / sqrtf(var + 64e-5f) * lnw + lnb;
            if (version == 7) {
                const float bonus = wave_sum(sh_r[tid] * sh_k[tid] * rk_p);
                y += bonus * sh_v[tid];
            }
            y *= gt;
            _Float16 hh, ll;
            split_hilo(y, hh, ll);
            // lanes pair up, the even lane stores both halves (one 4-byte store per pair instead of two 2-byte stores)
            const unsigned hb = __builtin_bit_cast(unsigned short, hh), lb = __builtin_bit_cast(unsigned short, ll);
            const unsigned hn = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hb, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
            const unsigned ln2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lb, 0xB1, 0xf, 0xf, true);
            if ((tid & 1) == 0) {
                const long yo = opd_off(t, cb + tid, a.ldh);     // channels tid, tid+1 are adjacent halfs of one 16-byte piece
                __builtin_amdgcn_raw_buffer_store_b32(hb | (hn << 16), byh, (unsigned)(yo * 2), 0, ACT_SC1);
                if (a.ylo) __builtin_amdgcn_raw_buffer_store_b32(lb | (ln2 << 16), byl, (unsigned)(yo * 2), 0, ACT_SC1);
            }
        }
    }
    TRACE_K(1, 4);
#pragma unroll
    for (int aa = 0; aa < 4; ++aa) __builtin_nontemporal_store(__builtin_bit_cast(f32x4, T[aa]), (f32x4 *)(st + (aa * 16 + ig) * 64 + jg * 4));
    TRACE_K(1, 5);
}

// =====================================================================================
// WKV for steps in which a sequence has several rows (prefill chunks).  Same arithmetic as wkv_kernel, re-ordered so
// that nothing but the recurrence itself is sequential.  Per chunk of WKV_CH tokens of the block's (slot, head):
//   A (parallel over tokens, one wave per token): r/k/v loads, the decay  w = exp(-exp(decay + D2 td))  (V6), V7's
//     kappa normalisation / k, v transforms  ->  LDS rows;
//   B (sequential over tokens, NO barrier, NO global access): state update and output from LDS rows, 16-lane DPP sums,
//     raw head outputs -> LDS;
//   C (parallel over tokens): GroupNorm over the head, bonus (V7), gate, operand emit.
// The per-token chain of the decode kernel (3 barriers, an L2 round trip, two exp, three wave reductions: ~1.5 us) shrinks
// to ~20 LDS reads + 48 FMAs + 16 DPP adds.
// =====================================================================================
constexpr int WKV_CH = 32;                               // tokens per chunk: 8 per wave in the parallel phases
// CH = 8 (round 5): the form for steps whose sequences have at most 8 rows each — the embeddings job at `token_chunk_size` 256 hands each of
// 32 slots 8 tokens per call.  A quarter of the LDS rows and of the phase-A registers: five blocks per CU instead of two or three, so the
// 1280 (slot, head) blocks of such a step are resident in ONE round instead of 1.7-2.5.
template <int VER, int DD, int CH = WKV_CH>
__global__ __launch_bounds__(256, CH == 8 ? 5 : (VER == 7 ? 2 : 3)) void wkv_chunk_kernel(const WkvArgs a) {
    constexpr int WKV_CH = CH;                               // (shadows the namespace constant inside this kernel)
    __shared__ __attribute__((aligned(16))) float s_r[WKV_CH][64], s_k[WKV_CH][64], s_v[WKV_CH][64], s_w[WKV_CH][64];
    // kappa rows ([0]) and kappa*a rows ([1]) exist for V7 only: V5 / V6 keep 40 KiB of LDS and run three blocks per CU (a launch of
    // 1280 blocks is 1.7 rounds of 768 instead of 2.5 of 512)
    __shared__ __attribute__((aligned(16))) float s_kka[VER == 7 ? 2 : 1][VER == 7 ? WKV_CH : 1][64], s_o[WKV_CH][64];
    __shared__ __attribute__((aligned(16))) float s_u[64];
    auto &s_kk = s_kka[0];
    auto &s_ka = s_kka[VER == 7 ? 1 : 0];
    const int seq = blockIdx.x, h = blockIdx.y;
    const int tid = threadIdx.x, ig = tid >> 4, jg = tid & 15, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    TRACE_K(3, 0);
    const int slot = a.seq_slot[seq], row0 = a.seq_begin[seq], nrow = a.seq_len[seq];
    const int C = a.C, cb = h * 64, Dd = a.Dd;
    float *st = a.state + (long)slot * a.slot_stride + (long)h * 4096;
    float4 T[4];
#pragma unroll
    for (int aa = 0; aa < 4; ++aa) T[aa] = __builtin_bit_cast(float4, __builtin_nontemporal_load((const f32x4 *)(st + (aa * 16 + ig) * 64 + jg * 4)));
    if (VER != 7 && tid < 64) s_u[tid] = a.u[cb + tid];
    // per-channel parameters of phase A (channel = lane)
    float kk_p = 0.f, ka_p = 0.f, wconst = 0.f;
    if (VER == 7) { kk_p = a.k_k[cb + lane]; ka_p = a.k_a[cb + lane]; }
    if (VER == 5) wconst = a.wdec_or_decay[cb + lane];

    // V6: this wave's rows of D2 and its decay constants do not depend on the chunk — requested here, behind the state, so that phase A2 finds them
    // in registers instead of starting an L2 round trip of its own after A1 (round 6: the phase trace showed A1 -> A2 -> C each opening with one)
    constexpr int KS6 = DD / 32, NTILE6 = WKV_CH >= 16 ? WKV_CH / 16 : 1;
    f16x8 af[VER == 6 ? KS6 : 1];
    float4 dec4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // (not in the short-chunk form: its 96 registers per lane spill under the extra live ranges — measured 18.7 -> 22.6 us at 256 rows, where the long
    // form went 63.9 -> 59.1 at 2048; the short form keeps its loads inside the phases)
    constexpr bool AHEAD = CH != 8;
    auto load_d2 = [&]() {
        const _Float16 *d2 = a.D2 + (long)(cb + wave * 16 + (lane & 15)) * Dd + (lane >> 4) * 8;
#pragma unroll
        for (int ks = 0; ks < KS6; ++ks) af[ks] = *(const f16x8 *)(d2 + ks * 32);
        dec4 = *(const float4 *)(a.wdec_or_decay + cb + wave * 16 + (lane >> 4) * 4);
    };
    if (VER == 6 && AHEAD) load_d2();

    for (int c0 = 0; c0 < nrow; c0 += WKV_CH) {
        const int n = min(WKV_CH, nrow - c0);
        // V6: the chunk's first-stage decay rows (td) for the MFMA of A2, requested with the chunk's other loads
        float4 td0[VER == 6 ? NTILE6 : 1][VER == 6 ? KS6 : 1], td1[VER == 6 ? NTILE6 : 1][VER == 6 ? KS6 : 1];
        auto load_td = [&]() {
#pragma unroll
            for (int tile = 0; tile < NTILE6; ++tile) {
                const float *tdp = a.td + (long)(row0 + c0 + min(tile * 16 + (lane & 15), n - 1)) * Dd + (lane >> 4) * 8;
#pragma unroll
                for (int ks = 0; ks < KS6; ++ks) { td0[tile][ks] = *(const float4 *)(tdp + ks * 32); td1[tile][ks] = *(const float4 *)(tdp + ks * 32 + 4); }
            }
        };
        // ---- phase A: wave w prepares tokens w, w+4, ... (lane = channel).  A1: every global load of the chunk is issued
        //      back to back and parked RAW in the LDS rows (a lane only touches its own elements: no barrier);
        //      A2: a rolled loop transforms the rows in place.
        {
            float q0[WKV_CH / 4], q1[WKV_CH / 4], q2[WKV_CH / 4], q3[WKV_CH / 4], q4[WKV_CH / 4], q5[WKV_CH / 4], q6[WKV_CH / 4];
#pragma unroll
            for (int i = 0; i < WKV_CH / 4; ++i) {               // rows clamped, not predicated: straight-line loads
                const long rb = (long)(row0 + c0 + min(wave + 4 * i, n - 1)) * C + cb + lane;
                q0[i] = a.r[rb]; q1[i] = a.k[rb]; q2[i] = a.v[rb];
                if (VER == 7) {
                    q3[i] = a.a7[rb]; q4[i] = a.w7[rb];
                    if (a.layer != 0) { q5[i] = a.v_first[rb]; q6[i] = a.vg7[rb]; }
                }
            }
            if (VER == 6 && AHEAD) load_td();
            if (VER == 7) {
                // V7's per-token transforms on the eight tokens of this wave AT ONCE, in the registers the loads landed in (kappa =
                // normalised k * k_k, k <- k (1 + (a - 1) k_a), v <- v + (v_first - v) gate): eight independent L2-norm reductions
                // interleave instead of one dependent LDS round trip + reduction per token (the rolled form was phase A2 of round 3).
                // The same operations on the same values in the same order per token: bit-identical.
                float kk[WKV_CH / 4], ss[WKV_CH / 4];
#pragma unroll
                for (int i = 0; i < WKV_CH / 4; ++i) { kk[i] = q1[i] * kk_p; ss[i] = kk[i] * kk[i]; }
#pragma unroll
                for (int i = 0; i < WKV_CH / 4; ++i) ss[i] = wave_sum(ss[i]);
#pragma unroll
                for (int i = 0; i < WKV_CH / 4; ++i) {
                    const int tt = wave + 4 * i;
                    if (tt < n) {
                        const float av = q3[i];
                        const float kn = kk[i] / fmaxf(sqrtf(ss[i]), 1e-12f);
                        float v = q2[i];
                        if (a.layer == 0) a.v_first[(long)(row0 + c0 + tt) * C + cb + lane] = v;
                        else v = v + (q5[i] - v) * q6[i];
                        s_r[tt][lane] = q0[i];
                        s_k[tt][lane] = q1[i] * (1.0f + (av - 1.0f) * ka_p);
                        s_v[tt][lane] = v;
                        s_w[tt][lane] = q4[i];
                        s_kk[tt][lane] = -kn;
                        s_ka[tt][lane] = kn * av;
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < WKV_CH / 4; ++i) {
                    const int tt = wave + 4 * i;
                    if (tt < n) { s_r[tt][lane] = q0[i]; s_k[tt][lane] = q1[i]; s_v[tt][lane] = q2[i]; }
                }
            }
        }
        if (c0 == 0) TRACE_K(3, 1);
        if (VER == 6) {
            // decay LoRA stage 2 on the matrix pipe: dd[ch][t] = sum_d D2[ch][d] * td[t][d] is a 64 x 32 x Dd product per chunk.
            // Wave w owns channels 16w .. 16w+15 (A = its rows of D2, f16 as stored), both 16-token tiles; td (fp32) goes in as
            // an f16 (hi, lo) pair, so a product is exact and the sum is an fp32 accumulation as before — in MFMA order instead of
            // the decode kernel's four partial sums, which the oracle's tolerance covers.  (The VALU form — every lane 64 FMAs +
            // 64 converts per token over LDS broadcasts — was 4 us of a chunk's 15, profiles/r3_trace_wkv_chunk.log.)
            constexpr int KS = DD / 32;
            if (!AHEAD) { load_d2(); load_td(); }
#pragma unroll
            for (int tile = 0; tile < (WKV_CH >= 16 ? WKV_CH / 16 : 1); ++tile) {
                const int tt = tile * 16 + (lane & 15);
                f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const float4 t0k = td0[tile][ks], t1k = td1[tile][ks];
                    const float tv[8] = {t0k.x, t0k.y, t0k.z, t0k.w, t1k.x, t1k.y, t1k.z, t1k.w};
                    f16x8 bh, bl;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { _Float16 hh, ll; split_hilo(tv[e], hh, ll); bh[e] = hh; bl[e] = ll; }
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ks], bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ks], bl, acc, 0, 0, 0);
                }
                if (tt < n) {
                    float4 wv;
                    wv.x = expf(-expf(dec4.x + acc[0])); wv.y = expf(-expf(dec4.y + acc[1]));
                    wv.z = expf(-expf(dec4.z + acc[2])); wv.w = expf(-expf(dec4.w + acc[3]));
                    *(float4 *)(&s_w[tt][wave * 16 + (lane >> 4) * 4]) = wv;
                }
            }
        }
        if (VER == 5) {
            for (int tt = wave; tt < n; tt += 4) s_w[tt][lane] = wconst;
        }
        // phase C's gate values (thread = (token tid>>3, 8 channels)): requested here, a recurrence ahead of their use
        float4 gpre0 = make_float4(0.f, 0.f, 0.f, 0.f), gpre1 = gpre0;
        constexpr bool GATE_AHEAD = AHEAD;
        if (GATE_AHEAD) {
            const int ttc = tid >> 3;
            if (ttc < n) {
                const float *gp = a.g + (long)(row0 + c0 + ttc) * C + cb + (tid & 7) * 8;
                gpre0 = *(const float4 *)gp; gpre1 = *(const float4 *)(gp + 4);
            }
        }
        if (c0 == 0) TRACE_K(3, 2);
        __syncthreads();
        if (c0 == 0) TRACE_K(3, 3);
        // ---- phase B: the recurrence; thread (ig, jg) owns T[p = aa*16+ig][q = jg*4 .. +4]
        float4 uq = make_float4(0.f, 0.f, 0.f, 0.f);
        if (VER != 7) uq = *(const float4 *)(s_u + jg * 4);
        // one wave per SIMD: no other wave hides the LDS latency, so four tokens' LDS rows are fetched per trip
        // two state columns per instruction (v_pk_mul_f32 / v_pk_fma_f32 run at twice the scalar rate, and the recurrence is
        // VALU-issue-bound: two waves per SIMD, nothing else to hide behind): every element sees the operations it saw before
        // (kv = k*v; o += r*(u*kv + S); S = kv + w*S), only the four products of an output are added pairwise
        auto step = [&](int tt, const float4 &rq, const float4 &kq, const float4 &wq, const float4 &nk, const float4 &ka, const float (&vp4)[4]) {
            float outp[4];
            const f32x2 r01 = {rq.x, rq.y}, r23 = {rq.z, rq.w}, k01 = {kq.x, kq.y}, k23 = {kq.z, kq.w}, w01 = {wq.x, wq.y}, w23 = {wq.z, wq.w};
            if (VER != 7) {
                const f32x2 u01 = {uq.x, uq.y}, u23 = {uq.z, uq.w};
#pragma unroll
                for (int aa = 0; aa < 4; ++aa) {
                    const f32x2 vp = {vp4[aa], vp4[aa]};
                    f32x2 S0 = {T[aa].x, T[aa].y}, S1 = {T[aa].z, T[aa].w};
                    const f32x2 kv0 = k01 * vp, kv1 = k23 * vp;
                    f32x2 o2 = r01 * __builtin_elementwise_fma(u01, kv0, S0);
                    o2 = __builtin_elementwise_fma(r23, __builtin_elementwise_fma(u23, kv1, S1), o2);
                    S0 = __builtin_elementwise_fma(w01, S0, kv0);
                    S1 = __builtin_elementwise_fma(w23, S1, kv1);
                    T[aa] = make_float4(S0[0], S0[1], S1[0], S1[1]);
                    outp[aa] = o2[0] + o2[1];
                }
            } else {
                const f32x2 n01 = {nk.x, nk.y}, n23 = {nk.z, nk.w}, a01 = {ka.x, ka.y}, a23 = {ka.z, ka.w};
#pragma unroll
                for (int aa = 0; aa < 4; ++aa) {
                    const f32x2 vp = {vp4[aa], vp4[aa]};
                    f32x2 S0 = {T[aa].x, T[aa].y}, S1 = {T[aa].z, T[aa].w};
                    const f32x2 d2 = __builtin_elementwise_fma(S1, n23, S0 * n01);
                    const float sa = sum16(d2[0] + d2[1]);
                    const f32x2 sa2 = {sa, sa};
                    S0 = __builtin_elementwise_fma(vp, k01, __builtin_elementwise_fma(sa2, a01, S0 * w01));
                    S1 = __builtin_elementwise_fma(vp, k23, __builtin_elementwise_fma(sa2, a23, S1 * w23));
                    T[aa] = make_float4(S0[0], S0[1], S1[0], S1[1]);
                    const f32x2 o2 = __builtin_elementwise_fma(S1, r23, S0 * r01);
                    outp[aa] = o2[0] + o2[1];
                }
            }
            const float q = row_sum16x4(outp[0], outp[1], outp[2], outp[3], jg);    // lane jg < 4 holds output aa = jg
            if (jg < 4) s_o[tt][jg * 16 + ig] = q;
        };
        // (the short-chunk form runs five waves per SIMD: occupancy hides the LDS latency and two tokens per trip keep it under 102 registers)
        constexpr int UB = CH == 8 ? 2 : 4;
        for (int t4 = 0; t4 < n; t4 += UB) {
            float4 rq[UB], kq[UB], wq[UB], nk[UB], ka[UB];
            float vp[UB][4];
#pragma unroll
            for (int u = 0; u < UB; ++u) {                       // rows beyond n are read (LDS, harmless) and never used
                const int tt = min(t4 + u, WKV_CH - 1);
                rq[u] = *(const float4 *)(&s_r[tt][jg * 4]);
                kq[u] = *(const float4 *)(&s_k[tt][jg * 4]);
                wq[u] = *(const float4 *)(&s_w[tt][jg * 4]);
                if (VER == 7) { nk[u] = *(const float4 *)(&s_kk[tt][jg * 4]); ka[u] = *(const float4 *)(&s_ka[tt][jg * 4]); }
#pragma unroll
                for (int aa = 0; aa < 4; ++aa) vp[u][aa] = s_v[tt][aa * 16 + ig];
            }
#pragma unroll
            for (int u = 0; u < UB; ++u)
                if (t4 + u < n) step(t4 + u, rq[u], kq[u], wq[u], nk[u], ka[u], vp[u]);
        }
        if (c0 == 0) TRACE_K(3, 4);
        __syncthreads();
        // ---- phase C: GroupNorm over the head (eps 64e-5), bonus (V7), gate, operand emit — all tokens of the chunk at
        //      once: thread = (token tid>>3, 8 channels (tid&7)*8 ..), sums over the 8 lanes of a token
        {
            const int tt = tid >> 3, c8 = (tid & 7) * 8;
            auto sum8 = [](float v) {                            // the 8 lanes sharing tid>>3 (half a DPP row)
                v = quad_sum(v);
                return v + __shfl_xor(v, 4, 64);
            };
            if (tt < n) {                                         // the 8 lanes of a token take the branch together
                const int t = row0 + c0 + tt;
                float4 g0 = gpre0, g1 = gpre1;
                if (!GATE_AHEAD) { g0 = *(const float4 *)(a.g + (long)t * C + cb + c8); g1 = *(const float4 *)(a.g + (long)t * C + cb + c8 + 4); }
                const float4 o0 = *(const float4 *)(&s_o[tt][c8]), o1 = *(const float4 *)(&s_o[tt][c8 + 4]);
                float o[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
                const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                float sm = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) sm += o[e];
                const float mean = sum8(sm) * (1.0f / 64.0f);
                float sq = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { o[e] -= mean; sq += o[e] * o[e]; }
                const float rstd = 1.0f / sqrtf(sum8(sq) * (1.0f / 64.0f) + 64e-5f);
                float bonus = 0.f;
                if (VER == 7) {
                    float bs = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) bs += s_r[tt][c8 + e] * s_k[tt][c8 + e] * a.r_k[cb + c8 + e];
                    bonus = sum8(bs);
                }
                f16x8 hh, ll;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float y = o[e] * rstd * a.lnx_w[cb + c8 + e] + a.lnx_b[cb + c8 + e];
                    if (VER == 7) y += bonus * s_v[tt][c8 + e];
                    y *= gg[e];
                    _Float16 h, l;
                    split_hilo(y, h, l);
                    hh[e] = h; ll[e] = l;
                }
                const long yo = opd_off(t, cb + c8, a.ldh);           // 8 consecutive k of one token: one 16-byte piece
                *(f16x8 *)(a.yhi + yo) = hh;
                if (a.ylo) *(f16x8 *)(a.ylo + yo) = ll;
            }
        }
        if (c0 == 0) TRACE_K(3, 5);
        __syncthreads();                                   // the next chunk overwrites the LDS rows
    }
    TRACE_K(3, 6);
#pragma unroll
    for (int aa = 0; aa < 4; ++aa) __builtin_nontemporal_store(__builtin_bit_cast(f32x4, T[aa]), (f32x4 *)(st + (aa * 16 + ig) * 64 + jg * 4));
    TRACE_K(3, 7);
}
void launch_wkv(const WkvArgs &a, bool multi_row, hipStream_t s) {
    // every sequence of the step has <= 8 rows: the five-blocks-per-CU form.  V6's decay LoRA is compiled for Dd = 64 and 128 only (the KS of
    // its MFMA stage): any other Dd the loader accepts (multiples of 4 up to 128) falls through to the generic per-token kernel below,
    // like the long form does (a <6, 128, 8> instance would read D2 / td rows with the wrong extent).
    if (multi_row && a.max_rows > 0 && a.max_rows <= 8 && (a.version != 6 || a.Dd == 64 || a.Dd == 128)) {
        if (a.version == 5) hipLaunchKernelGGL((wkv_chunk_kernel<5, 64, 8>), dim3(a.n_seq, a.H), dim3(256), 0, s, a);
        else if (a.version == 6 && a.Dd == 64) hipLaunchKernelGGL((wkv_chunk_kernel<6, 64, 8>), dim3(a.n_seq, a.H), dim3(256), 0, s, a);
        else if (a.version == 6) hipLaunchKernelGGL((wkv_chunk_kernel<6, 128, 8>), dim3(a.n_seq, a.H), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((wkv_chunk_kernel<7, 64, 8>), dim3(a.n_seq, a.H), dim3(256), 0, s, a);
        return;
    }
    if (multi_row && a.version == 5) hipLaunchKernelGGL((wkv_chunk_kernel<5, 64>), dim3(a.n_seq, a.H), dim3(256), 0, s, a);
    else if (multi_row && a.version == 6 && a.Dd == 64) hipLaunchKernelGGL((wkv_chunk_kernel<6, 64>), dim3(a.n_seq, a.H), dim3(256), 0, s, a);
    else if (multi_row && a.version == 6 && a.Dd == 128) hipLaunchKernelGGL((wkv_chunk_kernel<6, 128>), dim3(a.n_seq, a.H), dim3(256), 0, s, a);
    else if (multi_row && a.version == 7) hipLaunchKernelGGL((wkv_chunk_kernel<7, 64>), dim3(a.n_seq, a.H), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(wkv_kernel, dim3(a.n_seq, a.H), dim3(256), 0, s, a);
}

#endif  // part 3: row kernels + WKV

#if RWKV_PART_ON(4)
// =====================================================================================
// State slab <-> internal.  slab[l][0][c]=sx_att, slab[l][1+i][h*64+j]=S_h[i][j], slab[l][65][c]=sx_ffn
// internal wkv T[l][h][p][q]:  transposed (v5/v6): S[i][j] = T[p=j][q=i];  v7: S[i][j] = T[p=i][q=j]
// =====================================================================================
__global__ void state_pack_kernel(const StatePackArgs a) {
    const int C = a.C, N = 64;
    const long per_layer = (long)(N + 2) * C;
    const long total = a.layer_only >= 0 ? (long)N * C : (long)a.L * per_layer;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int l, row, c;
        if (a.layer_only >= 0) { l = a.layer_only; row = 1 + (int)(idx / C); c = (int)(idx % C); }
        else { l = (int)(idx / per_layer); long r2 = idx % per_layer; row = (int)(r2 / C); c = (int)(r2 % C); }
        float *src;
        if (row == 0) src = a.sxa + (long)l * C + c;
        else if (row == N + 1) src = a.sxf + (long)l * C + c;
        else {
            const int i = row - 1, h = c >> 6, j = c & 63;
            const int p = a.transposed ? j : i, q = a.transposed ? i : j;
            src = a.wkv + (((long)l * a.H + h) * 64 + p) * 64 + q;
        }
        if (a.to_slab) a.slab[idx] = *src; else *src = a.slab[idx];
    }
}
void launch_state_pack(const StatePackArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(state_pack_kernel, dim3(1024), dim3(256), 0, s, a);
}

// =====================================================================================
// softmax / argmax over the vocabulary (one block per row)
// =====================================================================================
__global__ __launch_bounds__(256) void softmax_kernel(const float *in, float *out, int V) {
    __shared__ float red[4];
    const float *x = in + (long)blockIdx.x * V;
    float *y = out + (long)blockIdx.x * V;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < V; i += 256) m = fmaxf(m, x[i]);
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) m = fmaxf(m, __shfl_xor(m, k, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 0.f;
    for (int i = threadIdx.x; i < V; i += 256) s += expf(x[i] - m);
    s = block_sum256(s, red);
    const float inv = 1.0f / s;
    for (int i = threadIdx.x; i < V; i += 256) y[i] = expf(x[i] - m) * inv;
}
void launch_softmax(const float *in, float *out, int n_rows, int V, hipStream_t s) {
    hipLaunchKernelGGL(softmax_kernel, dim3(n_rows), dim3(256), 0, s, in, out, V);
}

constexpr int ARGMAX_SEG = 32;                                  // segments per row in stage 1
__device__ __forceinline__ void argmax_block(float &best, int &idx, float *bv, int *bi) {
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) {
        const float ov = __shfl_xor(best, k, 64);
        const int oi = __shfl_xor(idx, k, 64);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
    }
}
__global__ __launch_bounds__(256) void argmax_stage1(const float *logits, int V, float *pv, int *pi) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    const int row = blockIdx.x, seg = blockIdx.y;
    const int per = (V + ARGMAX_SEG - 1) / ARGMAX_SEG;
    const int lo = seg * per, hi = min(V, lo + per);
    const float *x = logits + (long)row * V;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int i = lo + threadIdx.x; i < hi; i += 256) {
        const float v = x[i];
        if (v > best) { best = v; idx = i; }               // ascending per thread: first max kept
    }
    argmax_block(best, idx, bv, bi);
    if (threadIdx.x == 0) { pv[row * ARGMAX_SEG + seg] = best; pi[row * ARGMAX_SEG + seg] = idx; }
}
__global__ __launch_bounds__(64) void argmax_stage2(const float *pv, const int *pi, int *out_tok) {
    const int row = blockIdx.x;
    float best = threadIdx.x < ARGMAX_SEG ? pv[row * ARGMAX_SEG + threadIdx.x] : -INFINITY;
    int idx = threadIdx.x < ARGMAX_SEG ? pi[row * ARGMAX_SEG + threadIdx.x] : 0x7fffffff;
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) {
        const float ov = __shfl_xor(best, k, 64);
        const int oi = __shfl_xor(idx, k, 64);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if (threadIdx.x == 0) out_tok[row] = idx;                // lowest index on ties == np.argmax
}
void launch_argmax(const float *logits, int n_rows, int V, int *out_tok, float *scratch_v, int *scratch_i, hipStream_t s) {
    hipLaunchKernelGGL(argmax_stage1, dim3(n_rows, ARGMAX_SEG), dim3(256), 0, s, logits, V, scratch_v, scratch_i);
    hipLaunchKernelGGL(argmax_stage2, dim3(n_rows), dim3(64), 0, s, (const float *)scratch_v, (const int *)scratch_i, out_tok);
}

// =====================================================================================
// On-device sampling front-end: sparse logit adjustments + softmax + nucleus (top-k / top-p /
// temperature) sampling with a caller-supplied uniform draw — crates/ai00-core/src/sampler/nucleus.rs:69-101 after
// run.rs:664-697.  One 1024-thread block per row; the row's probabilities live in registers (V <= 65536).
// =====================================================================================
__global__ void logit_adjust_kernel(float *logits, int V, const int *rows, const int *toks, const float *vals, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && toks[i] >= 0 && toks[i] < V) logits[(long)rows[i] * V + toks[i]] += vals[i];   // host merges duplicates
}

constexpr int NUC_THREADS = 1024, NUC_EPT = 64, NUC_CAND = 1024;

__device__ __forceinline__ float block_reduce_1024(float v, float *red, bool is_max) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { const float o = __shfl_xor(v, m, 64); v = is_max ? fmaxf(v, o) : v + o; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int w = 1; w < NUC_THREADS / 64; ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
    return r;
}

// NC = candidate capacity; MIRO = the Mirostat instantiation (kind 2 rows only; the other one takes kinds 0 and 1).
// Mirostat (mirostat.rs:44-90): sort descending, k = 1 + #(tokens whose surprise -log2 p does not exceed max_surprise)
// (exact while that fits NC = 8192, i.e. max_surprise < 13; beyond, the tail below 2^-13 is cut), no temperature,
// draw u * sum against the running sum; `out_prob` carries the token surprise log2(sum) - log2(p) the host needs for
// its update of max_surprise.
// FULL: V == 65536 exactly (the World vocabulary padded): no bound predicates at all — with them hipcc keeps 64 exec masks
// alive across the kernel (spilled to VGPR lanes) and wraps every load in its own branch.
template <int NC, bool MIRO, bool FULL>
__global__ __launch_bounds__(NUC_THREADS) void nucleus_kernel(const float *logits, int V, const SampleRow *sp, int *out_tok,
                                                               float *out_prob) {
    extern __shared__ __attribute__((aligned(16))) unsigned char nuc_smem[];
    __shared__ float red[16];
    __shared__ unsigned hist[256];
    __shared__ unsigned sel[3];                                    // prefix, remaining k, candidate counter
    unsigned long long *cand = (unsigned long long *)nuc_smem;     // [NC] (key bits << 32) | ~id  -> sort descending
    float *qv = (float *)(cand + NC);                              // [NC]
    const int row = blockIdx.x, tid = threadIdx.x;
    const SampleRow P = sp[row];
    if ((P.kind == 2) != MIRO) return;                             // uniform per block
    const float *x = logits + (long)row * V;
    float p[NUC_EPT];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < NUC_EPT; ++j) {
        const int i = j * NUC_THREADS + tid;
        const float xv = x[FULL ? i : (i < V ? i : V - 1)];      // clamped, not predicated: the 64 loads are issued back to back
        p[j] = (FULL || i < V) ? xv : -INFINITY;
        m = fmaxf(m, p[j]);
    }
    m = block_reduce_1024(m, red, true);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NUC_EPT; ++j) { p[j] = expf(p[j] - m); s += p[j]; }            // out of range: exp(-inf) = 0
    s = block_reduce_1024(s, red, false);
#pragma unroll
    for (int j = 0; j < NUC_EPT; ++j) p[j] = p[j] / s;

    // ---- Typical sampling (typical.rs:70-108): the sort key is |(-ln p) - H| ascending over p > 0, H = sum p (-ln p).
    //      The registers are re-used for the key, stored as 0xFFFFFFFF - bits(key) (0 for p == 0), so that the
    //      "k largest, descending" machinery below selects the k smallest keys in ascending order; the few surviving
    //      probabilities are recomputed from the logits with the same two operations.
    const bool typical = P.kind == 1;
    if (typical) {
        float h = 0.f;
#pragma unroll
        for (int j = 0; j < NUC_EPT; ++j) h += p[j] > 0.f ? p[j] * -logf(p[j]) : 0.f;
        h = block_reduce_1024(h, red, false);
#pragma unroll
        for (int j = 0; j < NUC_EPT; ++j)
            p[j] = __uint_as_float(p[j] > 0.f ? 0xFFFFFFFFu - __float_as_uint(fabsf(-logf(p[j]) - h)) : 0u);
    }
    auto prob_of = [&](unsigned long long e) {                     // probability of a sorted candidate
        if (!typical) return __uint_as_float((unsigned)(e >> 32));
        const int id = (int)(0xFFFFFFFFu - (unsigned)(e & 0xFFFFFFFFull));
        return expf(x[id] - m) / s;
    };
    const float cut = typical ? P.tau : P.top_p;
    // ---- the k-th largest key (positive floats / complemented keys: bit patterns are order-preserving).
    // A histogram radix select is the textbook choice and was the first version: all 65 536 keys of a row share a handful of
    // exponent bytes, so its LDS atomics serialise on a few bins (137 us per launch).  Instead: a search over the key bits,
    // two bits per step; a step counts, per thread over its 64 registers, the keys >= each of three trial thresholds and sums
    // the three counts over the block (DPP wave sums, one barrier).  16 steps, no atomics, deterministic.
    if (!MIRO && P.top_k < 1) {                                     // `.take(0)`: nothing kept, find_or_first -> None -> token 0 (nucleus.rs:78-101)
        if (tid == 0) { out_tok[row] = 0; if (out_prob) out_prob[row] = 0.f; }
        return;
    }
    int k = P.top_k > 256 ? 256 : P.top_k;
    if (MIRO) {
        float c = 0.f;
#pragma unroll
        for (int j = 0; j < NUC_EPT; ++j) c += (p[j] > 0.f && !(-log2f(p[j]) > P.tau)) ? 1.f : 0.f;   // counts < 2^24: exact in fp32
        k = (int)block_reduce_1024(c, red, false) + 1;              // ... and the first token beyond max_surprise
        if (k > NC) k = NC;
    }
    if (k > V) k = V;
    float *cnt = (float *)hist;                                     // [2][16 waves][4] per-wave partial counts, double-buffered
    // per-wave counts arrive as wave-uniform integers: a count is s_bcnt1 of the compare mask (v_cmp -> SALU), no cross-lane sum.
    // Written as asm: left to itself hipcc batches the 192 compares of a step and spills their masks to VGPR lanes (17 k lines).
    // (compare, count and accumulate in one statement: with the count as an asm OUTPUT hipcc still parks every count in a lane)
    auto wcount3 = [](int &c1, int &c2, int &c3, unsigned key, unsigned t1, unsigned t2, unsigned t3) {   // c_i += #lanes(key >= t_i)
        asm volatile("v_cmp_le_u32 vcc, %3, %6\n\ts_bcnt1_i32_b64 vcc_lo, vcc\n\ts_add_i32 %0, %0, vcc_lo\n\t"
                     "v_cmp_le_u32 vcc, %4, %6\n\ts_bcnt1_i32_b64 vcc_lo, vcc\n\ts_add_i32 %1, %1, vcc_lo\n\t"
                     "v_cmp_le_u32 vcc, %5, %6\n\ts_bcnt1_i32_b64 vcc_lo, vcc\n\ts_add_i32 %2, %2, vcc_lo"
                     : "+s"(c1), "+s"(c2), "+s"(c3) : "s"(t1), "s"(t2), "s"(t3), "v"(key) : "vcc", "scc");
    };
    auto count3 = [&](int buf, float c1, float c2, float c3, float &s1, float &s2, float &s3) {
        float *cb = cnt + buf * 64;
        if ((tid & 63) == 0) { cb[(tid >> 6) * 4 + 0] = c1; cb[(tid >> 6) * 4 + 1] = c2; cb[(tid >> 6) * 4 + 2] = c3; }
        __syncthreads();
        s1 = s2 = s3 = 0.f;
#pragma unroll
        for (int w = 0; w < NUC_THREADS / 64; ++w) { s1 += cb[w * 4 + 0]; s2 += cb[w * 4 + 1]; s3 += cb[w * 4 + 2]; }
    };
    unsigned thr = 0u;                                              // invariant: count(key >= thr) >= k
#pragma unroll 1
    for (int step = 0; step < 16; ++step) {
        const int lo = 30 - 2 * step;
        const unsigned tu = __builtin_amdgcn_readfirstlane(thr);    // block-uniform by construction; make it an SGPR
        const unsigned t1 = tu | (1u << lo), t2 = tu | (2u << lo), t3 = tu | (3u << lo);
        int c1 = 0, c2 = 0, c3 = 0;
#pragma unroll
        for (int j = 0; j < NUC_EPT; ++j) {
            const unsigned key = __float_as_uint(p[j]);             // elements beyond V hold 0 and never count (thresholds > 0)
            wcount3(c1, c2, c3, key, t1, t2, t3);
        }
        float s1, s2, s3;
        count3(step & 1, (float)c1, (float)c2, (float)c3, s1, s2, s3);   // sums <= 65536: exact in fp32
        thr = s3 >= (float)k ? t3 : (s2 >= (float)k ? t2 : (s1 >= (float)k ? t1 : thr));
    }
    // thr = bits of the k-th largest key (0 when fewer than k keys are non-zero).  Candidates: every key > thr, and of the keys
    // == thr the lowest ids (ties order lower id first, as the oracle's restatement does: the reference sorts with an unstable
    // radix sort, nucleus.rs:76, so its tie order is not defined).  If more keys tie at thr than the candidate buffer holds,
    // a second search of the same kind over the id picks the id bound, so the set never depends on thread timing.
    float n_ge, n_gt, n_dummy;
    {
        const unsigned tu = __builtin_amdgcn_readfirstlane(thr);
        const unsigned tge = tu < 1u ? 1u : tu, tgt = tu + 1u;     // key >= thr and non-zero; key > thr  (thr < 2^32 - 1: a key of all ones is never the k-th... guarded below)
        int cge = 0, cgt = 0, cxx = 0;
#pragma unroll
        for (int j = 0; j < NUC_EPT; ++j) wcount3(cge, cgt, cxx, __float_as_uint(p[j]), tge, tgt, tgt);
        if (tu == 0xFFFFFFFFu) cgt = 0;                             // tgt wrapped to 0: nothing is greater than all ones
        count3(0, (float)cge, (float)cgt, 0.f, n_ge, n_gt, n_dummy);
    }
    unsigned id_bound = 0xFFFFFFFFu;                                // keys == thr are admitted while id <= id_bound
    if (n_ge > (float)NC) {
        const float need = (float)k - n_gt;                         // >= 1 ties needed; there are more than NC - n_gt of them
        unsigned lim = 0u;                                          // the largest t with count(key == thr, id < t) < need
#pragma unroll 1
        for (int step = 0; step < 9; ++step) {                      // 18 bits, two per step (ids < 2^16)
            const int lo = 16 - 2 * step;
            const unsigned lu = __builtin_amdgcn_readfirstlane(lim);
            const unsigned t1 = lu | (1u << lo), t2 = lu | (2u << lo), t3 = lu | (3u << lo);
            int c1 = 0, c2 = 0, c3 = 0;                             // counts of ties with id < t (monotone in t)
#pragma unroll
            for (int j = 0; j < NUC_EPT; ++j) {
                const unsigned id = (unsigned)(j * NUC_THREADS + tid);
                // a tie counts when id < t, i.e. NOT (id >= t): fold the tie test into the compared value (non-ties compare as 2^32-1)
                const unsigned idv = (__float_as_uint(p[j]) == thr) ? id : 0xFFFFFFFFu;   // out-of-range elements hold key 0 != thr
                wcount3(c1, c2, c3, idv, t1, t2, t3);                // counts of NOT (id < t); flipped below
            }
            c1 = NUC_EPT * 64 - c1; c2 = NUC_EPT * 64 - c2; c3 = NUC_EPT * 64 - c3;
            float s1, s2, s3;
            count3((step + 1) & 1, (float)c1, (float)c2, (float)c3, s1, s2, s3);
            // keep the LARGEST lim whose count is still < need: then lim is the last id bound that is one short
            lim = s3 < need ? t3 : (s2 < need ? t2 : (s1 < need ? t1 : lim));
        }
        id_bound = lim;                                             // then id `lim` is a tie and the ties with id <= lim are exactly `need`
    }
    if (tid == 0) sel[2] = 0u;
    for (int i = tid; i < NC; i += NUC_THREADS) cand[i] = 0ull;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NUC_EPT; ++j) {
        const int i = j * NUC_THREADS + tid;
        const unsigned key = __float_as_uint(p[j]);
        if (key != 0u && (key > thr || (key == thr && (unsigned)i <= id_bound))) {   // out-of-range elements hold key 0
            const unsigned slot = atomicAdd(&sel[2], 1u);           // slot order is arbitrary, the SET is not; sorted below
            if (slot < NC) cand[slot] = ((unsigned long long)key << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)i);
        }
    }
    __syncthreads();
    for (int size = 2; size <= NC; size <<= 1) {                    // bitonic sort, descending
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < NC; i += NUC_THREADS) {
                const int j = i ^ stride;
                if (j > i) {
                    const bool desc = (i & size) == 0;
                    const unsigned long long a = cand[i], b = cand[j];
                    if ((a < b) == desc) { cand[i] = b; cand[j] = a; }
                }
            }
            __syncthreads();
        }
    }
    const int ncand = min((int)min(sel[2], (unsigned)NC), k);
    // ---- nucleus: keep while the cumulative probability BEFORE the element is <= top_p (nucleus.rs:84-91)
    if (tid == 0) {
        float cum = 0.f;
        int n = 0;
        for (; n < ncand; ++n) {
            if (!MIRO && cum > cut) break;
            cum += prob_of(cand[n]);
        }
        sel[0] = (unsigned)(n < 1 ? 1 : n);
    }
    __syncthreads();
    const int n = (int)sel[0];
    for (int i = tid; i < n; i += NUC_THREADS)
        qv[i] = MIRO ? prob_of(cand[i]) : powf(prob_of(cand[i]), 1.0f / P.temperature);   // nucleus.rs:92, typical.rs:96
    __syncthreads();
    if (tid == 0) {
        float sum = 0.f;
        for (int i = 0; i < n; ++i) sum += qv[i];
        float c = 0.f;
        int pick = 0;                                               // find_or_first: nothing found -> first element
        const float r = P.uniform * sum;                            // mirostat.rs:78-79
        for (int i = 0; i < n; ++i) {
            if (MIRO) { c += qv[i]; if (r <= c) { pick = i; break; } }
            else { c += qv[i] / sum; if (P.uniform <= c) { pick = i; break; } }
        }
        const unsigned long long e = cand[pick];
        out_tok[row] = (int)(0xFFFFFFFFu - (unsigned)(e & 0xFFFFFFFFull));
        if (out_prob) out_prob[row] = MIRO ? log2f(sum) - log2f(prob_of(e)) : prob_of(e);
    }
}

// formatter masks (run.rs:676-679, sampler/bnf.rs:35-38): row rows[j] keeps only the tokens with allow[j][token] != 0
__global__ __launch_bounds__(256) void logit_mask_kernel(float *logits, int V, const int *rows, const unsigned char *allow) {
    const int j = blockIdx.y;
    float *x = logits + (long)rows[j] * V;
    const unsigned char *a = allow + (long)j * V;
    for (int i = (blockIdx.x * 256 + threadIdx.x) * 4; i < V; i += gridDim.x * 1024) {
        if (i + 3 < V) {
            const unsigned m = *(const unsigned *)(a + i);
            float4 v = *(float4 *)(x + i);
            if (!(m & 0xFFu)) v.x = -INFINITY;
            if (!(m & 0xFF00u)) v.y = -INFINITY;
            if (!(m & 0xFF0000u)) v.z = -INFINITY;
            if (!(m & 0xFF000000u)) v.w = -INFINITY;
            *(float4 *)(x + i) = v;
        } else {
            for (int k = i; k < V; ++k) if (!a[k]) x[k] = -INFINITY;
        }
    }
}
void launch_logit_mask(float *logits, int V, const int *rows, const unsigned char *allow, int n, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(logit_mask_kernel, dim3(16, n), dim3(256), 0, s, logits, V, rows, allow);
}
void launch_logit_adjust(float *logits, int V, const int *rows, const int *toks, const float *vals, int n, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(logit_adjust_kernel, dim3((n + 255) / 256), dim3(256), 0, s, logits, V, rows, toks, vals, n);
}
void launch_nucleus(const float *logits, int n_rows, int V, const SampleRow *sp, bool any_nucleus_typical, bool any_mirostat,
                    int *out_tok, float *out_prob, hipStream_t s) {
    constexpr int NC0 = NUC_CAND, NC2 = 8192;
    static bool attr_done[16] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_done[dev & 15]) {
        (void)hipFuncSetAttribute((const void *)nucleus_kernel<NC2, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, NC2 * 12);
        (void)hipFuncSetAttribute((const void *)nucleus_kernel<NC2, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, NC2 * 12);
        attr_done[dev & 15] = true;
    }
    const bool full = V == NUC_EPT * NUC_THREADS;
#define NUC_LAUNCH(nc, miro) do { if (full) hipLaunchKernelGGL((nucleus_kernel<nc, miro, true>), dim3(n_rows), dim3(NUC_THREADS), nc * 12, s, logits, V, sp, out_tok, out_prob); \
                                  else hipLaunchKernelGGL((nucleus_kernel<nc, miro, false>), dim3(n_rows), dim3(NUC_THREADS), nc * 12, s, logits, V, sp, out_tok, out_prob); } while (0)
    if (any_nucleus_typical) NUC_LAUNCH(NC0, false);
    if (any_mirostat) NUC_LAUNCH(NC2, true);
#undef NUC_LAUNCH
}

// =====================================================================================
// Load-time layout kernels
// =====================================================================================
// fp16 tile: out uint4 index ((strip*KT + kt)*64 + lane) <- W[strip*16 + (lane&15)][kt*32 + (lane>>4)*8 .. +8]
__global__ void tile_f16_kernel(const _Float16 *raw, int rows_valid, int rows, int K, uint4 *out) {
    const int KT = K >> 5;
    const long total = (long)(rows >> 4) * KT * 64;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        const long tile = idx >> 6;
        const int kt = (int)(tile % KT), strip = (int)(tile / KT);
        const int row = strip * 16 + (lane & 15), k = kt * 32 + (lane >> 4) * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < rows_valid) v = *(const uint4 *)(raw + (long)row * K + k);
        out[idx] = v;
    }
}
void launch_tile_f16(const _Float16 *raw, int rows_valid, int rows, int K, void *out, hipStream_t s) {
    hipLaunchKernelGGL(tile_f16_kernel, dim3(2048), dim3(256), 0, s, raw, rows_valid, rows, K, (uint4 *)out);
}

// int8: one thread per (row, 128-block).  q = rint((x-b)/a), a = fp16((max-min)/255), b = fp16(min)
// tiled byte position of element (row, k): tile kt=k/64, lane=(row&15)+16*kc with kc=(k%32)/8, byte=(k%64>=32?8:0)+(k%8)
__global__ void quant_int8_kernel(const _Float16 *raw, int rows, int K, unsigned char *out, f16x2 *scales) {
    const int nb = K >> 7;
    const long total = (long)rows * nb;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int row = (int)(idx / nb), blk = (int)(idx % nb);
        const _Float16 *x = raw + (long)row * K + blk * 128;
        float mn = INFINITY, mx = -INFINITY;
        for (int i = 0; i < 128; ++i) { const float v = (float)x[i]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
        const _Float16 ah = (_Float16)((mx - mn) / 255.0f);
        const _Float16 bh = (_Float16)mn;
        const float af = (float)ah, bf = (float)bh;
        const float safe = af > 0.f ? af : 1.0f;
        const int strip = row >> 4, r16 = row & 15;
        const int KT = K >> 6, NG = K >> 8;
        for (int i = 0; i < 128; ++i) {
            const int k = blk * 128 + i;
            float q = rintf(((float)x[i] - bf) / safe);
            q = fminf(fmaxf(q, 0.f), 255.f);
            const int kt = k >> 6, kin = k & 63;
            const int lane = r16 + 16 * ((kin & 31) >> 3);
            const int byte = (kin >= 32 ? 8 : 0) + (kin & 7);
            out[(((long)strip * KT + kt) * 64 + lane) * 16 + byte] = (unsigned char)q;
        }
        // scales layout [strip][K/256][16][2]{a,b}
        scales[(((long)strip * NG + (blk >> 1)) * 16 + r16) * 2 + (blk & 1)] = (f16x2){ah, bh};
    }
}
void launch_quant_int8(const _Float16 *raw, int rows, int K, void *out, void *scales, hipStream_t s) {
    hipLaunchKernelGGL(quant_int8_kernel, dim3(2048), dim3(256), 0, s, raw, rows, K, (unsigned char *)out, (f16x2 *)scales);
}

// nf4: one thread per (row, 64-block).  tiled nibble position of (row,k): tile kt=k/128, lane=(row&15)+16*kc,
// kc=(k%32)/8, kstep=(k%128)/32, e=k%8: byte = kstep*4 + (e&3), nibble = e>>2 (0 = low)
__global__ void quant_nf4_kernel(const _Float16 *raw, int rows, int K, unsigned char *out, _Float16 *scales,
                                 const float *mids) {
    const int nb = K >> 6;
    const long total = (long)rows * nb;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int row = (int)(idx / nb), blk = (int)(idx % nb);
        const _Float16 *x = raw + (long)row * K + blk * 64;
        float am = 0.f;
        for (int i = 0; i < 64; ++i) am = fmaxf(am, fabsf((float)x[i]));
        const _Float16 amh = (_Float16)am;                 // exact: inputs are fp16
        const float safe = am > 0.f ? am : 1.0f;
        const int strip = row >> 4, r16 = row & 15;
        const int KT = K >> 7, NG = K >> 8;
        for (int i = 0; i < 64; i += 8) {
            // 8 consecutive k share (kt, kstep, kc): bytes kstep*4 + 0..3, lo nibble e=0..3, hi nibble e=4..7
            const int k = blk * 64 + i;
            const int kt = k >> 7, kin = k & 127;
            const int kstep = kin >> 5, kc = (kin & 31) >> 3;
            const int lane = r16 + 16 * kc;
            unsigned char code[8];
            for (int e = 0; e < 8; ++e) {
                const float xn = (float)x[i + e] / safe;
                int c = 0;
                for (int m = 0; m < 15; ++m) c += (xn > mids[m]) ? 1 : 0;
                code[e] = (unsigned char)c;
            }
            unsigned char *dst = out + (((long)strip * KT + kt) * 64 + lane) * 16 + kstep * 4;
            for (int b = 0; b < 4; ++b) dst[b] = (unsigned char)(code[b] | (code[b + 4] << 4));
        }
        // scales layout [strip][K/256][16][4] half
        scales[(((long)strip * NG + (blk >> 2)) * 16 + r16) * 4 + (blk & 3)] = amh;
    }
}
static float *g_nf4_mid_dev[16] = {nullptr};
void launch_quant_nf4(const _Float16 *raw, int rows, int K, void *out, void *scales, hipStream_t s) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!g_nf4_mid_dev[dev & 15]) {
        // exact fp32 midpoints, computed like the oracle: (Q[i+1] + Q[i]) * 0.5f
        static const float Q[16] = {-1.0f, -0.6961928009986877f, -0.5250730514526367f, -0.39491748809814453f,
                                    -0.28444138169288635f, -0.18477343022823334f, -0.09105003625154495f, 0.0f,
                                    0.07958029955625534f, 0.16093020141124725f, 0.24611230194568634f, 0.33791524171829224f,
                                    0.44070982933044434f, 0.5626170039176941f, 0.7229568362236023f, 1.0f};
        float mids[15];
        for (int i = 0; i < 15; ++i) mids[i] = (Q[i + 1] + Q[i]) * 0.5f;
        float *d = nullptr;
        (void)hipMalloc(&d, sizeof(mids));
        (void)hipMemcpy(d, mids, sizeof(mids), hipMemcpyHostToDevice);
        g_nf4_mid_dev[dev & 15] = d;
    }
    hipLaunchKernelGGL(quant_nf4_kernel, dim3(2048), dim3(256), 0, s, raw, rows, K, (unsigned char *)out,
                       (_Float16 *)scales, (const float *)g_nf4_mid_dev[dev & 15]);
}

__global__ void f16_to_f32_kernel(const _Float16 *in, float *out, long n, int op) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = (float)in[i];
        if (op == 1) v = expf(-expf(v));
        out[i] = v;
    }
}
void launch_f16_to_f32(const _Float16 *in, float *out, long n, int op, hipStream_t s) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(f16_to_f32_kernel, dim3(blocks), dim3(256), 0, s, in, out, n, op);
}

// LoRA on a non-matrix tensor (`LoraBlend::full(alpha)` matches every tensor, lib.rs:466-482): a LoRA file that carries a tensor of
// the SAME name blends it in whole, v = alpha * l + (1 - alpha) * v (a lerp, like the matrices' W += alpha B A^T only for alpha = 1), on the fp32 copy and BEFORE any load-time transform (op 1: exp(-exp(v))).
__global__ void vec_blend_kernel(float *v, const _Float16 *l, long n, float alpha) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) v[i] = alpha * (float)l[i] + (1.0f - alpha) * v[i];
}
__global__ void vec_op_kernel(float *v, long n, int op) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        if (op == 1) v[i] = expf(-expf(v[i]));
}
void launch_vec_blend(float *v, const _Float16 *l, long n, float alpha, hipStream_t s) {
    const int blocks = (int)std::max<long>(1, std::min<long>(1024, (n + 255) / 256));
    hipLaunchKernelGGL(vec_blend_kernel, dim3(blocks), dim3(256), 0, s, v, l, n, alpha);
}
void launch_vec_op(float *v, long n, int op, hipStream_t s) {
    const int blocks = (int)std::max<long>(1, std::min<long>(1024, (n + 255) / 256));
    hipLaunchKernelGGL(vec_op_kernel, dim3(blocks), dim3(256), 0, s, v, n, op);
}

__global__ void lora_blend_kernel(_Float16 *W, const _Float16 *B, const _Float16 *A, int rows, int K, int r, float alpha) {
    const long total = (long)rows * K;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int row = (int)(idx / K), k = (int)(idx % K);
        float s = 0.f;
        for (int j = 0; j < r; ++j) s += (float)B[(long)row * r + j] * (float)A[(long)k * r + j];
        W[idx] = (_Float16)((float)W[idx] + alpha * s);
    }
}
void launch_lora_blend(_Float16 *W, const _Float16 *B, const _Float16 *A, int rows, int K, int r, float alpha, hipStream_t s) {
    hipLaunchKernelGGL(lora_blend_kernel, dim3(2048), dim3(256), 0, s, W, B, A, rows, K, r, alpha);
}

#endif  // part 4: state pack, softmax / arg-max / samplers, load-time kernels
}  // namespace rwkv
