/* oracle/cpu_backend.c — TEST INFRASTRUCTURE (a second, compiled restatement of the oracle; never linked into the product).
 *
 * The lock-step decode step of oracle/rwkv_ref.py (`RwkvRefBatch.step`: B independent slots advance one token each, run.rs:1121-1132)
 * for RWKV V5.2, V6 and V7, in plain C with OpenMP: fp16 weights (exactly the values the oracle holds: checkpoint tensors rounded through
 * fp16, quantised layers fake-quantised to the fp16 value the GPU dequantises to), fp32 activations and accumulation.  It exists for
 * two reasons: (1) a CPU baseline on the SAME configuration as the GPU line (`bench.py` `cpu_baseline`, SURVEY 8d "CPU reference
 * timing": threaded fp16 GEMV over the fake-quantised weights), (2) an independent implementation of the same published formulas
 * that tests/test_oracle.py holds against the numpy restatement (two codes, one arithmetic).  Formula references: SURVEY Appendix A
 * (BlinkDL's RWKV-5.2 / RWKV-6 inference), data contract as in rwkv_ref.py (state slab [L][N+2][C]: row 0 att shift, rows 1..N the
 * WKV matrices with slab[1+i][h*N+j] = S_h[i][j], row N+1 ffn shift; V5 lerp `xx*mu + sx*(1-mu)`, V6 `xx + (sx-xx)*mu`).
 *
 * Build: gcc -O3 -mavx2 -mfma -mf16c -fopenmp -shared -fPIC oracle/cpu_backend.c -o oracle/_build/libcpu_backend.so -lm
 */
#define _GNU_SOURCE
#include <immintrin.h>
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    const float *ln1w, *ln1b, *ln2w, *ln2b;
    const float *mix_x, *mix_w, *mix_k, *mix_v, *mix_r, *mix_g;   /* V5: mix_k/v/r/g only */
    const uint16_t *mix_w1, *mix_w2;                               /* V6: [5*Dm][C], [5][C][Dm] */
    const float *decay, *first;                                    /* [C] */
    const uint16_t *decay_w1, *decay_w2;                           /* V6: [Dd][C], [C][Dd] */
    const uint16_t *Wr, *Wk, *Wv, *Wg, *Wo;                        /* [C][C] */
    const float *lnxw, *lnxb;
    const float *fmix_k, *fmix_r;
    const uint16_t *Fk, *Fv, *Fr;                                  /* [F][C], [C][F], [C][C] */
    /* V7 (mix_r/w/k/v/g double as x_r/x_w/x_k/x_v/x_g, fmix_k as ffn.x_k; no Wg, no Fr) */
    const float *mix_a, *w0, *a0, *v0, *k_k, *k_a, *r_k;           /* [C] */
    const uint16_t *w1, *w2, *a1, *a2, *v1, *v2, *g1, *g2;         /* x1: [D][C], x2: [C][D] */
    int32_t Dw, Da, Dv, Dg;
} CpuLayer;

typedef struct {
    int32_t version, L, C, F, V, H, Dm, Dd;
    const uint16_t *emb, *head;                                    /* [V][C] */
    const float *ln0w, *ln0b, *lnow, *lnob;
    const CpuLayer *layers;
} CpuModel;

static void master_enter(void);
static void master_leave(void);
#define N_HEAD 64
#define LN_EPS 1e-5f
#define GN_EPS 64e-5f   /* GroupNorm eps of the reference models: 1e-5 * head_size_divisor^2 (8^2), as in rwkv_ref.GN_EPS */

/* Y[b][r] = sum_k W[r][k] X[b][k];  W fp16 row-major [rows][K], K % 8 == 0;  slots in groups of 8 so the accumulators stay in registers */
/* Called by EVERY thread of the step's team (orphaned worksharing loop, implicit barrier at its end): one persistent team per step
 * instead of a fork per GEMM.  The static schedule over rows is the one rwkv_cpu_place used to first-touch W, so a thread reads the
 * rows that live on its own NUMA node. */
/* Error-attribution switch (scripts/fp16_error_attribution.py; off = 0 in every parity test and in the bench): bit `cls` set ->
 * the X operand of the GEMMs of that class is rounded to fp16 on the way in, which is what `Precision::Fp16` does on the GPU
 * (reload.rs:89-94: f16 operands, fp32 accumulate).  Lets the CPU say WHICH operand class carries a model's Fp16 error. */
enum { CLS_ATT = 0, CLS_LORA1 = 1, CLS_LORA2 = 2, CLS_WO = 3, CLS_FFN1 = 4, CLS_FV = 5, CLS_MIX1 = 6, CLS_MIX2 = 7, CLS_DECAY2 = 8, CLS_HEAD = 9,
       /* round 6: V6's time-mix projections one by one (bit CLS_ATT rounds all five; bits 10..14 round one each) — which of them must read hi + lo */
       CLS_ATT_R = 10, CLS_ATT_K = 11, CLS_ATT_V = 12, CLS_ATT_G = 13, CLS_ATT_W = 14 };
static int g_f16_mask = 0;
void rwkv_cpu_set_operand_rounding(int mask) { g_f16_mask = mask; }
static inline __m256 round_f16(__m256 x) {
    x = _mm256_min_ps(_mm256_max_ps(x, _mm256_set1_ps(-65504.0f)), _mm256_set1_ps(65504.0f));
    return _mm256_cvtph_ps(_mm256_cvtps_ph(x, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC));
}
/* Round 6: weight-reusing batched form.  Two rows at a time, K in panels of GEMM_KP: the panel of both rows is converted fp16 -> fp32 ONCE into
 * an L1-resident buffer and multiplied against all B slots (four slots per pass: 8 accumulators + 2 weight + 4 operand vectors = 14 of the 16
 * AVX2 registers; 6 loads per 8 FMAs), so a weight is read from memory once and converted once per step whatever the batch.  (Until round 5
 * every group of 8 slots re-read and re-converted the row, one load per FMA: 27 GB/s of weights at 32 slots on 16 cores.)  The arithmetic of
 * one (row, slot) dot product is unchanged — eight lane-wise partial sums over k in k order, then the same horizontal sum — so every result
 * is bit-identical to the old loop nest (tests/test_oracle.py holds the backend against the numpy restatement either way). */
#define GEMM_KP 1024
static inline float hsum8(__m256 v) {
    __m128 s = _mm_add_ps(_mm256_castps256_ps128(v), _mm256_extractf128_ps(v, 1));
    s = _mm_add_ps(s, _mm_movehl_ps(s, s));
    s = _mm_add_ss(s, _mm_shuffle_ps(s, s, 1));
    return _mm_cvtss_f32(s);
}
static void gemm_f16(const uint16_t *W, long rows, long K, const float *X, long ldx, float *Y, long ldy, int B, int cls) {
    if (B > 64) {                                                      /* (the carried partial sums below are sized for 64 slots) */
        for (int b = 0; b < B; b += 64) gemm_f16(W, rows, K, X + (long)b * ldx, ldx, Y + (long)b * ldy, ldy, B - b < 64 ? B - b : 64, cls);
        return;
    }
    const int rnd = ((g_f16_mask >> cls) & 1) | ((cls >= CLS_ATT_R && cls <= CLS_ATT_W) ? (g_f16_mask & 1) : 0);
    const long npair = (rows + 1) / 2;
#pragma omp for schedule(static)
    for (long p = 0; p < npair; ++p) {
        const long r0 = 2 * p;
        const int nr = rows - r0 < 2 ? 1 : 2;
        float wf[2][GEMM_KP] __attribute__((aligned(32)));
        __m256 accb[2][64];                                            /* lane-wise partial sums of (row, slot), carried across the K panels; B <= 64 */
        for (long k0 = 0; k0 < K; k0 += GEMM_KP) {
            const long kn = K - k0 < GEMM_KP ? K - k0 : GEMM_KP;
            for (int i = 0; i < nr; ++i)
                for (long k = 0; k < kn; k += 8)
                    _mm256_store_ps(&wf[i][k], _mm256_cvtph_ps(_mm_loadu_si128((const __m128i *)(W + (r0 + i) * K + k0 + k))));
            if (nr == 1) for (long k = 0; k < kn; k += 8) _mm256_store_ps(&wf[1][k], _mm256_setzero_ps());
            for (int b0 = 0; b0 < B; b0 += 4) {
                const int nb = B - b0 < 4 ? B - b0 : 4;
                __m256 a00, a01, a02, a03, a10, a11, a12, a13;
                if (k0 == 0) { a00 = a01 = a02 = a03 = a10 = a11 = a12 = a13 = _mm256_setzero_ps(); }
                else { a00 = accb[0][b0]; a01 = accb[0][b0 + 1]; a02 = accb[0][b0 + 2]; a03 = accb[0][b0 + 3];
                       a10 = accb[1][b0]; a11 = accb[1][b0 + 1]; a12 = accb[1][b0 + 2]; a13 = accb[1][b0 + 3]; }
                const float *x0 = X + (long)b0 * ldx + k0, *x1 = X + (long)(b0 + (nb > 1 ? 1 : 0)) * ldx + k0;      /* (slots past B re-read a valid one; their sums are dropped) */
                const float *x2 = X + (long)(b0 + (nb > 2 ? 2 : 0)) * ldx + k0, *x3 = X + (long)(b0 + (nb > 3 ? 3 : 0)) * ldx + k0;
                if (!rnd) {
                    for (long k = 0; k < kn; k += 8) {
                        const __m256 w0 = _mm256_load_ps(&wf[0][k]), w1 = _mm256_load_ps(&wf[1][k]);
                        const __m256 v0 = _mm256_loadu_ps(x0 + k), v1 = _mm256_loadu_ps(x1 + k), v2 = _mm256_loadu_ps(x2 + k), v3 = _mm256_loadu_ps(x3 + k);
                        a00 = _mm256_fmadd_ps(w0, v0, a00); a01 = _mm256_fmadd_ps(w0, v1, a01); a02 = _mm256_fmadd_ps(w0, v2, a02); a03 = _mm256_fmadd_ps(w0, v3, a03);
                        a10 = _mm256_fmadd_ps(w1, v0, a10); a11 = _mm256_fmadd_ps(w1, v1, a11); a12 = _mm256_fmadd_ps(w1, v2, a12); a13 = _mm256_fmadd_ps(w1, v3, a13);
                    }
                } else {
                    for (long k = 0; k < kn; k += 8) {
                        const __m256 w0 = _mm256_load_ps(&wf[0][k]), w1 = _mm256_load_ps(&wf[1][k]);
                        const __m256 v0 = round_f16(_mm256_loadu_ps(x0 + k)), v1 = round_f16(_mm256_loadu_ps(x1 + k));
                        const __m256 v2 = round_f16(_mm256_loadu_ps(x2 + k)), v3 = round_f16(_mm256_loadu_ps(x3 + k));
                        a00 = _mm256_fmadd_ps(w0, v0, a00); a01 = _mm256_fmadd_ps(w0, v1, a01); a02 = _mm256_fmadd_ps(w0, v2, a02); a03 = _mm256_fmadd_ps(w0, v3, a03);
                        a10 = _mm256_fmadd_ps(w1, v0, a10); a11 = _mm256_fmadd_ps(w1, v1, a11); a12 = _mm256_fmadd_ps(w1, v2, a12); a13 = _mm256_fmadd_ps(w1, v3, a13);
                    }
                }
                if (k0 + kn < K) {
                    accb[0][b0] = a00; accb[0][b0 + 1] = a01; accb[0][b0 + 2] = a02; accb[0][b0 + 3] = a03;
                    accb[1][b0] = a10; accb[1][b0 + 1] = a11; accb[1][b0 + 2] = a12; accb[1][b0 + 3] = a13;
                } else {
                    const __m256 r0v[4] = {a00, a01, a02, a03}, r1v[4] = {a10, a11, a12, a13};
                    for (int i = 0; i < nb; ++i) {
                        Y[(long)(b0 + i) * ldy + r0] = hsum8(r0v[i]);
                        if (nr == 2) Y[(long)(b0 + i) * ldy + r0 + 1] = hsum8(r1v[i]);
                    }
                }
            }
        }
    }
}

static void layernorm(const float *x, const float *w, const float *b, float *y, int n, float eps) {
    float m = 0.f;
    for (int i = 0; i < n; ++i) m += x[i];
    m /= (float)n;
    float v = 0.f;
    for (int i = 0; i < n; ++i) v += (x[i] - m) * (x[i] - m);
    v /= (float)n;
    const float inv = 1.0f / sqrtf(v + eps);
    for (int i = 0; i < n; ++i) y[i] = (x[i] - m) * inv * w[i] + b[i];
}
static inline float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }
static inline float h2f(uint16_t h) { return _cvtsh_ss(h); }

/* one decode step: tokens[B]; states [B][L][N+2][C] updated in place; logits [B][V] or NULL.  Returns 0, or -1 on bad arguments / no memory. */
int rwkv_cpu_step(const CpuModel *m, const int32_t *tokens, int B, float *states, float *logits) {
    if (!m || !tokens || !states || B <= 0 || m->C != m->H * N_HEAD || (m->version != 5 && m->version != 6 && m->version != 7)) return -1;
    const int C = m->C, F = m->F, H = m->H, N = N_HEAD, L = m->L, Dm = m->Dm, Dd = m->Dd;
    const long slab = (long)L * (N + 2) * C;
    const long big = F > C ? F : C;
    long maxD = 8;
    if (m->version == 7)
        for (int l = 0; l < L; ++l) {
            const CpuLayer *q = &m->layers[l];
            const long d4[4] = {q->Dw, q->Da, q->Dv, q->Dg};
            for (int i = 0; i < 4; ++i) if (d4[i] > maxD) maxD = d4[i];
        }
    float *buf = (float *)malloc(sizeof(float) * (size_t)B * (size_t)(16 * (long)C + 2 * big + 5L * (Dm > 0 ? Dm : 1) + (Dd > 0 ? Dd : 1) + maxD));
    if (!buf) return -1;
    float *x = buf, *xx = x + (long)B * C, *sx = xx + (long)B * C, *t0 = sx + (long)B * C, *t1 = t0 + (long)B * C, *t2 = t1 + (long)B * C,
          *t3 = t2 + (long)B * C, *t4 = t3 + (long)B * C, *r = t4 + (long)B * C, *k = r + (long)B * C, *v = k + (long)B * C, *g = v + (long)B * C,
          *hid = g + (long)B * C, *hid2 = hid + (long)B * big, *mm = hid2 + (long)B * big, *td = mm + (long)B * 5 * (Dm > 0 ? Dm : 1),
          *vfirst = td + (long)B * (Dd > 0 ? Dd : 1), *aa = vfirst + (long)B * C, *kk = aa + (long)B * C, *xa = kk + (long)B * C,
          *lora = xa + (long)B * C;                                   /* [B][maxD] */
    master_enter();
#pragma omp parallel
    {
#pragma omp for
    for (int b = 0; b < B; ++b) {
        float *xb = x + (long)b * C;
        const uint16_t *e = m->emb + (long)tokens[b] * C;
        for (int c = 0; c < C; ++c) xb[c] = h2f(e[c]);
        layernorm(xb, m->ln0w, m->ln0b, xb, C, LN_EPS);
    }
    for (int l = 0; l < L; ++l) {
        const CpuLayer *p = &m->layers[l];
        /* ---- time mix */
#pragma omp for
        for (int b = 0; b < B; ++b) {
            float *st = states + (long)b * slab + (long)l * (N + 2) * C;
            layernorm(x + (long)b * C, p->ln1w, p->ln1b, xx + (long)b * C, C, LN_EPS);
            memcpy(sx + (long)b * C, st, sizeof(float) * C);
            memcpy(st, xx + (long)b * C, sizeof(float) * C);
        }
        float *xr = t0, *xk = t1, *xv = t2, *xg = t3, *xw = t4;
        if (m->version == 7) {
            /* x_n = xx + dx * mu_n, n in (r,w,k,v,a,g);  r,k,v projections;  w/a/g/v LoRAs;  kappa = normalised k*k_k per head;
             * k <- k (1 + (a-1) k_a);  v <- v + (v_first - v) sigmoid(v0 + V2 V1 x_v) (layers > 0);  decay = exp(-0.606531 sigmoid(w0 + W2 tanh(W1 x_w)));
             * S <- S diag(decay) + (S (-kappa)) (kappa a)^T + v k^T;  out = S r;  y = GN(out) + (sum_j r_j k_j r_k_j) v;  att = Wo (y g) */
#pragma omp for
            for (int b = 0; b < B; ++b)
                for (int c = 0; c < C; ++c) {
                    const long i = (long)b * C + c;
                    const float d = sx[i] - xx[i];
                    xr[i] = xx[i] + d * p->mix_r[c]; xw[i] = xx[i] + d * p->mix_w[c]; xk[i] = xx[i] + d * p->mix_k[c];
                    xv[i] = xx[i] + d * p->mix_v[c]; xa[i] = xx[i] + d * p->mix_a[c]; xg[i] = xx[i] + d * p->mix_g[c];
                }
            gemm_f16(p->Wr, C, C, xr, C, r, C, B, CLS_ATT);
            gemm_f16(p->Wk, C, C, xk, C, k, C, B, CLS_ATT);
            gemm_f16(p->Wv, C, C, xv, C, v, C, B, CLS_ATT);
            float *wdec = hid2;
            gemm_f16(p->w1, p->Dw, C, xw, C, lora, p->Dw, B, CLS_LORA1);
#pragma omp for
            for (long i = 0; i < (long)B * p->Dw; ++i) lora[i] = tanhf(lora[i]);
            gemm_f16(p->w2, C, p->Dw, lora, p->Dw, wdec, C, B, CLS_LORA2);
            gemm_f16(p->a1, p->Da, C, xa, C, lora, p->Da, B, CLS_LORA1);
            gemm_f16(p->a2, C, p->Da, lora, p->Da, aa, C, B, CLS_LORA2);
            gemm_f16(p->g1, p->Dg, C, xg, C, lora, p->Dg, B, CLS_LORA1);
#pragma omp for
            for (long i = 0; i < (long)B * p->Dg; ++i) lora[i] = sigmoidf(lora[i]);
            gemm_f16(p->g2, C, p->Dg, lora, p->Dg, g, C, B, CLS_LORA2);
            float *vgate = hid;                                      /* [B][C] */
            if (l > 0) {
                gemm_f16(p->v1, p->Dv, C, xv, C, lora, p->Dv, B, CLS_LORA1);
                gemm_f16(p->v2, C, p->Dv, lora, p->Dv, vgate, C, B, CLS_LORA2);
            }
            float *out = t0;
#pragma omp for collapse(2)
            for (int b = 0; b < B; ++b)
                for (int h = 0; h < H; ++h) {
                    const long o = (long)b * C + (long)h * N;
                    float kap[N_HEAD], ka[N_HEAD], kh[N_HEAD], vh[N_HEAD], wh[N_HEAD], oo[N_HEAD];
                    float nrm = 0.f;
                    for (int j = 0; j < N; ++j) {
                        const int c = h * N + j;
                        const float a = sigmoidf(p->a0[c] + aa[o + j]);
                        kap[j] = k[o + j] * p->k_k[c];
                        nrm += kap[j] * kap[j];
                        kh[j] = k[o + j] * (1.0f + (a - 1.0f) * p->k_a[c]);
                        ka[j] = a;
                        float vv = v[o + j];
                        if (l == 0) vfirst[o + j] = vv;
                        else vv = vv + (vfirst[o + j] - vv) * sigmoidf(p->v0[c] + vgate[o + j]);
                        vh[j] = vv;
                        wh[j] = expf(-0.606531f * sigmoidf(p->w0[c] + wdec[o + j]));
                    }
                    nrm = sqrtf(nrm);
                    if (nrm < 1e-12f) nrm = 1e-12f;
                    for (int j = 0; j < N; ++j) { kap[j] /= nrm; ka[j] *= kap[j]; }
                    float *S = states + (long)b * slab + (long)l * (N + 2) * C + (long)C + (long)h * N;   /* S[i][j] at S[i*C + j] */
                    float dot = 0.f;
                    for (int j = 0; j < N; ++j) dot += r[o + j] * kh[j] * p->r_k[h * N + j];
                    for (int i = 0; i < N; ++i) {
                        float *Si = S + (long)i * C;
                        float sa = 0.f;
                        for (int j = 0; j < N; ++j) sa -= Si[j] * kap[j];
                        float acc = 0.f;
                        for (int j = 0; j < N; ++j) {
                            const float sn = Si[j] * wh[j] + sa * ka[j] + vh[i] * kh[j];
                            Si[j] = sn;
                            acc += sn * r[o + j];
                        }
                        oo[i] = acc;
                    }
                    float mean = 0.f, var = 0.f;
                    for (int j = 0; j < N; ++j) mean += oo[j];
                    mean /= (float)N;
                    for (int j = 0; j < N; ++j) var += (oo[j] - mean) * (oo[j] - mean);
                    var /= (float)N;
                    const float inv = 1.0f / sqrtf(var + GN_EPS);
                    for (int j = 0; j < N; ++j) {
                        const int c = h * N + j;
                        out[o + j] = ((oo[j] - mean) * inv * p->lnxw[c] + p->lnxb[c] + dot * vh[j]) * g[o + j];
                    }
                }
            gemm_f16(p->Wo, C, C, out, C, t1, C, B, CLS_WO);
        } else {
        if (m->version == 5) {
#pragma omp for
            for (int b = 0; b < B; ++b)
                for (int c = 0; c < C; ++c) {
                    const long i = (long)b * C + c;
                    xr[i] = xx[i] * p->mix_r[c] + sx[i] * (1.0f - p->mix_r[c]);
                    xk[i] = xx[i] * p->mix_k[c] + sx[i] * (1.0f - p->mix_k[c]);
                    xv[i] = xx[i] * p->mix_v[c] + sx[i] * (1.0f - p->mix_v[c]);
                    xg[i] = xx[i] * p->mix_g[c] + sx[i] * (1.0f - p->mix_g[c]);
                }
        } else {
            /* z = xx + dx*mix_x;  m = tanh(W1 z) [5*Dm];  x_c = xx + dx*(mix_c + W2_c m_c), c in (w,k,v,r,g) */
#pragma omp for
            for (int b = 0; b < B; ++b)
                for (int c = 0; c < C; ++c) {
                    const long i = (long)b * C + c;
                    hid[i] = xx[i] + (sx[i] - xx[i]) * p->mix_x[c];
                }
            gemm_f16(p->mix_w1, 5L * Dm, C, hid, C, mm, 5L * Dm, B, CLS_MIX1);
#pragma omp for
            for (long i = 0; i < (long)B * 5 * Dm; ++i) mm[i] = tanhf(mm[i]);
            float *dst[5] = {xw, xk, xv, xr, xg};
            const float *mu[5] = {p->mix_w, p->mix_k, p->mix_v, p->mix_r, p->mix_g};
            for (int c5 = 0; c5 < 5; ++c5) {
                gemm_f16(p->mix_w2 + (long)c5 * C * Dm, C, Dm, mm + (long)c5 * Dm, 5L * Dm, hid, C, B, CLS_MIX2);
#pragma omp for
                for (int b = 0; b < B; ++b)
                    for (int c = 0; c < C; ++c) {
                        const long i = (long)b * C + c;
                        dst[c5][i] = xx[i] + (sx[i] - xx[i]) * (mu[c5][c] + hid[i]);
                    }
            }
        }
        gemm_f16(p->Wr, C, C, xr, C, r, C, B, m->version == 6 ? CLS_ATT_R : CLS_ATT);
        gemm_f16(p->Wk, C, C, xk, C, k, C, B, m->version == 6 ? CLS_ATT_K : CLS_ATT);
        gemm_f16(p->Wv, C, C, xv, C, v, C, B, m->version == 6 ? CLS_ATT_V : CLS_ATT);
        gemm_f16(p->Wg, C, C, xg, C, g, C, B, m->version == 6 ? CLS_ATT_G : CLS_ATT);
        float *wdec = hid2;                                         /* [B][C] */
        if (m->version == 5) {
#pragma omp for
            for (int b = 0; b < B; ++b)
                for (int c = 0; c < C; ++c) wdec[(long)b * C + c] = expf(-expf(p->decay[c]));
        } else {
            gemm_f16(p->decay_w1, Dd, C, xw, C, td, Dd, B, CLS_ATT_W);
#pragma omp for
            for (long i = 0; i < (long)B * Dd; ++i) td[i] = tanhf(td[i]);
            gemm_f16(p->decay_w2, C, Dd, td, Dd, wdec, C, B, CLS_DECAY2);
#pragma omp for
            for (int b = 0; b < B; ++b)
                for (int c = 0; c < C; ++c) wdec[(long)b * C + c] = expf(-expf(p->decay[c] + wdec[(long)b * C + c]));
        }
        /* WKV: out_j = sum_i r_i (u_i k_i v_j + S_ij);  S_ij = k_i v_j + w_i S_ij */
        float *out = t0;                                            /* xr is dead */
#pragma omp for collapse(2)
        for (int b = 0; b < B; ++b)
            for (int h = 0; h < H; ++h) {
                float *S = states + (long)b * slab + (long)l * (N + 2) * C + (long)C + (long)h * N;   /* row 1, column h*N; row stride C */
                const float *rb = r + (long)b * C + h * N, *kb = k + (long)b * C + h * N, *vb = v + (long)b * C + h * N;
                const float *wb = wdec + (long)b * C + h * N, *u = p->first + h * N;
                float o[N_HEAD];
                for (int j = 0; j < N; ++j) o[j] = 0.f;
                for (int i = 0; i < N; ++i) {
                    float *Si = S + (long)i * C;
                    const float ri = rb[i], ki = kb[i], ui = u[i], wi = wb[i];
                    for (int j = 0; j < N; ++j) {
                        const float a = ki * vb[j];
                        o[j] += ri * (ui * a + Si[j]);
                        Si[j] = a + wi * Si[j];
                    }
                }
                /* GroupNorm over the head, then the gate */
                float mean = 0.f, var = 0.f;
                for (int j = 0; j < N; ++j) mean += o[j];
                mean /= (float)N;
                for (int j = 0; j < N; ++j) var += (o[j] - mean) * (o[j] - mean);
                var /= (float)N;
                const float inv = 1.0f / sqrtf(var + GN_EPS);
                for (int j = 0; j < N; ++j) {
                    const int c = h * N + j;
                    const float gg = g[(long)b * C + c];
                    out[(long)b * C + c] = ((o[j] - mean) * inv * p->lnxw[c] + p->lnxb[c]) * (gg * sigmoidf(gg));
                }
            }
        gemm_f16(p->Wo, C, C, out, C, t1, C, B, CLS_WO);
        }
        /* ---- channel mix */
#pragma omp for
        for (int b = 0; b < B; ++b) {
            float *st = states + (long)b * slab + (long)l * (N + 2) * C + (long)(N + 1) * C;
            float *xb = x + (long)b * C;
            for (int c = 0; c < C; ++c) xb[c] += t1[(long)b * C + c];
            layernorm(xb, p->ln2w, p->ln2b, xx + (long)b * C, C, LN_EPS);
            memcpy(sx + (long)b * C, st, sizeof(float) * C);
            memcpy(st, xx + (long)b * C, sizeof(float) * C);
            for (int c = 0; c < C; ++c) {
                const long i = (long)b * C + c;
                if (m->version == 7) {
                    t2[i] = xx[i] + (sx[i] - xx[i]) * p->fmix_k[c];
                } else if (m->version == 5) {
                    t2[i] = xx[i] * p->fmix_k[c] + sx[i] * (1.0f - p->fmix_k[c]);
                    t3[i] = xx[i] * p->fmix_r[c] + sx[i] * (1.0f - p->fmix_r[c]);
                } else {
                    t2[i] = xx[i] + (sx[i] - xx[i]) * p->fmix_k[c];
                    t3[i] = xx[i] + (sx[i] - xx[i]) * p->fmix_r[c];
                }
            }
        }
        gemm_f16(p->Fk, F, C, t2, C, hid, F, B, CLS_FFN1);
#pragma omp for schedule(static)
        for (long i = 0; i < (long)B * F; ++i) { const float a = hid[i] > 0.f ? hid[i] : 0.f; hid[i] = a * a; }
        gemm_f16(p->Fv, C, F, hid, F, t4, C, B, CLS_FV);
        if (m->version == 7) {
#pragma omp for
            for (int b = 0; b < B; ++b)
                for (int c = 0; c < C; ++c) x[(long)b * C + c] += t4[(long)b * C + c];
        } else {
            gemm_f16(p->Fr, C, C, t3, C, r, C, B, CLS_FFN1);
#pragma omp for
            for (int b = 0; b < B; ++b)
                for (int c = 0; c < C; ++c) x[(long)b * C + c] += sigmoidf(r[(long)b * C + c]) * t4[(long)b * C + c];
        }
    }
    if (logits) {
#pragma omp for
        for (int b = 0; b < B; ++b) layernorm(x + (long)b * C, m->lnow, m->lnob, xx + (long)b * C, C, LN_EPS);
        gemm_f16(m->head, m->V, C, xx, C, logits, m->V, B, CLS_HEAD);
    }
    }   /* omp parallel */
    master_leave();
    free(buf);
    return 0;
}

/* ---- Int8 fake-quantisation in place, bit for bit what rwkv_ref.fake_quant(w, QUANT_INT8) returns (rwkv_ref.quant_int8 /
 * dequant_int8): per 128-block a = f16((max-min)/255), b = f16(min), q = clip(rint((x-b)/a)), value = f16(double(a)*q + double(b)). */
static uint16_t f64_to_f16(double d) {                             /* round to nearest even, like numpy's float64 -> float16 */
    if (d != d) return 0x7e00;
    const uint16_t sign = d < 0 ? 0x8000 : 0;
    double a = fabs(d);
    if (a == 0.0) return sign;
    if (a >= 65520.0) return (uint16_t)(sign | 0x7c00);             /* rounds to infinity */
    int e;
    (void)frexp(a, &e);                                            /* a = f * 2^e, f in [0.5, 1) -> a in [2^(e-1), 2^e) */
    int exp = e - 1;                                               /* floor(log2 a) */
    if (exp < -14) exp = -14;                                      /* subnormal quantum 2^-24 */
    const double q = ldexp(1.0, exp - 10);                         /* spacing of halfs in this binade */
    double mant = nearbyint(a / q);                                /* exact division by a power of two; ties to even (default mode) */
    if (mant >= 2048.0) { mant /= 2.0; exp += 1; }                 /* rounded up into the next binade */
    if (exp > 15) return (uint16_t)(sign | 0x7c00);
    if (mant < 1024.0) return (uint16_t)(sign | (uint16_t)mant);   /* subnormal (exp == -14, no implicit bit) */
    return (uint16_t)(sign | ((uint16_t)(exp + 15) << 10) | ((uint16_t)mant & 0x3ff));
}
static uint16_t f32_to_f16(float f) { return f64_to_f16((double)f); }   /* exact widening, one rounding */

void rwkv_cpu_fake_quant_int8(uint16_t *w, long rows, long K) {
#pragma omp parallel for schedule(static)
    for (long rr = 0; rr < rows; ++rr)
        for (long k0 = 0; k0 < K; k0 += 128) {
            uint16_t *blk = w + rr * K + k0;
            float x[128], mn = 0.f, mx = 0.f;
            for (int i = 0; i < 128; ++i) { x[i] = h2f(blk[i]); if (i == 0 || x[i] < mn) mn = x[i]; if (i == 0 || x[i] > mx) mx = x[i]; }
            const uint16_t ah = f32_to_f16((mx - mn) / 255.0f), bh = f32_to_f16(mn);
            const float a32 = h2f(ah), b32 = h2f(bh), safe = a32 > 0.f ? a32 : 1.0f;
            for (int i = 0; i < 128; ++i) {
                float q = nearbyintf((x[i] - b32) / safe);
                q = q < 0.f ? 0.f : q > 255.f ? 255.f : q;
                blk[i] = f64_to_f16((double)a32 * (double)q + (double)b32);
            }
        }
}
/* NF4 fake-quantisation in place (rwkv_ref.quant_nf4 / dequant_nf4): per 64-block absmax in fp16, index = number of the 15 fp32
 * midpoints below x / absmax, value = f16(double(absmax) * double(table[index])).  The tables come from rwkv_ref so there is one copy. */
void rwkv_cpu_fake_quant_nf4(uint16_t *w, long rows, long K, const float *mid15, const uint16_t *table16) {
#pragma omp parallel for schedule(static)
    for (long rr = 0; rr < rows; ++rr)
        for (long k0 = 0; k0 < K; k0 += 64) {
            uint16_t *blk = w + rr * K + k0;
            float x[64], am = 0.f;
            for (int i = 0; i < 64; ++i) { x[i] = h2f(blk[i]); const float a = fabsf(x[i]); if (a > am) am = a; }
            const uint16_t amh = f32_to_f16(am);
            const float am32 = h2f(amh), safe = am32 > 0.f ? am32 : 1.0f;
            for (int i = 0; i < 64; ++i) {
                const float xn = x[i] / safe;
                int idx = 0;
                for (int t = 0; t < 15; ++t) idx += xn > mid15[t];
                blk[i] = f64_to_f16((double)am32 * (double)h2f(table16[idx]));
            }
        }
}
/* ---- threads and memory placement (the baseline should be limited by the host's DRAM, not by where its pages happen to live).
 * rwkv_cpu_pin: team size = n, thread t pinned to logical CPU cpus[t] (the caller passes one CPU per physical core, socket by socket);
 * libgomp keeps its pool, so the pinning holds for every later parallel region of this process.
 * rwkv_cpu_place: a copy of a weight matrix whose pages are FIRST TOUCHED by the thread that gemm_f16's static row schedule will hand
 * those rows to — on a multi-socket host every thread then streams from its own NUMA node.  (numpy had first-touched every page from
 * one thread: all weights on one node.) */
#include <sched.h>
#include <pthread.h>
static int g_master_cpu = -1;                                      /* the caller's thread is pinned only while it works in the team */
static cpu_set_t g_master_saved;
static void master_enter(void) {
    if (g_master_cpu < 0) return;
    cpu_set_t set;
    (void)pthread_getaffinity_np(pthread_self(), sizeof(g_master_saved), &g_master_saved);
    CPU_ZERO(&set);
    CPU_SET(g_master_cpu, &set);
    (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
}
static void master_leave(void) {
    if (g_master_cpu >= 0) (void)pthread_setaffinity_np(pthread_self(), sizeof(g_master_saved), &g_master_saved);
}
int rwkv_cpu_pin(const int32_t *cpus, int n) {
    if (n <= 0) return -1;
    omp_set_dynamic(0);
    omp_set_num_threads(n);
    g_master_cpu = cpus ? cpus[0] : -1;
    int bad = 0;
#pragma omp parallel num_threads(n) reduction(+ : bad)
    {
        const int t = omp_get_thread_num();
        if (cpus && t > 0) {                                       /* pool threads stay where they are put; the caller's thread: master_enter */
            cpu_set_t set;
            CPU_ZERO(&set);
            CPU_SET(cpus[t], &set);
            if (pthread_setaffinity_np(pthread_self(), sizeof(set), &set) != 0) bad += 1;
        }
    }
    return bad;
}
uint16_t *rwkv_cpu_place(const uint16_t *src, long rows, long K) {
    void *mem = NULL;
    if (posix_memalign(&mem, 4096, (size_t)rows * (size_t)K * 2 + 64) != 0) return NULL;
    uint16_t *dst = (uint16_t *)mem;
    master_enter();
#pragma omp parallel for schedule(static)
    for (long r = 0; r < rows; ++r) memcpy(dst + r * K, src + r * K, (size_t)K * 2);
    master_leave();
    return dst;
}
void rwkv_cpu_free(void *p) { free(p); }

int rwkv_cpu_threads(void) {
    int n = 1;
#pragma omp parallel
    {
#pragma omp master
        n = omp_get_num_threads();
    }
    return n;
}
