// dev tool (round 5): can a decode GEMM RUN AHEAD of the row kernel that produces its operand?
//   hipcc --offload-arch=gfx950 -O3 scripts/runahead_bench.hip -o scripts/runahead_bench.bin && scripts/runahead_bench.bin
//
// A decode layer is a chain  ... -> P (row kernel: LayerNorm / token shift / mix; moves no weight bytes, ~5-7 us of latency) -> C (GEMM:
// streams 27 MB of weights once, needs P's 164 KB operand X) -> ...   Stream-ordered, C's weight stream starts when P has exited.  The
// weights do not depend on P: launched on a second branch of the graph, C can pull every weight tile into registers WHILE P runs and
// wait on a flag only for X (the micro-architecture guide's `prefetch-credit` row: 4.8-5.2 us per edge).  This bench prices exactly that
// edge with stand-in kernels of the real geometry, as a graph of NPAIR (P, C) pairs, three ways:
//   serial   P -> C on one stream (what the engine does today; C loads X first, then its weights)
//   gated    C forked off in front of P, weights first, then ONE lane per block polls the hand-off word, then X
//   gated2   the same with the C node captured BEFORE / AFTER the P node (dispatch order is the driver's choice: both are measured)
// Hand-off protocols (MI355X_MICROARCH.md, "valid forms"):
//   fence    P: plain X stores, __syncthreads, lane 0 release fence (agent) + s_waitcnt vmcnt(0) + relaxed atomic add;
//            C: relaxed sc1 poll + s_sleep, ONE acquire fence (agent), __syncthreads, plain X loads
//   wt       P: write-through (sc0 sc1) X stores, s_waitcnt vmcnt(0), __syncthreads, lane 0 relaxed atomic add;
//            C: poll, __syncthreads, sc1 X loads (L1-bypassing), no fence on either side
// Poll target: the arrival counter itself ("ctr") or one of 8 per-XCD flags stored by the last arriver ("xcd").
// Every X word is checked against (pair, replay) — a stale or torn hand-off is counted, never silent; every spin is bounded (2 ms).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);}}while(0)
typedef unsigned int u32;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int XWORDS = 164 * 1024 / 16;        // X operand: 164 KB as u32x4 words (32 rows x 2560 k x 2 B)
constexpr int MAXT = 16;                       // weight tiles (1 KiB) a wave holds: 215 blocks x 10 waves x 13 KiB = 27.9 MB

struct Sync {                                  // one per pair, 128-byte lines apart
    u32 ctr; u32 pad0[31];
    u32 flag[8 * 32];                          // per-XCD flags, one line each
};
struct PArgs { const float *in; u32x4 *X; Sync *sy; const u32 *epoch; int pair, nblk, dur_ticks, proto, poll; };
struct CArgs { const u32x4 *W; const u32x4 *X; float *out; Sync *sy; const u32 *epoch; u32 *err; int pair, tiles, target, proto, poll, gated; };

__device__ __forceinline__ u32 xval(int pair, u32 epoch, int i) { return (u32)pair * 0x9E3779B1u + epoch * 0x85EBCA6Bu + (u32)i; }
__device__ __forceinline__ u32 ld_relaxed(const u32 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int xcc_id() { u32 v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return (int)(v & 7); }

__global__ void bump_epoch(u32 *e) { if (threadIdx.x == 0) *e += 1; }

// ---- P: the row-kernel stand-in.  Reads what the previous C wrote (a dependent fetch), stays busy until `dur` after its start, writes its
// share of X, arrives.
__device__ __forceinline__ void p_body(const PArgs &a) {
    const unsigned long long t0 = wall_clock64();
    const u32 epoch = *a.epoch;
    float acc = 0.f;
    for (int i = threadIdx.x; i < 2560; i += blockDim.x) acc += a.in[(size_t)(blockIdx.x % 32) * 2560 + i];
    while ((long long)(wall_clock64() - t0) < a.dur_ticks) __builtin_amdgcn_s_sleep(1);
    const int per = (XWORDS + a.nblk - 1) / a.nblk, lo = blockIdx.x * per, hi = min(XWORDS, lo + per);
    const u32 bias = acc == 12345.678f ? 1u : 0u;                  // keeps the fetch alive, never true
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const u32 v = xval(a.pair, epoch, i) + bias;
        const u32x4 w = {v, v ^ 0x55555555u, v + 7u, ~v};
        if (a.proto == 1) {
            u32x4 *dst = a.X + i;
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(dst), "v"(w) : "memory");
        } else {
            a.X[i] = w;
        }
    }
    if (a.proto == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (a.proto == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const u32 old = __hip_atomic_fetch_add(&a.sy->ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a.poll == 1 && old == (u32)a.nblk - 1) {
            for (int x = 0; x < 8; ++x) __hip_atomic_store(&a.sy->flag[x * 32], epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- C: the GEMM stand-in.  10 waves; every wave holds `tiles` 1 KiB weight tiles (non-temporal, read once), a 16-tile slice of X,
// multiplies (two MFMAs per tile, like NT = 2), parks in LDS, one barrier, 2 KiB of output per block.
__global__ __launch_bounds__(1024) void p_kernel(const PArgs a) { p_body(a); }

__device__ __forceinline__ void c_body(const CArgs &a, int bid) {
    __shared__ f32x4 red[10 * 64];
    __shared__ u32 s_bad;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 epoch = *a.epoch;
    const u32x4 *w = a.W + ((size_t)(bid * 10 + wave) * a.tiles) * 64 + lane;
    u32x4 wt[MAXT], xb[16];
    u32 bad = 0;
    auto load_w = [&]() {
#pragma unroll
        for (int j = 0; j < MAXT; ++j) if (j < a.tiles) wt[j] = __builtin_nontemporal_load(w + (size_t)j * 64);
    };
    auto load_x = [&](bool sc1) {
        const u32x4 *x = a.X + (size_t)wave * 16 * 64 + lane;     // this wave's K slice of the operand: 16 KiB
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (sc1) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(xb[j]) : "v"(x + j * 64) : "memory");
            else xb[j] = x[j * 64];
        }
        if (sc1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    if (threadIdx.x == 0) s_bad = 0;
    if (a.gated) {
        load_w();
        if (threadIdx.x == 0) {
            const unsigned long long t0 = wall_clock64();
            const u32 *word = a.poll == 1 ? &a.sy->flag[xcc_id() * 32] : &a.sy->ctr;
            const u32 want = a.poll == 1 ? epoch + 1u : (u32)a.target;
            bool ok = false;
            while (!(ok = (a.poll == 1 ? ld_relaxed(word) == want : ld_relaxed(word) >= want))) {
                if ((long long)(wall_clock64() - t0) > 200000) break;       // 2 ms at 100 MHz: never hang the box
                __builtin_amdgcn_s_sleep(2);
            }
            if (!ok) atomicAdd(a.err + 1, 1u);
            if (a.proto == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        load_x(a.proto == 1);
    } else {
        load_x(false);
        load_w();
    }
    // check the operand: word i of X belongs to (pair, epoch)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int i = (wave * 16 + j) * 64 + lane;
        const u32 v = xval(a.pair, epoch, i);
        bad += (xb[j].x != v) | (xb[j].y != (v ^ 0x55555555u)) | (xb[j].z != v + 7u) | (xb[j].w != ~v);
    }
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
        if (j < a.tiles) {
            const f16x8 af = __builtin_bit_cast(f16x8, wt[j]);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, __builtin_bit_cast(f16x8, xb[j & 15]), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, __builtin_bit_cast(f16x8, xb[(j + 8) & 15]), acc1, 0, 0, 0);
        }
    }
    red[wave * 64 + lane] = acc0 + acc1;
    if (bad) atomicAdd(&s_bad, bad);
    __syncthreads();
    if (wave < 2) {
        f32x4 v = red[lane];
        for (int w2 = 1; w2 < 10; ++w2) v += red[w2 * 64 + lane];
        // 32 rows x 2560 floats of "output" shared by the grid (the next P reads it)
        float *o = a.out + ((size_t)(bid * 2 + wave) * 256 + lane * 4) % (32 * 2560);
        *(f32x4 *)o = v;
    }
    if (threadIdx.x == 0 && s_bad) atomicAdd(a.err, s_bad);
}
__global__ __launch_bounds__(640) void c_kernel(const CArgs a) { c_body(a, blockIdx.x); }
// ---- fused: ONE launch, blocks [0, nblk) are the row kernel, the rest the gated GEMM.  Workgroups are dispatched in index order, so the
// producers are resident before any consumer can spin; the grid stays <= 256 so that every block is resident at once (one per CU).
struct FArgs { PArgs p; CArgs c; };
__global__ __launch_bounds__(640) void f_kernel(const FArgs a) {
    if ((int)blockIdx.x < a.p.nblk) p_body(a.p);
    else c_body(a.c, (int)blockIdx.x - a.p.nblk);
}

struct Cfg { const char *name; int mode, order, proto, poll; };   // mode 0 serial, 1 gated; order 0: C node first, 1: P node first

int main(int argc, char **argv) {
    const int NPAIR = 32, REPS = 20;
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    const int Gmax = 256;
    const size_t wbytes = (size_t)Gmax * 10 * MAXT * 1024;   // one phase of weights (upper bound)           // one phase of weights
    const int NBUF = 12;                                              // > Infinity Cache when rotated (12 x 34 MB)
    u32x4 *W; CK(hipMalloc(&W, wbytes * NBUF)); CK(hipMemset(W, 0x3c, wbytes * NBUF));
    u32x4 *X; CK(hipMalloc(&X, (size_t)XWORDS * 16 * 2));
    float *out; CK(hipMalloc(&out, 32 * 2560 * 4 + 4096)); CK(hipMemset(out, 0, 32 * 2560 * 4 + 4096));
    Sync *sy; CK(hipMalloc(&sy, sizeof(Sync) * NPAIR));
    u32 *epoch, *err; CK(hipMalloc(&epoch, 64)); CK(hipMalloc(&err, 64));
    CK(hipMemset(epoch, 0, 64)); CK(hipMemset(err, 0, 64));
    hipEvent_t e0, e1, ef[NPAIR], ej[NPAIR];
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < NPAIR; ++i) { CK(hipEventCreateWithFlags(&ef[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej[i], hipEventDisableTiming)); }

    const Cfg cfgs[] = {
        {"serial (today)            ", 0, 0, 0, 0},
        {"gated fence/ctr  C first  ", 1, 0, 0, 0},
        {"gated fence/ctr  P first  ", 1, 1, 0, 0},
        {"gated fence/xcd  P first  ", 1, 1, 0, 1},
        {"gated wt/ctr     C first  ", 1, 0, 1, 0},
        {"gated wt/ctr     P first  ", 1, 1, 1, 0},
        {"gated wt/xcd     P first  ", 1, 1, 1, 1},
        {"FUSED launch fence/ctr    ", 2, 0, 0, 0},
        {"FUSED launch fence/xcd    ", 2, 0, 0, 1},
        {"FUSED launch wt/ctr       ", 2, 0, 1, 0},
        {"FUSED launch wt/xcd       ", 2, 0, 1, 1},
    };
    struct Geo { const char *name; int pblk, pthr; float dur_us; int G, tiles; };
    const Geo geos[] = {
        {"P = 32 x 1024 (ln_shift), 3.0 us busy; C = 215 x 640, 27.9 MB", 32, 1024, 3.0f, 215, 13},
        {"P = 32 x 1024 (ln_shift), 3.0 us busy; C = 241 x 640, 28.9 MB", 32, 1024, 3.0f, 241, 12},
        {"P = 100 x 512 (v6_mix),   5.5 us busy; C = 215 x 640, 27.9 MB", 100, 512, 5.5f, 215, 13},
        {"P = 100 x 512 (v6_mix),   5.5 us busy; C = 224 x 640, 26.8 MB", 100, 512, 5.5f, 224, 12},
        {"P = 32 x 1024 (ln_shift), 1.0 us busy; C = 215 x 640, 27.9 MB", 32, 1024, 1.0f, 215, 13},
        {"P = 40 x 512 (v6_mix),    5.5 us busy; C = 215 x 640, 27.9 MB", 40, 512, 5.5f, 215, 13},
        {"P = 40 x 512 (v6_mix),    7.0 us busy; C = 215 x 640, 27.9 MB", 40, 512, 7.0f, 215, 13},
        {"P = 32 x 1024 (ln_shift), 3.0 us busy; C = 180 x 640, 29.5 MB (16 tiles per wave)", 32, 1024, 3.0f, 180, 16},
        {"P = 32 x 1024 (ln_shift), 3.0 us busy; C = 224 x 640, 30.3 MB", 32, 1024, 3.0f, 224, 13},
    };
    (void)argc; (void)argv;
    for (const Geo &g : geos) {
        printf("== %s\n", g.name);
        for (const Cfg &c : cfgs) {
            if (c.mode == 2 && g.pblk + g.G > 256) { printf("  %s    (grid %d > 256: skipped)\n", c.name, g.pblk + g.G); continue; }
            if (c.mode == 1 && c.order == 0) continue;
            hipGraph_t gr; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
            CK(hipMemsetAsync(sy, 0, sizeof(Sync) * NPAIR, s0));
            hipLaunchKernelGGL(bump_epoch, dim3(1), dim3(64), 0, s0, epoch);
            for (int p = 0; p < NPAIR; ++p) {
                PArgs pa{out, X + (size_t)(p & 1) * XWORDS, sy + p, epoch, p, g.pblk, (int)(g.dur_us * 100.f), c.proto, c.poll};
                CArgs ca{W + (wbytes / 16) * (p % NBUF), X + (size_t)(p & 1) * XWORDS, out, sy + p, epoch, err, p, g.tiles, g.pblk, c.proto, c.poll, c.mode};
                if (c.mode == 0) {
                    hipLaunchKernelGGL(p_kernel, dim3(g.pblk), dim3(g.pthr), 0, s0, pa);
                    hipLaunchKernelGGL(c_kernel, dim3(g.G), dim3(640), 0, s0, ca);
                } else if (c.mode == 2) {
                    ca.gated = 1;
                    FArgs fa{pa, ca};
                    hipLaunchKernelGGL(f_kernel, dim3(g.pblk + g.G), dim3(640), 0, s0, fa);
                } else {
                    CK(hipEventRecord(ef[p], s0));
                    CK(hipStreamWaitEvent(s1, ef[p], 0));
                    if (c.order == 0) {
                        hipLaunchKernelGGL(c_kernel, dim3(g.G), dim3(640), 0, s1, ca);
                        hipLaunchKernelGGL(p_kernel, dim3(g.pblk), dim3(g.pthr), 0, s0, pa);
                    } else {
                        hipLaunchKernelGGL(p_kernel, dim3(g.pblk), dim3(g.pthr), 0, s0, pa);
                        hipLaunchKernelGGL(c_kernel, dim3(g.G), dim3(640), 0, s1, ca);
                    }
                    CK(hipEventRecord(ej[p], s1));
                    CK(hipStreamWaitEvent(s0, ej[p], 0));
                }
            }
            CK(hipStreamEndCapture(s0, &gr));
            CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
            CK(hipMemset(err, 0, 64));
            for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s0));
            CK(hipStreamSynchronize(s0));
            CK(hipEventRecord(e0, s0));
            for (int i = 0; i < REPS; ++i) CK(hipGraphLaunch(ge, s0));
            CK(hipEventRecord(e1, s0));
            CK(hipEventSynchronize(e1));
            float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
            u32 herr[2]; CK(hipMemcpy(herr, err, 8, hipMemcpyDeviceToHost));
            printf("  %s %7.2f us per (P, C) pair   stale words %u  timeouts %u\n", c.name, ms * 1000.f / (REPS * NPAIR), herr[0], herr[1]);
            fflush(stdout);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(gr));
        }
    }
    return 0;
}
