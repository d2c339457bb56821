// dev tool (round 6): the decode GEMM as an LDS-DMA loader / consumer stream (Next #3) against today's K-stationary VGPR stream.
//   hipcc --offload-arch=gfx950 -O3 scripts/ldsdma_bench.hip -o scripts/ldsdma_bench.bin && scripts/ldsdma_bench.bin
//
// Stand-in kernels at the real geometry of the V6-3B Int8 r/k/v/g/decay launch (645 strips of 16 rows x K = 2560 as 1 KiB tiles of 64 k:
// 26.4 MB per launch), one MFMA per (k-step, token tile), as a graph of dependent launches over rotating weight buffers (12 x 26.4 MB > the
// 256 MB Infinity Cache):
//   kstat<NT>      today's structure: 215 blocks x 10 waves, wave = 3 strips x 256 k, all 12 weight tiles + the wave's X slice (8 NT KiB from
//                  L2) in flight at once in VGPRs, LDS park + barrier + 10-way reduce.  NT = token tiles (1: <= 16 rows, 2: <= 32 rows).
//   ldsdma<NC,NT>  one LOADER wave per block issues `global_load_lds_dwordx4 ... nt` for every tile of the block's strips (<= 3 strips = 120 KiB:
//                  the whole block's weights fit in LDS, no ring reuse), publishing its progress in an LDS word after a counted
//                  `s_waitcnt vmcnt(56)` (seven 8-tile groups stay in flight); NC CONSUMER waves each own the k-tiles {c, c + NC, ...} of every
//                  strip (X slice in VGPRs, loaded once from L2 like today), poll the progress word, read a landed tile with one ds_read_b128,
//                  multiply, and park / reduce like kstat (NC-way).  No VMEM instruction in a consumer after its X loads, no VGPR holds a
//                  weight in flight.  Grids: 215 blocks x 3 strips (today's) and 256 blocks x 2-3 strips (one per CU).
// What to read: (i) does one loader wave per CU stream as fast as ten VGPR-loading waves; (ii) what the launch costs beyond the stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);}}while(0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int K = 2560, KT = K / 64;            // Int8-sized tiles: 64 k per 1 KiB tile -> 40 tiles per strip

struct Args { const u32x4 *W; const u32x4 *X; float *out; int strips; int nblk; };

// ---- today: K-stationary, weights and X through VGPRs
template <int NT, bool WFIRST = false>
__global__ __launch_bounds__(640) void kstat(const Args a) {
    __shared__ f32x4 red[3 * 10 * 64 * NT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int strip0 = blockIdx.x * 3;
    u32x4 xb[8 * NT], w[12];
    if constexpr (!WFIRST) {
#pragma unroll
        for (int j = 0; j < 8 * NT; ++j) xb[j] = a.X[(size_t)(wave * 8 * NT + j) * 64 + lane];             // this wave's K slice of X from L2
        asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            w[s * 4 + j] = __builtin_nontemporal_load(a.W + ((size_t)min(strip0 + s, a.strips - 1) * KT + wave * 4 + j) * 64 + lane);
    if constexpr (WFIRST) {                                  // the HBM stream is requested first, the L2-resident operand behind it
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 8 * NT; ++j) xb[j] = a.X[(size_t)(wave * 8 * NT + j) * 64 + lane];
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f16x8 af = __builtin_bit_cast(f16x8, w[s * 4 + j]);
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, __builtin_bit_cast(f16x8, xb[(2 * j + q) * NT + t]), acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) red[((s * 10 + wave) * NT + t) * 64 + lane] = acc[t];
    }
    __syncthreads();
    if (wave < 3 && strip0 + wave < a.strips) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x4 v = red[((wave * 10) * NT + t) * 64 + lane];
            for (int w2 = 1; w2 < 10; ++w2) v += red[((wave * 10 + w2) * NT + t) * 64 + lane];
            *(f32x4 *)(a.out + (((size_t)(strip0 + wave) * NT + t) * 64 + lane) * 4) = v;
        }
    }
}

// ---- LDS-DMA loader + NC consumers.  LDS: [tiles of the block, 1 KiB each][progress word]; the park area re-uses the tile area.
__device__ __forceinline__ void dma16(const void *gsrc_lane, unsigned lds_byte) {          // lane l -> LDS byte lds_byte + 16 l, non-temporal
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc_lane), "s"(lds_byte) : "memory");
}
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int GROUP = 8, LAG = 7;                  // tiles per published group; groups in flight behind the one being waited for (7 x 8 = 56 <= 63)
template <int NC, int NT, int NL = 1>
__global__ __launch_bounds__((NC + NL) * 64) void ldsdma(const Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // strips of this block: as even as possible over the grid
    const int s0 = (int)((long)blockIdx.x * a.strips / a.nblk), s1 = (int)((long)(blockIdx.x + 1) * a.strips / a.nblk);
    const int ns = s1 - s0, ntile = ns * KT;
    volatile int *progress = (volatile int *)(smem + 3 * KT * 1024);        // per loader: groups landed so far
    if (threadIdx.x < NL) progress[threadIdx.x] = 0;
    __syncthreads();
    constexpr int TPC = (KT + NC - 1) / NC;                                    // k-tiles per consumer (last may own fewer)
    f32x4 acc[3][NT];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[s][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (wave < NL) {
        // ---------------- loader l: tiles l, l + NL, ... of the block (strip-major / k-major); tile i lands at LDS byte i * 1024
        const char *src = (const char *)(a.W + (size_t)s0 * KT * 64 + lane);
        const unsigned lds0 = (unsigned)(uintptr_t)smem;
        const int mine = (ntile - wave + NL - 1) / NL;                         // tiles of this loader
        const int ngroup = (mine + GROUP - 1) / GROUP;
        for (int g = 0; g < ngroup; ++g) {
#pragma unroll
            for (int j = 0; j < GROUP; ++j) {
                const int i = min((g * GROUP + j) * NL + wave, ntile - 1);     // (a short last group re-fetches the last tile: keeps the count exact)
                dma16(src + (size_t)i * 1024, lds0 + (unsigned)i * 1024u);
            }
            if (g >= LAG) { vm_wait<GROUP * LAG>(); if (lane == 0) progress[wave] = g - LAG + 1; }
        }
        // drain: the last LAG groups (literal counts)
#define DRAIN(n) if (ngroup >= (n) + 1) { vm_wait<GROUP * (n)>(); if (lane == 0) progress[wave] = ngroup - (n); }
        DRAIN(6) DRAIN(5) DRAIN(4) DRAIN(3) DRAIN(2) DRAIN(1) DRAIN(0)
#undef DRAIN
    } else {
        // ---------------- consumer c: k-tiles {c, c + NC, ...} of every strip; its X slice (2 k-steps per tile, NT token tiles) in registers
        const int c = wave - NL;
        u32x4 xb[TPC * 2 * NT];
#pragma unroll
        for (int j = 0; j < TPC; ++j) {
            const int kt = min(c + j * NC, KT - 1);
#pragma unroll
            for (int q = 0; q < 2 * NT; ++q) xb[j * 2 * NT + q] = a.X[(size_t)((kt * 2) * NT + q) * 64 + lane];
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if (s < ns) {
#pragma unroll
                for (int j = 0; j < TPC; ++j) {
                    const int kt = c + j * NC;
                    if (kt < KT) {
                        const int i = s * KT + kt, need = (i / NL) / GROUP + 1;
                        while (progress[i % NL] < need) __builtin_amdgcn_s_sleep(1);
                        asm volatile("" ::: "memory");
                        const f16x8 af = *(const f16x8 *)(smem + (size_t)i * 1024 + lane * 16);
#pragma unroll
                        for (int q = 0; q < 2; ++q)
#pragma unroll
                            for (int t = 0; t < NT; ++t)
                                acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, __builtin_bit_cast(f16x8, xb[(j * 2 + q) * NT + t]), acc[s][t], 0, 0, 0);
                    }
                }
            }
        }
    }
    __syncthreads();                       // every tile consumed: the tile area becomes the park area
    f32x4 *red = (f32x4 *)smem;
    if (wave >= NL) {
        const int c = wave - NL;
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int t = 0; t < NT; ++t) red[((s * NC + c) * NT + t) * 64 + lane] = acc[s][t];
    }
    __syncthreads();
    if (wave >= NL && wave < NL + 3 && wave - NL < ns) {
        const int s = wave - NL;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x4 v = red[((s * NC) * NT + t) * 64 + lane];
            for (int w2 = 1; w2 < NC; ++w2) v += red[((s * NC + w2) * NT + t) * 64 + lane];
            *(f32x4 *)(a.out + (((size_t)(s0 + s) * NT + t) * 64 + lane) * 4) = v;
        }
    }
}

template <class F>
static void bench(const char *label, F launch, hipStream_t st, const u32x4 *W, size_t wvec, const u32x4 *X, float *out, int strips) {
    const int NL = 48, NBUF = 12, REPS = 20;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int p = 0; p < NL; ++p) { Args a{W + wvec * (p % NBUF), X, out, strips, 0}; launch(a); }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    float best = 1e30f, sum = 0.f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < REPS; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best; sum += ms;
    }
    const double us = best * 1000.0 / (REPS * NL), usm = sum / 5 * 1000.0 / (REPS * NL), mb = (double)strips * KT * 1024 / 1e6;
    printf("  %-66s %6.2f us per launch (mean of 5: %6.2f)   %5.1f MB -> %4.2f TB/s\n", label, us, usm, mb, mb / us);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int strips = 645;
    const size_t wvec = (size_t)648 * KT * 64;                   // u32x4 per weight buffer (rounded up to whole blocks)
    u32x4 *W; CK(hipMalloc(&W, wvec * 16 * 12));
    {   // random bytes (a constant fill lets the chip clock higher than real data does)
        const size_t n = wvec * 16 * 12 / 8;
        uint64_t *h = (uint64_t *)malloc(n * 8), s = 0x9E3779B97F4A7C15ull;
        for (size_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = s & 0x3BFF3BFF3BFF3BFFull; }      // finite halfs below 1
        CK(hipMemcpy(W, h, n * 8, hipMemcpyHostToDevice)); free(h);
    }
    const size_t xbytes = (size_t)2 * K * 32 * 2;               // 32 rows of f16 (two token tiles)
    u32x4 *X; CK(hipMalloc(&X, xbytes)); CK(hipMemset(X, 0x3c, xbytes));
    float *out; CK(hipMalloc(&out, (size_t)648 * 64 * 16 * 2));
    const size_t lds = 3 * KT * 1024 + 64;          // tiles + progress words
#define LD(nl, nc, nt, grid) do { \
        CK(hipFuncSetAttribute((const void *)ldsdma<nc, nt, nl>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        char lb[128]; snprintf(lb, sizeof lb, "ldsdma  %d loader(s) + %2d consumers, %d token tile(s), %d blocks", nl, nc, nt, grid); \
        bench(lb, [&](const Args &a0) { Args a = a0; a.nblk = grid; hipLaunchKernelGGL((ldsdma<nc, nt, nl>), dim3(grid), dim3((nc + nl) * 64), lds, st, a); }, st, W, wvec, X, out, strips); } while (0)
    for (int rep = 0; rep < 2; ++rep) {
        printf("== pass %d: r/k/v/g/decay-sized launch (645 strips x K = 2560, Int8-sized tiles = 26.4 MB)\n", rep);
        bench("kstat<1> (today, <= 16 rows): 215 blocks x 10 waves", [&](const Args &a) { hipLaunchKernelGGL(kstat<1>, dim3((strips + 2) / 3), dim3(640), 0, st, a); }, st, W, wvec, X, out, strips);
        bench("kstat<2> (today, <= 32 rows): 215 blocks x 10 waves", [&](const Args &a) { hipLaunchKernelGGL(kstat<2>, dim3((strips + 2) / 3), dim3(640), 0, st, a); }, st, W, wvec, X, out, strips);
        bench("kstat<1>, weights requested before the operand", [&](const Args &a) { hipLaunchKernelGGL((kstat<1, true>), dim3((strips + 2) / 3), dim3(640), 0, st, a); }, st, W, wvec, X, out, strips);
        bench("kstat<2>, weights requested before the operand", [&](const Args &a) { hipLaunchKernelGGL((kstat<2, true>), dim3((strips + 2) / 3), dim3(640), 0, st, a); }, st, W, wvec, X, out, strips);
        LD(1, 5, 1, 215); LD(1, 10, 1, 215); LD(1, 10, 1, 256); LD(1, 3, 1, 256);
        LD(2, 5, 1, 215); LD(2, 10, 1, 215); LD(4, 5, 1, 215); LD(4, 8, 1, 215); LD(4, 10, 1, 215); LD(4, 10, 1, 256);
        LD(1, 10, 2, 215); LD(2, 10, 2, 215); LD(4, 5, 2, 215); LD(4, 10, 2, 215); LD(4, 10, 2, 256);
    }
    return 0;
}
