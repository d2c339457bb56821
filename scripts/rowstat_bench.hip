// dev tool (round 5): would a ROW-STATIONARY decode GEMM beat the K-stationary one?  (Next #2 (i))
//   hipcc --offload-arch=gfx950 -O3 scripts/rowstat_bench.hip -o scripts/rowstat_bench.bin && scripts/rowstat_bench.bin
//
// The engine's decode GEMM (T <= 16) is K-stationary: a block owns 3 strips of 16 output rows, its ten waves split K = 2560 into
// 256-k slices, every wave has all its weight tiles in flight at once, partial accumulators are parked in LDS and reduced after ONE
// barrier (measured: 1.5 us of compute tail + 1.0 us of barrier / reduce / epilogue behind the last byte).  Row-stationary: X (<= 80 KB
// of f16 at 16 rows) is staged once per block in LDS, each WAVE owns whole strips over the FULL K and streams their tiles through a ring,
// accumulates in registers and stores straight from the accumulators: no park, no barrier behind the stream, no reduce.  The catch is
// geometry: the r/k/v/g/decay launch has 644 strips, so row-stationary has 644 waves for the chip (2.5 per CU) where K-stationary has
// 2,150 — and what a CU pulls from HBM grows with its waves (profiles/r3_exp_stream_waves_x_loads.log).  This bench prices both with
// stand-in kernels of the real byte counts (Int8-sized tiles: 26.4 MB per launch, 645 strips), one MFMA pair per tile, as a graph of
// dependent launches over rotating weight buffers:
//   kstat      215 blocks x 10 waves, wave = (3 strips) x (256 k), 12 tiles in flight, LDS park + barrier + 10-way reduce   [today]
//   rowstat/R  215 blocks x 3 waves, wave = 1 strip x full K (40 tiles) through a ring of R loads, X from LDS, no reduce
//   rowstat2/R 215 blocks x 6 waves, two waves per strip (half K each, 20 tiles), pair reduce through LDS (one barrier)
//   rowstat5/R 129 blocks x 10 waves, two waves per strip over 5 strips (the most waves per CU this strip count allows at 10 per block)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);}}while(0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int K = 2560, KT = K / 64;            // Int8-sized tiles: 64 k per 1 KiB tile -> 40 tiles per strip
constexpr int XBYTES = 16 * K * 2;               // 16 rows of f16: 80 KiB

struct Args { const u32x4 *W; const u32x4 *X; float *out; int strips; };

// ---- today: K-stationary
__global__ __launch_bounds__(640) void kstat(const Args a) {
    __shared__ f32x4 red[3 * 10 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int strip0 = blockIdx.x * 3;
    u32x4 xb[8], w[12];
#pragma unroll
    for (int j = 0; j < 8; ++j) xb[j] = a.X[(size_t)(wave * 8 + j) * 64 + lane];                  // this wave's K slice of X: 8 KiB from L2
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            w[s * 4 + j] = __builtin_nontemporal_load(a.W + ((size_t)min(strip0 + s, a.strips - 1) * KT + wave * 4 + j) * 64 + lane);
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = acc;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f16x8 af = __builtin_bit_cast(f16x8, w[s * 4 + j]);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, __builtin_bit_cast(f16x8, xb[2 * j]), acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, __builtin_bit_cast(f16x8, xb[2 * j + 1]), acc2, 0, 0, 0);
        }
        red[(s * 10 + wave) * 64 + lane] = acc + acc2;
    }
    __syncthreads();
    if (wave < 3 && strip0 + wave < a.strips) {
        f32x4 v = red[(wave * 10) * 64 + lane];
        for (int w2 = 1; w2 < 10; ++w2) v += red[(wave * 10 + w2) * 64 + lane];
        *(f32x4 *)(a.out + ((size_t)(strip0 + wave) * 64 + lane) * 4) = v;
    }
}

// ---- row-stationary: WPS waves per strip (1 or 2), SPB strips per block; X staged in LDS once
template <int WPS, int SPB, int R>
__global__ __launch_bounds__(WPS * SPB * 64) void rowstat(const Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4 *xl = (u32x4 *)smem;                                  // X in B-fragment order: K/32 tiles of 1 KiB
    f32x4 *red = (f32x4 *)(smem + XBYTES);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nthr = WPS * SPB * 64;
    const int strip = min((int)blockIdx.x * SPB + wave / WPS, a.strips - 1), half = wave % WPS;
    constexpr int NT = KT / WPS;                                // tiles of this wave
    const u32x4 *wp = a.W + ((size_t)strip * KT + half * NT) * 64 + lane;
    u32x4 ring[R];
#pragma unroll
    for (int j = 0; j < R; ++j) if (j < NT) ring[j] = __builtin_nontemporal_load(wp + (size_t)j * 64);      // the stream starts before X is staged
    for (int i = threadIdx.x; i < XBYTES / 16; i += nthr) xl[i] = a.X[i];
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = acc;
    for (int t0 = 0; t0 < NT; t0 += R) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int t = t0 + j;
            if (t < NT) {
                const f16x8 af = __builtin_bit_cast(f16x8, ring[j]);
                const int kt = (half * NT + t) * 2;
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, __builtin_bit_cast(f16x8, xl[(size_t)kt * 64 + lane]), acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, __builtin_bit_cast(f16x8, xl[(size_t)(kt + 1) * 64 + lane]), acc2, 0, 0, 0);
                if (t + R < NT) ring[j] = __builtin_nontemporal_load(wp + (size_t)(t + R) * 64);
            }
        }
    }
    f32x4 v = acc + acc2;
    if constexpr (WPS == 2) {
        if (half == 1) red[(wave / 2) * 64 + lane] = v;
        __syncthreads();
        if (half == 0) v += red[(wave / 2) * 64 + lane];
    }
    if (half == 0 && (int)blockIdx.x * SPB + wave / WPS < a.strips) *(f32x4 *)(a.out + ((size_t)strip * 64 + lane) * 4) = v;
}

template <class F>
static void bench(const char *label, F launch, hipStream_t st, const u32x4 *W, size_t wvec, const u32x4 *X, float *out, int strips) {
    const int NL = 48, NBUF = 12, REPS = 20;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int p = 0; p < NL; ++p) { Args a{W + wvec * (p % NBUF), X, out, strips}; launch(a); }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < REPS; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / (REPS * NL), mb = (double)strips * KT * 1024 / 1e6;
    printf("  %-52s %6.2f us per launch   %5.1f MB -> %4.2f TB/s\n", label, us, mb, mb / us);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int strips = 645;
    const size_t wvec = (size_t)648 * KT * 64;                   // u32x4 per weight buffer (rounded up to whole blocks)
    u32x4 *W; CK(hipMalloc(&W, wvec * 16 * 12)); CK(hipMemset(W, 0x3c, wvec * 16 * 12));
    u32x4 *X; CK(hipMalloc(&X, XBYTES)); CK(hipMemset(X, 0x3c, XBYTES));
    float *out; CK(hipMalloc(&out, (size_t)648 * 64 * 16));
#define RS(wps, spb, r) do { \
        CK(hipFuncSetAttribute((const void *)rowstat<wps, spb, r>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        char lb[96]; snprintf(lb, sizeof lb, "rowstat  %d wave(s)/strip, %d strips/block, ring %d", wps, spb, r); \
        bench(lb, [&](const Args &a) { hipLaunchKernelGGL((rowstat<wps, spb, r>), dim3((strips + spb - 1) / spb), dim3(wps * spb * 64), XBYTES + spb * 1024, st, a); }, st, W, wvec, X, out, strips); } while (0)
    for (int rep = 0; rep < 2; ++rep) {
        printf("== pass %d: r/k/v/g/decay-sized launch (645 strips x K = 2560, Int8-sized tiles = 26.4 MB), 16 rows\n", rep);
        bench("kstat (today): 215 blocks x 10 waves", [&](const Args &a) { hipLaunchKernelGGL(kstat, dim3((strips + 2) / 3), dim3(640), 0, st, a); }, st, W, wvec, X, out, strips);
        RS(1, 3, 8); RS(1, 3, 16); RS(1, 3, 40);
        RS(2, 3, 10); RS(2, 3, 20);
        RS(2, 5, 10); RS(2, 5, 20);
        RS(1, 5, 16); RS(1, 10, 16);
    }
    return 0;
}
