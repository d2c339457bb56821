// dev tool: what one launch of a dependent chain can ingest on gfx950, by how the bytes are spread over workgroups.
//   hipcc --offload-arch=gfx950 -O3 scripts/stream_bench.hip -o scripts/stream_bench.bin && scripts/stream_bench.bin
// A phase = one graph-captured launch that streams `total` bytes of "weights" (read once, non-temporal, 1 KiB per wave
// instruction, like the tiles of gemm_kernel) split evenly over G workgroups of W waves, each wave keeping R loads in flight;
// phases rotate through a buffer larger than the Infinity Cache.  Optional: every workgroup also re-reads the same `xkb` KiB
// (an L2-resident activation operand) first; a block-wide LDS reduce + 2 KiB store at the end (the GEMM's tail).
// Questions: (1) is ~25 GB/s per CU a per-CU cap or the chip's HBM rate / 256?  (grid < 256)  (2) what does imbalance cost
// (215 vs 256 workgroups)?  (3) many small workgroups vs one big one per CU?  (4) price of the shared operand re-read and tail.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);}}while(0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Args {
    const u32x4 *W;      // phase p: W + p * phase_vec
    const u32x4 *X;      // shared operand (L2 resident), xvec u32x4 per block read
    float *out;
    size_t phase_vec;    // u32x4 per phase
    int tiles_per_wave;  // 1 KiB tiles each wave streams
    int xtiles;          // 1 KiB tiles of X each WAVE reads first (0 = none)
    int tail;            // 1: LDS reduce + store
};

template <int R>
__global__ __launch_bounds__(1024) void stream_k(const Args a, int p) {
    __shared__ float red[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const u32x4 *w = a.W + (size_t)p * a.phase_vec + ((size_t)(blockIdx.x * nw + wave) * a.tiles_per_wave) * 64 + lane;
    unsigned acc = 0;
    // shared operand first (as gemm_kernel does): L2 hits
    for (int i = 0; i < a.xtiles; i += 8) {
        u32x4 x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) if (i + j < a.xtiles) x[j] = a.X[(size_t)((wave * a.xtiles + i + j) * 64 + lane)];
#pragma unroll
        for (int j = 0; j < 8; ++j) if (i + j < a.xtiles) acc ^= x[j].x ^ x[j].w;
    }
    u32x4 r[R];
    const int n = a.tiles_per_wave;
#pragma unroll
    for (int j = 0; j < R; ++j) if (j < n) r[j] = __builtin_nontemporal_load(w + (size_t)j * 64);
    for (int i = 0; i < n; i += R) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if (i + j < n) {
                acc ^= r[j].x ^ r[j].y ^ r[j].z ^ r[j].w;
                if (i + j + R < n) r[j] = __builtin_nontemporal_load(w + (size_t)(i + j + R) * 64);
            }
        }
    }
    float v = (float)(acc & 1);
    if (a.tail) {
        for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wave] = v;
        __syncthreads();
        if (wave == 0) {
            float s = 0.f;
            for (int i = 0; i < nw; ++i) s += red[i];
            a.out[(size_t)blockIdx.x * 64 + lane] = s;                 // 256 B per block
        }
    } else if (acc == 0x12345u) a.out[0] = v;
}

static void run(const char *label, size_t total, int G, int waves, int R, int xkb, int tail, bool hot, hipStream_t st) {
    const int nphase = 48;
    // tiles per wave (1 KiB each), rounded up
    const size_t tiles = total / 1024;
    const int tpw = (int)((tiles + (size_t)G * waves - 1) / ((size_t)G * waves));
    const size_t phase_vec = (size_t)G * waves * tpw * 64;
    const int nbuf = hot ? 1 : 24;
    u32x4 *W; CK(hipMalloc(&W, phase_vec * 16 * nbuf)); CK(hipMemset(W, 1, phase_vec * 16 * nbuf));
    u32x4 *X; CK(hipMalloc(&X, 1 << 20)); CK(hipMemset(X, 2, 1 << 20));
    float *out; CK(hipMalloc(&out, (size_t)G * 256 + 256));
    Args a{W, X, out, phase_vec, tpw, xkb / waves, tail};
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int p = 0; p < nphase; ++p) {
#define L(r) hipLaunchKernelGGL((stream_k<r>), dim3(G), dim3(waves * 64), 0, st, a, p % nbuf)
        if (R == 4) L(4); else if (R == 8) L(8); else if (R == 16) L(16); else L(2);
#undef L
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    const double us = best * 1e3 / nphase, bytes = (double)phase_vec * 16;
    printf("%-10s total %5.1f MB  G=%4d waves=%2d R=%2d xKB=%3d tail=%d %s: %6.2f us/phase  %5.2f TB/s  %5.1f GB/s per block  (%.1f KB/block)\n",
           label, bytes / 1e6, G, waves, R, xkb, tail, hot ? "hot " : "cold", us, bytes / us / 1e6, bytes / G / us / 1e3, bytes / G / 1e3);
    fflush(stdout);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipFree(W)); CK(hipFree(X)); CK(hipFree(out));
}

int main(int argc, char **argv) {
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t MB = 1000000;
    if (argc > 1 && argv[1][0] == 'w') {      // waves x loads-in-flight matrix at one workgroup per CU
        for (int w : {2, 4, 5, 8, 10, 16})
            for (int r : {2, 4, 8, 16}) run("wxr", 27400 * 1000, 256, w, r, 0, 1, false, st);
        for (int w : {4, 8}) for (int r : {4, 8}) run("wxr-512", 27400 * 1000, 512, w, r, 0, 1, false, st);
        return 0;
    }
    // (0) floor: near-empty launches
    run("empty", 256 * 1024, 256, 4, 4, 0, 0, false, st);
    // (1) per-CU cap: fixed 100 KB per block, fewer and fewer blocks
    for (int G : {32, 64, 128, 192, 256}) run("percu", (size_t)G * 100 * 1024, G, 8, 8, 0, 0, false, st);
    for (int G : {32, 64, 128, 256}) run("percu-w16", (size_t)G * 100 * 1024, G, 16, 8, 0, 0, false, st);
    // (2) 27.4 MB (V6-3B Int8 r/k/v/g/D1) by grid
    for (int G : {215, 256, 322, 512, 644, 1024, 2048}) run("k3-w5", 27400 * 1000, G, 5, 8, 0, 1, false, st);
    for (int G : {215, 256, 512, 1024}) run("k3-w10", 27400 * 1000, G, 10, 4, 0, 1, false, st);
    for (int G : {256, 512, 1024, 2048}) run("k3-w4", 27400 * 1000, G, 4, 8, 0, 1, false, st);
    for (int G : {256, 512}) run("k3-w8r16", 27400 * 1000, G, 8, 16, 0, 1, false, st);
    for (int G : {256, 512}) run("k3-w16", 27400 * 1000, G, 16, 4, 0, 1, false, st);
    // (3) with the shared operand re-read (T = 32: 160 KiB per block) and without the tail
    for (int x : {0, 16, 80, 160}) run("k3+x", 27400 * 1000, 256, 10, 4, x, 1, false, st);
    for (int x : {0, 160}) run("k3+x-215", 27400 * 1000, 215, 10, 4, x, 1, false, st);
    run("k3-notail", 27400 * 1000, 256, 10, 4, 0, 0, false, st);
    // (4) Infinity-Cache resident source
    for (int G : {128, 256, 512}) run("k3-hot", 27400 * 1000, G, 8, 8, 0, 1, true, st);
    // (5) small launches: Wo (6.8 MB), row-kernel sized (2 MB)
    for (int G : {160, 256, 320, 640, 800}) run("wo", 6800 * 1000, G, 4, 8, 0, 1, false, st);
    for (int G : {32, 64, 256}) run("rows", 2 * MB, G, 16, 4, 0, 1, true, st);
    return 0;
}
