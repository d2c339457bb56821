// coldstart_bench.hip — dev tool (GPU box): what a decode-shaped kernel pays BEFORE its first useful load, per launch of a dependent chain.
//   (1) straight-line code: every launch starts with cold instruction caches (the dispatch's acquire invalidates them); a wave that executes
//       N KB of unrolled code fetches N KB through the shared instruction cache — cost per launch by code size, same kernel repeated vs two
//       kernels alternating;
//   (2) dependent scalar loads from the kernarg segment: a prologue that walks a by-value argument struct (`L.p[i].block_begin` for i < nprob,
//       then the fields of `L.p[pi]`) is a chain of scalar-cache misses; cost per launch by number of DEPENDENT round trips, and the same
//       fields fetched in one independent batch;
//   (3) block shape: 256 x 640-thread blocks with 0 / 64 / 150 KiB of dynamic LDS against 1024 x 256-thread blocks.
// Each case: a graph of 200 dependent launches, 256 workgroups, timed with events; the number printed is microseconds per launch.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/coldstart_bench.bin scripts/coldstart_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// ---- (1) straight-line code of ~N KB: unrolled dependent integer ops the compiler cannot fold (each v_mad/v_xor is 8 bytes)
template <int KB, int SALT>
__global__ __launch_bounds__(640) void code_kernel(unsigned *out, unsigned seed) {
    unsigned a = threadIdx.x + seed, b = blockIdx.x * 2654435761u + SALT;
#pragma unroll
    for (int i = 0; i < KB * 64; ++i) {           // two 8-byte VALU instructions per iteration = 16 B -> 64 iterations per KB
        a = a * 1664525u + b;
        b ^= a >> 7;
    }
    if (a == 0x12345678u) out[blockIdx.x] = b;   // never true in practice: keeps the chain alive without a store on the timed path
}

// ---- (2) kernarg walks
struct Prob { const void *W, *S, *x; int a[20]; int block_begin; int pad[15]; };        // 168 B like GemmProb: one field of interest per entry
struct Launch { Prob p[8]; int nprob, T, total; int pad[13]; };
__global__ __launch_bounds__(640) void walk_kernel(const Launch L, unsigned *out) {      // the product's pattern: scan block_begin, then P's fields
    if ((int)blockIdx.x >= L.total) return;
    int pi = 0;
    for (int i = 1; i < L.nprob; ++i)
        if ((int)blockIdx.x >= L.p[i].block_begin) pi = i;
    const Prob &P = L.p[pi];
    const unsigned v = (unsigned)(size_t)P.W + (unsigned)P.a[3] + (unsigned)P.a[19] + (unsigned)(size_t)P.x;
    if (v == 0x12345678u) out[blockIdx.x] = v;
}
struct Flat { int block_begin[8]; int nprob, T, total, pad[5]; Prob p[8]; };             // header first: one 64-byte line
__global__ __launch_bounds__(640) void flat_kernel(const Flat L, unsigned *out) {
    if ((int)blockIdx.x >= L.total) return;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i)
        if (i < L.nprob && (int)blockIdx.x >= L.block_begin[i]) pi = i;
    const Prob &P = L.p[pi];
    const unsigned v = (unsigned)(size_t)P.W + (unsigned)P.a[3] + (unsigned)P.a[19] + (unsigned)(size_t)P.x;
    if (v == 0x12345678u) out[blockIdx.x] = v;
}
__global__ __launch_bounds__(640) void gridy_kernel(const Flat L, unsigned *out) {        // problem index = blockIdx.y: no selection loads at all
    const Prob &P = L.p[blockIdx.y];
    if ((int)blockIdx.x >= P.block_begin) return;                                       // here: blocks of this problem
    const unsigned v = (unsigned)(size_t)P.W + (unsigned)P.a[3] + (unsigned)P.a[19] + (unsigned)(size_t)P.x;
    if (v == 0x12345678u) out[blockIdx.x] = v;
}
__global__ void empty_kernel(unsigned *out) { if (out == nullptr) out[0] = 0; }

template <class F>
static int time_chain(const char *label, hipStream_t st, int n, F &&launch) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n; ++i) launch(i);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    printf("%-62s %6.2f us per launch\n", label, best * 1000.0f / n);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return 0;
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    unsigned *out; CK(hipMalloc(&out, 1 << 20));
    const int N = 200;
    if (time_chain("empty kernel, 1 x 64", st, N, [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st, out); })) return 1;
    if (time_chain("empty kernel, 256 x 640", st, N, [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(640), 0, st, out); })) return 1;
    if (time_chain("empty kernel, 256 x 640, 64 KiB LDS", st, N, [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(640), 64 << 10, st, out); })) return 1;
    if (time_chain("empty kernel, 1024 x 256", st, N, [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(1024), dim3(256), 0, st, out); })) return 1;
#define CODE(KB) \
    if (time_chain("straight-line code " #KB " KB, same kernel, 256 x 640", st, N, [&](int i) { hipLaunchKernelGGL((code_kernel<KB, 0>), dim3(256), dim3(640), 0, st, out, (unsigned)i); })) return 1; \
    if (time_chain("straight-line code " #KB " KB, two kernels alternating", st, N, [&](int i) { if (i & 1) hipLaunchKernelGGL((code_kernel<KB, 1>), dim3(256), dim3(640), 0, st, out, (unsigned)i); else hipLaunchKernelGGL((code_kernel<KB, 0>), dim3(256), dim3(640), 0, st, out, (unsigned)i); })) return 1; \
    if (time_chain("straight-line code " #KB " KB, same kernel, 256 x 64 (one wave)", st, N, [&](int i) { hipLaunchKernelGGL((code_kernel<KB, 0>), dim3(256), dim3(64), 0, st, out, (unsigned)i); })) return 1;
    CODE(1) CODE(4) CODE(8) CODE(16) CODE(32)
    Launch L{}; Flat F{};
    for (int i = 0; i < 8; ++i) { L.p[i].block_begin = i * 32; L.p[i].W = out; L.p[i].x = out; F.p[i] = L.p[i]; F.block_begin[i] = i * 32; F.p[i].block_begin = 32; }
    L.total = F.total = 256;
    for (int np : {1, 2, 5, 8}) {
        L.nprob = F.nprob = np;
        char lab[128];
        snprintf(lab, sizeof lab, "kernarg walk (scan block_begin of %d problems, then P)", np);
        if (time_chain(lab, st, N, [&](int) { hipLaunchKernelGGL(walk_kernel, dim3(256), dim3(640), 0, st, L, out); })) return 1;
        snprintf(lab, sizeof lab, "kernarg flat header (%d problems: one line, then P)", np);
        if (time_chain(lab, st, N, [&](int) { hipLaunchKernelGGL(flat_kernel, dim3(256), dim3(640), 0, st, F, out); })) return 1;
    }
    if (time_chain("kernarg: problem = blockIdx.y (P only), grid 32 x 8", st, N, [&](int) { hipLaunchKernelGGL(gridy_kernel, dim3(32, 8), dim3(640), 0, st, F, out); })) return 1;
    return 0;
}
