// dev tool: can the Infinity Cache (256 MiB, memory side) serve as a prefetch buffer for the weight stream of a
// dependent launch chain on gfx950?
//   hipcc --offload-arch=gfx950 -O3 scripts/mall_bench.hip -o scripts/mall_bench.bin && scripts/mall_bench.bin
// scripts/stream_bench.hip showed a 27 MB launch streaming at ~7 us from HBM and 4.0-5.3 us when its source was already
// cache resident.  Here:
//  (A) sequential graph  P(buf p) -> C(buf p):  a prefetch kernel touches buffer p (variants: 16 B per lane; 4 B per lane at a
//      64 B or 128 B stride, i.e. one word per line — the line reaches L2 / Infinity Cache, only 4 B reach the CU; default / nt
//      policy), then the consumer C streams it exactly like stream_bench's k3 kernel.  Reported: time of the pair minus the
//      time of P alone = what C costs on a prefetched buffer, against C cold.
//  (B) persistent prefetcher: ONE long-running kernel (PB blocks x PW waves, launched beside the chain) walks the buffers in
//      chain order, never more than LEAD phases ahead of the consumer (progress word bumped by block 0 of every C launch),
//      while the chain C(0) -> C(1) -> ... runs as graph launches.  Reported: us per phase of the chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);}}while(0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int NBUF = 24;
constexpr size_t BUF_BYTES = 27525120;            // 256 blocks x 10 waves x 10.5 KiB -> 26880 tiles of 1 KiB

struct CArgs { const u32x4 *W; float *out; unsigned *progress; int tiles_per_wave; };

// consumer: as stream_bench's stream_k<R=4>, 256 blocks x 10 waves, LDS reduce + 256 B store
__global__ __launch_bounds__(640) void consume_k(const CArgs a, int p, int signal) {
    __shared__ float red[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (signal && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(a.progress, (unsigned)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u32x4 *w = a.W + (size_t)(p % NBUF) * (BUF_BYTES / 16) + ((size_t)(blockIdx.x * nw + wave) * a.tiles_per_wave) * 64 + lane;
    constexpr int R = 4;
    unsigned acc = 0;
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
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (wave == 0) {
        float s = 0.f;
        for (int i = 0; i < nw; ++i) s += red[i];
        a.out[(size_t)blockIdx.x * 64 + lane] = s;
    }
}

// touch `bytes` at `base`: MODE 0: 16 B per lane (everything crosses into the CU); MODE 1: 4 B per lane, 64 B stride;
// MODE 2: 4 B per lane, 128 B stride.  NT: non-temporal policy.  Each wave keeps up to 16 loads in flight.
template <int MODE, bool NT>
__device__ __forceinline__ unsigned touch(const char *base, size_t bytes, int gw, int ngw, int lane) {
    constexpr size_t STEP = MODE == 0 ? 1024 : (MODE == 1 ? 4096 : 8192);     // bytes covered per wave instruction
    constexpr int LSTRIDE = MODE == 0 ? 16 : (MODE == 1 ? 64 : 128);
    unsigned acc = 0;
    const size_t nstep = bytes / STEP;
    for (size_t s = gw; s < nstep; s += (size_t)ngw * 16) {
        unsigned v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const size_t ss = s + (size_t)j * ngw;
            if (ss < nstep) {
                const char *p = base + ss * STEP + (size_t)lane * LSTRIDE;
                if (MODE == 0) {
                    const u32x4 q = NT ? __builtin_nontemporal_load((const u32x4 *)p) : *(const u32x4 *)p;
                    v[j] = q.x ^ q.w;
                } else {
                    v[j] = NT ? __builtin_nontemporal_load((const unsigned *)p) : *(const unsigned *)p;
                }
            } else v[j] = 0;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) acc ^= v[j];
    }
    return acc;
}

template <int MODE, bool NT>
__global__ __launch_bounds__(256) void prefetch_k(const char *W, int p, unsigned *sink) {
    const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const unsigned acc = touch<MODE, NT>(W + (size_t)(p % NBUF) * BUF_BYTES, BUF_BYTES, blockIdx.x * nw + (threadIdx.x >> 6), gridDim.x * nw, lane);
    if (acc == 0x1234567u) sink[0] = acc;
}

// persistent prefetcher: phases 1 .. nphase-1 (phase 0 is cold by construction), throttled by the consumer's progress word
template <int MODE, bool NT>
__global__ __launch_bounds__(256) void persist_k(const char *W, int nphase, int lead, unsigned *progress, unsigned *sink, unsigned *lag) {
    const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    unsigned acc = 0, waited = 0;
    for (int q = 1; q < nphase; ++q) {
        // do not run more than `lead` buffers ahead of the phase the chain is executing
        if (threadIdx.x == 0) {
            unsigned spins = 0;
            while ((int)__hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + lead < q + 1) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1u << 20)) break;
            }
            waited += spins;
        }
        __syncthreads();
        acc ^= touch<MODE, NT>(W + (size_t)(q % NBUF) * BUF_BYTES, BUF_BYTES, blockIdx.x * nw + (threadIdx.x >> 6), gridDim.x * nw, lane);
    }
    if (acc == 0x1234567u) sink[0] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) lag[0] = waited;
}

int main() {
    hipStream_t st, st2; CK(hipStreamCreate(&st)); CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
    char *W; CK(hipMalloc(&W, BUF_BYTES * NBUF)); CK(hipMemset(W, 1, BUF_BYTES * NBUF));
    float *out; CK(hipMalloc(&out, 256 * 256 + 256));
    unsigned *misc; CK(hipMalloc(&misc, 4096)); CK(hipMemset(misc, 0, 4096));
    unsigned *progress = misc, *sink = misc + 64, *lag = misc + 128;
    CArgs ca{(const u32x4 *)W, out, progress, (int)(BUF_BYTES / 1024 / (256 * 10))};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nphase = 48;

    auto time_graph = [&](auto record) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        record();
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        return best * 1e3 / nphase;
    };
#define PF(mode, nt, grid, p) hipLaunchKernelGGL((prefetch_k<mode, nt>), dim3(grid), dim3(256), 0, st, (const char *)W, p, sink)
    const double c_cold = time_graph([&] { for (int p = 0; p < nphase; ++p) hipLaunchKernelGGL(consume_k, dim3(256), dim3(640), 0, st, ca, p, 0); });
    printf("C cold (27.5 MB, 256 x 10 waves):                      %6.2f us/phase\n", c_cold);
    // (A) sequential P -> C
    struct V { int mode; bool nt; int grid; const char *name; };
    const V vs[] = {{0, false, 256, "16B/lane default, 256 blocks"}, {0, true, 256, "16B/lane nt,      256 blocks"},
                    {1, false, 256, " 4B/64B  default, 256 blocks"}, {1, true, 256, " 4B/64B  nt,      256 blocks"},
                    {2, false, 256, " 4B/128B default, 256 blocks"}, {2, true, 256, " 4B/128B nt,      256 blocks"},
                    {1, false, 64, " 4B/64B  default,  64 blocks"},  {2, false, 64, " 4B/128B default,  64 blocks"}};
    for (const V &v : vs) {
        auto launchP = [&](int p) {
            if (v.mode == 0) { if (v.nt) PF(0, true, v.grid, p); else PF(0, false, v.grid, p); }
            else if (v.mode == 1) { if (v.nt) PF(1, true, v.grid, p); else PF(1, false, v.grid, p); }
            else { if (v.nt) PF(2, true, v.grid, p); else PF(2, false, v.grid, p); }
        };
        const double p_only = time_graph([&] { for (int p = 0; p < nphase; ++p) launchP(p); });
        const double pair = time_graph([&] { for (int p = 0; p < nphase; ++p) { launchP(p); hipLaunchKernelGGL(consume_k, dim3(256), dim3(640), 0, st, ca, p, 0); } });
        // prefetch two buffers ahead: C(p) runs after P(p+2) -> does the line survive 55 MB of other traffic?
        const double pair2 = time_graph([&] { launchP(0); launchP(1); for (int p = 0; p < nphase; ++p) { launchP(p + 2); hipLaunchKernelGGL(consume_k, dim3(256), dim3(640), 0, st, ca, p, 0); } });
        printf("(A) P = %s:  P alone %6.2f   P+C %6.2f   => C after P %6.2f   (2 ahead: %6.2f)\n", v.name, p_only, pair, pair - p_only, pair2 - p_only);
        fflush(stdout);
    }
    // (B) persistent prefetcher beside the chain
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int p = 0; p < nphase; ++p) hipLaunchKernelGGL(consume_k, dim3(256), dim3(640), 0, st, ca, p, 1);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    struct B { int mode; bool nt; int pb, pw, lead; };
    const B bs[] = {{1, false, 64, 4, 2}, {1, false, 128, 4, 2}, {1, false, 256, 1, 2}, {1, false, 256, 2, 2}, {1, false, 256, 4, 2}, {1, false, 256, 4, 4},
                    {2, false, 256, 2, 2}, {2, false, 64, 4, 2}, {0, false, 256, 2, 2}, {0, true, 256, 2, 2}, {1, true, 256, 2, 2}, {1, false, 32, 4, 2},
                    {1, false, 256, 2, 1}, {1, false, 256, 2, 6}};
    for (const B &b : bs) {
        float best = 1e30f; unsigned hl = 0;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipMemsetAsync(progress, 0, 4, st)); CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
#define PK(mode, nt) hipLaunchKernelGGL((persist_k<mode, nt>), dim3(b.pb), dim3(b.pw * 64), 0, st2, (const char *)W, nphase, b.lead, progress, sink, lag)
            if (b.mode == 0) { if (b.nt) PK(0, true); else PK(0, false); }
            else if (b.mode == 1) { if (b.nt) PK(1, true); else PK(1, false); }
            else { if (b.nt) PK(2, true); else PK(2, false); }
            CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            CK(hipStreamSynchronize(st2));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
            CK(hipMemcpy(&hl, lag, 4, hipMemcpyDeviceToHost));
        }
        printf("(B) persistent prefetcher mode %d %s  %3d blocks x %d waves, lead %d:  chain %6.2f us/phase  (cold %5.2f)  prefetcher spins %u\n",
               b.mode, b.nt ? "nt     " : "default", b.pb, b.pw, b.lead, best * 1e3 / nphase, c_cold, hl);
        fflush(stdout);
    }
    return 0;
}
