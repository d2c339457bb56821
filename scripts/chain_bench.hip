// dev tool: what does ONE phase of a dependent chain cost on gfx950 when the chain runs as a persistent task-queue kernel
// (tasks dequeued in order, weights prefetched BEFORE the wait on the previous phase's completion counter, activations
// handed over with sc1 stores / sc1 loads, no fences) — against the same phases as graph-captured launches?
//   hipcc --offload-arch=gfx950 -O3 scripts/chain_bench.hip -o scripts/chain_bench.bin && scripts/chain_bench.bin
// A phase = N tasks; a task streams LOADS x 4 KiB of "weights" (read once, non-temporal), reads XR x 4 KiB of the previous
// phase's output, writes 2 KiB of output.  Every handed-over word is checked (value = phase index).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);}}while(0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t act_t;
__device__ __forceinline__ act_t act_buf(const void *p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, -1, 0x00020000); }

struct Args {
    const u32x4 *W;          // [nphase][N][LOADS][256]
    float *x[2];             // ping-pong activations, XF floats each
    unsigned *ctr;           // [nphase] completion counters, then head, then err
    unsigned *head, *err;
    int nphase, N, xf;       // xf = floats per activation buffer
    int prefetch_first;      // 1: weights before the wait (overlap), 0: after
    int sleep;
    unsigned *bar;           // persistent mode: 32-word-spaced cells: [0] gc, [1] g0, [2+x] nx, [10+x] xc, [18+x] xf, [26] nxcd
    unsigned *xcc_log;       // [grid] XCC_ID each block ran on
};

template <int LOADS, int XR, bool QUEUE>
__device__ __forceinline__ void task(const Args &a, int p, int j, float &sink) {
    const int tid = threadIdx.x;
    u32x4 w[LOADS];
    const u32x4 *wp = a.W + (((size_t)p * a.N + j) * LOADS) * 256 + tid;
    if (a.prefetch_first || !QUEUE) {
#pragma unroll
        for (int i = 0; i < LOADS; ++i) w[i] = __builtin_nontemporal_load(wp + i * 256);
    }
    if (QUEUE && p > 0) {
        if (tid == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(a.ctr + p - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)a.N) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 22)) { __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
        __syncthreads();
    }
    if (QUEUE && !a.prefetch_first) {
#pragma unroll
        for (int i = 0; i < LOADS; ++i) w[i] = __builtin_nontemporal_load(wp + i * 256);
    }
    // activations of the previous phase: XR x 16 B per thread
    const act_t bx = act_buf(a.x[(p + 1) & 1]);
    f32x4 xv[XR];
    const unsigned base = (unsigned)(((size_t)j * XR * 1024) % (size_t)a.xf);
#pragma unroll
    for (int i = 0; i < XR; ++i) {
        const unsigned off = ((base + i * 1024 + tid * 4) % (unsigned)a.xf) * 4;
        if (QUEUE) xv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(bx, off, 0, 16));
        else xv[i] = *(const f32x4 *)((const char *)a.x[(p + 1) & 1] + off);
    }
    float acc = 0.f;
    unsigned bad = 0;
#pragma unroll
    for (int i = 0; i < XR; ++i) {
        bad |= (xv[i].x != (float)p) | (xv[i].y != (float)p) | (xv[i].z != (float)p) | (xv[i].w != (float)p);
        acc += xv[i].x;
    }
#pragma unroll
    for (int i = 0; i < LOADS; ++i) acc += (float)(w[i].x ^ w[i].y ^ w[i].z ^ w[i].w) * 1e-30f;
    if (bad) atomicAdd(a.err + 1, 1u);
    sink += acc;
    // output: 2 KiB per task (tid < 128), value p + 1, covering the whole buffer once per phase (N tasks x 512 floats <= xf)
    const act_t bo = act_buf(a.x[p & 1]);
    const float v = (float)(p + 1) + acc * 0.f;
    for (unsigned o = (unsigned)j * 512 + tid * 4; o < (unsigned)a.xf; o += (unsigned)a.N * 512) {
        if (tid < 128) {
            if (QUEUE) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, (f32x4){v, v, v, v}), bo, o * 4, 0, 16);
            else *(f32x4 *)(a.x[p & 1] + o) = (f32x4){v, v, v, v};
        }
    }
    if (QUEUE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(a.ctr + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int LOADS, int XR>
__global__ __launch_bounds__(256, 4) void chain_queue(const Args a, float *sink) {
    __shared__ int s_t;
    float acc = 0.f;
    const int total = a.nphase * a.N;
    for (;;) {
        if (threadIdx.x == 0) s_t = (int)__hip_atomic_fetch_add(a.head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int t = s_t;
        __syncthreads();
        if (t >= total) break;
        task<LOADS, XR, true>(a, t / a.N, t % a.N, acc);
    }
    if (acc == 12345.f) sink[0] = acc;
}
template <int LOADS, int XR>
__global__ __launch_bounds__(256, 4) void chain_launch(const Args a, int p, float *sink) {
    float acc = 0.f;
    task<LOADS, XR, false>(a, p, blockIdx.x, acc);
    if (acc == 12345.f) sink[0] = acc;
}

// ---- persistent kernel, static task assignment (task j of every phase -> block j % grid), two-level barrier:
// arrive = one atomic on the block's own XCD's counter (the line never leaves that XCD's L2), the last block of an XCD adds
// one to the device-wide counter (8 adds per phase); release = REL 0: XCD leader polls the device counter and publishes the
// phase number in its XCD's flag, the others poll that flag with group-scope (sc0) loads; REL 1: same, polled with
// agent-scope loads; REL 2: everybody polls the device counter.  Weights of phase p+1 are requested before the wait on phase p.
#define CELL(k) ((k) * 32)
__device__ __forceinline__ unsigned ld_agent(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <int LOADS, int XR, int REL>
__global__ __launch_bounds__(256, 4) void chain_persist(const Args a, float *sink) {
    const int tid = threadIdx.x;
    __shared__ unsigned s_n, s_nx, s_lead;
    const unsigned xcd = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;      // HW_REG_XCC_ID[3:0]
    unsigned *gc = a.bar + CELL(0), *g0 = a.bar + CELL(1), *nx = a.bar + CELL(2 + xcd), *xc = a.bar + CELL(10 + xcd),
             *xf = a.bar + CELL(18 + xcd);
    if (tid == 0) {
        a.xcc_log[blockIdx.x] = xcd;
        __hip_atomic_fetch_add(nx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(g0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (ld_agent(g0) < gridDim.x) { __builtin_amdgcn_s_sleep(4); if (++spins > (1u << 18)) { a.err[0] = 1; break; } }
        s_n = ld_agent(nx);
        unsigned k = 0;
        for (int x = 0; x < 8; ++x) k += ld_agent(a.bar + CELL(2 + x)) != 0;
        s_nx = k;
    }
    __syncthreads();
    const unsigned nblk = s_n, nxcd = s_nx;
    const act_t bxf = act_buf(xf);
    float acc = 0.f;
    bool lead = false, dead = false;
    for (int p = 0; p < a.nphase; ++p) {
        for (int j = blockIdx.x; j < a.N; j += gridDim.x) {
            u32x4 w[LOADS];
            const u32x4 *wp = a.W + (((size_t)p * a.N + j) * LOADS) * 256 + tid;
#pragma unroll
            for (int i = 0; i < LOADS; ++i) w[i] = __builtin_nontemporal_load(wp + i * 256);
            if (p > 0 && j == (int)blockIdx.x) {                                  // wait for phase p-1, once per phase
                if (tid == 0 && !dead) {
                    unsigned spins = 0;
                    if (REL == 2 || lead) {
                        while (ld_agent(gc) < nxcd * (unsigned)p) { if (a.sleep) __builtin_amdgcn_s_sleep(1); if (++spins > (1u << 18)) { a.err[0] = 1; dead = true; break; } }
                        if (REL != 2) __hip_atomic_store(xf, (unsigned)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else if (REL == 0) {
                        while (__builtin_amdgcn_raw_buffer_load_b32(bxf, 0, 0, 1) < (unsigned)p) { asm volatile("" ::: "memory"); if (a.sleep) __builtin_amdgcn_s_sleep(1); if (++spins > (1u << 18)) { a.err[0] = 1; dead = true; break; } }
                    } else {
                        while (ld_agent(xf) < (unsigned)p) { if (a.sleep) __builtin_amdgcn_s_sleep(1); if (++spins > (1u << 18)) { a.err[0] = 1; dead = true; break; } }
                    }
                }
                __syncthreads();
            }
            const act_t bx = act_buf(a.x[(p + 1) & 1]);
            f32x4 xv[XR];
            const unsigned base = (unsigned)(((size_t)j * XR * 1024) % (size_t)a.xf);
#pragma unroll
            for (int i = 0; i < XR; ++i) {
                const unsigned off = ((base + i * 1024 + tid * 4) % (unsigned)a.xf) * 4;
                xv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(bx, off, 0, 16));
            }
            unsigned bad = 0;
#pragma unroll
            for (int i = 0; i < XR; ++i) {
                bad |= (xv[i].x != (float)p) | (xv[i].y != (float)p) | (xv[i].z != (float)p) | (xv[i].w != (float)p);
                acc += xv[i].x;
            }
#pragma unroll
            for (int i = 0; i < LOADS; ++i) acc += (float)(w[i].x ^ w[i].y ^ w[i].z ^ w[i].w) * 1e-30f;
            if (bad) atomicAdd(a.err + 1, 1u);
            const act_t bo = act_buf(a.x[p & 1]);
            const float v = (float)(p + 1) + acc * 0.f;
            for (unsigned o = (unsigned)j * 512 + tid * 4; o < (unsigned)a.xf; o += (unsigned)a.N * 512)
                if (tid < 128) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, (f32x4){v, v, v, v}), bo, o * 4, 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(xc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool last = old + 1 == nblk * (unsigned)(p + 1);
            if (last) __hip_atomic_fetch_add(gc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_lead = last;
        }
        __syncthreads();
        lead = s_lead != 0;
    }
    if (acc == 12345.f) sink[0] = acc;
}

template <int LOADS, int XR>
void run(int nphase, int N, int grid, int mode, hipStream_t st) {
    Args a{};
    const size_t wbytes = (size_t)nphase * N * LOADS * 4096;
    u32x4 *W; CK(hipMalloc(&W, wbytes)); CK(hipMemset(W, 1, wbytes));
    a.W = W; a.nphase = nphase; a.N = N; a.xf = N * 512;
    for (int i = 0; i < 2; ++i) { CK(hipMalloc(&a.x[i], (size_t)a.xf * 4)); }
    std::vector<float> zeros(a.xf, 0.f);
    unsigned *ctr; CK(hipMalloc(&ctr, (nphase + 8) * 4));
    a.ctr = ctr; a.head = ctr + nphase; a.err = ctr + nphase + 1;
    float *sink; CK(hipMalloc(&sink, 4));
    CK(hipMalloc(&a.bar, 32 * 32 * 4)); CK(hipMalloc(&a.xcc_log, grid * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipGraphExec_t exec = nullptr;
    if (mode == 2) {
        hipGraph_t g;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int p = 0; p < nphase; ++p) hipLaunchKernelGGL((chain_launch<LOADS, XR>), dim3(N), dim3(256), 0, st, a, p, sink);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
    }
    float best = 1e30f; unsigned herr[2] = {0, 0};
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemcpy(a.x[1], zeros.data(), (size_t)a.xf * 4, hipMemcpyHostToDevice));   // phase 0 reads x[1] == 0
        CK(hipMemsetAsync(ctr, 0, (nphase + 8) * 4, st));
        CK(hipMemsetAsync(a.bar, 0, 32 * 32 * 4, st));
        a.prefetch_first = mode == 0;
        a.sleep = mode >= 6;
        CK(hipEventRecord(e0, st));
        if (mode == 2) CK(hipGraphLaunch(exec, st));
        else if (mode == 3 || mode == 6) hipLaunchKernelGGL((chain_persist<LOADS, XR, 0>), dim3(grid), dim3(256), 0, st, a, sink);
        else if (mode == 4 || mode == 7) hipLaunchKernelGGL((chain_persist<LOADS, XR, 1>), dim3(grid), dim3(256), 0, st, a, sink);
        else if (mode == 5 || mode == 8) hipLaunchKernelGGL((chain_persist<LOADS, XR, 2>), dim3(grid), dim3(256), 0, st, a, sink);
        else hipLaunchKernelGGL((chain_queue<LOADS, XR>), dim3(grid), dim3(256), 0, st, a, sink);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
        unsigned h[2]; CK(hipMemcpy(h, a.err, 8, hipMemcpyDeviceToHost)); herr[0] |= h[0]; herr[1] += h[1];
    }
    const char *names[9] = {"queue, weights BEFORE wait", "queue, weights after wait ", "graph of launches          ",
                            "persist, xcd flag sc0 poll ", "persist, xcd flag sc1 poll ", "persist, all poll device  ",
                            "persist+sleep, xcd flag sc0", "persist+sleep, xcd flag sc1", "persist+sleep, all poll dev"};
    if (mode >= 3) {
        std::vector<unsigned> xl(grid); CK(hipMemcpy(xl.data(), a.xcc_log, grid * 4, hipMemcpyDeviceToHost));
        int cnt[8] = {0}, rr = 0;
        for (int b = 0; b < grid; ++b) { cnt[xl[b] & 7]++; rr += (int)(xl[b] & 7) == b % 8; }
        static bool once = false;
        if (!once) { once = true; printf("XCC_ID histogram:"); for (int x = 0; x < 8; ++x) printf(" %d", cnt[x]); printf("; blocks with xcc == b %% 8: %d of %d\n", rr, grid); }
    }
    printf("%s LOADS=%2d XR=%2d N=%4d grid=%4d: %6.2f us/phase, %5.2f TB/s, timeout=%u stale=%u\n", names[mode], LOADS, XR, N, grid,
           best * 1e3 / nphase, (double)wbytes / (best * 1e-3) / 1e12, herr[0], herr[1]);
    if (exec) CK(hipGraphExecDestroy(exec));
    CK(hipFree(a.bar)); CK(hipFree(a.xcc_log));
    CK(hipFree(W)); CK(hipFree(a.x[0])); CK(hipFree(a.x[1])); CK(hipFree(ctr)); CK(hipFree(sink));
}

int main(int argc, char **argv) {
    hipStream_t st; CK(hipStreamCreate(&st));
    const int nphase = 96;
    if (argc > 1 && argv[1][0] == 'p') {     // persistent-kernel study
        for (int mode : {2, 3, 4, 5, 6, 7, 8}) {
            run<10, 8>(nphase, 512, 512, mode, st);      // 20 MB per phase, 2 blocks per CU
            run<5, 8>(nphase, 1024, 1024, mode, st);     // 20 MB per phase, 4 blocks per CU
            run<10, 8>(nphase, 256, 256, mode, st);      // 10 MB per phase, 1 block per CU
            run<1, 1>(nphase, 256, 256, mode, st);       // barrier cost, 1 block per CU
            run<1, 1>(nphase, 1024, 1024, mode, st);     // barrier cost, 4 blocks per CU
            run<2, 8>(nphase, 512, 512, mode, st);       // 4 MB per phase
            run<16, 8>(nphase, 1024, 1024, mode, st);    // 64 MB per phase
        }
        return 0;
    }
    // ~20 MB per phase (a V6-3B Int8 layer GEMM): N tasks x LOADS x 4 KiB
    for (int mode = 0; mode < 3; ++mode) {
        run<10, 8>(nphase, 512, 1024, mode, st);
        run<10, 8>(nphase, 512, 768, mode, st);
        run<10, 8>(nphase, 512, 512, mode, st);
        run<5, 8>(nphase, 1024, 1024, mode, st);
        run<10, 2>(nphase, 512, 1024, mode, st);
        run<2, 8>(nphase, 512, 1024, mode, st);      // 4 MB per phase: latency floor
        run<2, 2>(nphase, 128, 1024, mode, st);      // 1 MB per phase, few tasks
        run<16, 8>(nphase, 1024, 1024, mode, st);    // 64 MB per phase: bandwidth regime
    }
    return 0;
}
