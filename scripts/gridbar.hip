// dev tool: cost and correctness of a software grid barrier on gfx950 (is a persistent per-step kernel viable?)
//   hipcc --offload-arch=gfx950 -O3 scripts/gridbar.hip -o scripts/gridbar.bin && scripts/gridbar.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

__device__ __forceinline__ bool grid_barrier(unsigned *ctr, unsigned target) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > 4000000) { ok = false; break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    return ok;
}

// MODE 0: barrier only.  MODE 1: every block writes a 1 KiB row before the barrier and reads another block's row after it.
template <int MODE>
__global__ __launch_bounds__(256) void bar_test(unsigned *ctr, float *buf, int iters, long *out, int *err) {
    const int G = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
    long t0 = wall_clock64();
    float acc = 0.f;
    for (int k = 0; k < iters; ++k) {
        if (MODE == 1) buf[((size_t)(k & 1) * G + b) * 256 + tid] = (float)(b + k);
        if (!grid_barrier(ctr, (unsigned)(k + 1) * G)) { if (tid == 0) atomicAdd(err, 1000000); break; }
        if (MODE == 1) {
            const int src = (b + 1 + k * 37) % G;
            const float v = buf[((size_t)(k & 1) * G + src) * 256 + tid];
            if (v != (float)(src + k)) { if (tid == 0) atomicAdd(err, 1); }
            acc += v;
        }
    }
    long t1 = wall_clock64();
    if (tid == 0 && b == 0) { out[0] = t1 - t0; out[1] = (long)acc; }
}

int main() {
    unsigned *ctr; float *buf; long *out; int *err;
    CK(hipMalloc(&ctr, 64)); CK(hipMalloc(&buf, (size_t)2 * 2048 * 256 * 4)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&err, 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    const int iters = 2000;
    for (int mode = 0; mode < 2; ++mode)
        for (int G : {64, 256, 512, 1024}) {
            CK(hipMemset(ctr, 0, 64)); CK(hipMemset(err, 0, 4));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, st));
            if (mode == 0) hipLaunchKernelGGL(bar_test<0>, dim3(G), dim3(256), 0, st, ctr, buf, iters, out, err);
            else hipLaunchKernelGGL(bar_test<1>, dim3(G), dim3(256), 0, st, ctr, buf, iters, out, err);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            long h[2]; int he; CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(&he, err, 4, hipMemcpyDeviceToHost));
            printf("mode %d grid %4d: %.3f us per barrier (in-kernel %.3f us), errors %d\n", mode, G, ms * 1e3 / iters, h[0] * 0.01 / iters, he);
        }
    return 0;
}
