// Round 6 dev tool (GPU box): what shader clock does the chip run at WHILE another process keeps it busy?  One wave spins ~1 ms and reports
// clock64 (shader cycles) against wall_clock64 (the 100 MHz constant counter); sampled every 50 ms for `seconds`.  Run it beside the workload:
//   ./clock_probe.bin 8 > idle.txt;   (./clock_probe.bin 20 > decode.txt &) ; python bench.py --decode-only ...
// A wave's clock is its XCD's; the probe lands on whichever CU the dispatcher picks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
__global__ void spin(long n, long *out) {
    const long c0 = clock64(), w0 = wall_clock64();
    float a = 1.f;
    for (long i = 0; i < n; ++i) a = a * 1.0000001f + 1e-9f;
    const long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long)a; }
}
int main(int argc, char **argv) {
    const double seconds = argc > 1 ? std::atof(argv[1]) : 5.0;
    long *d, h[4];
    CK(hipMalloc(&d, 32));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const auto t0 = std::chrono::steady_clock::now();
    double lo = 1e9, hi = 0, sum = 0; int n = 0;
    static int series[4096];
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, 400000L, d);
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(h, d, 24, hipMemcpyDeviceToHost));
        const double mhz = (double)h[0] / (double)h[1] * 100.0;
        lo = mhz < lo ? mhz : lo; hi = mhz > hi ? mhz : hi; sum += mhz; if (n < 4096) series[n] = (int)mhz; ++n;
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
    std::printf("shader clock over %.1f s, %d samples: mean %.0f MHz, min %.0f, max %.0f\n", seconds, n, sum / n, lo, hi);
    std::printf("  series (MHz, one sample per ~51 ms):");
    for (int i = 0; i < n && i < 4096; ++i) std::printf(" %d", series[i]);
    std::printf("\n");
    return 0;
}
