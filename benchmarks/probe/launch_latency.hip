// What does a timestep pay between "the host decides to launch" and "the host sees the last kernel's result" for a chain of N
// dependent kernels of ~W microseconds each, submitted (a) as ONE hipGraphLaunch of a captured graph, (b) as N direct
// hipLaunchKernel calls issued back to back by one host call? The per-timestep path (NAFAgent.step -> act: seven launches,
// ~45 us of GPU work) runs (a); this probe says what (b) would buy.   hipcc -O2 --offload-arch=gfx950 -o launch_latency launch_latency.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void work_kernel(int* chain, int k, long long ticks, volatile unsigned* seq, unsigned val) {
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) {          // a dependency on the launch in front; chain[.] counts the submissions
        const int v = chain[k] + (k == 0);
        if (k == 0) chain[0] = v;
        chain[k + 1] = v;
    }
    while (wall_clock64() - t0 < ticks) { }
    if (seq && threadIdx.x == 0 && blockIdx.x == 0) *seq = val ? val : (unsigned)chain[k + 1];
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 7, W_us = argc > 2 ? atoi(argv[2]) : 6, iters = 2000;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    int* chain;
    CK(hipMalloc(&chain, 64 * sizeof(int)));
    CK(hipMemset(chain, 0, 64 * sizeof(int)));
    unsigned* seq;
    CK(hipHostMalloc(&seq, 64, hipHostMallocDefault));
    *seq = 0;
    const long long ticks = 100LL * W_us;
    auto body = [&](unsigned val) {
        for (int k = 0; k < N; ++k)
            work_kernel<<<k == 0 ? 1 : 40, 256, 0, st>>>(chain, k, ticks, k == N - 1 ? seq : nullptr, val);
    };
    // (a) graph: the value the last kernel writes is a launch-time constant, so the host waits for a counter in `chain` instead
    hipGraph_t g;
    hipGraphExec_t ge;
    body(1);
    CK(hipStreamSynchronize(st));
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int k = 0; k < N; ++k) work_kernel<<<k == 0 ? 1 : 40, 256, 0, st>>>(chain, k, ticks, k == N - 1 ? seq : nullptr, 0);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    std::vector<double> tg, tgl, td, tdl, ts;
    for (int it = 0; it < iters; ++it) {
        auto t0 = now();
        CK(hipGraphLaunch(ge, st));
        auto t1 = now();
        CK(hipStreamSynchronize(st));
        auto t2 = now();
        ts.push_back(us(t0, t2));
    }
    // chain[k + 1] += chain[k]'s increment: make chain[0] grow by one per replay so that the last kernel's word changes
    for (int it = 0; it < iters; ++it) {
        const unsigned before = *(volatile unsigned*)seq;
        auto t0 = now();
        CK(hipGraphLaunch(ge, st));
        auto t1 = now();
        while (*(volatile unsigned*)seq == before) { }
        auto t2 = now();
        tg.push_back(us(t0, t2));
        tgl.push_back(us(t0, t1));
        CK(hipStreamSynchronize(st));
    }
    for (int it = 0; it < iters; ++it) {
        const unsigned val = 1000 + it;
        auto t0 = now();
        body(val);
        auto t1 = now();
        while (*(volatile unsigned*)seq != val) { }
        auto t2 = now();
        td.push_back(us(t0, t2));
        tdl.push_back(us(t0, t1));
    }
    // graph + spin on a pinned word is what the product does; measure that too: last node writes seq through a device counter
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("N = %d kernels of %d us: ideal %d us of GPU work\n", N, W_us, N * W_us);
    printf("  graph : launch call %.1f us, launch -> stream synchronised %.1f us, launch -> last kernel's pinned word seen %.1f us (median of %d)\n", med(tgl), med(ts), med(tg), iters);
    printf("  direct: launch calls %.1f us, launch -> last kernel's pinned word seen %.1f us\n", med(tdl), med(td));
    return 0;
}
