// The pipelined per-timestep path is GPU-bound with a light environment: a timestep's graph (6 launches, ~37 us) is launched while
// the previous one still runs, and the host only waits for the FIRST kernel of each (the action). What does the boundary between
// two such submissions cost — as two hipGraphLaunch calls, or as 2 x 6 direct launches on the same stream?
//   hipcc -O2 --offload-arch=gfx950 -o pipeline_gap pipeline_gap.hip && ./pipeline_gap [N=6] [first_us=7] [rest_us=5] [host_us=9]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void work_kernel(int* chain, int k, long long ticks, volatile unsigned* seq, unsigned* ctr) {
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) chain[k + 1] = chain[k] + 1;        // a dependency on the launch in front
    while (wall_clock64() - t0 < ticks) { }
    if (seq && threadIdx.x == 0 && blockIdx.x == 0) {
        const unsigned v = *ctr + 1;                                                // (the ordinal lives on the device: graphs replay constants)
        *ctr = v;
        __hip_atomic_store((unsigned*)seq, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 6, first_us = argc > 2 ? atoi(argv[2]) : 7, rest_us = argc > 3 ? atoi(argv[3]) : 5;
    const double host_us = argc > 4 ? atof(argv[4]) : 9.0;
    const int iters = 3000;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    int* chain;
    unsigned* ctr;
    CK(hipMalloc(&chain, 64 * sizeof(int)));
    CK(hipMemset(chain, 0, 64 * sizeof(int)));
    CK(hipMalloc(&ctr, sizeof(unsigned)));
    CK(hipMemset(ctr, 0, sizeof(unsigned)));
    unsigned* seq;
    CK(hipHostMalloc(&seq, 64, hipHostMallocDefault));
    *seq = 0;
    auto body = [&]() {
        for (int k = 0; k < N; ++k)
            work_kernel<<<40, 256, 0, st>>>(chain, k, 100LL * (k == 0 ? first_us : rest_us), k == 0 ? seq : nullptr, ctr);
    };
    body();
    CK(hipStreamSynchronize(st));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    body();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    printf("N = %d kernels (first %d us, the others %d us: %d us of GPU work per timestep), %.0f us of host work between the action and the "
           "next submission\n", N, first_us, rest_us, first_us + (N - 1) * rest_us, host_us);
    for (int mode = 0; mode < 2; ++mode) {
        CK(hipStreamSynchronize(st));
        unsigned expect = *(volatile unsigned*)seq;
        std::vector<double> call;
        auto t_start = now();
        for (int it = 0; it < iters; ++it) {
            auto t0 = now();
            if (mode == 0) CK(hipGraphLaunch(ge, st));
            else body();
            auto t1 = now();
            call.push_back(us(t0, t1));
            ++expect;
            while (*(volatile unsigned*)seq != expect) { }                        // the action
            auto t2 = now();
            while (us(t2, now()) < host_us) { }                                   // the environment, Python
        }
        CK(hipStreamSynchronize(st));
        const double per = us(t_start, now()) / iters;
        std::sort(call.begin(), call.end());
        printf("  %s: %.1f us per timestep (GPU work %d), submission call %.1f us (median)\n", mode == 0 ? "one hipGraphLaunch per timestep " : "direct launches, same stream   ",
               per, first_us + (N - 1) * rest_us, call[call.size() / 2]);
    }
    return 0;
}
