// The 5 - 6 us between two hipGraphLaunch'es on ONE stream (pipeline_gap.hip) are the queue's: would the graphs of consecutive timesteps
// on TWO alternating streams, ordered by a device word instead — the first kernel of graph t + 1 polls for the count of completed
// graphs that the last kernel of graph t raises behind a release — give them back?
//   hipcc -O2 --offload-arch=gfx950 -o two_stream_gap two_stream_gap.hip && ./two_stream_gap [N=5] [first_us=8] [rest_us=6] [host_us=14]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// need: mapped host word = the ordinal of this graph (graphs that must have completed before it); done[0]: graphs completed;
// done[32]: workgroups of the last kernel that are through
__global__ void work_kernel(int* chain, int k, long long ticks, volatile unsigned* seq, unsigned* ctr, const unsigned* need, unsigned* done,
                            int first, int last, int flagged) {
    if (flagged && first) {
        if (threadIdx.x == 0) {
            const unsigned n = __hip_atomic_load(need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const long long t1 = wall_clock64();
            while (__hip_atomic_load(&done[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n) {
                if (wall_clock64() - t1 > 100000000LL) break;                      // (1 s: never in a healthy run)
                __builtin_amdgcn_s_sleep(2);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) chain[k + 1] = chain[k] + 1;
    while (wall_clock64() - t0 < ticks) { }
    if (seq && threadIdx.x == 0 && blockIdx.x == 0) {
        const unsigned v = *ctr + 1;
        *ctr = v;
        __hip_atomic_store((unsigned*)seq, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (flagged && last && threadIdx.x == 0) {
        if (__hip_atomic_fetch_add(&done[32], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
            __hip_atomic_store(&done[32], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&done[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 5, first_us = argc > 2 ? atoi(argv[2]) : 8, rest_us = argc > 3 ? atoi(argv[3]) : 6;
    const double host_us = argc > 4 ? atof(argv[4]) : 14.0;
    const int iters = 3000;
    hipStream_t st[2];
    CK(hipStreamCreate(&st[0]));
    CK(hipStreamCreate(&st[1]));
    int* chain;
    unsigned *ctr, *done, *seq, *need;
    CK(hipMalloc(&chain, 64 * sizeof(int)));
    CK(hipMemset(chain, 0, 64 * sizeof(int)));
    CK(hipMalloc(&ctr, sizeof(unsigned)));
    CK(hipMalloc(&done, 64 * sizeof(unsigned)));
    CK(hipHostMalloc(&seq, 64, hipHostMallocDefault));
    CK(hipHostMalloc(&need, 64, hipHostMallocDefault));
    auto body = [&](hipStream_t s, int flagged) {
        for (int k = 0; k < N; ++k)
            work_kernel<<<40, 256, 0, s>>>(chain, k, 100LL * (k == 0 ? first_us : rest_us), k == 0 ? seq : nullptr, ctr, need, done, k == 0, k == N - 1,
                                           flagged);
    };
    hipGraphExec_t ge[2][2];                    // [flagged][stream]
    for (int f = 0; f < 2; ++f)
        for (int s = 0; s < 2; ++s) {
            hipGraph_t g;
            CK(hipStreamBeginCapture(st[s], hipStreamCaptureModeGlobal));
            body(st[s], f);
            CK(hipStreamEndCapture(st[s], &g));
            CK(hipGraphInstantiate(&ge[f][s], g, nullptr, nullptr, 0));
        }
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    printf("N = %d kernels (first %d us, the others %d us: %d us of GPU work per timestep), %.0f us of host work between the action and the "
           "next submission\n", N, first_us, rest_us, first_us + (N - 1) * rest_us, host_us);
    for (int mode = 0; mode < 2; ++mode) {
        CK(hipDeviceSynchronize());
        CK(hipMemset(ctr, 0, sizeof(unsigned)));
        CK(hipMemset(done, 0, 64 * sizeof(unsigned)));
        CK(hipDeviceSynchronize());
        *(volatile unsigned*)seq = 0;
        unsigned expect = 0;
        std::vector<double> call;
        auto t_start = now();
        for (int it = 0; it < iters; ++it) {
            auto t0 = now();
            if (mode == 0) CK(hipGraphLaunch(ge[0][0], st[0]));
            else {
                *(volatile unsigned*)need = (unsigned)it;                          // graphs 0 .. it - 1 must be through
                CK(hipGraphLaunch(ge[1][it & 1], st[it & 1]));
            }
            auto t1 = now();
            call.push_back(us(t0, t1));
            ++expect;
            while (*(volatile unsigned*)seq != expect) { }                        // the action
            auto t2 = now();
            while (us(t2, now()) < host_us) { }                                   // the environment, Python
        }
        CK(hipDeviceSynchronize());
        const double per = us(t_start, now()) / iters;
        int last = 0;
        CK(hipMemcpy(&last, chain + N, sizeof(int), hipMemcpyDeviceToHost));
        std::sort(call.begin(), call.end());
        printf("  %s: %.1f us per timestep (GPU work %d), submission call %.1f us (median)\n",
               mode == 0 ? "one stream, ordered by the queue           " : "two alternating streams, ordered by a flag ", per,
               first_us + (N - 1) * rest_us, call[call.size() / 2]);
    }
    return 0;
}
