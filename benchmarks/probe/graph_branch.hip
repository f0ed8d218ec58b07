// Round 6: what does a PARALLEL BRANCH cost inside the per-timestep graph? The pipelined timestep is a chain of 6 launches
// (~7 us + 5 x 5 us); the prefetch of a later minibatch (one workgroup, ~13 us) should run beside the chain, not in front of it.
// Three ways to submit "chain + one side kernel" once per timestep, GPU-bound, the host waiting for the first kernel only:
//   0  the chain alone as one graph                                   (the floor)
//   1  the chain with the side kernel as the first launch's tail      (round 5: the side work serialised in front of launch 2)
//   2  one graph with a fork at the root and a join at the end        (stream capture: event fork / join)
//   3  one graph with a fork BEHIND the first kernel and a join at the end
//   4  the chain as a graph + the side kernel launched directly on a second stream, no dependency between them
//   hipcc -O2 --offload-arch=gfx950 -o graph_branch graph_branch.hip && ./graph_branch [N=6] [first_us=7] [rest_us=5] [side_us=13] [host_us=9]
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
__global__ void side_kernel(int* out, long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (threadIdx.x == 0) out[0] += 1;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 6, first_us = argc > 2 ? atoi(argv[2]) : 7, rest_us = argc > 3 ? atoi(argv[3]) : 5;
    const int side_us = argc > 4 ? atoi(argv[4]) : 13;
    const double host_us = argc > 5 ? atof(argv[5]) : 9.0;
    const int iters = 3000;
    hipStream_t st, s2;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    int *chain, *side;
    unsigned* ctr;
    CK(hipMalloc(&chain, 64 * sizeof(int)));
    CK(hipMemset(chain, 0, 64 * sizeof(int)));
    CK(hipMalloc(&side, 64 * sizeof(int)));
    CK(hipMemset(side, 0, 64 * sizeof(int)));
    CK(hipMalloc(&ctr, sizeof(unsigned)));
    CK(hipMemset(ctr, 0, sizeof(unsigned)));
    unsigned* seq;
    CK(hipHostMalloc(&seq, 64, hipHostMallocDefault));
    *seq = 0;
    hipEvent_t ef, ej;
    CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    auto chain_body = [&](int first_extra_us, int fork_at) -> int {
        for (int k = 0; k < N; ++k) {
            if (k == fork_at) {
                CK(hipEventRecord(ef, st));
                CK(hipStreamWaitEvent(s2, ef, 0));
                side_kernel<<<1, 1024, 0, s2>>>(side, 100LL * side_us);
                CK(hipEventRecord(ej, s2));
            }
            work_kernel<<<40, 256, 0, st>>>(chain, k, 100LL * (k == 0 ? first_us + first_extra_us : rest_us), k == 0 ? seq : nullptr, ctr);
        }
        if (fork_at >= 0) CK(hipStreamWaitEvent(st, ej, 0));
        return 0;
    };
    chain_body(0, -1);
    side_kernel<<<1, 1024, 0, s2>>>(side, 100);
    CK(hipDeviceSynchronize());
    hipGraph_t g[4];
    hipGraphExec_t ge[4];
    for (int v = 0; v < 4; ++v) {
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        // v = 1: the side work as the tail of the first launch: the first launch lasts max(first, side) but announces at `first`
        if (v == 0) { if (chain_body(0, -1)) return 1; }
        else if (v == 1) {
            // (the announcement comes at first_us; the kernel then idles on to side_us: modelled by a second kernel of the difference)
            for (int k = 0; k < N; ++k) {
                work_kernel<<<40, 256, 0, st>>>(chain, k, 100LL * (k == 0 ? first_us : rest_us), k == 0 ? seq : nullptr, ctr);
                if (k == 0 && side_us > first_us) side_kernel<<<1, 1024, 0, st>>>(side, 100LL * (side_us - first_us - 2));
            }
        } else if (chain_body(0, v == 2 ? 0 : 1)) return 1;
        CK(hipStreamEndCapture(st, &g[v]));
        CK(hipGraphInstantiate(&ge[v], g[v], nullptr, nullptr, 0));
    }
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    printf("N = %d kernels (first %d us, the others %d us: %d us of chain per timestep), side kernel %d us, %.0f us of host work\n", N, first_us,
           rest_us, first_us + (N - 1) * rest_us, side_us, host_us);
    const char* names[5] = {"chain alone, one graph                      ", "side work serialised behind launch 1 (r05)  ",
                            "graph: fork at the root, join at the end    ", "graph: fork behind launch 1, join at the end",
                            "graph + direct launch on a second stream    "};
    for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 5; ++mode) {
        CK(hipDeviceSynchronize());
        unsigned expect = *(volatile unsigned*)seq;
        std::vector<double> call;
        auto t_start = now();
        for (int it = 0; it < iters; ++it) {
            auto t0 = now();
            CK(hipGraphLaunch(ge[mode == 4 ? 0 : mode], st));
            if (mode == 4) side_kernel<<<1, 1024, 0, s2>>>(side, 100LL * side_us);
            auto t1 = now();
            call.push_back(us(t0, t1));
            ++expect;
            while (*(volatile unsigned*)seq != expect) { }                        // the action
            auto t2 = now();
            while (us(t2, now()) < host_us) { }                                   // the environment, Python
        }
        CK(hipDeviceSynchronize());
        const double per = us(t_start, now()) / iters;
        std::sort(call.begin(), call.end());
        printf("  %s: %.1f us per timestep, submission call %.1f us (median)\n", names[mode], per, call[call.size() / 2]);
    }
    return 0;
}
