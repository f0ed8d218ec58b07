// Does a kernel ALWAYS see what the host stored into device memory (large BAR, memcpy + sfence) just before it launched the graph
// the kernel is the first node of — read with system-scope loads (sc0 sc1), by workgroups on every XCD, while the PREVIOUS graph's
// tail may still be running (the pipelined per-timestep path's pattern)? One stale read in the product is a wrong observation or a
// wrong transition row; this probe counts them over many iterations.
//   hipcc -O2 --offload-arch=gfx950 -o bar_visibility bar_visibility.hip && ./bar_visibility [iters=200000] [tail_us=25] [host_us=8]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <emmintrin.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define NWG 48
#define WORDS 68   // [row: 64 words | count | 3 pad] as the product publishes it (272 bytes, three 128-byte lines)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mkbuf(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000);
}
// first node: every workgroup reads all 68 words with system-scope loads and records (a) how many differ from the value the count
// word says, (b) the count word itself
__global__ void reader(const unsigned* pub, unsigned* seen, unsigned* bad, volatile unsigned* seq_host, unsigned* ctr) {
    const int tid = threadIdx.x, wg = blockIdx.x;
    unsigned v = 0;
    if (tid < WORDS) v = __builtin_amdgcn_raw_buffer_load_b32(mkbuf(pub), 4u * (unsigned)tid, 0, 17);
    __shared__ unsigned sv[WORDS];
    if (tid < WORDS) sv[tid] = v;
    __syncthreads();
    if (tid == 0) {
        const unsigned want = sv[64];
        unsigned nb = 0;
        for (int i = 0; i < 64; ++i) nb += sv[i] != want;
        seen[wg] = want;
        if (nb) atomicAdd(&bad[1], 1u);
        if (wg == 0) {
            const unsigned c = *ctr + 1;
            *ctr = c;
            __hip_atomic_store((unsigned*)seq_host, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// the rest of the "graph": a few dependent kernels that keep the GPU busy (the chain) and touch memory (so L2s have traffic)
__global__ void tail(float* scratch, int n, long long ticks) {
    const long long t0 = wall_clock64();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) scratch[i] = scratch[i] * 1.0001f + 1.0f;
    while (wall_clock64() - t0 < ticks) { }
}
// last node: compare what the readers saw with the expected ordinal (kept on the device)
__global__ void check(const unsigned* seen, unsigned* bad, const unsigned* ctr) {
    const unsigned want = *ctr;
    if (threadIdx.x < NWG && seen[threadIdx.x] != want) atomicAdd(&bad[0], 1u);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200000;
    const int tail_us = argc > 2 ? atoi(argv[2]) : 25;
    const double host_us = argc > 3 ? atof(argv[3]) : 8.0;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    unsigned *pub, *seen, *bad, *ctr, *seq;
    float* scratch;
    CK(hipMalloc(&pub, 4096));
    CK(hipMemset(pub, 0, 4096));
    CK(hipMalloc(&seen, NWG * 4));
    CK(hipMalloc(&bad, 16));
    CK(hipMemset(bad, 0, 16));
    CK(hipMalloc(&ctr, 4));
    CK(hipMemset(ctr, 0, 4));
    CK(hipMalloc(&scratch, 1 << 22));
    CK(hipMemset(scratch, 0, 1 << 22));
    CK(hipHostMalloc(&seq, 64, hipHostMallocDefault));
    *seq = 0;
    auto body = [&]() {
        reader<<<NWG, 128, 0, st>>>(pub, seen, bad, seq, ctr);
        for (int k = 0; k < 5; ++k) tail<<<256, 256, 0, st>>>(scratch, 1 << 16, 100LL * tail_us / 5);
        check<<<1, 64, 0, st>>>(seen, bad, ctr);
    };
    unsigned host_row[WORDS];
    auto publish = [&](unsigned v) {
        for (int i = 0; i < WORDS; ++i) host_row[i] = v;
        memcpy(pub, host_row, sizeof(host_row));        // straight into device memory (large BAR)
        _mm_sfence();
    };
    publish(1);
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
    unsigned expect = 1;
    auto t_start = now();
    for (int it = 0; it < iters; ++it) {
        ++expect;
        publish(expect);
        CK(hipGraphLaunch(ge, st));
        while (*(volatile unsigned*)seq != expect) { }   // the first kernel's word (the action): the tail still runs
        auto t2 = now();
        while (us(t2, now()) < host_us) { }
    }
    CK(hipStreamSynchronize(st));
    const double per = us(t_start, now()) / iters;
    unsigned hb[4];
    CK(hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost));
    printf("%d timesteps (%.1f us each; %d workgroups x 68 words read per timestep, graph tail %d us, host %.0f us): "
           "%u workgroup reads saw a STALE count word, %u saw a row that did not match its count word\n",
           iters, per, NWG, tail_us, host_us, hb[0], hb[1]);
    return 0;
}
