// Probe for the "one persistent launch confined to a single XCD" design: what do its ingredients
// cost on MI355X?  (1) a census kernel: which XCC does each workgroup run on (HW_REG_XCC_ID) — used to find the CU-mask
// convention of hipExtStreamCreateWithCUMask and to check that a masked stream (also under hipGraph replay) really
// confines kernels; (2) a persistent kernel that confines ITSELF to one XCD (workgroups elsewhere exit, the 32 on the
// chosen XCD take tickets) and runs `iters` rounds of {write a payload, 32-workgroup barrier, read a neighbour's payload},
// in two flavours: the portable agent-scope release / acquire, and an XCD-local form (stores drained with
// s_waitcnt vmcnt(0), sc1 loads: the XCD's L2 is the coherence point, no L2 write-back) — errors are counted.
// Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC persistent_probe.hip -o libpp.so (benchmarks/persistent_probe.py)
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ static inline int xcc_id() {
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15;
}

__global__ void census_kernel(int* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

extern "C" int pp_census(int* out, int n_blocks, int threads, int lds_bytes, void* stream) {
    census_kernel<<<n_blocks, threads, lds_bytes, (hipStream_t)stream>>>(out);
    return (int)hipGetLastError();
}

extern "C" int pp_make_masked_stream(void** stream_out, const uint32_t* mask, int words) {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask);
    *stream_out = (void*)s;
    return (int)e;
}

extern "C" int pp_destroy_stream(void* s) { return (int)hipStreamDestroy((hipStream_t)s); }

// "Blockers": 1024-thread workgroups, two per CU = all 32 wave slots of the CU; the ones on XCCs in `free_mask` leave at once,
// the others wait (one lane polls, the rest sit at the barrier) until the host raises *flag (pinned host memory) or
// `timeout_ticks` of the 100 MHz wall clock pass. While they are resident no other wave can be placed on their CUs: the way
// this probe confines the REAL update chain to one XCD (hipExtStreamCreateWithCUMask is ignored on this box: the census
// finds all 8 XCCs whatever the mask).
__global__ __launch_bounds__(1024) void blocker_kernel(const int* flag, long long timeout_ticks, int free_mask,
                                                       unsigned long long* resident) {
    if ((free_mask >> xcc_id()) & 1) return;
    if (threadIdx.x == 0) {
        atomicAdd(resident, 1ull);
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0 && wall_clock64() - t0 < timeout_ticks)
            __builtin_amdgcn_s_sleep(64);
    }
    __syncthreads();
}

extern "C" int pp_blocker(const int* host_flag, double timeout_s, int free_mask, unsigned long long* resident, int n_blocks,
                          void* stream) {
    if (!(timeout_s > 0.0) || timeout_s > 30.0) return -1;
    blocker_kernel<<<n_blocks, 1024, 0, (hipStream_t)stream>>>(host_flag, (long long)(timeout_s * 1e8), free_mask, resident);
    return (int)hipGetLastError();
}

#define PP_THREADS 256
#define PP_WG 32
// ctrl: [0] tickets, [1] barrier arrivals, [2] errors, [3] time-outs, [4] workers seen, [8..] per-xcc counters
template <int LOCAL>
__global__ __launch_bounds__(PP_THREADS) void barrier_probe_kernel(unsigned long long* ctrl, float* payload, int iters,
                                                                   int want_xcc, int payload_floats) {
    extern __shared__ float lds_pad[];      // sized by the host so that one workgroup fits per CU
    __shared__ int s_w;
    const int tid = threadIdx.x;
    if (xcc_id() != want_xcc) return;
    if (tid == 0) s_w = (int)atomicAdd(&ctrl[0], 1ull);
    __syncthreads();
    const int w = s_w;
    if (w >= PP_WG) return;                  // more than 32 workgroups landed here: the extras leave
    // wait until all 32 workers exist (bounded)
    if (tid == 0) {
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(&ctrl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)PP_WG) {
            if (wall_clock64() - t0 > 200000000ll) { atomicAdd(&ctrl[3], 1ull); break; }   // 2 s
            __builtin_amdgcn_s_sleep(4);
        }
        atomicAdd(&ctrl[4], 1ull);
    }
    __syncthreads();
    if (__hip_atomic_load(&ctrl[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    unsigned errors = 0;
    for (int it = 0; it < iters; ++it) {
        float* mine = payload + ((size_t)(it & 1) * PP_WG + w) * payload_floats;
        for (int i = tid; i < payload_floats; i += PP_THREADS) {
            const float v = (float)(it * 64 + w) + (float)(i & 7);
            if (LOCAL) __builtin_nontemporal_store(v, mine + i);     // (any store: visibility comes from the drain below)
            else mine[i] = v;
        }
        // ---- barrier over the 32 workgroups ----
        if (LOCAL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have reached the XCD's L2
        __syncthreads();
        if (tid == 0) {
            if (!LOCAL) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __hip_atomic_fetch_add(&ctrl[1], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long want = (unsigned long long)PP_WG * (unsigned long long)(it + 1);
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(&ctrl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (wall_clock64() - t0 > 200000000ll) { atomicAdd(&ctrl[3], 1ull); break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (!LOCAL) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __syncthreads();
        // ---- read the neighbour's payload ----
        const int nb = (w + 1) & (PP_WG - 1);
        const float* theirs = payload + ((size_t)(it & 1) * PP_WG + nb) * payload_floats;
        for (int i = tid; i < payload_floats; i += PP_THREADS) {
            float v;
            if (LOCAL) v = __hip_atomic_load(theirs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1: past the L1
            else v = theirs[i];
            if (v != (float)(it * 64 + nb) + (float)(i & 7)) ++errors;
        }
    }
    if (errors) atomicAdd(&ctrl[2], (unsigned long long)errors);
}

extern "C" int pp_barrier_probe(unsigned long long* ctrl, float* payload, int iters, int want_xcc, int payload_floats,
                                int local, int n_blocks, int lds_bytes, void* stream) {
    hipError_t e = hipFuncSetAttribute(local ? (const void*)barrier_probe_kernel<1> : (const void*)barrier_probe_kernel<0>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return (int)e;
    if (local)
        barrier_probe_kernel<1><<<n_blocks, PP_THREADS, lds_bytes, (hipStream_t)stream>>>(ctrl, payload, iters, want_xcc, payload_floats);
    else
        barrier_probe_kernel<0><<<n_blocks, PP_THREADS, lds_bytes, (hipStream_t)stream>>>(ctrl, payload, iters, want_xcc, payload_floats);
    return (int)hipGetLastError();
}
