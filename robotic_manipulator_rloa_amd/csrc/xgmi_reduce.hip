// One-shot sum all-reduce of the flat gradient between the GPUs of one node, over peer-mapped (hipIpc) memory:
// the data-parallel exchange of SURVEY.md §8e without a ring. Every rank PUSHES its 4*P-byte gradient straight into a
// receive slot on each of its W-1 peers (7 xGMI links in parallel at W = 8), raises one epoch flag per peer, waits for
// its own W-1 flags and then sums the W contributions in RANK ORDER, so every replica computes bit-identical sums.
// The reducing kernel also emits the sum-of-squares partials clip_grad_norm_ needs (the separate norm launch of the
// RCCL path disappears) and advances the optimizer step count.
//
//   ONE launch, xgmi_allreduce_kernel, grid (chunks): every workgroup pushes its 4-KB chunk (16-B stores into each
//   peer's memory) -> system-scope release fence -> arrival counter, the LAST workgroup to arrive publishes the epoch flag
//   on every peer -> wave 0 polls the W-1 local flags (bounded by wall clock: a missing peer ends in a recorded time-out,
//   never in a hang) -> system-scope acquire -> rank-ordered sum of the chunk -> gradient + sumsq partial
//
// Receive slots and flags live in device memory allocated UNCACHED (hipDeviceMallocUncached: not kept in any L2, so a
// line written by a remote GPU is what the next local load returns); slots are double-buffered by epoch parity. A peer
// can never be two epochs ahead: its launch of epoch e+2 is stream-ordered behind its launch of e+1, whose every
// workgroup waited for this rank's flag of e+1, raised only after this rank's launch of e had finished.
// Nothing here allocates or synchronises after naf_xgmi_connect; the launch is a plain kernel launch and can be captured
// into a hipGraph (the epoch lives on the device).
#include <stdlib.h>
#include <string.h>
#include "xgmi_dev.h"

#define XG_THREADS 256
#define XG_CHUNK (XG_THREADS * 4)          // floats per workgroup: one float4 per thread
#define XG_FLAG_STRIDE 128                 // bytes: one flag per line
#define XG_TICKS_PER_S 100000000ll         // wall_clock64() runs at 100 MHz

struct XgPeers {
    char* base[NAF_XGMI_MAX_WORLD];        // slab of every rank as mapped into THIS process (base[rank] = local)
};

struct XgmiComm {
    int rank, world, mem_kind;
    size_t n, n_pad, data_off, slab_bytes;
    char* local;
    XgPeers peers;
    bool opened[NAF_XGMI_MAX_WORLD];
    uint64_t* ctrl;                        // device, ordinary memory: [0] epoch [1] arrivals [2] time-outs [3] spare
    uint64_t* host_timeouts;               // pinned host word the kernel bumps on a time-out: the host reads it WITHOUT a sync
    long long timeout_ticks;
};

// One launch per all-reduce. No workgroup waits for another workgroup of its own launch (the last one to ARRIVE raises
// the flags; nobody polls the arrival counter), so the only waits are on the peers' flags, which depend on nothing but the
// peers' own pushes: no circular wait whatever the residency of the grid.
template <int world>
__global__ __launch_bounds__(XG_THREADS) void xgmi_allreduce_kernel(XgPeers peers, const float* __restrict__ grad_in,
                                                                    float* __restrict__ grad_out, size_t n, size_t n_pad,
                                                                    size_t data_off, int rank,
                                                                    uint64_t* __restrict__ ctrl,
                                                                    float* __restrict__ sumsq_partials, int32_t* step_dev,
                                                                    long long timeout_ticks, size_t pushed_lo,
                                                                    size_t skip_lo, size_t skip_hi,
                                                                    uint64_t* __restrict__ host_timeouts) {
    // pushed_lo: grad_in[pushed_lo, n) has already been pushed for this epoch (naf_xgmi_push_early or the layer-1
    // backward kernel's extra workgroups, in an earlier launch of this stream); pushed_lo = n: nothing has.
    // [skip_lo, skip_hi): a second range that went ahead (the row-split chain's finish launch pushes the two weight-gradient
    // segments W2 and Wh, which do not touch in the flat buffer: Wh = [pushed_lo, n), W2 = [skip_lo, skip_hi))
    __shared__ float red[XG_THREADS / 64];
    __shared__ int last;
    __shared__ int timed_out;
    if (threadIdx.x == 0) timed_out = 0;   // (published by the barrier behind the pushes)
    const uint64_t e = ctrl[0] + 1;        // nobody writes ctrl[0] before every workgroup has arrived below
    const size_t i = (size_t)blockIdx.x * XG_CHUNK + (size_t)threadIdx.x * 4;
    const bool on = i < n;                 // n is a multiple of 4 (checked on the host)
    // ---- push: this workgroup's chunk of the local gradient into the slot `rank` of every peer ---------------------
    xg_f4 mine = {0.f, 0.f, 0.f, 0.f};
    if (on) mine = *(const xg_f4*)(grad_in + i);
    if (on && i < pushed_lo && !(i >= skip_lo && i < skip_hi)) {
        // branch-free on purpose (the own slab gets a copy nobody reads): with `if (p != rank)` around each store the
        // compiler put an s_waitcnt vmcnt(0) in front of every one of them — W-1 SERIAL round trips over xGMI
#pragma unroll
        for (int p = 0; p < world; ++p) *(xg_f4*)(xg_slot(peers.base[p], data_off, n_pad, world, e, rank) + i) = mine;
    }
    // every wave: its stores have reached the peers before it arrives. A RELEASE is all this side needs
    // (__threadfence_system() = release + acquire would also invalidate this CU's caches: ~1.7 us for nothing)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    if (threadIdx.x == 0) last = (atomicAdd((unsigned long long*)&ctrl[1], 1ull) == (unsigned long long)gridDim.x - 1);
    __syncthreads();
    if (last) {
        if (threadIdx.x < world && (int)threadIdx.x != rank) {
            uint64_t* flag = (uint64_t*)(peers.base[threadIdx.x] + (size_t)rank * XG_FLAG_STRIDE);
            __hip_atomic_store(flag, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (threadIdx.x == 0) {
            ctrl[1] = 0;
            ctrl[0] = e;
        }
    }
    // ---- wait for the W-1 peers' flags (bounded), then the rank-ordered sum of this chunk ------------------------------
    if (threadIdx.x < world && (int)threadIdx.x != rank) {
        const uint64_t* flag = (const uint64_t*)(peers.base[rank] + (size_t)threadIdx.x * XG_FLAG_STRIDE);
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < e) {
            if (wall_clock64() - t0 > timeout_ticks) {       // the exit every wave reaches: peer missing or dead
                atomicAdd((unsigned long long*)&ctrl[2], 1ull);
                if (host_timeouts) __hip_atomic_fetch_add(host_timeouts, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                timed_out = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");            // system scope: drop anything this CU still holds
    float ss = 0.f;
    if (on) {
        // all W-1 peer contributions in flight together (the slot of the own rank is never written: its address is
        // read like the others and the value replaced, which keeps the loop free of divergent addressing)
        const float* slot0 = xg_slot(peers.base[rank], data_off, n_pad, world, e, 0) + i;
        xg_f4 v[world];
#pragma unroll
        for (int s = 0; s < world; ++s) v[s] = *(const xg_f4*)(slot0 + (size_t)s * n_pad);
        xg_f4 acc = (rank == 0) ? mine : v[0];
#pragma unroll
        for (int s = 1; s < world; ++s) acc = acc + (s == rank ? mine : v[s]);
        *(xg_f4*)(grad_out + i) = acc;
        ss = acc.x * acc.x + acc.y * acc.y + acc.z * acc.z + acc.w * acc.w;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int k = 0; k < XG_THREADS / 64; ++k) s += red[k];
        // a contribution is missing: what was summed is not the gradient. The partial is POISONED (a sum of squares is
        // never negative): naf_adam_polyak_fused skips the whole update when the norm it folds comes out negative, so
        // a slow or dead peer cannot push a wrong step into the weights; the time-out is counted where the host sees
        // it without synchronising (naf_xgmi_timeouts_nowait) and the training loop raises on it
        // (the partials of workgroups that did NOT time out stay below 1e30, so that their sum with a poisoned one is -inf
        //  and never inf - inf = NaN, which the optimizer's `norm < 0` test would let through; a NaN partial — a NaN
        //  gradient — stays NaN and fails loudly as it does on one GPU)
        if (sumsq_partials) sumsq_partials[blockIdx.x] = timed_out ? -__builtin_huge_valf() : (s > 1e30f ? 1e30f : s);
        if (blockIdx.x == 0 && step_dev && !timed_out) *step_dev += 1;
    }
}

static inline XgmiComm* xg_comm(void* h) { return (XgmiComm*)h; }

extern "C" int naf_xgmi_chunk_floats(void) { return XG_CHUNK; }

extern "C" int naf_xgmi_create(int rank, int world, size_t n_floats, double timeout_s, void** handle) {
    if (!handle || world < 2 || world > NAF_XGMI_MAX_WORLD || rank < 0 || rank >= world || n_floats == 0 ||
        (n_floats & 3) != 0 || !(timeout_s > 0.0))
        return NAF_ERR_ARG;
    XgmiComm* c = new XgmiComm();
    c->rank = rank;
    c->world = world;
    c->n = n_floats;
    c->n_pad = (n_floats + XG_CHUNK - 1) / XG_CHUNK * XG_CHUNK;
    c->data_off = ((size_t)world * XG_FLAG_STRIDE + 4095) / 4096 * 4096;
    c->slab_bytes = c->data_off + (size_t)2 * world * c->n_pad * sizeof(float);
    c->timeout_ticks = (long long)(timeout_s * (double)XG_TICKS_PER_S);
    for (int p = 0; p < NAF_XGMI_MAX_WORLD; ++p) {
        c->peers.base[p] = nullptr;
        c->opened[p] = false;
    }
    void* p = nullptr;
    c->mem_kind = 2;
    hipError_t e = hipExtMallocWithFlags(&p, c->slab_bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c->mem_kind = 1;
        e = hipExtMallocWithFlags(&p, c->slab_bytes, hipDeviceMallocFinegrained);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        delete c;
        return (int)e;
    }
    c->local = (char*)p;
    c->peers.base[rank] = c->local;
    e = hipMalloc((void**)&c->ctrl, 64);
    if (e == hipSuccess) e = hipMemset(c->ctrl, 0, 64);
    c->host_timeouts = nullptr;
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->host_timeouts, 64, hipHostMallocMapped);
    if (e == hipSuccess) *c->host_timeouts = 0;
    if (e == hipSuccess) e = hipMemset(c->local, 0, c->slab_bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void)hipFree(c->local);
        if (c->ctrl) (void)hipFree(c->ctrl);
        if (c->host_timeouts) (void)hipHostFree(c->host_timeouts);
        delete c;
        return (int)e;
    }
    *handle = c;
    return NAF_OK;
}

extern "C" int naf_xgmi_set_timeout(void* handle, double timeout_s) {
    if (!handle || !(timeout_s > 0.0)) return NAF_ERR_ARG;
    xg_comm(handle)->timeout_ticks = (long long)(timeout_s * (double)XG_TICKS_PER_S);
    return NAF_OK;
}

extern "C" int naf_xgmi_mem_kind(void* handle) { return handle ? xg_comm(handle)->mem_kind : NAF_ERR_STATE; }

extern "C" int naf_xgmi_export(void* handle, void* out_handle_bytes) {
    if (!handle || !out_handle_bytes) return NAF_ERR_ARG;
    static_assert(sizeof(hipIpcMemHandle_t) == NAF_XGMI_HANDLE_BYTES, "ipc handle size");
    hipIpcMemHandle_t h;
    hipError_t e = hipIpcGetMemHandle(&h, xg_comm(handle)->local);
    if (e != hipSuccess) return (int)e;
    memcpy(out_handle_bytes, &h, sizeof(h));
    return NAF_OK;
}

extern "C" int naf_xgmi_connect(void* handle, const void* all_handle_bytes, const int* peer_devices) {
    if (!handle || !all_handle_bytes) return NAF_ERR_ARG;
    XgmiComm* c = xg_comm(handle);
    // the peers' devices (peer_devices[rank], W entries; NULL: every other device of the node) must be reachable before
    // their slabs are mapped (errors such as "already enabled" are not failures; hipIpcOpenMemHandle decides)
    int cur = 0, ndev = 0;
    if (hipGetDevice(&cur) == hipSuccess && hipGetDeviceCount(&ndev) == hipSuccess) {
        for (int k = 0; k < (peer_devices ? c->world : ndev); ++k) {
            const int d = peer_devices ? peer_devices[k] : k;
            int can = 0;
            if (d >= 0 && d < ndev && d != cur && hipDeviceCanAccessPeer(&can, cur, d) == hipSuccess && can)
                (void)hipDeviceEnablePeerAccess(d, 0);
        }
        (void)hipGetLastError();
    }
    for (int p = 0; p < c->world; ++p) {
        if (p == c->rank || c->opened[p]) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, (const char*)all_handle_bytes + (size_t)p * NAF_XGMI_HANDLE_BYTES, sizeof(h));
        void* mapped = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&mapped, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return (int)e;
        c->peers.base[p] = (char*)mapped;
        c->opened[p] = true;
    }
    return NAF_OK;
}

// Communicators of ONE process (a rehearsal of world sizes a one-GPU box cannot host as processes: the pool allows six processes on a
// card, north_star's world is eight): every rank's slab is this process's own memory — no hipIpc, the same kernels, the same
// protocol. all_handles[world]: the communicators of ranks 0 .. world - 1, all created with the same (world, n_floats).
extern "C" int naf_xgmi_connect_local(void* handle, void* const* all_handles) {
    if (!handle || !all_handles) return NAF_ERR_ARG;
    XgmiComm* c = xg_comm(handle);
    for (int p = 0; p < c->world; ++p) {
        const XgmiComm* o = xg_comm(all_handles[p]);
        if (!o || o->world != c->world || o->rank != p || o->n != c->n) return NAF_ERR_ARG;
        if (p == c->rank) continue;
        c->peers.base[p] = o->local;
        c->opened[p] = false;              // (nothing to close: not a mapping)
    }
    return NAF_OK;
}

static void xg_fill_desc(const XgmiComm* c, naf_xgmi_push_t* d) {
    for (int p = 0; p < NAF_XGMI_MAX_WORLD; ++p) d->peer_base[p] = c->peers.base[p];
    d->ctrl = c->ctrl;
    d->data_off = c->data_off;
    d->n_pad = c->n_pad;
    d->rank = c->rank;
    d->world = c->world;
    d->timeout_ticks = c->timeout_ticks;
    d->host_timeouts = c->host_timeouts;
}

extern "C" int naf_xgmi_push_desc(void* handle, naf_xgmi_push_t* out) {
    if (!handle || !out) return NAF_ERR_ARG;
    XgmiComm* c = xg_comm(handle);
    for (int p = 0; p < c->world; ++p)
        if (!c->peers.base[p]) return NAF_ERR_STATE;
    xg_fill_desc(c, out);
    return NAF_OK;
}

__global__ __launch_bounds__(XG_THREADS) void xgmi_push_early_kernel(naf_xgmi_push_t d, const float* __restrict__ grad,
                                                                     size_t lo, size_t hi) {
    xg_push_range<NAF_XGMI_MAX_WORLD>(d, grad, lo, hi, blockIdx.x, XG_THREADS, threadIdx.x);
}

extern "C" int naf_xgmi_push_early(void* handle, const float* grad_in, size_t lo, size_t hi, void* stream) {
    if (!handle || !grad_in) return NAF_ERR_ARG;
    XgmiComm* c = xg_comm(handle);
    if ((lo & 3) || (hi & 3) || lo >= hi || hi > c->n || ((uintptr_t)grad_in & 15)) return NAF_ERR_ARG;
    naf_xgmi_push_t d;
    int rc = naf_xgmi_push_desc(handle, &d);
    if (rc != NAF_OK) return rc;
    const unsigned blocks = (unsigned)((hi - lo + XG_CHUNK - 1) / XG_CHUNK);
    xgmi_push_early_kernel<<<blocks, XG_THREADS, 0, (hipStream_t)stream>>>(d, grad_in, lo, hi);
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}

extern "C" int naf_xgmi_allreduce_sum(void* handle, const float* grad_in, float* grad_out, float* sumsq_partials,
                                      int32_t* step_dev, void* stream) {
    if (!handle) return NAF_ERR_ARG;
    return naf_xgmi_allreduce_sum_from(handle, grad_in, grad_out, sumsq_partials, step_dev, xg_comm(handle)->n, stream);
}

extern "C" int naf_xgmi_allreduce_sum_from(void* handle, const float* grad_in, float* grad_out, float* sumsq_partials,
                                           int32_t* step_dev, size_t pushed_lo, void* stream) {
    return naf_xgmi_allreduce_sum_from2(handle, grad_in, grad_out, sumsq_partials, step_dev, pushed_lo, 0, 0, stream);
}

extern "C" int naf_xgmi_allreduce_sum_from2(void* handle, const float* grad_in, float* grad_out, float* sumsq_partials,
                                            int32_t* step_dev, size_t pushed_lo, size_t skip_lo, size_t skip_hi, void* stream) {
    if (!handle || !grad_in || !grad_out) return NAF_ERR_ARG;
    XgmiComm* c = xg_comm(handle);
    if ((pushed_lo & 3) || pushed_lo > c->n || (skip_lo & 3) || (skip_hi & 3) || skip_lo > skip_hi || skip_hi > c->n) return NAF_ERR_ARG;
    if ((((uintptr_t)grad_in | (uintptr_t)grad_out) & 15) != 0) return NAF_ERR_ARG;
    for (int p = 0; p < c->world; ++p)
        if (!c->peers.base[p]) return NAF_ERR_STATE;           // naf_xgmi_connect has not mapped every peer
    const unsigned chunks = (unsigned)(c->n_pad / XG_CHUNK);
#define XG_REDUCE(W)                                                                                                 \
    case W:                                                                                                          \
        xgmi_allreduce_kernel<W><<<chunks, XG_THREADS, 0, (hipStream_t)stream>>>(                                    \
            c->peers, grad_in, grad_out, c->n, c->n_pad, c->data_off, c->rank, c->ctrl, sumsq_partials, step_dev,    \
            c->timeout_ticks, pushed_lo, skip_lo, skip_hi, c->host_timeouts);                                                          \
        break;
    switch (c->world) {
        XG_REDUCE(2) XG_REDUCE(3) XG_REDUCE(4) XG_REDUCE(5) XG_REDUCE(6) XG_REDUCE(7) XG_REDUCE(8)
        default: return NAF_ERR_STATE;
    }
#undef XG_REDUCE
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}

extern "C" int naf_xgmi_status(void* handle, uint64_t* epoch, uint64_t* timeouts) {
    if (!handle) return NAF_ERR_ARG;
    uint64_t host[4];
    hipError_t e = hipMemcpy(host, xg_comm(handle)->ctrl, sizeof(host), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return (int)e;
    if (epoch) *epoch = host[0];
    if (timeouts) *timeouts = host[2];
    return NAF_OK;
}

extern "C" int naf_xgmi_timeouts_nowait(void* handle, uint64_t* timeouts) {
    if (!handle || !timeouts) return NAF_ERR_ARG;
    *timeouts = *(volatile uint64_t*)xg_comm(handle)->host_timeouts;   // pinned host memory the kernel writes: no sync
    return NAF_OK;
}

// Teardown hygiene. Root cause of round 1's "stale slab" (reproduced with tests/xgmi_worker.py, NAF_XGMI_TEST_ORDER=1,0,1,0,
// 4 ranks sharing one GPU: the learner built right after a communicator had been freed took different updates in every
// run; the one-shot exchange itself was bit-identical throughout): the pages of a released slab come back to the next
// allocation while L2 still holds lines of them from the slab's days — written through the peers' hipIpc mappings, read
// uncached by the owner, so no ordinary kernel-boundary invalidate ever dropped them. Closing mappings before freeing, with a
// barrier in between, does not help (tried); time does not help; streaming a buffer through every L2 right after the teardown
// removes the effect (10 of 10 runs against 1 of 8 without). So both halves of the teardown end with a SCRUB: a grid over all
// XCDs reads 64 MiB (twice the 32 MiB of L2 on the chip), and
// system-scope fences bracket it. The raw memory life cycle alone does not reproduce it (benchmarks/probe/ipc_stale_repro.cpp).
__global__ __launch_bounds__(256) void xg_scrub_kernel(const float4* __restrict__ p, size_t n4, float* sink) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 12345.678f) *sink = s;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
}

static void xg_scrub_caches() {
    const long mb = 64;
    void* buf = nullptr;
    const size_t bytes = (size_t)mb << 20;
    if (hipMalloc(&buf, bytes + 256) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    (void)hipMemsetAsync(buf, 0, bytes + 256, 0);
    xg_scrub_kernel<<<2048, 256, 0, 0>>>((const float4*)buf, bytes / 16, (float*)((char*)buf + bytes));
    (void)hipDeviceSynchronize();
    (void)hipFree(buf);
}

extern "C" int naf_xgmi_disconnect(void* handle) {
    if (!handle) return NAF_ERR_ARG;
    XgmiComm* c = xg_comm(handle);
    hipError_t e = hipDeviceSynchronize();
    bool any = false;
    for (int p = 0; p < c->world; ++p) any = any || c->opened[p];
    if (any) xg_scrub_caches();            // before the peers' slabs are unmapped: this device's L2s forget them
    for (int p = 0; p < c->world; ++p)
        if (c->opened[p]) {
            (void)hipIpcCloseMemHandle(c->peers.base[p]);
            c->opened[p] = false;
            c->peers.base[p] = nullptr;
        }
    return e == hipSuccess ? NAF_OK : (int)e;
}

extern "C" int naf_xgmi_destroy(void* handle) {
    if (!handle) return NAF_ERR_ARG;
    XgmiComm* c = xg_comm(handle);
    (void)naf_xgmi_disconnect(handle);
    // The own slab is PARKED until the process ends (<= 5.3 MB per communicator; round 1's behaviour, 30 of 30 runs
    // bit-identical): returning it to the allocator is safe only behind the cache scrub above, whose evidence is empirical
    // (10 of 10 runs clean) — NAF_XGMI_FREE_SLAB=1 opts into scrub + free.
    const char* fr = getenv("NAF_XGMI_FREE_SLAB");
    if (fr && fr[0] == '1') {
        xg_scrub_caches();
        (void)hipFree(c->local);
    }
    (void)hipFree(c->ctrl);
    (void)hipHostFree(c->host_timeouts);
    delete c;
    return NAF_OK;
}
