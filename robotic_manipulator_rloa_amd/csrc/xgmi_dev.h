// Device-side pieces of the one-shot peer-memory all-reduce (xgmi_reduce.hip) that other kernels embed: the layer-1
// backward kernel pushes the already final part of the gradient from extra workgroups of its own launch, so that the
// wire time of the exchange runs under it instead of behind it.
#pragma once
#include "common.h"
#include "../../include/naf_hip.h"

typedef float xg_f4 __attribute__((ext_vector_type(4)));

__device__ static inline float* xg_slot(char* base, size_t data_off, size_t n_pad, int world, uint64_t epoch, int sender) {
    return (float*)(base + data_off) + ((size_t)(epoch & 1) * world + sender) * n_pad;
}

// Push grad[lo, hi) (multiples of 4) into this rank's slot on every peer for the NEXT epoch (ctrl[0] + 1; ctrl[0] moves
// only in the all-reduce launch that follows). Workgroup `wg` of `nthreads` threads covers nthreads * 4 floats. Branch-free
// stores (the own slab gets a copy nobody reads), then a system-scope release: when the calling launch has finished,
// every byte has reached its peer; the flags are raised later, by naf_xgmi_allreduce_sum(_from).
template <int WORLD_MAX>
__device__ static inline void xg_push_range(const naf_xgmi_push_t& d, const float* __restrict__ grad, size_t lo, size_t hi,
                                            int wg, int nthreads, int tid) {
    const uint64_t e = d.ctrl[0] + 1;
    const size_t i = lo + ((size_t)wg * nthreads + tid) * 4;
    if (i < hi) {
        const xg_f4 v = *(const xg_f4*)(grad + i);
#pragma unroll
        for (int p = 0; p < WORLD_MAX; ++p)
            if (p < d.world) *(xg_f4*)(xg_slot((char*)d.peer_base[p], d.data_off, d.n_pad, d.world, e, d.rank) + i) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
}
