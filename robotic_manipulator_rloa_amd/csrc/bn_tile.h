// Column-ownership tile shared by bn_relu.hip and fused_layers.hip.
#pragma once
#include "common.h"

// Which column tile a workgroup owns. Workgroups are dealt round-robin over the 8 XCDs (XCD = blockIdx.x % 8; for the
// two-net grids too, their x extent being a multiple of 8), and an 8-column f32 tile is a quarter of a 128-B line: with
// tile = blockIdx.x the four tiles sharing every line of the activation matrices sit on four different XCDs, and each
// of those L2s pulls the whole line over the fabric — for data the previous kernel has only just written (the dominant
// cost of an update, round 2's benchmarks/chain_probe.py). Dealing CONTIGUOUS runs of tiles to an XCD lets one L2 fetch a line
// once for all its tiles. Placement is speed only: results do not depend on it.
__device__ static inline int naf_xcd_tile(int b, int n) { return (n & 7) == 0 ? (b & 7) * (n >> 3) + (b >> 3) : b; }

// tile shape: TX feature columns x TY row phases per workgroup (TX*TY threads, TX <= 64, TX*TY % 64 == 0).
// Column sums: lanes of a wave that share a column (lane = phase*TX + tx) fold by xor shuffles, the TX*TY/64 wave
// results meet in LDS. Fixed order -> bitwise reproducible.
// FRESH = the LDS array(s) passed in have not been touched by this workgroup before: the leading barrier ("the previous
// use is over") is skipped. Every reduction of a kernel gets its own array for exactly that reason — these kernels are
// chains of short barrier-separated phases, and a barrier costs about as much as the phase it guards.
template <int BN_TX, int BN_TY, bool FRESH = false>
__device__ static inline void bn_col_reduce2(float pa, float pb, float (*red)[BN_TX + 1], float (*red2)[BN_TX + 1], int tx,
                                             int ty, float* oa, float* ob) {
    constexpr int NW = BN_TX * BN_TY / 64;
#pragma unroll
    for (int o = BN_TX; o < 64; o <<= 1) {
        pa += __shfl_xor(pa, o);
        pb += __shfl_xor(pb, o);
    }
    const int tid = ty * BN_TX + tx;
    if (!FRESH) __syncthreads();  // previous use of red/red2 is over
    if ((tid & 63) < BN_TX) {
        red[tid >> 6][tx] = pa;
        red2[tid >> 6][tx] = pb;
    }
    __syncthreads();
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        sa += red[k][tx];
        sb += red2[k][tx];
    }
    *oa = sa;
    *ob = sb;
}

template <int BN_TX, int BN_TY, bool FRESH = false>
__device__ static inline float bn_col_reduce(float part, float (*red)[BN_TX + 1], int tx, int ty) {
    constexpr int NW = BN_TX * BN_TY / 64;
#pragma unroll
    for (int o = BN_TX; o < 64; o <<= 1) part += __shfl_xor(part, o);
    const int tid = ty * BN_TX + tx;
    if (!FRESH) __syncthreads();
    if ((tid & 63) < BN_TX) red[tid >> 6][tx] = part;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NW; ++k) s += red[k][tx];
    return s;
}


// Sum of one float per thread over a whole workgroup of NT threads (NT % 64 == 0), result valid in thread 0.
// Fixed order (xor shuffles inside a wave, then the wave results in index order) -> bitwise reproducible.
template <int NT, bool FRESH = false>
__device__ static inline float block_sum_to_thread0(float v, float* sh /* >= NT/64 floats */, int tid) {
    v = naf_sum64(v);
    if (!FRESH) __syncthreads();
    if ((tid & 63) == 0) sh[tid >> 6] = v;
    __syncthreads();
    float s = 0.f;
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < NT / 64; ++k) s += sh[k];
    }
    return s;
}
