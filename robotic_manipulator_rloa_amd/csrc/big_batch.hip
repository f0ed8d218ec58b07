// learn() at LARGE batches (B > 512: BASELINE configs[3] B = 1024, configs[4] B = 2048) for gfx950.
//
// The kernels of fused_layers.hip own whole feature columns (8 columns x ALL B rows per workgroup) so that BatchNorm's
// batch statistics never leave the workgroup: 32-64 workgroups, ceil(B/64) rows per thread in registers — right for
// B <= 512 where an update is launch-latency bound, wrong beyond it (a quarter of the chip busy, register spills).
// Here the batch is split into 64-row blocks that spread over the whole chip, and BatchNorm becomes TWO-STAGE:
//   stage 1  the kernel that produces a pre-activation tile also writes, per 64-row block and column, the block's
//            (sum, M2 = sum of squared deviations from the BLOCK mean)           -> partials[net][block][column]
//   stage 2  every consumer folds the B/64 partials of the columns it touches in block order (Chan's pairwise formula,
//            fixed order => bitwise reproducible, no atomics) in its prologue, then normalises on the fly.
// The backward statistics (sum dy, sum dy*xhat) go the same way. Replaces, for both networks in one launch each,
// naf_neural_network.py:76-87 (forward), its autograd, and the BatchNorm1d training-mode statistics of torch.
//
// Chain of one update (Learner.learn_rows, fuse = "bb"):
//   bb_layer1_stats -> bb_layer1_apply -> bb_linear_stats (f32 MFMA GEMM 2, statistics in the epilogue) ->
//   bb_bn_relu_heads_partial (BN2 + ReLU + heads GEMM split over 4 column slices) -> naf_head_kernel<.., 4 slabs> ->
//   bb_heads_bwd_stage1 (dA2 = dH Wh on the fly, ReLU mask, backward partials) -> bb_bn_bwd_stage2 (dZ2) ->
//   gemm_bundle (dWh, dW2, dA1) -> bb_layer1_bwd (column-owning, streaming) -> grad norm -> Adam + Polyak
#include <string.h>
#include <stdlib.h>
#include "bn_tile.h"
#include "head_body.h"
#include "adam_body.h"
#include "moments_body.h"
#include "bn2bwd_fold.h"   // gemm_bn2bwd_poll_record: the readers' side of a self-validating 16-byte record
#include "xgmi_dev.h"
#include "../../include/naf_hip.h"

#define BB_ROWS NAF_BB_ROWS      // rows per statistics block
#define BB_COLS 64               // feature columns per workgroup (row-split kernels)
#define BB_THREADS 256
#define BB_MAX_K4 8              // layer 1: K <= 32

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- stage 2 of the batch statistics: fold the NB block partials of one column, in block order -----------------
// NB <= BB_MAX_NB (B <= 2048): every partial is requested up front (indices clamped, adds predicated), ONE round trip
// instead of NB / 8 dependent ones — this runs in the prologue of every consumer.
#define BB_MAX_NB 32
NAF_TL_DECL(g_tl_bb);
NAF_TL_READER(naf_tl_read_bb, g_tl_bb)
#define BB_L1_TL(slot, is_first, is_last) NAF_TL_FL(g_tl_bb, NAF_TL_BB_LAYER1, slot, is_first, is_last)
#define BB_L1_TL_T(slot, is_first, is_last, thread) NAF_TL_FL_T(g_tl_bb, NAF_TL_BB_LAYER1, slot, is_first, is_last, thread)
#include "layer1_body.h"     // bb_place_rows, bb_l1_tile, bb_layer1_impl
template <bool FULL>
__device__ static inline void bb_fold_stats(const float2* __restrict__ p, int H, int NB, int B, int col, float* mean,
                                            float* var) {
    float2 v[BB_MAX_NB];
#pragma unroll
    for (int rb = 0; rb < BB_MAX_NB; ++rb) v[rb] = p[(int64_t)(rb < NB ? rb : 0) * H + col];
    float S = 0.f;
#pragma unroll
    for (int rb = 0; rb < BB_MAX_NB; ++rb) S += rb < NB ? v[rb].x : 0.f;
    const float m = S / (float)B;
    float M2 = 0.f;
    // (the last block holds B - 64 (NB - 1) rows: 64 where the batch is whole blocks (FULL) — then this is the arithmetic it always was)
    const float n_last = FULL ? (float)BB_ROWS : (float)(B - BB_ROWS * (NB - 1)), inv_last = FULL ? 1.0f / BB_ROWS : 1.0f / n_last;
#pragma unroll
    for (int rb = 0; rb < BB_MAX_NB; ++rb) {
        const bool lastb = !FULL && rb == NB - 1;
        const float d = v[rb].x * (lastb ? inv_last : 1.0f / BB_ROWS) - m;
        M2 += rb < NB ? v[rb].y + (lastb ? n_last : (float)BB_ROWS) * d * d : 0.f;
    }
    *mean = m;
    *var = M2 / (float)B;      // biased: what normalises
}

// The same fold for kernels whose prologue is bound by VALU issue (8 waves = two per SIMD, each running this once per
// thread): one instance per size class behind a UNIFORM branch, so a batch of 256 (NB = 4) executes 4 block steps, not
// 32 predicated ones, and buffer loads (common.h: naf_buf_*): wave-uniform resource and block offset, one lane offset.
// p = the partials; the thread's (block 0) element at byte wave_off + lane_off. Same arithmetic and order as above.
template <int NMAX, bool FULL>
__device__ __forceinline__ static void bb_fold_stats_n(__amdgpu_buffer_rsrc_t p, unsigned lane_off, unsigned wave_off, int H, int NB,
                                                       int B, float* mean, float* var) {
    naf_f32x2 v[NMAX];
#pragma unroll
    for (int rb = 0; rb < NMAX; ++rb) v[rb] = naf_buf_f2(p, lane_off, wave_off + (unsigned)(rb < NB ? rb : 0) * (unsigned)H * 8u);
    float S = 0.f;
#pragma unroll
    for (int rb = 0; rb < NMAX; ++rb) S += rb < NB ? v[rb].x : 0.f;
    const float m = S / (float)B;
    float M2 = 0.f;
    // (FULL: whole 64-row blocks — no last-block weights and no division; the same bits as the general form gives there)
    const float n_last = FULL ? (float)BB_ROWS : (float)(B - BB_ROWS * (NB - 1)), inv_last = FULL ? 1.0f / BB_ROWS : 1.0f / n_last;
#pragma unroll
    for (int rb = 0; rb < NMAX; ++rb) {
        const bool lastb = !FULL && rb == NB - 1;
        const float d = v[rb].x * (lastb ? inv_last : 1.0f / BB_ROWS) - m;
        M2 += rb < NB ? v[rb].y + (lastb ? n_last : (float)BB_ROWS) * d * d : 0.f;
    }
    *mean = m;
    *var = M2 / (float)B;
}
template <bool FULL>
__device__ __forceinline__ static void bb_fold_stats_u(__amdgpu_buffer_rsrc_t p, unsigned lane_off, unsigned wave_off, int H, int NB,
                                                       int B, float* mean, float* var) {
    if (NB <= 4) bb_fold_stats_n<4, FULL>(p, lane_off, wave_off, H, NB, B, mean, var);            // B <= 256 (uniform branches)
    else if (NB <= 8) bb_fold_stats_n<8, FULL>(p, lane_off, wave_off, H, NB, B, mean, var);
    else if (NB <= 16) bb_fold_stats_n<16, FULL>(p, lane_off, wave_off, H, NB, B, mean, var);
    else bb_fold_stats_n<BB_MAX_NB, FULL>(p, lane_off, wave_off, H, NB, B, mean, var);
}

// More than BB_MAX_NB blocks (2048 < B <= 4096: up to BB_MAX_NB2): the partials do not fit a thread's registers at once, so they are
// read twice — once for the sum, once (out of L2) for the squares about the mean — 16 at a time. The same sums in the same order
// as above, extended; both the folding workgroups of bb_layer2_head and a thread that folds for itself come here, so the two agree
// bit for bit as they do below 2048.
#define BB_MAX_NB2 64
__device__ static inline void bb_fold_stats_big(const float2* p, int H, int NB, int B, int col, float* mean, float* var) {
    float S = 0.f;
    for (int r0 = 0; r0 < NB; r0 += 16) {
        float2 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = p[(int64_t)(r0 + i < NB ? r0 + i : 0) * H + col];
#pragma unroll
        for (int i = 0; i < 16; ++i) S += r0 + i < NB ? v[i].x : 0.f;
    }
    const float m = S / (float)B;
    const float n_last = (float)(B - BB_ROWS * (NB - 1)), inv_last = 1.0f / n_last;
    float M2 = 0.f;
    for (int r0 = 0; r0 < NB; r0 += 16) {
        float2 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = p[(int64_t)(r0 + i < NB ? r0 + i : 0) * H + col];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int rb = r0 + i;
            const bool lastb = rb == NB - 1;
            const float d = v[i].x * (lastb ? inv_last : 1.0f / BB_ROWS) - m;
            M2 += rb < NB ? v[i].y + (lastb ? n_last : (float)BB_ROWS) * d * d : 0.f;
        }
    }
    *mean = m;
    *var = M2 / (float)B;
}

// ------------------------------------------------------------------------------------------------------------
// The moments of layer 1's inputs for ALL minibatches of a chunk (grid = minibatches x nets), one launch behind the gather; the
// arithmetic is csrc/moments_body.h (shared with the per-timestep launch of csrc/step_path.hip).
// ------------------------------------------------------------------------------------------------------------
template <int K4>
__global__ __launch_bounds__(BM_THREADS) void bb_moments_kernel(const float* __restrict__ x, int64_t batch_stride,
                                                                int64_t x_net_stride, int ldx, float* __restrict__ mom,
                                                                int B) {
    constexpr int KP = 4 * K4, REC = KP + KP * KP;
    __shared__ __attribute__((aligned(16))) BmShared S;
    const float* xb = x + blockIdx.x * batch_stride + blockIdx.y * x_net_stride;
    float* out = mom + ((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * REC;
    bb_moments_body<K4>([&](int row, int q) { return ((const float4*)(xb + (int64_t)row * ldx))[q]; }, S, threadIdx.x, out, B);
}

// finish: TWO columns per workgroup, two waves per column, lane = (k of 32, half): the column's p_slabs blocks are dealt in
// four contiguous runs to the (wave, half) quarters — at most 16 loads per lane, all in flight at once — and meet in a fixed
// order (quarters 0 + 1 by shuffle, 2 + 3 likewise, the pairs through LDS). The column's block sums are loaded ONE block per
// lane and folded by an xor-shuffle tree. (The first version had 8 columns x 32 k lanes per workgroup, every lane walking all
// 64 blocks of three arrays in two rounds and recomputing w_c C with 32 shuffles: 6.6 / 8.4 us at B = 1024 / 2048, most of it
// waiting — benchmarks/kernel_timeline.py.) w_c C comes from the forward pass (wc, [H][KP]).
// Behind the finish blocks, workgroups that add the split-K slabs of the bundle's weight gradients (dW2, dWh) in slab order,
// 1024 floats each — every gradient element leaves this launch final, with its sum-of-squares partial.
#define BF_COLS 2
#define BB_MAX_NB1 64            // 32-row blocks of the dA1 product held at once (B <= 2048); up to twice as many behind a uniform branch
struct BbSlabSeg {
    const float* src;       // slab 0; slab s at src + s * stride
    float* dst;
    int64_t stride;
    int n, n_slabs, block0; // n floats (multiple of 4); first reduce block of this segment
};
struct BbSlabs {
    BbSlabSeg seg[2];
    int n_seg, n_finish_blocks;
};
#define BB_MAX_SLABS 8
struct FinishArgs {
    const float* p_slabs;
    int KP, K;
    const float2* partials1;
    int NB1;
    const float* dz2_col_partials;
    int NB;
    const float *mom, *wc, *gamma, *save_invstd;
    float *d_W, *d_gamma, *d_beta, *d_bias, *d_bias2;
    const float *d_gamma2, *d_beta2;
    float* sumsq_partials;       // one float per workgroup
    int32_t* step_dev;
    int B, H;
    BbSlabs slabs;
    int* fold_flag;
    int n_blocks;                // finish blocks + slab-reduce blocks
    // data parallel over peer memory (csrc/xgmi_reduce.hip): the slab-reduce workgroups also store what they finalise — the two
    // weight-gradient segments W2 and Wh, 91 % of the flat gradient — into this rank's slot on every peer, so that the
    // all-reduce launch behind this one has only the layer-1 / BatchNorm segments left to send before it raises its flags
    naf_xgmi_push_t push;
    const float* grad_base;      // the flat gradient the segments' dst pointers lie in; NULL: no push
    // merge != 0: the WHOLE exchange happens in this launch (bb_finish_exchange): what is not a slab segment — the layer-1 and
    // BatchNorm ranges [r_lo[i], r_hi[i]) of the flat gradient, 27 KB — is pushed by the launch's last-arriving workgroup, which
    // then raises the epoch flags; every workgroup that holds gradient elements waits for the peers' flags and leaves its elements
    // as the rank-ordered sum, with the sum-of-squares partials of the REDUCED gradient — no all-reduce launch behind this one
    int merge;
    size_t r_lo[2], r_hi[2];
};
// a float4 of this rank's gradient into its slot on a peer, inside the launch that also raises the flags (merge)
#ifndef BB_PUSH_MODE
// 0: plain stores + a system-scope release fence per workgroup in front of its ticket (the protocol of csrc/xgmi_reduce.hip);
// 1: sc1 stores, 2: sc0 sc1 stores (written through, no fence). Shared-GPU rehearsal, W = 2, us per update (benchmarks/
// ab_push_mode.sh; the peers' slabs are LOCAL memory there, cached by the writer's L2 unless written through): 61.8 | 94.8 | 95.0 —
// against 48.1 with the all-reduce as a launch of its own (NAF_DP_EXCHANGE=oneshot), which is why the start-up autotune (Learner.autotune_exchange) ranks it first until a
// multi-GPU box has measured both over real xGMI (where a peer's memory is not cached locally and the ranking may differ).
#define BB_PUSH_MODE 0
#endif
__device__ __forceinline__ static void bb_push_st(float* sl, xg_f4 v) {
#if BB_PUSH_MODE == 0
    *(xg_f4*)sl = v;
#elif BB_PUSH_MODE == 1
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(naf_u32x4, v), naf_buf(sl, 16), 0, 0, 16);
#else
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(naf_u32x4, v), naf_buf(sl, 16), 0, 0, 17);
#endif
}

// a gradient element of the column workgroups: written THROUGH (sc1) when the launch's last-arriving workgroup will read it
__device__ __forceinline__ static void bb_st(float* p, float v, int through) {
    if (through) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), naf_buf(p, 4), 0, 0, 16);
    else *p = v;
}

// The gradient exchange of data parallel INSIDE the finish launch (round 4; until then a launch of its own behind it,
// csrc/xgmi_reduce.hip, whose protocol — slots by epoch parity, one epoch flag per peer, wall-clock bounded waits, a poisoned norm
// partial on a time-out — this is, dealt over the workgroups that already hold the gradient):
//   every workgroup: its pushes into the peers' slabs released at system scope and acknowledged (BB_PUSH_MODE), its own elements
//     of the small ranges written through (sc1) -> a ticket.
//   the LAST to arrive: reads the ranges no slab workgroup covers (layer 1 and the BatchNorm vectors, sc1 loads: the column
//     workgroups wrote them through), pushes them, waits for the acknowledgements, raises this rank's flag on every peer, advances
//     the epoch.
//   every workgroup that holds gradient elements (slab workgroups; the last one for the small ranges): waits for the W - 1 peers'
//     flags, then leaves grad = the sum over ranks in RANK ORDER (its own contribution from registers), and its sum-of-squares partial.
// No workgroup waits for another workgroup of its own launch — the ticket is not polled — only for the peers' flags, which depend
// on nothing but the peers' own launches: no circular wait whatever the residency of the grid.
#define BB_EX_SMALL 8            // float4 per thread of the last workgroup: the small ranges are at most 8 KB floats
__device__ static inline void bb_finish_exchange(const FinishArgs& F, int block, int tid, float4 ex_a, size_t ex_o, bool ex_on, float* sQ) {
    __shared__ int s_last, s_timed;
    const naf_xgmi_push_t& X = F.push;
    const int W = X.world, rank = X.rank;
    const uint64_t e = X.ctrl[0] + 1;          // (read before this workgroup's ticket: only the last arrival writes ctrl[0])
    if (tid == 0) s_timed = 0;
#if BB_PUSH_MODE == 0
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_last = atomicAdd((unsigned long long*)&X.ctrl[1], 1ull) == (unsigned long long)F.n_blocks - 1;
    __syncthreads();
    const bool last = s_last != 0;
    const bool slab_wg = block >= F.slabs.n_finish_blocks;
    float* grad = (float*)F.grad_base;
    xg_f4 mine[BB_EX_SMALL];
    size_t moff[BB_EX_SMALL];
    const size_t n0 = (F.r_hi[0] - F.r_lo[0]) >> 2, n1 = (F.r_hi[1] - F.r_lo[1]) >> 2;
    if (last) {
        const __amdgpu_buffer_rsrc_t gr = naf_buf(grad);
#pragma unroll
        for (int q = 0; q < BB_EX_SMALL; ++q) {
            const size_t j = (size_t)tid + (size_t)BB_THREADS * q;
            const bool on = j < n0 + n1;
            moff[q] = !on ? 0 : (j < n0 ? F.r_lo[0] + 4 * j : F.r_lo[1] + 4 * (j - n0));
            mine[q] = naf_buf_f4_sc1(gr, (unsigned)(moff[q] * 4), 0);
        }
#pragma unroll
        for (int q = 0; q < BB_EX_SMALL; ++q) {
            const size_t j = (size_t)tid + (size_t)BB_THREADS * q;
            if (j < n0 + n1) {
#pragma unroll
                for (int p = 0; p < NAF_XGMI_MAX_WORLD; ++p)
                    if (p < W)
                        bb_push_st(xg_slot((char*)X.peer_base[p], X.data_off, X.n_pad, W, e, rank) + moff[q], mine[q]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid < W && tid != rank) {
            uint64_t* flag = (uint64_t*)((char*)X.peer_base[tid] + (size_t)rank * 128);
            __hip_atomic_store(flag, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (tid == 0) {
            X.ctrl[1] = 0;
            X.ctrl[0] = e;
        }
    }
    if (!slab_wg && !last) {                   // a column workgroup that is not the last: nothing of the sum is its to form
        if (tid == 0 && F.sumsq_partials) F.sumsq_partials[block] = 0.f;
        return;
    }
    if (tid < W && tid != rank) {
        const uint64_t* flag = (const uint64_t*)((char*)X.peer_base[rank] + (size_t)tid * 128);
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < e) {
            if (wall_clock64() - t0 > X.timeout_ticks) {         // the exit every wave reaches: peer missing or dead
                atomicAdd((unsigned long long*)&X.ctrl[2], 1ull);
                if (X.host_timeouts) __hip_atomic_fetch_add(X.host_timeouts, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                s_timed = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");            // system scope: drop anything this CU still holds of the slots
    const bool timed = s_timed != 0;
    float ss = 0.f;
    // the rank-ordered sum of one float4: every peer's contribution in flight together, the own one from registers
    auto reduce4 = [&](size_t off, xg_f4 own) {
        const float* slot0 = xg_slot((char*)X.peer_base[rank], X.data_off, X.n_pad, W, e, 0) + off;
        xg_f4 v[NAF_XGMI_MAX_WORLD];
#pragma unroll
        for (int s_ = 0; s_ < NAF_XGMI_MAX_WORLD; ++s_) v[s_] = *(const xg_f4*)(slot0 + (size_t)(s_ < W ? s_ : 0) * X.n_pad);
        xg_f4 acc = rank == 0 ? own : v[0];
#pragma unroll
        for (int s_ = 1; s_ < NAF_XGMI_MAX_WORLD; ++s_)
            if (s_ < W) acc = acc + (s_ == rank ? own : v[s_]);
        *(xg_f4*)(grad + off) = acc;
        ss += acc.x * acc.x + acc.y * acc.y + acc.z * acc.z + acc.w * acc.w;
    };
    if (slab_wg && ex_on) reduce4(ex_o, (xg_f4){ex_a.x, ex_a.y, ex_a.z, ex_a.w});
    const float tot = block_sum_to_thread0<BB_THREADS, true>(ss, sQ, tid);
    // a contribution is missing: the partial is POISONED (csrc/xgmi_reduce.hip: naf_adam_polyak_fused skips the whole update on a
    // negative norm; partials of workgroups that did not time out stay below 1e30, so the sum is -inf, never inf - inf)
    if (tid == 0 && F.sumsq_partials) F.sumsq_partials[block] = timed ? -__builtin_huge_valf() : (tot > 1e30f ? 1e30f : tot);
    if (last) {
        // the small ranges' sum of squares goes to an entry of ITS OWN behind the workgroups' (index n_blocks): WHICH workgroup
        // arrives last differs from rank to rank and from launch to launch, and the optimizer adds the partials in index order —
        // folded into the last workgroup's own entry, the norm rounded differently on different ranks and the replicas drifted apart
        ss = 0.f;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < BB_EX_SMALL; ++q)
            if ((size_t)tid + (size_t)BB_THREADS * q < n0 + n1) reduce4(moff[q], mine[q]);
        const float tot2 = block_sum_to_thread0<BB_THREADS, true>(ss, sQ, tid);
        if (tid == 0) {
            if (F.sumsq_partials) F.sumsq_partials[F.n_blocks] = timed ? -__builtin_huge_valf() : (tot2 > 1e30f ? 1e30f : tot2);
            if (F.step_dev && !timed) *F.step_dev += 1;
        }
    }
}

// One workgroup (BB_THREADS threads) of the finish work. MERGE: the gradient exchange of data parallel inside this launch
// (F.merge; a kernel of its own so that the launch of one GPU — and of the separate all-reduce — carries none of its code: with a
// run-time flag the single-GPU update was 0.25 us slower, A/B on one box).
template <bool MERGE, bool BIG>
__device__ static inline void bb_finish_block(const FinishArgs& F, int block, int tid, float* sQ, float (*sP)[2][32]) {
    // (no FP contraction: the gradient of a build rounds the same way whatever the optimizer makes of this body)
#pragma clang fp contract(off)
    float sq = 0.f;
    // every field of the (by-value) argument either branch needs, requested NOW in one batch of scalar loads: fetched where they are
    // first used they came in three dependent batches in front of the first vector load (~0.25 us each: the scalar cache is cold at
    // the start of a launch, and this launch is nothing but one round trip to fresh data and a store)
    asm volatile("" ::"s"(F.p_slabs), "s"(F.KP), "s"(F.K), "s"(F.partials1), "s"(F.NB1), "s"(F.dz2_col_partials), "s"(F.NB), "s"(F.mom),
                 "s"(F.wc), "s"(F.gamma), "s"(F.save_invstd), "s"(F.d_W), "s"(F.d_gamma), "s"(F.d_beta), "s"(F.d_bias), "s"(F.d_bias2),
                 "s"(F.d_gamma2), "s"(F.d_beta2), "s"(F.sumsq_partials), "s"(F.step_dev), "s"(F.B), "s"(F.H), "s"(F.fold_flag),
                 "s"(F.slabs.n_finish_blocks), "s"(F.slabs.n_seg), "s"(F.grad_base));
    asm volatile("" ::"s"(F.slabs.seg[0].src), "s"(F.slabs.seg[0].dst), "s"(F.slabs.seg[0].stride), "s"(F.slabs.seg[0].n),
                 "s"(F.slabs.seg[0].n_slabs), "s"(F.slabs.seg[0].block0), "s"(F.slabs.seg[1].src), "s"(F.slabs.seg[1].dst),
                 "s"(F.slabs.seg[1].stride), "s"(F.slabs.seg[1].n), "s"(F.slabs.seg[1].n_slabs), "s"(F.slabs.seg[1].block0));
    // the launch number the bundle's folded constants are tagged with (naf_gemm_bn2bwd_t.epoch): a new one for the next update
    if (F.fold_flag && block == 0 && tid == 64) *F.fold_flag += 1;
    float4 ex_a = make_float4(0.f, 0.f, 0.f, 0.f);          // merge: this thread's float4 of a slab segment and its flat offset
    size_t ex_o = 0;
    bool ex_on = false;
    if (block >= F.slabs.n_finish_blocks) {
        const int rbk = block - F.slabs.n_finish_blocks;
        const BbSlabSeg& sg = (F.slabs.n_seg > 1 && rbk >= F.slabs.seg[1].block0) ? F.slabs.seg[1] : F.slabs.seg[0];
        const int i = (rbk - sg.block0) * (BB_THREADS * 4) + tid * 4;
        if (i < sg.n) {
            float4 v[BB_MAX_SLABS];
#pragma unroll
            for (int s_ = 0; s_ < BB_MAX_SLABS; ++s_)
                v[s_] = *(const float4*)(sg.src + (int64_t)(s_ < sg.n_slabs ? s_ : 0) * sg.stride + i);
            float4 a = v[0];
#pragma unroll
            for (int s_ = 1; s_ < BB_MAX_SLABS; ++s_)
                if (s_ < sg.n_slabs) { a.x += v[s_].x; a.y += v[s_].y; a.z += v[s_].z; a.w += v[s_].w; }
            *(float4*)(sg.dst + i) = a;
            sq = a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
            if (F.grad_base) {
                // branch-free over the ranks (the own slab gets a copy nobody reads), as xg_push_range: with `if (p != rank)`
                // around each store the compiler waits for every one of them in turn
                const uint64_t e = F.push.ctrl[0] + 1;           // the epoch of the all-reduce launch that follows
                const size_t o = (size_t)(sg.dst - F.grad_base) + (size_t)i;
                const xg_f4 v = {a.x, a.y, a.z, a.w};
                ex_a = a; ex_o = o; ex_on = true;
                // (merge: the flags go up inside THIS launch, with no launch boundary in between to write the L2 back — where a
                //  peer's slab is mapped cacheable, as with several ranks on one GPU, the stores must be written through themselves)
#pragma unroll
                for (int p = 0; p < NAF_XGMI_MAX_WORLD; ++p)
                    if (p < F.push.world) {
                        float* sl = xg_slot((char*)F.push.peer_base[p], F.push.data_off, F.push.n_pad, F.push.world, e, F.push.rank) + o;
                        if (MERGE) bb_push_st(sl, v);
                        else *(xg_f4*)sl = v;
                    }
            }
        }
        // Every pushing wave waits until its stores to the peers' (uncached) slabs are acknowledged; no release fence: the flags
        // that tell the peers to read are raised by the NEXT launch of this stream, behind the launch boundary and behind that
        // launch's own system-scope release. (A release fence in each of the 73 pushing workgroups — an L2 write-back each, of
        // lines that have nothing to do with the slabs — cost 7 - 15 us per update in the shared-GPU rehearsal.)
        if (F.grad_base) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        const int lane = tid & 63, wave = tid >> 6;
        const int cl = wave >> 1, wq = wave & 1, k = lane & 31, hf = lane >> 5;
        const int col = block * BF_COLS + cl;
        const bool col_on = col < F.H;
        const int colc = col_on ? col : F.H - 1;
        const int kc = k < F.KP ? k : 0;
        // everything requested up front, branch-free (indices clamped, sums predicated)
        const int Q = (F.NB1 + 3) >> 2, rb0 = (2 * wq + hf) * Q;     // this quarter's run of blocks
        float pv[BB_MAX_NB1 / 4], pw[BB_MAX_NB1 / 4];
#pragma unroll
        for (int i = 0; i < BB_MAX_NB1 / 4; ++i) {
            const int rb = rb0 + i;
            pv[i] = F.p_slabs[((int64_t)((i < Q && rb < F.NB1) ? rb : 0) * F.H + colc) * F.KP + kc];
        }
        const bool big = BIG;                              // 2048 < B <= 4096 (more than 64 dA1 blocks): the second half of each quarter's run
        if (big) {
#pragma unroll
            for (int i = 0; i < BB_MAX_NB1 / 4; ++i) {
                const int j = BB_MAX_NB1 / 4 + i, rb = rb0 + j;
                pw[i] = F.p_slabs[((int64_t)((j < Q && rb < F.NB1) ? rb : 0) * F.H + colc) * F.KP + kc];
            }
        }
        // block sums: wave 0 of the column takes F.partials1 (block = lane), wave 1 the layer-2 bias partials
        float2 av = make_float2(0.f, 0.f);
        float dv = 0.f;
        float2 aw = make_float2(0.f, 0.f);
        if (wq == 0) av = F.partials1[(int64_t)(lane < F.NB1 ? lane : 0) * F.H + colc];
        if (wq == 0 && big) aw = F.partials1[(int64_t)(64 + lane < F.NB1 ? 64 + lane : 0) * F.H + colc];
        else if (F.NB > 0) dv = F.dz2_col_partials[(int64_t)(lane < F.NB ? lane : 0) * F.H + colc];   // (NB = 0: no such array)
        const float invstd = F.save_invstd[colc], gm = F.gamma[colc];
        const float sxk = F.mom[kc], wck = F.wc[(int64_t)colc * F.KP + kc];
        float g2 = 0.f, b2 = 0.f;
        if (F.d_gamma2) {
            g2 = F.d_gamma2[colc];                           // written by bb_bn_bwd_stage2, an earlier launch
            b2 = F.d_beta2[colc];
        }
        float P = 0.f;
#pragma unroll
        for (int i = 0; i < BB_MAX_NB1 / 4; ++i) P += (i < Q && rb0 + i < F.NB1) ? pv[i] : 0.f;
        if (big) {
#pragma unroll
            for (int i = 0; i < BB_MAX_NB1 / 4; ++i) P += (BB_MAX_NB1 / 4 + i < Q && rb0 + BB_MAX_NB1 / 4 + i < F.NB1) ? pw[i] : 0.f;
        }
        {
            const float other = __shfl_xor(P, 32);          // quarters (0, 1) of wave 0, (2, 3) of wave 1: lower + upper
            P = hf ? other + P : P + other;
        }
        if (hf == 0) sP[cl][wq][k] = P;
        float sdy = (wq == 0 && lane < F.NB1) ? av.x : 0.f, sdx = (wq == 0 && lane < F.NB1) ? av.y : 0.f;
        if (big && wq == 0 && 64 + lane < F.NB1) {          // blocks 64 .. 127: the lane's second block
            sdy += aw.x;
            sdx += aw.y;
        }
        float db2 = (wq == 1 && lane < F.NB) ? dv : 0.f;
        sdy = naf_sum64(sdy);
        sdx = naf_sum64(sdx);
        db2 = naf_sum64(db2);
        __syncthreads();
            if (col_on) {
            if (wq == 0) {
                if (hf == 0 && k < F.K) {
                    const float invB = 1.0f / (float)F.B;
                    const float Pt = sP[cl][0][k] + sP[cl][1][k];
                    const float g = (gm * invstd) * (Pt - (sdy * invB) * sxk - (sdx * invB) * (invstd * wck));
                    bb_st(F.d_W + (int64_t)col * F.K + k, g, MERGE);
                    sq = g * g;
                } else if (lane == 32) {                      // F.d_gamma = sum dy*xhat, F.d_beta = sum dy; F.d_bias = 0 (see above)
                    bb_st(F.d_gamma + col, sdx, MERGE);
                    bb_st(F.d_beta + col, sdy, MERGE);
                    bb_st(F.d_bias + col, 0.f, MERGE);
                    sq = sdx * sdx + sdy * sdy;
                }
            } else if (lane == 0) {
                bb_st(F.d_bias2 + col, db2, MERGE);
                sq = db2 * db2;
            } else if (lane == 1) {
                sq = g2 * g2 + b2 * b2;
            }
        }
    }
    if (MERGE) {
        bb_finish_exchange(F, block, tid, ex_a, ex_o, ex_on, sQ);
        return;
    }
    if (F.sumsq_partials) {
        const float tot = block_sum_to_thread0<BB_THREADS, true>(sq, sQ, tid);
        if (tid == 0) {
            F.sumsq_partials[block] = tot;
            if (block == 0 && F.step_dev) *F.step_dev += 1;   // read by the NEXT launch (Adam) only
        }
    }
}

template <bool MERGE, bool BIG>
__global__ __launch_bounds__(BB_THREADS) void bb_layer1_bwd_finish_kernel(const FinishArgs F) {
    __shared__ float sQ[BB_THREADS / 64];
    __shared__ float sP[BF_COLS][2][32];
    NAF_TL(g_tl_bb, NAF_TL_BB_FINISH, 0);
    bb_finish_block<MERGE, BIG>(F, (int)blockIdx.x, (int)threadIdx.x, sQ, sP);
    NAF_TL(g_tl_bb, NAF_TL_BB_FINISH, 1);
}

// layer 1: the body is csrc/layer1_body.h (bb_layer1_impl); this is it as a launch of its own
template <int K4, bool ADAM, bool FULL>
__global__ __launch_bounds__(ADAM ? 2 * BB_THREADS : BB_THREADS) void bb_layer1_kernel(
    const float* __restrict__ x, int64_t x_net_stride, int ldx, int K, const float* __restrict__ W,
    const float* __restrict__ bias, const float* __restrict__ gamma, const float* __restrict__ beta,
    int64_t param_net_stride, const float* __restrict__ mom, float* __restrict__ running_mean,
    float* __restrict__ running_var, int64_t stat_net_stride, float* __restrict__ out, int64_t out_net_stride, int ldo,
    float* __restrict__ save_mean, float* __restrict__ save_invstd, float* __restrict__ wc_out, int B, int H, float momentum,
    float eps, int n_main, const AdamArgs ad, int64_t l1_4, int64_t n4, int n_adam, int xcd_rows, float* __restrict__ xhat_out) {
    __shared__ BbL1Shared<K4, ADAM> SM;
    bb_layer1_impl<K4, ADAM, FULL>(SM, (int)blockIdx.x, x, x_net_stride, ldx, K, W, bias, gamma, beta, param_net_stride, mom, running_mean, running_var, stat_net_stride, out, out_net_stride, ldo, save_mean, save_invstd, wc_out, B, H, momentum, eps, n_main, ad, l1_4, n4, n_adam, xcd_rows, xhat_out);
}

// ------------------------------------------------------------------------------------------------------------
// GEMM 2 (and any Linear with K <= 256): Z[net] = A[net] W[net]^T + bias on v_mfma_f32_16x16x4_f32, 64 x 32 output tile
// per workgroup (4 waves: wave = (wm, wn) owns rows 32 wm .. +31 = two 16-row MFMA tiles, columns 16 wn .. +15), K
// staged through LDS in chunks of 128 with the next chunk's loads in flight under the MFMAs, and the column statistics
// of the 64-row block in the epilogue. 2 B / 64 x H / 32 workgroups: 256 at B = 1024, H = 256.
// ------------------------------------------------------------------------------------------------------------
#define BL_BM 64
#define BL_BN 32
#define BL_KC 128                // k per staged chunk: (64 + 32) x 132 x 4 B = 50 KB of LDS, three workgroups per CU
#define BL_LD (BL_KC + 4)
// chunk loads through buffer loads (common.h): row (tid >> 5) + 8 i, float4 (tid & 31) — lane offsets la / lw computed once
__device__ __forceinline__ static void bl_load_chunk_buf(f32x4 (&va)[8], f32x4 (&vb)[4], __amdgpu_buffer_rsrc_t ab, unsigned la, int lda,
                                                        __amdgpu_buffer_rsrc_t wb, unsigned lw, int K, int k0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) va[i] = naf_buf_f4(ab, la, (unsigned)(8 * i * lda + k0) * 4u);
#pragma unroll
    for (int i = 0; i < 4; ++i) vb[i] = naf_buf_f4(wb, lw, (unsigned)(8 * i * K + k0) * 4u);
}
__device__ __forceinline__ static void bl_mfma_chunk(const float* __restrict__ pa0, const float* __restrict__ pa1,
                                            const float* __restrict__ pb, f32x4& c00, f32x4& c01, f32x4& c10, f32x4& c11) {
#pragma unroll 2
    for (int kk = 0; kk < BL_KC; kk += 16) {
        const float4 a0 = *(const float4*)(pa0 + kk), a1 = *(const float4*)(pa1 + kk), b = *(const float4*)(pb + kk);
        c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b.x, c00, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b.x, c10, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b.y, c01, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b.y, c11, 0, 0, 0);
        c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b.z, c00, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b.z, c10, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b.w, c01, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b.w, c11, 0, 0, 0);
    }
}
__device__ __forceinline__ static void bl_store_chunk(const f32x4 (&va)[8], const f32x4 (&vb)[4], float* __restrict__ sA,
                                                     float* __restrict__ sB, int tid) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = tid + BB_THREADS * i;
        *(f32x4*)(sA + (e >> 5) * BL_LD + 4 * (e & 31)) = va[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + BB_THREADS * i;
        *(f32x4*)(sB + (e >> 5) * BL_LD + 4 * (e & 31)) = vb[i];
    }
}
// (amdgpu_waves_per_eu: LDS admits 3 workgroups per CU; left alone the register allocator aimed at 4 waves per SIMD and
// spilled the second chunk's 48 registers to scratch: 8.0 -> 12.8 us per launch)
// ADAM (the deferred optimizer step, adam_body.h): the one-dimensional grid carries, behind its n_main GEMM workgroups (index =
// x + gx y of the former 2-D grid, gx = nets B/64), extra workgroups that step floats [0, 4 l1_4) of the flat buffers — the
// layer-1 segment, which the launch in front of this one read for the last time.
template <bool ADAM, bool FULL, bool KBIG = false>
__global__ __launch_bounds__(BB_THREADS) __attribute__((amdgpu_waves_per_eu(1, 3))) void bb_linear_stats_kernel(const float* __restrict__ a, int64_t a_net_stride,
                                                                     int lda, const float* __restrict__ W,
                                                                     const float* __restrict__ bias,
                                                                     int64_t param_net_stride, float* __restrict__ z,
                                                                     int64_t z_net_stride, int ldz,
                                                                     float2* __restrict__ partials, int B, int N, int K,
                                                                     int gx, int n_main, const AdamArgs ad, int64_t l1_4, int xcd_nets) {
    __shared__ __attribute__((aligned(16))) float sA[BL_BM * BL_LD];
    __shared__ __attribute__((aligned(16))) float sB[BL_BN * BL_LD];
    __shared__ float red[2][BL_BN];
    __shared__ AdamScalars shA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int widx = blockIdx.x;
    // (every argument in one batch of scalar loads: see bb_layer1_kernel)
    asm volatile("" ::"s"(a), "s"(a_net_stride), "s"(lda), "s"(W), "s"(bias), "s"(param_net_stride), "s"(z), "s"(z_net_stride), "s"(ldz),
                 "s"(partials), "s"(B), "s"(N), "s"(K), "s"(gx), "s"(n_main), "s"(l1_4), "s"(xcd_nets));
    if (ADAM)
        asm volatile("" ::"s"(ad.theta), "s"(ad.g), "s"(ad.m), "s"(ad.v), "s"(ad.target), "s"(ad.partials), "s"(ad.n_partials),
                     "s"(ad.max_norm), "s"(ad.lr), "s"(ad.beta1), "s"(ad.beta2), "s"(ad.eps), "s"(ad.tau), "s"(ad.one_minus_tau),
                     "s"(ad.step_dev), "s"(ad.inv_world), "s"(ad.bc));
    if (ADAM && __builtin_expect(widx >= n_main, 0)) {     // (unlikely: the riding step's code sits behind the kernel's own)
        adam_block<BB_THREADS>(ad, 0, (size_t)l1_4, widx - n_main, (int)gridDim.x - n_main, &shA, tid, true);
        return;
    }
    int bx = widx % gx, by = widx / gx;
#define BL_TL(slot) NAF_TL_FL(g_tl_bb, NAF_TL_BB_LINEAR_STATS, slot, widx == 0, widx == n_main - 1)
    const int NB = (B + BB_ROWS - 1) / BB_ROWS;
    int net = bx / NB, rb = bx - net * NB;
    if (xcd_nets && gx == 2 * NB) bb_place_rows(widx, 0, NB, n_main / gx, net, rb, by);     // rows by eighths (see bb_place_rows)
    // rows of this block that exist (64 but for the last block of a batch that is not whole blocks, B % 16 == 0): rows past them
    // read as zeros (the A resource ends there), are not stored (the Z resource ends there) and stay out of the statistics
    const int valid = FULL ? BB_ROWS : (B - rb * BB_ROWS < BB_ROWS ? B - rb * BB_ROWS : BB_ROWS);
    const int n0 = by * BL_BN;
    const float* an = a + net * a_net_stride + (int64_t)rb * BL_BM * lda;
    const float* wn_ = W + net * param_net_stride + (int64_t)n0 * K;     // [N][K] row-major
    // chunk = 128 k: A 64 rows x 32 float4 (8 per thread), B 32 rows x 32 float4 (4 per thread); every load of a chunk is
    // issued before its first LDS store, and the NEXT chunk's loads before this chunk's MFMAs
    f32x4 va[8], vb[4];
    const int r = lane & 15, g = lane >> 4;
    const int wm = wave & 1, wn = wave >> 1;
    const float bcol = bias[net * param_net_stride + n0 + 16 * wn + r];
    const float* pa0 = sA + (32 * wm + r) * BL_LD + 4 * g;
    const float* pa1 = pa0 + 16 * BL_LD;
    const float* pb = sB + (16 * wn + r) * BL_LD + 4 * g;
    f32x4 c00 = {0.f, 0.f, 0.f, 0.f}, c01 = c00, c10 = c00, c11 = c00;
    // the second chunk's loads are issued behind the first chunk's LDS stores and land in registers while its MFMAs run
    // (all 24 loads up front made the register allocator park 12 of them in scratch)
    f32x4 va1[8], vb1[4];
    BL_TL(0);
    const __amdgpu_buffer_rsrc_t ab = naf_buf(an, FULL ? 0x7fffffffu : (unsigned)valid * (unsigned)lda * 4u), wb = naf_buf(wn_);
    const unsigned la = ((unsigned)(tid >> 5) * (unsigned)lda + 4u * (unsigned)(tid & 31)) * 4u;
    const unsigned lw = ((unsigned)(tid >> 5) * (unsigned)K + 4u * (unsigned)(tid & 31)) * 4u;
    bl_load_chunk_buf(va, vb, ab, la, lda, wb, lw, K, 0);
    bl_store_chunk(va, vb, sA, sB, tid);
    __builtin_amdgcn_sched_barrier(0);                    // keep chunk 1's loads behind chunk 0's stores
    bl_load_chunk_buf(va1, vb1, ab, la, lda, wb, lw, K, BL_KC);
    __syncthreads();
    BL_TL(1);
    bl_mfma_chunk(pa0, pa1, pb, c00, c01, c10, c11);
    __syncthreads();                                      // first chunk fully consumed
    BL_TL(2);
    bl_store_chunk(va1, vb1, sA, sB, tid);
    // KBIG (K = 512, round 6: layer sizes up to 512 on this chain; a kernel of its own — as a run-time loop it changed the register
    // allocation of the K = 256 kernel the presets run): two more chunks per trip through the same two register sets, each chunk's
    // loads issued before the MFMAs of the chunk in front of it
    if (KBIG) bl_load_chunk_buf(va, vb, ab, la, lda, wb, lw, K, 2 * BL_KC);
    __syncthreads();
    BL_TL(3);
    bl_mfma_chunk(pa0, pa1, pb, c00, c01, c10, c11);
    if (KBIG) for (int k0 = 2 * BL_KC; k0 < K; k0 += 2 * BL_KC) {
        __syncthreads();                                  // the chunk in LDS fully consumed
        bl_store_chunk(va, vb, sA, sB, tid);
        __builtin_amdgcn_sched_barrier(0);
        bl_load_chunk_buf(va1, vb1, ab, la, lda, wb, lw, K, k0 + BL_KC);
        __syncthreads();
        bl_mfma_chunk(pa0, pa1, pb, c00, c01, c10, c11);
        __syncthreads();
        bl_store_chunk(va1, vb1, sA, sB, tid);
        if (k0 + 2 * BL_KC < K) bl_load_chunk_buf(va, vb, ab, la, lda, wb, lw, K, k0 + 2 * BL_KC);
        __syncthreads();
        bl_mfma_chunk(pa0, pa1, pb, c00, c01, c10, c11);
    }
    BL_TL(4);
    // C/D map: col = lane & 15, row = 4 (lane >> 4) + reg
    float v[2][4];
    float s = 0.f;
    const __amdgpu_buffer_rsrc_t zb_ = naf_buf(z + net * z_net_stride + (int64_t)(rb * BL_BM) * ldz + n0,
                                               FULL ? 0x7fffffffu : ((unsigned)(valid - 1) * (unsigned)ldz + BL_BN) * 4u);
    const unsigned lz_ = 4u * (unsigned)((32 * wm + 4 * g) * ldz + 16 * wn + r);
    bool on[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            on[mt][e] = FULL || 32 * wm + 16 * mt + 4 * g + e < valid;
            v[mt][e] = (mt ? c10[e] + c11[e] : c00[e] + c01[e]) + bcol;
            naf_buf_st_f1(zb_, lz_, (unsigned)((16 * mt + e) * ldz) * 4u, v[mt][e], B >= NAF_WT_MIN_B);
            s += on[mt][e] ? v[mt][e] : 0.f;
        }
    // column statistics of the 64-row block: 8 rows in the lane, 4 lane groups (bits 4, 5), 2 waves (wm) through LDS
    s = naf_xor32_add(naf_xor16_add(s));
    if (g == 0) red[wm][16 * wn + r] = s;
    __syncthreads();
    const float S = red[0][16 * wn + r] + red[1][16 * wn + r];
    const float mb = S * ((FULL || valid == BB_ROWS) ? 1.0f / BB_ROWS : 1.0f / (float)valid);
    float m2 = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t = v[mt][e] - mb;
            m2 += on[mt][e] ? t * t : 0.f;
        }
    m2 = naf_xor32_add(naf_xor16_add(m2));
    __syncthreads();
    if (g == 0) red[wm][16 * wn + r] = m2;
    __syncthreads();
    if (wm == 0 && g == 0)
        partials[((int64_t)net * NB + rb) * N + n0 + 16 * wn + r] = make_float2(S, red[0][16 * wn + r] + red[1][16 * wn + r]);
    BL_TL(5);
}

// The same GEMM with 64 x 16 tiles for small batches (B <= 512): the 64 x 32 grid is 2 B / 64 x 8 = 64 workgroups at B = 256 and
// each spends 2 x 1.2 us in its two MFMA phases — a quarter of the chip busy, latency all the way. Half as wide, twice as
// many workgroups (wave w = rows 16 w .. +15, one MFMA tile): the MFMA phases halve. The statistics blocks stay 64 rows, so
// the partials and every consumer are unchanged.
template <bool ADAM, bool FULL, bool KBIG = false>
__global__ __launch_bounds__(BB_THREADS) void bb_linear_stats16_kernel(const float* __restrict__ a, int64_t a_net_stride, int lda,
                                                                       const float* __restrict__ W, const float* __restrict__ bias,
                                                                       int64_t param_net_stride, float* __restrict__ z,
                                                                       int64_t z_net_stride, int ldz, float2* __restrict__ partials,
                                                                       int B, int N, int K, int gx, int n_main, const AdamArgs ad,
                                                                       int64_t l1_4, int xcd_nets) {
    constexpr int BN = 16;
    __shared__ __attribute__((aligned(16))) float sA[BL_BM * BL_LD];
    __shared__ __attribute__((aligned(16))) float sB[BN * BL_LD];
    __shared__ float red[4][BN];
    __shared__ AdamScalars shA;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int widx = blockIdx.x;
    // (every argument in one batch of scalar loads: see bb_layer1_kernel)
    asm volatile("" ::"s"(a), "s"(a_net_stride), "s"(lda), "s"(W), "s"(bias), "s"(param_net_stride), "s"(z), "s"(z_net_stride), "s"(ldz),
                 "s"(partials), "s"(B), "s"(N), "s"(K), "s"(gx), "s"(n_main), "s"(l1_4), "s"(xcd_nets));
    if (ADAM)
        asm volatile("" ::"s"(ad.theta), "s"(ad.g), "s"(ad.m), "s"(ad.v), "s"(ad.target), "s"(ad.partials), "s"(ad.n_partials),
                     "s"(ad.max_norm), "s"(ad.lr), "s"(ad.beta1), "s"(ad.beta2), "s"(ad.eps), "s"(ad.tau), "s"(ad.one_minus_tau),
                     "s"(ad.step_dev), "s"(ad.inv_world), "s"(ad.bc));
    if (ADAM && __builtin_expect(widx >= n_main, 0)) {     // (extra workgroups: the deferred step of the layer-1 segment, see above)
        adam_block<BB_THREADS>(ad, 0, (size_t)l1_4, widx - n_main, (int)gridDim.x - n_main, &shA, tid, true);
        return;
    }
    int bx = widx % gx, by = widx / gx;
    const int NB = (B + BB_ROWS - 1) / BB_ROWS;
    int net = bx / NB, rb = bx - net * NB;
    if (xcd_nets && gx == 2 * NB) bb_place_rows(widx, 0, NB, n_main / gx, net, rb, by);     // rows by eighths (see bb_place_rows)
    const int valid = FULL ? BB_ROWS : (B - rb * BB_ROWS < BB_ROWS ? B - rb * BB_ROWS : BB_ROWS);   // (see bb_linear_stats_kernel)
    const int n0 = by * BN;
    const int r = lane & 15, g = lane >> 4;
    // operands through buffer loads (common.h): A rows (tid >> 5) + 8 i, float4 (tid & 31); B rows (tid >> 5) + 8 i < 16
    const __amdgpu_buffer_rsrc_t ab = naf_buf(a + net * a_net_stride + (int64_t)rb * BL_BM * lda,
                                              FULL ? 0x7fffffffu : (unsigned)valid * (unsigned)lda * 4u);
    const __amdgpu_buffer_rsrc_t wb = naf_buf(W + net * param_net_stride + (int64_t)n0 * K);
    const unsigned la = ((unsigned)(tid >> 5) * (unsigned)lda + 4u * (unsigned)(tid & 31)) * 4u;
    const unsigned lw = ((unsigned)(tid >> 5) * (unsigned)K + 4u * (unsigned)(tid & 31)) * 4u;
    const float bcol = bias[net * param_net_stride + n0 + r];
    f32x4 va[8], vb[2], va1[8], vb1[2];
    BL_TL(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) va[i] = naf_buf_f4(ab, la, (unsigned)(8 * i * lda) * 4u);
#pragma unroll
    for (int i = 0; i < 2; ++i) vb[i] = naf_buf_f4(wb, lw, (unsigned)(8 * i * K) * 4u);
    float* sa_t = sA + (tid >> 5) * BL_LD + 4 * (tid & 31);
    float* sb_t = sB + (tid >> 5) * BL_LD + 4 * (tid & 31);
#pragma unroll
    for (int i = 0; i < 8; ++i) *(f32x4*)(sa_t + 8 * i * BL_LD) = va[i];
#pragma unroll
    for (int i = 0; i < 2; ++i) *(f32x4*)(sb_t + 8 * i * BL_LD) = vb[i];
    __builtin_amdgcn_sched_barrier(0);                    // keep chunk 1's loads behind chunk 0's stores
#pragma unroll
    for (int i = 0; i < 8; ++i) va1[i] = naf_buf_f4(ab, la, (unsigned)(8 * i * lda + BL_KC) * 4u);
#pragma unroll
    for (int i = 0; i < 2; ++i) vb1[i] = naf_buf_f4(wb, lw, (unsigned)(8 * i * K + BL_KC) * 4u);
    __syncthreads();
    BL_TL(1);
    const float* pa = sA + (16 * wave + r) * BL_LD + 4 * g;
    const float* pb = sB + r * BL_LD + 4 * g;
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
    auto mfma_chunk = [&]() {
#pragma unroll 4
        for (int kk = 0; kk < BL_KC; kk += 16) {
            const f32x4 av = *(const f32x4*)(pa + kk), bv = *(const f32x4*)(pb + kk);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[1], c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv[2], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv[3], c1, 0, 0, 0);
        }
    };
    mfma_chunk();
    __syncthreads();                                      // first chunk fully consumed
    BL_TL(2);
#pragma unroll
    for (int i = 0; i < 8; ++i) *(f32x4*)(sa_t + 8 * i * BL_LD) = va1[i];
#pragma unroll
    for (int i = 0; i < 2; ++i) *(f32x4*)(sb_t + 8 * i * BL_LD) = vb1[i];
    // K > 256 (round 6): two more chunks per trip, see bb_linear_stats_kernel
    auto load_chunk = [&](f32x4 (&xa)[8], f32x4 (&xb)[2], int k0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) xa[i] = naf_buf_f4(ab, la, (unsigned)(8 * i * lda + k0) * 4u);
#pragma unroll
        for (int i = 0; i < 2; ++i) xb[i] = naf_buf_f4(wb, lw, (unsigned)(8 * i * K + k0) * 4u);
    };
    auto store_chunk = [&](const f32x4 (&xa)[8], const f32x4 (&xb)[2]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *(f32x4*)(sa_t + 8 * i * BL_LD) = xa[i];
#pragma unroll
        for (int i = 0; i < 2; ++i) *(f32x4*)(sb_t + 8 * i * BL_LD) = xb[i];
    };
    if (KBIG) load_chunk(va, vb, 2 * BL_KC);
    __syncthreads();
    BL_TL(3);
    mfma_chunk();
    if (KBIG) for (int k0 = 2 * BL_KC; k0 < K; k0 += 2 * BL_KC) {
        __syncthreads();                                  // the chunk in LDS fully consumed
        store_chunk(va, vb);
        __builtin_amdgcn_sched_barrier(0);
        load_chunk(va1, vb1, k0 + BL_KC);
        __syncthreads();
        mfma_chunk();
        __syncthreads();
        store_chunk(va1, vb1);
        if (k0 + 2 * BL_KC < K) load_chunk(va, vb, k0 + 2 * BL_KC);
        __syncthreads();
        mfma_chunk();
    }
    BL_TL(4);
    // C/D map: col = lane & 15, row = 4 (lane >> 4) + reg. Z2 out, then the column statistics of the 64-row block: 4 rows in
    // the lane, 4 lane groups, 4 waves through LDS
    float v[4];
    float sum = 0.f;
    {
        const unsigned ldz4 = (unsigned)ldz * 4u;
        const int vw = valid - 16 * wave;                 // rows of this wave's 16 that exist (<= 0: none)
        const int vwc = vw > 16 ? 16 : vw;
        const __amdgpu_buffer_rsrc_t zb = naf_buf(z + net * z_net_stride + (int64_t)(rb * BL_BM + 16 * wave) * ldz + n0,
                                                  FULL ? 0x7fffffffu : vw > 0 ? ((unsigned)(vwc - 1) * (unsigned)ldz + BN) * 4u : 0u);
        const unsigned lz = (unsigned)(4 * g) * ldz4 + 4u * (unsigned)r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = (c0[e] + c1[e]) + bcol;
            naf_buf_st_f1(zb, lz, (unsigned)e * ldz4, v[e], B >= NAF_WT_MIN_B);
            sum += (FULL || 4 * g + e < vw) ? v[e] : 0.f;
        }
    }
    const int vw_ = valid - 16 * wave;
    sum = naf_xor32_add(naf_xor16_add(sum));
    if (g == 0) red[wave][r] = sum;
    __syncthreads();
    const float S = (red[0][r] + red[1][r]) + (red[2][r] + red[3][r]);
    const float mb = S * ((FULL || valid == BB_ROWS) ? 1.0f / BB_ROWS : 1.0f / (float)valid);
    float m2 = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float t = v[e] - mb;
        m2 += (FULL || 4 * g + e < vw_) ? t * t : 0.f;
    }
    m2 = naf_xor32_add(naf_xor16_add(m2));
    __syncthreads();
    if (g == 0) red[wave][r] = m2;
    __syncthreads();
    if (tid < BN) partials[((int64_t)net * NB + rb) * N + n0 + tid] = make_float2(S, (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]));
    BL_TL(5);
#undef BL_TL
}

// ------------------------------------------------------------------------------------------------------------
// layer 2 + heads + NAF head + first backward stage of layer 2 in ONE launch (H = 256): a workgroup owns 32 batch rows
// across ALL 256 features, so nothing between the layer-2 pre-activations and dY2 leaves the chip:
//   fold the layer-2 statistics of both nets (512 (net, column) pairs, one per thread) -> xhat tile of the main net in LDS
//   (A2 = ReLU(gamma xhat + beta) written out for the dWh GEMM), V'(s') of the target net as a per-row dot product ->
//   heads = A2 Wh^T on f32 MFMA (A2 formed from xhat while the fragments are read) -> naf_head_body (Q, TD target, MSE,
//   d_heads) -> dA2 = d_heads Wh on f32 MFMA -> ReLU mask, dY2 and the block sums (sum dy, sum dy*xhat) per column.
// Replaces bb_bn_relu_heads_partial + naf_head_kernel + bb_heads_bwd_stage1 (three launches, 19.9 us at B = 1024).
// The backward partials are per 32-row block here: partials_bw[B/32][H].
// ------------------------------------------------------------------------------------------------------------
#define FK_ROWS 32               // rows per workgroup; 16 for small batches (ROWS below)
#define FK_THREADS 512
#define FK_H 256
#define FK_LD (FK_H + 4)
#define FK_MAX_A 11              // joints the fused launch takes: NHP = 80 floats of heads at most (the Wh tile: 83 KB of LDS)
// ROWS = 16 (B <= 1024): twice the workgroups, each phase of this latency chain roughly half as long (rows per wave in the
// normalise phase, MFMA tiles per wave in the two products); the backward partials are then per 16-row block.
// FULL: whole 64-row blocks, at most 32 of them (every BASELINE config): the kernel as it was before round 4 took other batch sizes
// — every row of every workgroup is a sample, the statistics fold has no last-block weights and no two-pass form. The general form
// is a kernel of its own: inside one kernel its extra paths (never taken at these sizes) cost 0.2 us per update at B = 256.
// HALVES = 2 (round 6: layer sizes up to 512 — H = 512 columns): TWO workgroups per row block, workgroup 2 i + h owning columns
// 256 h .. 256 h + 255 of block i. Everything up to the heads GEMM and everything behind the head body is per column (the statistics,
// x-hat, A2, the Wh tile's columns, dA2, dY2, the block sums): the kernel above on a 256-column slice, reading and writing at a column
// offset. The heads (and V'(s')) are sums over ALL columns: each sibling takes its half's partial products and hands them to the other
// as self-validating 16-byte records {v, v, epoch, v} (the protocol of the statistics records: sc1 store, sc1 poll, bounded), both
// add the two halves and the bias — a + b = b + a: the same bits on both — and run the head body redundantly (sibling 0 writes Q,
// the loss parts and d_heads). Siblings are neighbours in the grid: a workgroup's sibling is resident or next in line for dispatch.
template <int PMODE, int NH4, int ROWS, bool FULL, int HALVES = 1>
__global__ __launch_bounds__(FK_THREADS) void bb_layer2_head_kernel(
    const float* __restrict__ z, int64_t z_net_stride, int ldz, const float* __restrict__ gamma,
    const float* __restrict__ beta, int64_t param_net_stride, const float2* __restrict__ partials, int NB64,
    float* __restrict__ running_mean, float* __restrict__ running_var, int64_t stat_net_stride,
    float* __restrict__ a2_out, int ldo, float* __restrict__ save_mean, float* __restrict__ save_invstd,
    const float* __restrict__ Wh, int64_t wh_net_stride, int ldw, const float* __restrict__ u, int ldu,
    const float* __restrict__ r, int ldr, float gamma_td, float* __restrict__ q_out, float* __restrict__ d_heads,
    float* __restrict__ loss_partials, float* __restrict__ dy_out, int ldd, float2* __restrict__ partials_bw, int B, int A,
    float momentum, float eps, int xcd_rows, float* stat_rec /* polled while other workgroups of the launch write it: NOT __restrict__ — with it the compiler
    hoisted the poll's load out of its loop (the asm memory clobber does not reach a noalias argument) and the wait never ended */,
    const int* epoch_p, unsigned long long* errors, int n_fold, float* xch) {
    constexpr int NHP = 4 * NH4, H = FK_H;
    constexpr int Ht = HALVES * FK_H;                      // columns of the layer; this workgroup's are c0 .. c0 + 255
    constexpr int MT = ROWS / 16;                          // 16-row MFMA tiles
    constexpr int RPW = ROWS / 8;                          // rows per wave where a wave owns whole rows
    // 9 .. 11 joints (round 6; NHP = 64 | 80: [mu | l | V] is 55 | 66 | 78 wide): one sample per 16-LANE group in the head body (G; the
    // 16 rows of a workgroup are its first 256 threads), a Wh tile of 66 | 83 KB, heads and d_heads rows for 32 lane groups instead of
    // 64. At 16 rows per workgroup (B <= 2048) that is 128 | 154 KB of the CU's 160; at 32 rows (beyond) the A2 tile goes (A2T below: the
    // heads product forms A2 from x-hat as it reads, as the matmul mode always does) and the matmul mode's L tiles (35 KB) do not fit: the
    // Hadamard head only. Everything else — the statistics, the products' tile loops, the halves' exchange, dA2 — walks NHP as it finds it.
    constexpr int G = NHP > HEAD_MAX_LDH ? 16 : 8;         // lanes per sample in the head body
    constexpr int SLOTS = FK_THREADS / G;                  // lane groups (ROWS of them carry samples)
    // K split of the heads GEMM (keeps the 8 waves busy): MT x NHP/16 tiles x KS ranges over 8 waves
    constexpr int KS = ROWS == 16 ? (NHP == 64 ? 2 : 4) : 2;
    constexpr int DH_ROWS = SLOTS > (KS - 1) * ROWS ? SLOTS : (KS - 1) * ROWS;   // sDH also holds the partial tiles of K ranges 1 .. KS - 1
    static_assert(ROWS == 16 || ROWS == 32, "rows per workgroup");
    static_assert(G == 8 || ROWS == 16 || PMODE != NAF_P_MATMUL, "9 .. 11 joints, 32 rows per workgroup: the Hadamard head only (no room for the L tiles)");
    static_assert(ROWS <= SLOTS && (NHP * (FK_H / 4)) % FK_THREADS == 0, "lane groups / Wh tile per thread");
    // A2 = ReLU(gamma xhat + beta) as a tile of its own when the LDS budget allows (not with the 18 KB L tiles of the matmul
    // mode): the heads GEMM then reads ONE operand row per macro-step instead of xhat + gamma + beta and forms nothing on the
    // VALU inside its MFMA loop — that loop was bound by LDS reads (4 x 16 B per lane per step, 8 waves), 1.7 us for 0.4 us of MFMA
    // (9 .. 11 joints at 32 rows per workgroup — B > 2048 —: the 66 | 83 KB heads tile leaves no room for it either)
    constexpr bool A2T = PMODE != NAF_P_MATMUL && !(G == 16 && ROWS == 32);
    __shared__ __attribute__((aligned(16))) float sXH[ROWS * FK_LD];
    __shared__ __attribute__((aligned(16))) float sA2[A2T ? ROWS * FK_LD : 4];
    __shared__ __attribute__((aligned(16))) float sW[NHP * FK_LD];
    __shared__ __attribute__((aligned(16))) float sHd[SLOTS * NHP];                 // heads rows (ROWS live)
    __shared__ __attribute__((aligned(16))) float sDH[DH_ROWS * NHP];               // d_heads rows
    __shared__ __attribute__((aligned(16))) float sStat[2][4][H];                   // [net][mean, invstd, gamma, beta]
    __shared__ __attribute__((aligned(16))) float sWv[H];
    __shared__ float sBias[NHP + 1];
    __shared__ float sV[FK_THREADS / 8];
    __shared__ float sL[PMODE == NAF_P_MATMUL ? (G == 8 ? SLOTS : ROWS) * G * (G + 1) : 1];   // (G = 16: the live rows' tiles only, head_body.h)
    __shared__ float sRed[FK_THREADS / 64];
    __shared__ float2 sP[MT][H];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // which rows: workgroup i runs on XCD i % 8, and the bundle's dA1 blocks — the readers of the dY2 / A2 rows written here —
    // sit on XCD x for the block rows of the x-th EIGHTH of the batch (gemm_bundle.hip; the same rows are the K range of the
    // dW2 blocks on that XCD). Workgroup i takes row chunk (i % 8) (chunks / 8) + i / 8: every chunk is then written on the XCD
    // that reads it, and the lines are still in its L2 behind the launch boundary. (Whole groups of 16 chunks only; placement
    // is speed only.)
    // n_fold > 0 (more than 8 statistics blocks, B > 512): the launch's first n_fold workgroups fold the layer-2 statistics ONCE —
    // 32 (net, column) pairs each, one thread per pair, the arithmetic of bb_fold_stats — and publish (mean, invstd, *epoch, var)
    // as one self-validating 16-byte record per pair; every other workgroup polls the 512 records instead of pulling all
    // 2 x NB x 256 partials itself (128 KB per workgroup at B = 2048, 2.6 of the prologue's 4 us at the ~75 GB/s a CU gets out of
    // L2). The protocol, its ordering argument and its way out of a wait that lasts are the bundle's (bn2bwd_fold.h).
    int rb = blockIdx.x;
    const int n_main = (int)gridDim.x - n_fold;
    if (n_fold) {
        if (__builtin_expect(rb < n_fold, 0)) {
            asm volatile("" ::"s"(partials), "s"(NB64), "s"(B), "s"(eps), "s"(epoch_p), "s"(stat_rec));   // (one batch of scalar loads)
            if (tid < 32) {
                const int pair = 32 * rb + tid, net = pair / Ht, col = pair % Ht;
                float mean, var;
                if (ROWS == 16) bb_fold_stats<FULL>(partials      /* (16 rows per workgroup <=> B <= 2048 <=> at most 32 blocks) */ + (int64_t)net * NB64 * Ht, Ht, NB64, B, col, &mean, &var);
                else bb_fold_stats_big(partials + (int64_t)net * NB64 * Ht, Ht, NB64, B, col, &mean, &var);
                const float invstd = 1.0f / sqrtf(var + eps);
                const int epoch = *epoch_p;
                const f32x4 rec = {mean, invstd, __builtin_bit_cast(float, epoch), var};
                naf_buf_st_f4_sc1(naf_buf(stat_rec), 16u * (unsigned)pair, 0, rec);
            }
            return;
        }
        rb -= n_fold;
    }
    const int half = HALVES == 2 ? (rb & 1) : 0, c0 = FK_H * half;
    const int n_blocks = HALVES == 2 ? n_main >> 1 : n_main;
    if (HALVES == 2) rb >>= 1;
    if (ROWS == 16 && (n_blocks & 15) == 0 && xcd_rows) rb = (rb & 7) * (n_blocks >> 3) + (rb >> 3);
    const int64_t s0 = (int64_t)rb * ROWS;
    const int T = A * (A + 1) / 2, v_col = A + T;
    // every kernel argument this prologue needs, fetched NOW: left to itself the compiler fetches an argument where it is
    // first used, behind the branches of this prologue — six dependent scalar round trips in front of the loads proper
    asm volatile("" ::"s"(z), "s"(gamma), "s"(beta), "s"(partials), "s"(running_mean), "s"(running_var), "s"(a2_out), "s"(save_mean),
                 "s"(save_invstd), "s"(Wh), "s"(u), "s"(r), "s"(z_net_stride), "s"(param_net_stride), "s"(stat_net_stride),
                 "s"(wh_net_stride), "s"(ldz), "s"(ldw), "s"(ldu), "s"(ldr), "s"(NB64), "s"(B), "s"(A));
    #define FK_TL(slot) NAF_TL_FL(g_tl_bb, NAF_TL_BB_LAYER2_HEAD, slot, (int)blockIdx.x == n_fold, blockIdx.x == gridDim.x - 1)
    FK_TL(0);
    // ---- phase 0: every global operand requested up front, in ONE batch: the Z2 tiles, the head weights (into registers,
    // native vectors), the statistics partials, the per-sample scalars — measured with the weights staged behind the
    // statistics fold this phase took 4.4 of the kernel's 10.8 us (two dependent round trips to fresh data) -------------
    // (addresses: wave-uniform bases in SGPRs + one lane offset; with per-thread 64-bit arithmetic the 52 loads of this
    // prologue cost ~460 vector instructions per wave, two waves per SIMD: 2 us of issue before the first byte arrived)
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const unsigned l16 = 16u * (unsigned)lane, l8 = 8u * (unsigned)lane, l4 = 4u * (unsigned)lane;
    f32x4 zm[RPW], zt[RPW];
    {
        const __amdgpu_buffer_rsrc_t zb = naf_buf(z + (s0 + wave_s) * ldz + c0), ztb = naf_buf(z + z_net_stride + (s0 + wave_s) * ldz + c0);
#pragma unroll
        for (int i = 0; i < RPW; ++i) {                   // one wave per row: 64 lanes x 4 columns; rows wave + 8 i
            zm[i] = naf_buf_f4(zb, l16, (unsigned)(8 * i * ldz) * 4u);
            zt[i] = naf_buf_f4(ztb, l16, (unsigned)(8 * i * ldz) * 4u);
        }
    }
    constexpr int WPT = NHP * (H / 4) / FK_THREADS;       // float4 of the Wh tile per thread: 2, 4 or 6
    f32x4 wreg[WPT];
    {
        const __amdgpu_buffer_rsrc_t wb = naf_buf(Wh + (int64_t)wave_s * ldw + c0);
#pragma unroll
        for (int i = 0; i < WPT; ++i) wreg[i] = naf_buf_f4(wb, l16, (unsigned)(8 * i * ldw) * 4u);   // row = e >> 6, e = tid + 512 i
    }
    const float bias_r = tid < NHP ? Wh[(int64_t)tid * ldw + Ht] : (tid == NHP ? Wh[wh_net_stride + (int64_t)v_col * ldw + Ht] : 0.f);
    f32x4 wv_r = {0.f, 0.f, 0.f, 0.f};
    if (tid < H / 4) wv_r = *(const f32x4*)(Wh + wh_net_stride + (int64_t)v_col * ldw + c0 + 4 * tid);
    const int s_loc_ = tid / G, i_ = tid & (G - 1);
    // rows of this workgroup that exist (the last workgroup of a batch that is not whole 16-row groups holds fewer): the others are
    // not samples — the head body leaves their d_heads zero (so dA2, dY2 and every block sum get nothing from them), their Z2 rows
    // are the zeros the buffer was allocated with, and nothing of the minibatch is read for them
    const int ns_ = FULL ? ROWS : (B - (int)s0 < ROWS ? B - (int)s0 : ROWS);
    const bool live_ = s_loc_ < ns_;
    const float u_val = (live_ && i_ < A) ? u[(s0 + s_loc_) * ldu + i_] : 0.f;
    const float r_val = (live_ && i_ == 0) ? r[(s0 + s_loc_) * ldr] : 0.f;
    {
        const int net = wave_s >> 2;                       // 512 threads = 2 nets x 256 columns: 4 waves per net
        const int cb = (wave_s & 3) * 64;                  // the wave's 64 columns
        const unsigned col = (unsigned)(cb + lane);
        const float gm_ = naf_buf_f1(naf_buf(gamma + net * param_net_stride + c0 + cb), l4, 0);
        const float bt_ = naf_buf_f1(naf_buf(beta + net * param_net_stride + c0 + cb), l4, 0);
        float rm_ = 0.f, rv_ = 0.f;
        float* rmp = running_mean + net * stat_net_stride + c0;
        float* rvp = running_var + net * stat_net_stride + c0;
        if (rb == 0) {
            rm_ = rmp[col];
            rv_ = rvp[col];
        }
        float mean, var, invstd;
        f32x4 c;
        // (uniform n_fold) the folded statistics, once their record carries this launch; a thread whose budget runs out — the GPU is
        // shared and this XCD's folding workgroup still queued (bn2bwd_fold.h) — folds its pair itself: same arithmetic, same bits
        if (n_fold && gemm_bn2bwd_poll_record(naf_buf(stat_rec), net * Ht + c0 + (int)col, *epoch_p, &c)) {
            mean = c[0];
            invstd = c[1];
            var = c[3];
        } else {
            if (n_fold && errors) __hip_atomic_fetch_add(errors, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (ROWS == 16) bb_fold_stats_u<FULL>(naf_buf(partials + (int64_t)net * NB64 * Ht + c0 + cb), l8, 0, Ht, NB64, B, &mean, &var);
            else bb_fold_stats_big(partials + (int64_t)net * NB64 * Ht, Ht, NB64, B, c0 + cb + lane, &mean, &var);
            invstd = 1.0f / sqrtf(var + eps);
        }
        sStat[net][0][col] = mean;
        sStat[net][1][col] = invstd;
        sStat[net][2][col] = gm_;
        sStat[net][3][col] = bt_;
        if (rb == 0) {
            const float unbiased = B > 1 ? var * ((float)B / (float)(B - 1)) : var;
            rmp[col] = (1.0f - momentum) * rm_ + momentum * mean;
            rvp[col] = (1.0f - momentum) * rv_ + momentum * unbiased;
            (save_mean + (int64_t)net * Ht + c0)[col] = mean;
            (save_invstd + (int64_t)net * Ht + c0)[col] = invstd;
        }
    }
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int e = tid + FK_THREADS * i;
        *(f32x4*)(sW + (e >> 6) * FK_LD + 4 * (e & 63)) = wreg[i];
    }
    if (tid <= NHP) sBias[tid] = bias_r;                   // bias = column H of Wh (the ones column of A2); [NHP] = the target's V bias
    if (tid < H / 4) *(f32x4*)(sWv + 4 * tid) = wv_r;
    for (int e = tid; e < DH_ROWS * NHP; e += FK_THREADS) sDH[e] = 0.f;
    __syncthreads();
    FK_TL(1);
    // ---- phase 1: normalise. main net -> xhat (LDS) and A2 (memory); target net -> V'(s') ---------------------------
    {
        const f32x4 m0 = *(const f32x4*)&sStat[0][0][4 * lane], i0 = *(const f32x4*)&sStat[0][1][4 * lane];
        const f32x4 g0 = *(const f32x4*)&sStat[0][2][4 * lane], b0 = *(const f32x4*)&sStat[0][3][4 * lane];
        const f32x4 m1 = *(const f32x4*)&sStat[1][0][4 * lane], i1 = *(const f32x4*)&sStat[1][1][4 * lane];
        const f32x4 g1 = *(const f32x4*)&sStat[1][2][4 * lane], b1 = *(const f32x4*)&sStat[1][3][4 * lane];
        const f32x4 wv = *(const f32x4*)(sWv + 4 * lane);
        float p[RPW];
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int row = wave + 8 * i;
            const f32x4 xh = (zm[i] - m0) * i0;
            f32x4 y;
#pragma unroll
            for (int c = 0; c < 4; ++c) y[c] = fmaxf(__builtin_fmaf(xh[c], g0[c], b0[c]), 0.f);
            *(f32x4*)(sXH + row * FK_LD + 4 * lane) = xh;
            if (A2T) *(f32x4*)(sA2 + row * FK_LD + 4 * lane) = y;
            naf_buf_st_f4(naf_buf(a2_out + (s0 + wave_s) * ldo + c0), l16, (unsigned)(8 * i * ldo) * 4u, y, B >= NAF_WT_MIN_B);
            const f32x4 xt = (zt[i] - m1) * i1;
            float q = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) q = __builtin_fmaf(fmaxf(__builtin_fmaf(xt[c], g1[c], b1[c]), 0.f), wv[c], q);
            p[i] = q;
        }
        // the four rows' wave reductions (DPP + permlane swaps: common.h)
#pragma unroll
        for (int i = 0; i < RPW; ++i) p[i] = naf_sum64(p[i]);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < RPW; ++i) sV[wave + 8 * i] = HALVES == 2 ? p[i] : p[i] + sBias[NHP];    // (HALVES: the bias behind the exchange)
        }
    }
    __syncthreads();
    FK_TL(2);
    // ---- phase 2: heads = A2 Wh^T + bias: MT x NHP/16 MFMA tiles, each cut into KS ranges of K, over the 8 waves (with whole
    // tiles half of the waves idled through the longest MFMA chain of the kernel: 2.2 of its 10.8 us). The ranges meet in
    // LDS, in K order -------------------------------------------------------------------------------------------------
    const int rr = lane & 15, gg = lane >> 4;
    constexpr int NT = MT * (NHP / 16);                    // tiles
    float* sHalf = sDH;                                    // partial tiles of K ranges 1 .. KS - 1 (sDH is zeroed again below)
    for (int t = wave; t < KS * NT; t += 8) {
        const int tile = t % NT, kh = t / NT;
        const int mt = tile % MT, nt = tile / MT;
        const float* pa = (A2T ? sA2 : sXH) + (16 * mt + rr) * FK_LD + 4 * gg;
        const float* pb = sW + (16 * nt + rr) * FK_LD + 4 * gg;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int kk = kh * (H / KS); kk < (kh + 1) * (H / KS); kk += 16) {
            const f32x4 xh = *(const f32x4*)(pa + kk), b = *(const f32x4*)(pb + kk);
            float a0 = xh.x, a1 = xh.y, a2 = xh.z, a3 = xh.w;
            if (!A2T) {
                const f32x4 g = *(const f32x4*)&sStat[0][2][kk + 4 * gg], be = *(const f32x4*)&sStat[0][3][kk + 4 * gg];
                a0 = fmaxf(__builtin_fmaf(xh.x, g.x, be.x), 0.f), a1 = fmaxf(__builtin_fmaf(xh.y, g.y, be.y), 0.f);
                a2 = fmaxf(__builtin_fmaf(xh.z, g.z, be.z), 0.f), a3 = fmaxf(__builtin_fmaf(xh.w, g.w, be.w), 0.f);
            }
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b.w, acc1, 0, 0, 0);
        }
        float* dst = kh ? sHalf + (kh - 1) * ROWS * NHP : sHd;
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[(16 * mt + 4 * gg + e) * NHP + 16 * nt + rr] = acc0[e] + acc1[e];
    }
    __syncthreads();
    FK_TL(3);
    for (int e = tid; e < ROWS * NHP; e += FK_THREADS) {
        float hsum = sHd[e];
#pragma unroll
        for (int h = 0; h < KS - 1; ++h) {
            hsum += sHalf[h * ROWS * NHP + e];
            sHalf[h * ROWS * NHP + e] = 0.f;
        }
        sHd[e] = HALVES == 2 ? hsum : hsum + sBias[e % NHP];
    }
    __syncthreads();
    if (HALVES == 2) {
        // the two column halves' partial heads (ROWS x NHP) and partial V'(s') (ROWS) meet: three values + the launch's number per
        // 16-byte record, written through; the sibling's polled until they carry this launch's number (bounded: a sibling is a
        // neighbour in the grid, resident or next in line)
        constexpr int NV = ROWS * NHP + ROWS, NREC = (NV + 2) / 3;
        const int epoch = *epoch_p;
        const __amdgpu_buffer_rsrc_t mine = naf_buf(xch + ((int64_t)rb * 2 + half) * NREC * 4);
        const __amdgpu_buffer_rsrc_t theirs = naf_buf(xch + ((int64_t)rb * 2 + (1 - half)) * NREC * 4);
        auto val = [&](int j) { return j < ROWS * NHP ? sHd[j] : (j < NV ? sV[j - ROWS * NHP] : 0.f); };
        for (int t = tid; t < NREC; t += FK_THREADS) {
            const f32x4 rec = {val(3 * t), val(3 * t + 1), __builtin_bit_cast(float, epoch), val(3 * t + 2)};
            naf_buf_st_f4_sc1(mine, 16u * (unsigned)t, 0, rec);
        }
        for (int t = tid; t < NREC; t += FK_THREADS) {
            f32x4 c = naf_buf_f4_sc1(theirs, 16u * (unsigned)t, 0);
            float tagf = c[2];
            const long long t0 = wall_clock64();
            while (__builtin_bit_cast(int, tagf) != epoch) {
                if (wall_clock64() - t0 > 200000LL) {      // 2 ms: a hang guard, not a schedule — the update is poisoned and counted
                    if (errors) __hip_atomic_fetch_add(errors, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    c[0] = c[1] = c[3] = __builtin_nanf("");
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");
                c = naf_buf_f4_sc1(theirs, 16u * (unsigned)t, 0);
                tagf = c[2];
            }
            const float other[3] = {c[0], c[1], c[3]};
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int j = 3 * t + q;
                if (j < ROWS * NHP) sHd[j] = (sHd[j] + other[q]) + sBias[j % NHP];
                else if (j < NV) sV[j - ROWS * NHP] = (sV[j - ROWS * NHP] + other[q]) + sBias[NHP];
            }
        }
        __syncthreads();
    }
    FK_TL(4);
    // ---- phase 3: the NAF head on the 32 rows (threads 0..255 carry samples; every thread joins the barriers) --------
    naf_head_body<PMODE, 2, FK_THREADS, G>(sHd, sDH, sL, sRed, NHP, u_val, r_val, live_ ? sV[s_loc_] : 0.f, 0.f, gamma_td,
                                        half ? nullptr : q_out, nullptr,
                                        // (the body stores its workgroup's loss part at [blockIdx.x]; with two workgroups per row block
                                        //  and the folding workgroups in front, the row block's own index keeps it inside the array)
                                        half ? nullptr : (HALVES == 2 && loss_partials ? loss_partials + (rb - (int)blockIdx.x) : loss_partials),
                                        B, A, s0, ns_);
    FK_TL(5);
    if (ROWS * NH4 <= FK_THREADS) {
        if (!half && tid < ROWS * NH4) ((float4*)(d_heads + s0 * NHP))[tid] = ((const float4*)sDH)[tid];
    } else if (!half) {                                    // (32 rows of 80 floats: 640 float4 for 512 threads)
        for (int e = tid; e < ROWS * NH4; e += FK_THREADS) ((float4*)(d_heads + s0 * NHP))[e] = ((const float4*)sDH)[e];
    }
    // ---- phase 4: dA2 = d_heads Wh (K = NHP), MT x 16 tiles, 2 MT per wave; ReLU mask, dY2, block sums ------------------
    {
        const int mt = wave_s % MT;
        const float* pa = sDH + (16 * mt + rr) * NHP + 4 * gg;
        const unsigned ldd4 = (unsigned)ldd * 4u;
        const __amdgpu_buffer_rsrc_t dyb = naf_buf(dy_out + s0 * ldd + c0);
        const unsigned ldy = (unsigned)(4 * gg) * ldd4 + 4u * (unsigned)rr;
#pragma unroll
        for (int j = 0; j < 2 * MT; ++j) {
            const int nt = wave_s / MT + (8 / MT) * j;
            const int col = 16 * nt + rr;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < NHP; kk += 16) {
                const float4 a = *(const float4*)(pa + kk);
                const float* q = sW + (kk + 4 * gg) * FK_LD + col;       // k-major operand: Wh[h][col]
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, q[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, q[FK_LD], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, q[2 * FK_LD], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, q[3 * FK_LD], acc, 0, 0, 0);
            }
            const float g = sStat[0][2][col], be = sStat[0][3][col];
            float s_dy = 0.f, s_dx = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = 16 * mt + 4 * gg + e;
                const float xh = sXH[row * FK_LD + col];
                const float dy = __builtin_fmaf(xh, g, be) > 0.f ? acc[e] : 0.f;      // the forward's own ReLU decision
                // dY2 leaves as whole rows behind the barrier where there is a tile to collect it in (A2's, free since the heads
                // GEMM): 16 wave-stores of 1 KB instead of 64 of 4 x 64 B per workgroup
                if (A2T) sA2[row * FK_LD + col] = dy;
                else naf_buf_st_f1(dyb, ldy, (unsigned)(16 * mt + e) * ldd4 + (unsigned)(16 * nt) * 4u, dy, B >= NAF_WT_MIN_B);
                s_dy += dy;
                s_dx += dy * xh;
            }
            s_dy = naf_xor32_add(naf_xor16_add(s_dy));
            s_dx = naf_xor32_add(naf_xor16_add(s_dx));
            if (gg == 0) sP[mt][col] = make_float2(s_dy, s_dx);
        }
    }
    __syncthreads();
    if (A2T) {
        const __amdgpu_buffer_rsrc_t dyr = naf_buf(dy_out + s0 * ldd + c0);
#pragma unroll
        for (int i = 0; i < ROWS * (H / 4) / FK_THREADS; ++i) {
            const int e = tid + FK_THREADS * i, row = e >> 6, q = e & 63;
            naf_buf_st_f4(dyr, (unsigned)(row * ldd + 4 * q) * 4u, 0, *(const f32x4*)(sA2 + row * FK_LD + 4 * q), B >= NAF_WT_MIN_B);
        }
    }
    FK_TL(6);
    if (tid < H) {
        float2 t = sP[0][tid];
        if (MT == 2) {
            t.x += sP[MT - 1][tid].x;
            t.y += sP[MT - 1][tid].y;
        }
        partials_bw[(int64_t)rb * Ht + c0 + tid] = t;
    }
    FK_TL(7);
#undef FK_TL
}

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
// (any B from 16 to 4096: the last 64-row block, the last 16-row workgroup of bb_layer2_head and the last MFMA tile may be partial —
//  rows past the batch read as zeros, are never stored and stay out of every sum; the caller's activation buffers hold whole 16-row
//  groups, zero-initialised, so that the rows past the batch ARE zeros wherever a later launch walks them as a K dimension)
static int bb_shape_ok(int B, int H) { return B >= 16 && B <= BB_MAX_NB2 * BB_ROWS && H >= BB_COLS && (H % BB_COLS) == 0; }
static int bb_blocks(int B) { return (B + BB_ROWS - 1) / BB_ROWS; }

extern "C" int naf_bb_moments_floats(int K) {
    if (K <= 0 || K > 4 * BB_MAX_K4) return NAF_ERR_ARG;
    const int kp = (K + 3) / 4 <= 6 ? 24 : 32;
    return kp + kp * kp;
}

extern "C" int naf_bb_moments(const float* x, int64_t batch_stride, int64_t x_net_stride, int ldx, int K, float* mom, int B,
                              int n_batches, int nets, void* stream) {
    if (!x || !mom || B <= 0 || n_batches <= 0 || nets <= 0 || K <= 0 || K > 4 * BB_MAX_K4) return NAF_ERR_ARG;
    const int k4 = (K + 3) / 4, k4d = k4 <= 6 ? 6 : 8;
    if (((uintptr_t)x & 15) != 0 || (ldx & 3) != 0 || ldx < 4 * k4d || (x_net_stride & 3) != 0 || (batch_stride & 3) != 0 ||
        ((uintptr_t)mom & 15) != 0)
        return NAF_ERR_ARG;
    dim3 grid(n_batches, nets);
    if (k4d == 6) bb_moments_kernel<6><<<grid, BM_THREADS, 0, (hipStream_t)stream>>>(x, batch_stride, x_net_stride, ldx, mom, B);
    else bb_moments_kernel<8><<<grid, BM_THREADS, 0, (hipStream_t)stream>>>(x, batch_stride, x_net_stride, ldx, mom, B);
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}

// the deferred optimizer step riding on a launch (include/naf_hip.h, naf_adam_args_t): the device-side arguments, the end of
// the layer-1 segment and of the buffers in float4, and how many extra workgroups step floats [lo, hi)
static bool bb_adam_setup(const naf_adam_args_t* adam, AdamArgs& ad, int64_t& l1_4, int64_t& n4) {
    memset(&ad, 0, sizeof(ad));
    l1_4 = n4 = 0;
    if (!adam) return true;
    if (!adam_args_from(*adam, ad) || adam->l1_floats <= 0 || (adam->l1_floats & 3) || adam->n <= adam->l1_floats || (adam->n & 3))
        return false;
    l1_4 = adam->l1_floats / 4;
    n4 = adam->n / 4;
    return true;
}
static int bb_adam_blocks(int64_t lo4, int64_t hi4, int threads) { return (int)((hi4 - lo4 + threads - 1) / threads); }

extern "C" int naf_bb_layer1_adam(const float* x, int64_t x_net_stride, int ldx, int K, const float* W, const float* bias,
                                  const float* gamma, const float* beta, int64_t param_net_stride, const float* mom,
                                  float* running_mean, float* running_var, int64_t stat_net_stride, float* out,
                                  int64_t out_net_stride, int ldo, float* save_mean, float* save_invstd, float* wc_out,
                                  float* xhat_out, int B, int H, int nets, float momentum, float eps, const naf_adam_args_t* adam,
                                  void* stream) {
    if (xhat_out && ((uintptr_t)xhat_out & 15)) return NAF_ERR_ARG;
    if (!x || !W || !bias || !mom || !gamma || !beta || !running_mean || !running_var || !out || !save_mean || !save_invstd ||
        !bb_shape_ok(B, H) || nets <= 0 || K <= 0 || K > 4 * BB_MAX_K4 || ldo < H || (ldo & 3))
        return NAF_ERR_ARG;
    const int k4 = (K + 3) / 4, k4d = k4 <= 6 ? 6 : 8;
    if (((uintptr_t)x & 15) != 0 || (ldx & 3) != 0 || ldx < 4 * k4d || (x_net_stride & 3) != 0) return NAF_ERR_ARG;
    if ((((uintptr_t)bias | (uintptr_t)out | (uintptr_t)mom | (uintptr_t)W) & 15) != 0 || (param_net_stride & 3) != 0 ||
        (out_net_stride & 3) != 0)
        return NAF_ERR_ARG;                              // (W: the 64-column runs of 64 K floats are read as float4)
    AdamArgs ad;
    int64_t l1_4, n4;
    if (!bb_adam_setup(adam, ad, l1_4, n4)) return NAF_ERR_ARG;
    if (adam) {
        // the parameters this launch reads must be the main network's layer-1 segment, the target's param_net_stride behind
        const float* lo = adam->theta, *hi = adam->theta + adam->l1_floats;
        if ((((uintptr_t)gamma | (uintptr_t)beta) & 15) != 0) return NAF_ERR_ARG;       // (read as float4 here)
        if (nets > 2 || W < lo || W + (int64_t)H * K > hi || bias < lo || bias + H > hi || gamma < lo || gamma + H > hi || beta < lo ||
            beta + H > hi || (nets == 2 && adam->theta_target != adam->theta + param_net_stride))
            return NAF_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    const int n_main = bb_blocks(B) * (H / BB_COLS) * nets;
    const int n_adam = adam ? bb_adam_blocks(l1_4, n4, 2 * BB_THREADS) : 0;
    const int grid = n_main + n_adam;
    const int xcd_rows = 1;      // rows by eighths (bb_place_rows)
#define BB_L1_(K4V, AD, FL)                                                                                                    \
    bb_layer1_kernel<K4V, AD, FL><<<grid, (AD) ? 2 * BB_THREADS : BB_THREADS, 0, st>>>(x, x_net_stride, ldx, K, W, bias, gamma, beta, param_net_stride, mom, \
                                                           running_mean, running_var, stat_net_stride, out, out_net_stride, ldo, \
                                                           save_mean, save_invstd, wc_out, B, H, momentum, eps, n_main, ad, l1_4, n4, n_adam, xcd_rows, xhat_out)
#define BB_L1(K4V, AD)                       \
    do {                                     \
        if (B % BB_ROWS == 0) BB_L1_(K4V, AD, true); \
        else BB_L1_(K4V, AD, false);         \
    } while (0)
    if (k4d == 6) { if (adam) BB_L1(6, true); else BB_L1(6, false); }
    else { if (adam) BB_L1(8, true); else BB_L1(8, false); }
#undef BB_L1
#undef BB_L1_
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}
extern "C" int naf_bb_linear_stats_adam(const float* a, int64_t a_net_stride, int lda, const float* W, const float* bias,
                                        int64_t param_net_stride, float* z, int64_t z_net_stride, int ldz, float* partials, int B,
                                        int N, int K, int nets, const naf_adam_args_t* adam, void* stream) {
    if (!a || !W || !bias || !z || !partials || !bb_shape_ok(B, N) || nets <= 0) return NAF_ERR_ARG;
    if (K < 2 * BL_KC || K > 4 * BL_KC || K % (2 * BL_KC) != 0 || lda < K || (lda & 3) || ldz < N) return NAF_ERR_ARG;   // pairs of 128-k chunks: H = 256 | 512
    if ((((uintptr_t)a | (uintptr_t)W) & 15) != 0 || (a_net_stride & 3) != 0 || (param_net_stride & 3) != 0 ||
        ((uintptr_t)partials & 7) != 0)
        return NAF_ERR_ARG;
    AdamArgs ad;
    int64_t l1_4, n4;
    if (!bb_adam_setup(adam, ad, l1_4, n4)) return NAF_ERR_ARG;
    const int extra = adam ? bb_adam_blocks(0, l1_4, BB_THREADS) : 0;
    const int xcd_nets = 1;      // rows by eighths (bb_place_rows)
    const int gx = nets * bb_blocks(B);
    hipStream_t st = (hipStream_t)stream;
#define BB_LS__(KERNEL, GY, FL, KB)                                                                                         \
    do {                                                                                                                    \
        const int n_main = gx * (GY);                                                                                       \
        if (adam) KERNEL<true, FL, KB><<<n_main + extra, BB_THREADS, 0, st>>>(a, a_net_stride, lda, W, bias, param_net_stride, z, \
                                                                      z_net_stride, ldz, (float2*)partials, B, N, K, gx, n_main, ad, l1_4, xcd_nets); \
        else KERNEL<false, FL, KB><<<n_main, BB_THREADS, 0, st>>>(a, a_net_stride, lda, W, bias, param_net_stride, z, z_net_stride, \
                                                          ldz, (float2*)partials, B, N, K, gx, n_main, ad, l1_4, xcd_nets);             \
    } while (0)
#define BB_LS_(KERNEL, GY, FL)                               \
    do {                                                     \
        if (K > 2 * BL_KC) BB_LS__(KERNEL, GY, FL, true);    \
        else BB_LS__(KERNEL, GY, FL, false);                 \
    } while (0)
#define BB_LS(KERNEL, GY)                            \
    do {                                             \
        if (B % BB_ROWS == 0) BB_LS_(KERNEL, GY, true); \
        else BB_LS_(KERNEL, GY, false);              \
    } while (0)
#ifndef BB_MAX16
#define BB_MAX16 512
#endif
    const int max16 = BB_MAX16;  // small batches: 64 x 16 tiles, twice the workgroups (see the kernel)
    if (B <= max16) BB_LS(bb_linear_stats16_kernel, N / 16);
    else BB_LS(bb_linear_stats_kernel, N / BL_BN);
#undef BB_LS
#undef BB_LS_
#undef BB_LS__
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}
// rows per workgroup = rows per block of partials_bw (its consumer, the bundle's BatchNorm-backward prologue, is told B / rows blocks)
extern "C" int naf_bb_layer2_head_rows(int B) {
    // (32 rows per workgroup measured slower at every batch size up to 2048: 16.9k -> 17.7k updates/s there with 16. Beyond 2048 it
    //  is 32 all the same: the bundle's BatchNorm-backward fold takes at most 128 blocks of backward partials, bn2bwd_fold.h)
    return B > 32 * BB_ROWS ? 32 : 16;
}

// the exchange area of the two-halves form (H = 512): per row block and half, ceil((rows NHP + rows) / 3) records of 4 floats
extern "C" int naf_bb_layer2_head_exchange_floats(int B, int NHP) {
    if (B <= 0 || NHP <= 0) return NAF_ERR_ARG;
    const int rows = naf_bb_layer2_head_rows(B);
    return ((B + rows - 1) / rows) * 2 * ((rows * NHP + rows + 2) / 3) * 4;
}

extern "C" int naf_bb_layer2_head(const float* z, int64_t z_net_stride, int ldz, const float* gamma, const float* beta,
                                  int64_t param_net_stride, const float* partials, float* running_mean, float* running_var,
                                  int64_t stat_net_stride, float* a2_out, int ldo, float* save_mean, float* save_invstd,
                                  const float* Wh, int64_t wh_net_stride, int ldw, int NHP, const float* u, int ldu,
                                  const float* r, int ldr, float gamma_td, float* q_out, float* d_heads, float* loss_partials,
                                  float* dy_out, int ldd, float* partials_bw, int B, int H, int A, int p_mode, float momentum,
                                  float eps, const naf_bb_stats_once_t* once, void* stream) {
    if (once && (!once->records || !once->epoch || ((uintptr_t)once->records & 15))) return NAF_ERR_ARG;
    if (!z || !gamma || !beta || !partials || !running_mean || !running_var || !a2_out || !save_mean || !save_invstd || !Wh ||
        !u || !r || !q_out || !d_heads || !dy_out || !partials_bw || !bb_shape_ok(B, H) || (H != FK_H && H != 2 * FK_H))
        return NAF_ERR_ARG;
    const bool two = H == 2 * FK_H;      // two workgroups per row block (HALVES = 2): needs the exchange area and the launch's number
    if (two && (!once || !once->exchange || ((uintptr_t)once->exchange & 15))) return NAF_ERR_ARG;
    // (A <= 8: NHP = 16 | 32 | 48, one sample per 8-lane group; 9 .. 11 joints: NHP = 64 | 80, per 16-lane group — beyond B = 2048, where
    //  a workgroup takes 32 rows, with the Hadamard head only: the matmul mode's L tiles do not fit beside the heads tile)
    if (A <= 0 || A > FK_MAX_A || NHP < A + A * (A + 1) / 2 + 1 || NHP != ((A + A * (A + 1) / 2 + 1 + 15) / 16) * 16) return NAF_ERR_ARG;
    if (A > NAF_MAX_A && naf_bb_layer2_head_rows(B) != 16 && p_mode != NAF_P_HADAMARD) return NAF_ERR_ARG;
    if (p_mode != NAF_P_HADAMARD && p_mode != NAF_P_MATMUL) return NAF_ERR_ARG;
    if (ldz < H || (ldz & 3) || ldo < H || (ldo & 3) || ldd < H || ldw <= H || (ldw & 3) || ldu < A || ldr < 1) return NAF_ERR_ARG;
    if ((((uintptr_t)z | (uintptr_t)a2_out | (uintptr_t)Wh | (uintptr_t)d_heads) & 15) != 0 || (z_net_stride & 3) ||
        (wh_net_stride & 3) || ((uintptr_t)partials & 7) || ((uintptr_t)partials_bw & 7))
        return NAF_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int rows = naf_bb_layer2_head_rows(B);
    // folded once per launch where a workgroup would pull more than 8 statistics blocks per column (B > 512) — and only there: the
    // readers wait ~2.6 us for the records (updates/s, A/B/A/B on one box: B = 512 30.4k -> 30.2k with it, 1024 25.9k -> 26.3k,
    // 1536 21.0k -> 21.65k, 2048 20.35k -> 20.58k)
    const int n_fold = (once && bb_blocks(B) > 8) ? 2 * H / 32 : 0;
    const int blocks = (two ? 2 : 1) * ((B + rows - 1) / rows) + n_fold;
    float* xch = once ? once->exchange : nullptr;
    float* rec = once ? once->records : nullptr;
    const int* epoch_p = once ? once->epoch : nullptr;
    unsigned long long* errors = once ? (unsigned long long*)once->errors : nullptr;
    const int xcd_rows = 1;      // row chunks dealt to the XCD whose dA1 blocks read them (+0.4 - 1 %)
#define BB_FK_ARGS z, z_net_stride, ldz, gamma, beta, param_net_stride, (const float2*)partials, bb_blocks(B), running_mean, running_var, \
        stat_net_stride, a2_out, ldo, save_mean, save_invstd, Wh, wh_net_stride, ldw, u, ldu, r, ldr, gamma_td, q_out, d_heads, \
        loss_partials, dy_out, ldd, (float2*)partials_bw, B, A, momentum, eps, xcd_rows, rec, epoch_p, errors, n_fold, xch
#define BB_FK_R(PM, NH4V, RW, FL)                                                                                           \
    do {                                                                                                                    \
        if (two) bb_layer2_head_kernel<PM, NH4V, RW, FL, 2><<<blocks, FK_THREADS, 0, st>>>(BB_FK_ARGS);                      \
        else bb_layer2_head_kernel<PM, NH4V, RW, FL, 1><<<blocks, FK_THREADS, 0, st>>>(BB_FK_ARGS);                          \
    } while (0)
#define BB_FK(PM, NH4V)                                                          \
    do {                                                                         \
        if (rows == 16 && B % BB_ROWS == 0) BB_FK_R(PM, NH4V, 16, true);          \
        else if (rows == 16) BB_FK_R(PM, NH4V, 16, false);                       \
        else BB_FK_R(PM, NH4V, 32, false);                                       \
    } while (0)
#define BB_FK_WIDE(PM, NH4V)                                                \
    do {                                                                    \
        if (rows == 16 && B % BB_ROWS == 0) BB_FK_R(PM, NH4V, 16, true);    \
        else if (rows == 16) BB_FK_R(PM, NH4V, 16, false);                  \
        else if ((PM) == NAF_P_HADAMARD) BB_FK_R(NAF_P_HADAMARD, NH4V, 32, false); \
    } while (0)
#define BB_FK_NH(PM)                          \
    do {                                      \
        if (NHP == 16) BB_FK(PM, 4);          \
        else if (NHP == 32) BB_FK(PM, 8);     \
        else if (NHP == 48) BB_FK(PM, 12);    \
        else if (NHP == 64) BB_FK_WIDE(PM, 16); \
        else BB_FK_WIDE(PM, 20);              \
    } while (0)
    if (p_mode == NAF_P_HADAMARD) BB_FK_NH(NAF_P_HADAMARD);
    else BB_FK_NH(NAF_P_MATMUL);
#undef BB_FK_NH
#undef BB_FK_WIDE
#undef BB_FK
#undef BB_FK_R
#undef BB_FK_ARGS
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}

extern "C" int naf_bb_layer1_bwd_kp(int K) {
    if (K <= 0 || K > 4 * BB_MAX_K4) return NAF_ERR_ARG;
    return (K + 3) / 4 <= 6 ? 24 : 32;
}

extern "C" int naf_bb_layer1_bwd_finish_blocks(int H) { return H > 0 ? (H + BF_COLS - 1) / BF_COLS : NAF_ERR_ARG; }

extern "C" int naf_bb_layer1_bwd_finish(const float* p_slabs, int K, const float* partials1, int nb1,
                                        const float* dz2_col_partials, int nb, const float* mom, const float* wc, const float* gamma, const float* save_invstd,
                                        float* d_W, float* d_gamma, float* d_beta, float* d_bias, float* d_bias2,
                                        const float* d_gamma2, const float* d_beta2, float* sumsq_partials, int32_t* step_dev,
                                        int B, int H, const naf_bb_slab_seg_t* segs, int n_segs, int* fold_flag,
                                        const naf_xgmi_push_t* push, const float* grad_base, size_t merge_total, void* stream) {
    if ((push != nullptr) != (grad_base != nullptr)) return NAF_ERR_ARG;
    if (merge_total && (!push || n_segs != 2 || (merge_total & 3) || merge_total > push->n_pad || !sumsq_partials || !step_dev ||
                        !(push->timeout_ticks > 0)))
        return NAF_ERR_ARG;
    if (push && (push->world < 2 || push->world > NAF_XGMI_MAX_WORLD || ((uintptr_t)grad_base & 15))) return NAF_ERR_ARG;
    if (!p_slabs || !partials1 || (nb > 0 && !dz2_col_partials) || !mom || !wc || !gamma || !save_invstd || !d_W || !d_gamma || !d_beta ||
        !d_bias || !d_bias2 || nb < 0 || nb > BB_MAX_NB || nb1 <= 0 || nb1 > 2 * BB_MAX_NB1 || H <= 0 || B <= 0 || K <= 0 || K > 32)
        return NAF_ERR_ARG;
    if (sumsq_partials && (!d_gamma2 || !d_beta2)) return NAF_ERR_ARG;
    if (n_segs < 0 || n_segs > 2 || (n_segs && !segs)) return NAF_ERR_ARG;
    BbSlabs sl;
    memset(&sl, 0, sizeof(sl));
    sl.n_finish_blocks = (H + BF_COLS - 1) / BF_COLS;
    sl.n_seg = n_segs;
    int blocks = 0;
    for (int i = 0; i < n_segs; ++i) {
        const naf_bb_slab_seg_t& g = segs[i];
        if (!g.src || !g.dst || g.n <= 0 || (g.n & 3) || g.n_slabs < 1 || g.n_slabs > BB_MAX_SLABS || g.stride < g.n || (g.stride & 3) ||
            (((uintptr_t)g.src | (uintptr_t)g.dst) & 15) != 0)
            return NAF_ERR_ARG;
        sl.seg[i].src = g.src; sl.seg[i].dst = g.dst; sl.seg[i].stride = g.stride;
        sl.seg[i].n = g.n; sl.seg[i].n_slabs = g.n_slabs; sl.seg[i].block0 = blocks;
        blocks += (g.n + BB_THREADS * 4 - 1) / (BB_THREADS * 4);
    }
    FinishArgs F;
    memset(&F, 0, sizeof(F));
    F.p_slabs = p_slabs; F.KP = naf_bb_layer1_bwd_kp(K); F.K = K; F.partials1 = (const float2*)partials1; F.NB1 = nb1;
    F.dz2_col_partials = dz2_col_partials; F.NB = nb; F.mom = mom; F.wc = wc; F.gamma = gamma; F.save_invstd = save_invstd;
    F.d_W = d_W; F.d_gamma = d_gamma; F.d_beta = d_beta; F.d_bias = d_bias; F.d_bias2 = d_bias2; F.d_gamma2 = d_gamma2;
    F.d_beta2 = d_beta2; F.sumsq_partials = sumsq_partials; F.step_dev = step_dev; F.B = B; F.H = H; F.slabs = sl;
    F.fold_flag = fold_flag; F.n_blocks = sl.n_finish_blocks + blocks;
    if (push) {
        for (int i = 0; i < n_segs; ++i)       // every pushed segment: inside the flat gradient, whole float4
            if (segs[i].dst < grad_base || (size_t)(segs[i].dst - grad_base) + (size_t)segs[i].n > push->n_pad ||
                ((segs[i].dst - grad_base) & 3))
                return NAF_ERR_ARG;
        F.push = *push;
        F.grad_base = grad_base;
        if (merge_total) {
            // what no slab workgroup covers: the flat gradient minus the two segments — [0, first), [first end, second) (and nothing
            // behind the second: it ends the buffer) — float4 ranges small enough for one workgroup
            size_t o[2] = {(size_t)(segs[0].dst - grad_base), (size_t)(segs[1].dst - grad_base)};
            int first = o[0] <= o[1] ? 0 : 1, second = 1 - first;
            if (o[second] + (size_t)segs[second].n != merge_total || o[first] + (size_t)segs[first].n > o[second]) return NAF_ERR_ARG;
            F.r_lo[0] = 0; F.r_hi[0] = o[first];
            F.r_lo[1] = o[first] + (size_t)segs[first].n; F.r_hi[1] = o[second];
            if ((F.r_hi[0] & 3) || (F.r_lo[1] & 3) || (F.r_hi[1] & 3) ||
                ((F.r_hi[0] - F.r_lo[0]) + (F.r_hi[1] - F.r_lo[1])) > (size_t)BB_EX_SMALL * BB_THREADS * 4)
                return NAF_ERR_ARG;
            F.merge = 1;
        }
    }
    const bool big1 = nb1 > BB_MAX_NB1;
    if (F.merge && big1) bb_layer1_bwd_finish_kernel<true, true><<<F.n_blocks, BB_THREADS, 0, (hipStream_t)stream>>>(F);
    else if (F.merge) bb_layer1_bwd_finish_kernel<true, false><<<F.n_blocks, BB_THREADS, 0, (hipStream_t)stream>>>(F);
    else if (big1) bb_layer1_bwd_finish_kernel<false, true><<<F.n_blocks, BB_THREADS, 0, (hipStream_t)stream>>>(F);
    else bb_layer1_bwd_finish_kernel<false, false><<<F.n_blocks, BB_THREADS, 0, (hipStream_t)stream>>>(F);
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}
