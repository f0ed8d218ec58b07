// The second stage of layer 2's BatchNorm backward inside the backward GEMM launch (include/naf_hip.h, naf_gemm_bn2bwd_t): the
// fold of the block sums by the launch's first workgroups and the readers' side of the hand-off. Shared by the two forms of the
// bundle (gemm_bundle.hip: one 32 x 32 block per workgroup; gemm_bundle_p.hip: persistent workgroups on 64 x 64 tiles); both
// run 512 threads per workgroup; the shipping one also 256 (THREADS).
#pragma once
#include "common.h"
#include "../../include/naf_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- the block sums are folded ONCE per launch ----------------------------------------------------------------------------------
// With every block folding for itself (round 2's first form) a dA1 block reads npb x 256 float2 — 32 KB at 16 row blocks, 128 KB at
// 64, on top of the 64 KB of its panels. Instead the
// first H / 32 workgroups of the launch fold 32 columns each (up to 64 row blocks: the k-major fold above with two blocks per
// part) and publish ONE 16-byte record per column to `cst`: (k1 c1, invstd k1 c2, epoch, 0) — the two constants that depend on
// the fold plus the number of this launch (*epoch, a device word that naf_bb_layer1_bwd_finish advances behind every bundle
// launch). The GEMM blocks request their panels, derive mean and k1 themselves (known since the forward pass), and every
// thread that needs a column's constants polls THAT record until it carries this launch's epoch: a self-validating 16-byte
// granule (one sc1 store instruction by one lane; sc1 loads observe it whole), so no flag, no barrier and no atomic sit
// between the fold and its readers (a counter + flag protocol cost 2.2 us in front of a block's first MFMA, this one ~1).
// A dependency INSIDE the launch, so: the folding workgroups are the launch's first (dispatched before any block that waits
// for them; nothing they do depends on another workgroup). That order holds per XCD, not across processes: with several processes
// sharing the GPU an XCD can be full of ANOTHER process's waiting blocks while this launch's folding workgroup for that XCD is still
// queued — a circular wait between launches (4 ranks on one GPU at B = 1024 ran into it). So a wait never blocks for long: a
// thread polls for GB_POLL_TICKS (20 us: ten times what the records take on an idle chip), then FOLDS FOR ITSELF — the same sums
// in the same order as the folding workgroup (gemm_bn2bwd_fold_column), so the result does not depend on which of the two
// happened — and counts the event in `errors`, a pinned host word the training loop can read (Learner.fold_fallbacks: a
// diagnostic of an over-subscribed GPU, no result depends on it). No wait can expire into a wrong number.
// Records and polls are sc1 only (MI355X_MICROARCH.md, hand-offs with sc1 loads in place of the acquire).
#define GB_POLL_TICKS 2000LL             // 20 us at 100 MHz
#ifndef GB_FOLD_COLS
#define GB_FOLD_COLS 32
#endif
template <int THREADS = 512>
__device__ static inline void gemm_bn2bwd_fold_block(const naf_gemm_bn2bwd_t& P, int f, int tid, float* scratch) {
    constexpr int NPAIR = GB_FOLD_COLS / 2, PARTS = THREADS / NPAIR, QMAX = 128 / PARTS;
    static_assert(PARTS * QMAX >= 128, "npb <= 128");
    const int col0 = f * GB_FOLD_COLS;
    const int pair = tid % NPAIR, part = tid / NPAIR;
    const int npb = P.npb, Q = (npb + PARTS - 1) / PARTS, rb0 = part * Q;
    const __amdgpu_buffer_rsrc_t pb = naf_buf(P.partials + 2 * col0);
    f32x4 v[QMAX];
#pragma unroll
    for (int i = 0; i < QMAX; ++i) {
        const int rb = rb0 + i;
        v[i] = naf_buf_f4(pb, 16u * (unsigned)pair, (unsigned)((i < Q && rb < npb) ? rb : 0) * (unsigned)P.H * 8u);
    }
    const int c = tid & (GB_FOLD_COLS - 1);
    const float gm = P.gamma[col0 + c], invstd = P.save_invstd[col0 + c];
    const int epoch = *P.epoch;
    f32x4 sm = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < QMAX; ++i)
        if (i < Q && rb0 + i < npb) sm += v[i];
    ((f32x4*)scratch)[part * NPAIR + pair] = sm;            // [part][column] float2
    __syncthreads();
    if (tid < GB_FOLD_COLS) {
        const float2* sp = (const float2*)scratch;
        float sdy = 0.f, sdx = 0.f;
#pragma unroll
        for (int q = 0; q < PARTS; ++q) {
            sdy += sp[q * GB_FOLD_COLS + c].x;                // (parts past the last block hold zeros)
            sdx += sp[q * GB_FOLD_COLS + c].y;
        }
        const float k1 = gm * invstd, invB = 1.0f / (float)P.B;
        const f32x4 out = {k1 * (sdy * invB), invstd * (k1 * (sdx * invB)), __builtin_bit_cast(float, epoch), 0.f};
        naf_buf_st_f4_sc1(naf_buf(P.cst), 16u * (unsigned)(col0 + c), 0, out);
        P.d_gamma[col0 + c] = sdx;                            // d_gamma = sum dy*xhat, d_beta = sum dy (read after the launch)
        P.d_beta[col0 + c] = sdy;
    }
}
// A POLLED BUFFER MUST NOT BE A `__restrict__` KERNEL ARGUMENT: the asm memory clobber in the loop below does not reach a noalias
// argument, the compiler hoists the load out of the loop and the wait never ends (bb_layer2_head_kernel's records, round 3: 45 -
// 75 % of the pollers ran into the hang guard until the qualifier was dropped; here the records come out of a struct field).
// (the record in *out; false: the budget ran out before the record carried this launch's epoch)
__device__ __forceinline__ static bool gemm_bn2bwd_poll_record(__amdgpu_buffer_rsrc_t rb, int col, int epoch, f32x4* out) {
    // (the tag through a scalar copy: __builtin_bit_cast applied to the vector ELEMENT c[2] reads element 0 — clang 22 takes
    //  the address of the vector for the element reference; seen in the IR, and as a wait that never ended)
    f32x4 c = naf_buf_f4_sc1(rb, 16u * (unsigned)col, 0);
    float tagf = c[2];
    bool ok = __builtin_bit_cast(int, tagf) == epoch;
    if (!ok) {
        const long long t0 = wall_clock64();
        while (true) {
            __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");       // a poll: the load below must be issued again every trip (the intrinsic is
            c = naf_buf_f4_sc1(rb, 16u * (unsigned)col, 0);   // not volatile: hoisted out of the loop, the wait never ended)
            tagf = c[2];
            if (__builtin_bit_cast(int, tagf) == epoch) { ok = true; break; }
            if (wall_clock64() - t0 > GB_POLL_TICKS) break;
        }
    }
    *out = c;
    return ok;
}
// what the folding workgroup publishes for one column, computed by the thread itself: the same parts, the same order
template <int THREADS>
__device__ static inline f32x4 gemm_bn2bwd_fold_column(const naf_gemm_bn2bwd_t& P, int col) {
    constexpr int PARTS = THREADS / (GB_FOLD_COLS / 2);
    const int npb = P.npb, Q = (npb + PARTS - 1) / PARTS;
    const float2* pp = (const float2*)P.partials;
    float sdy = 0.f, sdx = 0.f;
    for (int q = 0; q < PARTS; ++q) {
        float a = 0.f, b = 0.f;
        for (int i = 0; i < Q; ++i) {
            const int rb = q * Q + i;
            if (rb < npb) {
                const float2 v = pp[(int64_t)rb * P.H + col];
                a += v.x;
                b += v.y;
            }
        }
        sdy += a;
        sdx += b;
    }
    const float invstd = P.save_invstd[col], k1 = P.gamma[col] * invstd, invB = 1.0f / (float)P.B;
    return (f32x4){k1 * (sdy * invB), invstd * (k1 * (sdx * invB)), 0.f, 0.f};
}
// Every thread that needs a column's constants polls that column's record itself. (Measured against two alternatives, updates/s
// at B = 256 | 512 | 1024 | 2048: this 30.6k | 28.1k | 23.6k | 16.8k; a few lanes of the first wave polling one record per folding
// workgroup, a barrier, then the records: 30.2k | 27.8k | 23.5k | 16.2k; a counter the folding workgroups add to after their
// stores have landed, one polling lane, barrier, records: 30.1k | 27.3k | 24.0k on a box ~2 % faster | 16.4k. Every block folding
// for itself, the form before: 29.9k | 27.2k | not possible | not possible.)
// the waiting side: constants of the block's columns -> cst (LDS, [4][256]). The caller puts the barrier behind it.
// kbase (k-contiguous A only): the first of the 256 columns the array holds — 0 but for layer sizes beyond 256 (round 6), where a
// block walks K = H in 256-column stretches and refills the array at each.
template <bool AK, int THREADS>
__device__ static inline void gemm_bn2bwd_wait_constants(const naf_gemm_bn2bwd_t& P, int m0, int tid, float* cst, int kbase = 0) {
    constexpr int NCOL = AK ? 32 : 256;
    const int col0 = AK ? m0 : kbase;
    if (tid < NCOL) {
        const int col = col0 + tid;
        const int epoch = *P.epoch;
        const float mean = P.save_mean[col], k1 = P.gamma[col] * P.save_invstd[col];
        f32x4 c;
        if (!gemm_bn2bwd_poll_record(naf_buf(P.cst), col, epoch, &c)) {
            c = gemm_bn2bwd_fold_column<THREADS>(P, col);
            if (P.errors) __hip_atomic_fetch_add((unsigned long long*)P.errors, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        cst[tid] = mean;
        cst[256 + tid] = k1;
        cst[512 + tid] = c[0];
        cst[768 + tid] = c[1];
    }
}
