// The moments of layer 1's inputs (csrc/big_batch.hip, bb_moments_kernel; csrc/step_path.hip, step_prep_kernel): ONE body for the
// launch that serves a chunk of minibatches and for the per-timestep launch that samples, gathers and takes the moments of one
// minibatch — the same sums in the same order, so a minibatch's record is the same bits whichever launch produced it.
#pragma once
#include "common.h"

// ------------------------------------------------------------------------------------------------------------
// Layer 1 is LINEAR in the minibatch rows, so everything BatchNorm needs from the batch dimension follows from the first
// two moments of X = the state (net 0) / next-state (net 1) columns of the rows, which do not depend on the weights:
//   Sx[k] = sum_r x[r][k],  m = Sx / B,  C[j][k] = sum_r (x[r][j] - m_j)(x[r][k] - m_k)        (double accumulation)
//   mean_c = b_c + w_c . m          var_c = w_c^T C w_c / B                                     (forward statistics)
//   sum_r xhat[r][c] x[r][k] = invstd_c (w_c C)[k]                                               (backward, see finish)
// One launch computes them for ALL minibatches of a chunk (grid = minibatches x nets) right behind the gather, off the
// per-update critical path: the statistics launch of layer 1 and the second stage of its backward disappear.
// moments record (f32): [Sx (KP) | C (KP x KP)], KP = 24 or 32 (columns >= K meet zero weights).
// ------------------------------------------------------------------------------------------------------------
#ifndef BM_MARK
#define BM_MARK(slot) do { } while (0)    // (timeline hook of an including kernel: csrc/step_path.hip under -DNAF_TIMELINE)
#endif
#define BM_CHUNK 256
#define BM_THREADS 512
#define BM_XS 36                 // LDS row stride of a staged chunk: 32 columns (the MFMA tiles of pass 2) + 4 pad
typedef float bm_f32x4 __attribute__((ext_vector_type(4)));
struct BmShared {                // 41,088 bytes per (minibatch, net) group of BM_THREADS threads
    float sX[BM_CHUNK * BM_XS];
    double sRed[16][32];
    float sM[32];
};
// load4(row, q): float4 q (< K4) of the net's input columns of minibatch row `row` (< B). tid = the thread's index inside its
// group of BM_THREADS; every __syncthreads() below is reached by every thread of the WORKGROUP the same number of times (groups
// of one workgroup work on minibatches of the same B).
template <int K4, typename Load4>
__device__ __forceinline__ static void bb_moments_body(Load4 load4, BmShared& S, const int tid, float* __restrict__ out, const int B) {
    constexpr int KP = 4 * K4, XS = BM_XS;
    float* sX = S.sX;
    float* sM = S.sM;
    // a chunk = 256 rows x 8 float4 (the last 8 - K4 of a row are zeros): 4 per thread, ALL requested before the first LDS store
    // (as a load -> store loop the compiler kept one load in flight per trip: four dependent round trips per chunk and pass — 21 of
    // this launch's 21 us at B = 1024). `centre`: subtract the column means (pass 2).
    auto stage = [&](int row0, bool centre) {
        float4 v[BM_CHUNK * 8 / BM_THREADS];
#pragma unroll
        for (int i = 0; i < BM_CHUNK * 8 / BM_THREADS; ++i) {
            const int e = tid + BM_THREADS * i;
            const int r_ = e >> 3, q = e & 7;
            const int row = row0 + r_;
            const bool on = row < B && q < K4;
            v[i] = load4(on ? row : 0, on ? q : 0);
            if (!on) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < BM_CHUNK * 8 / BM_THREADS; ++i) {
            const int e = tid + BM_THREADS * i;
            const int r_ = e >> 3, q = e & 7;
            if (centre && row0 + r_ < B && q < K4) {
                v[i].x -= sM[4 * q + 0];
                v[i].y -= sM[4 * q + 1];
                v[i].z -= sM[4 * q + 2];
                v[i].w -= sM[4 * q + 3];
            }
            *(float4*)(sX + r_ * XS + 4 * q) = v[i];
        }
    };
    // pass 1: column sums (double)
    const int k1 = tid & 31, g1 = tid >> 5;
    double s = 0.0;
    for (int row0 = 0; row0 < B; row0 += BM_CHUNK) {
        __syncthreads();
        stage(row0, false);
        __syncthreads();
        BM_MARK(7);
        if (k1 < KP) {
            // (rows past the batch were staged as zeros: leaving them out adds nothing — the same bits, fewer trips at small B)
            const int rows = B - row0 < BM_CHUNK ? B - row0 : BM_CHUNK;
            float part = 0.f;
            for (int r_ = g1; r_ < rows; r_ += 16) part += sX[r_ * XS + k1];
            s += (double)part;
        }
    }
    S.sRed[g1][k1] = s;
    __syncthreads();
    BM_MARK(8);
    float sx_out = 0.f;                                     // (stored at the end, with the rest of the record)
    if (tid < 32) {
        double t = 0.0;
        for (int g = 0; g < 16; ++g) t += S.sRed[g][tid];
        sM[tid] = (float)(t / (double)B);
        sx_out = (float)t;
    }
    BM_MARK(9);
    // pass 2: centred second moments C = Xc^T Xc on MFMA (v_mfma_f32_16x16x4_f32): the 32 x 32 padding of C is 2 x 2 tiles, wave =
    // (tile, half of the chunk's rows); the chunk is staged CENTRED (columns >= KP zero), a chunk's 128-row partial accumulates in
    // f32 (the data are centred), the running sum over chunks in double; the two row halves meet through LDS, lower + upper.
    // (On the VALU — one triangle entry per thread walking every row — this launch took 59 us per 64 minibatches at B = 2048, 31 at
    // B = 1024: ~1 us per update of the large batches for 1.2 MFLOP.) C/D map: col = lane & 15, row = 4 (lane >> 4) + reg; both
    // off-diagonal tiles are computed, from the same products in the same order: C is symmetric bit for bit.
    {
        const int lane = tid & 63, wave = tid >> 6;
        const int tile = wave & 3, half = wave >> 2, tm = tile >> 1, tn = tile & 1;
        const int r = lane & 15, g = lane >> 4;
        double dacc[4] = {0.0, 0.0, 0.0, 0.0};
        for (int row0 = 0; row0 < B; row0 += BM_CHUNK) {
            __syncthreads();
            if (B <= BM_CHUNK) {
                // one chunk: pass 1 left it in LDS — centre it where it lies (the same subtraction on the same values as staging
                // it again would do; a second trip to memory was 1 of this body's 5 us at the batch sizes of a timestep)
#pragma unroll
                for (int i = 0; i < BM_CHUNK * 8 / BM_THREADS; ++i) {
                    const int e = tid + BM_THREADS * i;
                    const int r_ = e >> 3, q = e & 7;
                    if (r_ < B && q < K4) {
                        float4 v = *(float4*)(sX + r_ * XS + 4 * q);
                        v.x -= sM[4 * q + 0];
                        v.y -= sM[4 * q + 1];
                        v.z -= sM[4 * q + 2];
                        v.w -= sM[4 * q + 3];
                        *(float4*)(sX + r_ * XS + 4 * q) = v;
                    }
                }
            } else {
                stage(row0, true);
            }
            __syncthreads();
            BM_MARK(10);
            bm_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            // The chunk's rows that exist are dealt to the two row halves in whole steps of 4: 128 each for a full chunk, 32 each at
            // B = 64 (with fixed halves of 128 rows the second half's waves had nothing to do there and the first half's issued 32
            // MFMAs each, half of them on zeros: 2 of the 5.6 us of the per-timestep launch, which takes both nets' moments on ONE CU).
            // Rows past the batch are zeros in LDS (they add +0), the last step of a half may run into them.
            const int rows = B - row0 < BM_CHUNK ? B - row0 : BM_CHUNK;
            const int hrows = (((rows + 1) >> 1) + 3) & ~3;                  // <= 128
            const float* pa = sX + (hrows * half + g) * XS + 16 * tm + r;
            const float* pb = sX + (hrows * half + g) * XS + 16 * tn + r;
            int kend = rows - hrows * half;
            kend = kend < 0 ? 0 : (kend > hrows ? hrows : (kend + 3) & ~3);
            // (requesting eight steps' operands ahead of their MFMAs measured SLOWER: 49.7 against 48.7 us per timestep at B = 256,
            //  A/B/A/B on one box — the compiler's own schedule of this loop already overlaps the LDS reads with the matrix pipe)
#pragma unroll 4
            for (int kk = 0; kk < kend; kk += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[kk * XS], pb[kk * XS], acc, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) dacc[e] += (double)acc[e];
        }
        BM_MARK(11);
        __syncthreads();                                    // the last chunk is consumed: sX becomes the halves' meeting place
        double* sD = (double*)sX;                           // [tile][lane][4]
        if (half) {
#pragma unroll
            for (int e = 0; e < 4; ++e) sD[(tile * 64 + lane) * 4 + e] = dacc[e];
        }
        __syncthreads();
        BM_MARK(12);
        if (!half) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = 16 * tm + 4 * g + e, col = 16 * tn + r;
                if (row < KP && col < KP) out[KP + row * KP + col] = (float)(dacc[e] + sD[(tile * 64 + lane) * 4 + e]);
            }
        }
    }
    if (tid < KP) out[tid] = sx_out;
}
