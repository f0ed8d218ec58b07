// Layer 1 of the row-split chain (csrc/big_batch.hip) as a device function, so that TWO launches can run it: bb_layer1_kernel — the
// chain's first launch, with or without the previous update's optimizer step riding on it — and adam_act_kernel of
// csrc/step_path.hip, where it rides on the per-timestep path's first launch behind that launch's own optimizer step (round 6: the
// chain of a pipelined timestep then starts at GEMM 2, one launch boundary and the act() tail earlier). One body, the same bits.
#pragma once
#include "common.h"
#include "adam_body.h"
#include "../../include/naf_hip.h"

#ifndef BB_ROWS
#define BB_ROWS NAF_BB_ROWS      // rows per statistics block
#define BB_COLS 64               // feature columns per workgroup (row-split kernels)
#define BB_THREADS 256
#define BB_MAX_K4 8              // layer 1: K <= 32
typedef float f32x4 __attribute__((ext_vector_type(4)));
#endif
// timeline hooks of the including file (csrc/big_batch.hip under -DNAF_TIMELINE); nothing elsewhere
#ifndef BB_L1_TL
#define BB_L1_TL(slot, is_first, is_last) do { } while (0)
#define BB_L1_TL_T(slot, is_first, is_last, thread) do { } while (0)
#endif

// Rows by eighths: workgroup w of a launch runs on XCD (w + shift) % 8 (round-robin dispatch; `shift` = the workgroups in front of
// these in the grid), and XCD x is given the rows [x B/8, (x+1) B/8) of BOTH networks in every launch of the chain — the 64-row
// blocks of layer 1 and GEMM 2 here, the 16-row chunks of bb_layer2_head, the 32-row blocks of the bundle's dA1 product and the K
// ranges of its dW2 product — so that what a launch reads of the previous launch's output is still in ITS OWN L2 (lines written by
// a launch stay valid there behind the boundary; data from another XCD comes back over the fabric, ~2.5 us instead of ~1).
// NB = 64-row blocks per network (a multiple of 8: B = 512, 1024, 1536, 2048), GY = column tiles: XCD x takes blocks
// [x NB/8, (x+1) NB/8), all tiles, both networks. Updates/s, A/B/A/B on one box: B = 512 30.7k -> 31.2k, 1024 26.5k -> 26.9k,
// 2048 21.05k -> 21.35k. (Smaller batches, where a block straddles 8 / NB eighths and its tiles would be dealt to those XCDs:
// nothing at B = 256, -1.8 % at 64 — left alone.) Placement is speed only; false = leave the workgroup's own indices as they are.
__device__ __forceinline__ static bool bb_place_rows(int w, int shift, int NB, int GY, int& net, int& rb, int& ct) {
    if (NB < 8 || (NB & 7)) return false;
    const int s = shift & 7, x = (w + s) & 7;
    const int slot = ((w + s) >> 3) - (x < s ? 1 : 0);     // the slot-th workgroup of these on XCD x
    const int G = NB >> 3, per = G * GY;
    net = slot / per;
    const int rem = slot - net * per;
    ct = rem / G;
    rb = x * G + (rem - ct * G);
    return true;
}

// ------------------------------------------------------------------------------------------------------------
// layer 1 (K = state size <= 32): z tile of 64 rows x 64 columns, thread (ty = tid >> 4, tx = tid & 15) owns rows
// 4 ty .. +3 and columns 4 tx .. +3. Operands go through LDS TRANSPOSED ([k][row], [k][column]) so that a thread's four
// rows / four columns are one 16-byte LDS read per k. z = b + sum_k x_k w_k, k ascending — the SAME code in the
// statistics launch, the normalising launch and the backward, so z is the same bits everywhere.
// ------------------------------------------------------------------------------------------------------------
template <int K4>
__device__ static inline void bb_l1_tile(const float (*sXt)[BB_ROWS + 4], const float (*sWt)[BB_COLS + 4], const float4 bias4,
                                         int ty, int tx, float (&z)[4][4]) {
    const float bb[4] = {bias4.x, bias4.y, bias4.z, bias4.w};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) z[i][j] = bb[j];
#pragma unroll
    for (int k = 0; k < 4 * K4; ++k) {
        const float4 a = *(const float4*)&sXt[k][4 * ty];
        const float4 w = *(const float4*)&sWt[k][4 * tx];
        const float av[4] = {a.x, a.y, a.z, a.w}, wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) z[i][j] = __builtin_fmaf(av[i], wv[j], z[i][j]);
    }
}

// layer 1 forward for `nets` networks: statistics from the moments, z tile, normalise, ReLU -> out
// Prologue: every global operand — the row tile, the 64 columns' weights (ONE contiguous run of 64 K floats, read as float4
// and scattered to [k][column] in LDS), the moments record, the per-column parameters — is requested before the first LDS
// store. (As `for (e = tid; ...) lds[..] = global[..]` loops the compiler kept one load in flight per trip: nine dependent
// round trips, 2.5 of this kernel's 5.0 us — benchmarks/kernel_timeline.py.)
// ADAM (the deferred optimizer step, adam_body.h): the grid is the n_main layer-1 workgroups plus extra workgroups that step
// floats [4 l1_4, 4 n4) of the flat buffers in place; the layer-1 workgroups request, with their operands, what the step needs
// for the parameters they read (gradient, moments, the main network's old value for the target's workgroups) and evaluate
// them as the step will leave them. Costs one more barrier (the clip scale must exist before the weights go to LDS).
// The grid is one-dimensional: workgroup = rb + B/64 (column tile + H/64 net), the order the 3-D grid had.
// FULL: the batch is whole 64-row blocks (every BASELINE config) — `valid` is then the constant 64 and the row masks, clamps and
// resource bounds of the partial last block fold away at compile time (with them at run time an update at B = 256 was 0.3 us slower)
// (the kernel's body as a device function: bb_layer1_kernel below is it on a launch of its own; adam_act_kernel of csrc/step_path.hip
//  runs it — ADAM = false — in extra workgroups of the per-timestep path's first launch, behind that launch's own optimizer step)
template <int K4, bool ADAM>
struct BbL1Shared {
    static constexpr int KP = 4 * K4, REC = KP + KP * KP;
    __attribute__((aligned(16))) float sXt[KP][BB_ROWS + 4];
    __attribute__((aligned(16))) float sWt[KP][BB_COLS + 4];
    __attribute__((aligned(16))) float sMom[REC];
    float sStat[4][BB_COLS];
    AdamScalars shA[ADAM ? 1 : 0];                                        // (the riding optimizer step's scalars and the parameters as
    __attribute__((aligned(16))) float sPar[ADAM ? 3 : 0][BB_COLS];       //  it leaves them: ADAM only)
};
// Hooks — nothing in the launches of csrc/big_batch.hip; the riders of adam_act_kernel (csrc/step_path.hip) use both:
//   parameters_loaded()     one thread, once every parameter (and, ADAM, its gradient and optimizer state) this workgroup reads has
//                           been consumed: the launch it rides on may overwrite them from here on
//   before_running_stats()  the lanes about to WRITE the running statistics (row block 0, one lane per column): that launch's own
//                           readers of the statistics must be through
struct BbL1NoHook {
    __device__ __forceinline__ void parameters_loaded() const {}
    __device__ __forceinline__ void before_running_stats() const {}
};
template <int K4, bool ADAM, bool FULL, typename Hook = BbL1NoHook>
__device__ __forceinline__ static void bb_layer1_impl(BbL1Shared<K4, ADAM>& SM, const int block,
    const float* __restrict__ x, int64_t x_net_stride, int ldx, int K, const float* __restrict__ W,
    const float* __restrict__ bias, const float* __restrict__ gamma, const float* __restrict__ beta,
    int64_t param_net_stride, const float* __restrict__ mom, float* __restrict__ running_mean,
    float* __restrict__ running_var, int64_t stat_net_stride, float* __restrict__ out, int64_t out_net_stride, int ldo,
    float* __restrict__ save_mean, float* __restrict__ save_invstd, float* __restrict__ wc_out, int B, int H, float momentum,
    float eps, int n_main, const AdamArgs ad, int64_t l1_4, int64_t n4, int n_adam, int xcd_rows, float* __restrict__ xhat_out,
    const Hook hook = Hook()) {
    constexpr int KP = 4 * K4, REC = KP + KP * KP;
    constexpr int XN = (BB_ROWS * K4 + BB_THREADS - 1) / BB_THREADS;           // float4 of the row tile per thread: 2
    constexpr int MN = (REC / 4 + BB_THREADS - 1) / BB_THREADS;                // of the moments record: 1 or 2
    float (&sXt)[KP][BB_ROWS + 4] = SM.sXt;
    float (&sWt)[KP][BB_COLS + 4] = SM.sWt;
    float (&sMom)[REC] = SM.sMom;
    float (&sStat)[4][BB_COLS] = SM.sStat;  // mean, invstd, gamma, beta of this workgroup's columns
    AdamScalars& shA = SM.shA[0];           // (ADAM only: zero-length arrays otherwise, never touched)
    float (*sPar)[BB_COLS] = SM.sPar;       // ADAM: b, gamma, beta of the columns as the step leaves them
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    // every kernel argument, requested NOW in one batch of scalar loads: fetched where they are first used they came in five
    // dependent batches threaded through the prologue (~0.25 us each against a scalar cache that is cold when a launch starts), each
    // holding back the vector loads behind it
    asm volatile("" ::"s"(x), "s"(x_net_stride), "s"(ldx), "s"(K), "s"(W), "s"(bias), "s"(gamma), "s"(beta), "s"(param_net_stride),
                 "s"(mom), "s"(running_mean), "s"(running_var), "s"(stat_net_stride), "s"(out), "s"(out_net_stride), "s"(ldo),
                 "s"(save_mean), "s"(save_invstd), "s"(wc_out), "s"(B), "s"(H), "s"(momentum), "s"(eps), "s"(n_main), "s"(l1_4),
                 "s"(n4), "s"(n_adam));
    if (ADAM)
        asm volatile("" ::"s"(ad.theta), "s"(ad.g), "s"(ad.m), "s"(ad.v), "s"(ad.target), "s"(ad.partials), "s"(ad.n_partials),
                     "s"(ad.max_norm), "s"(ad.lr), "s"(ad.beta1), "s"(ad.beta2), "s"(ad.eps), "s"(ad.tau), "s"(ad.one_minus_tau),
                     "s"(ad.step_dev), "s"(ad.inv_world), "s"(ad.bc));
    // grid: [riding optimizer step | this kernel's own workgroups]
    const int n_ride = ADAM ? n_adam : 0;
    const int widx = block - n_ride;                       // >= 0: a layer-1 workgroup
    if (ADAM && __builtin_expect(widx < 0, 0)) {           // (unlikely: the riding code sits behind the kernel's own)
        const int ra = block;
        BB_L1_TL(11, ra == 0, ra == n_adam - 1);    // (raw slots 11, 12: the riding step)
        adam_block<2 * BB_THREADS>(ad, (size_t)l1_4, (size_t)n4, ra, n_adam, &shA, tid, true);
        BB_L1_TL(12, ra == 0, ra == n_adam - 1);
        return;
    }
    const int gx = (B + BB_ROWS - 1) / BB_ROWS, gy = H / BB_COLS;
    int rb = widx % gx, ct_ = (widx / gx) % gy, net = widx / (gx * gy);
    if (xcd_rows && n_main == 2 * gx * gy) bb_place_rows(widx, n_ride, gx, gy, net, rb, ct_);
    const int col0 = ct_ * BB_COLS;
    const int64_t po = net * param_net_stride;
#define L1_TL(slot) BB_L1_TL(slot, widx == 0, widx == n_main - 1)
    L1_TL(0);
    const float* xn = x + net * x_net_stride + (int64_t)rb * BB_ROWS * ldx;
    // rows of this block that exist (the last block of a batch that is not whole 64-row blocks holds fewer): rows past them are
    // read as copies of row 0 and never stored (the output resources end at the last row that exists)
    const int valid = FULL ? BB_ROWS : (B - rb * BB_ROWS < BB_ROWS ? B - rb * BB_ROWS : BB_ROWS);
    // ADAM: the workgroup has 512 threads. Threads 0 .. 255 are the layer-1 workgroup as ever; ALL 512 take part in evaluating
    // the parameters as the pending step leaves them (one float4 = four elements per thread: the update formula is ~100
    // instructions per element — a division and a square root, correctly rounded), then waves 4 .. 7 are done.
    const bool mainw = !ADAM || tid < BB_THREADS;
    f32x4 xv[XN], wv[2], mv[MN];
    const int wn4 = (BB_COLS * K) >> 2;                                          // 16 K float4 (<= 512)
    const f32x4* wsrc = (const f32x4*)(W + po + (int64_t)col0 * K);
    // the statistics of column cl are finished by lane (r < 4, g) of wave cl / 16 (see below): its constants
    const int lane = tid & 63, wave = (tid >> 6) & 3, mr = lane & 15, mg = lane >> 4;
    const int cl = 16 * wave + 4 * mg + (mr & 3);
    float4 b4v = make_float4(0.f, 0.f, 0.f, 0.f);
    float bias_cv = 0.f, gmv = 0.f, btv = 0.f;
    float rm_ = 0.f, rv_ = 0.f;
    if (mainw) {
#pragma unroll
        for (int i = 0; i < XN; ++i) {
            const int e = tid + BB_THREADS * i;
            const int row = e / K4, q = e - row * K4;
            xv[i] = ((const f32x4*)(xn + (int64_t)(row < valid ? row : 0) * ldx))[q];
        }
    }
    if (!ADAM) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + BB_THREADS * i;
            wv[i] = wsrc[e < wn4 ? e : 0];
        }
    } else {
        wv[0] = wsrc[tid < wn4 ? tid : 0];
    }
    if (mainw) {
#pragma unroll
        for (int i = 0; i < MN; ++i) {
            const int e = tid + BB_THREADS * i;
            mv[i] = ((const f32x4*)(mom + (int64_t)net * REC))[e < REC / 4 ? e : 0];
        }
        if (!ADAM) {
            b4v = *(const float4*)(bias + po + col0 + 4 * tx);
            bias_cv = bias[po + col0 + cl];
            gmv = gamma[po + col0 + cl];
            btv = beta[po + col0 + cl];
        }
        if (rb == 0 && mr < 4) {
            rm_ = running_mean[net * stat_net_stride + col0 + cl];
            rv_ = running_var[net * stat_net_stride + col0 + cl];
        }
    }
    // deferred step: everything it needs for the parameters this workgroup reads, in flight with the operands above. Every
    // parameter is evaluated ONCE per workgroup and handed on through LDS (evaluated where they are used — the per-column
    // parameters 16 x redundantly — a thread ran the formula 15 times: 2.6 us in front of the first barrier): thread t takes
    // float4 t of the weight tile (t < 16 K), the last 48 threads one float4 each of [b | gamma | beta].
    const bool tgt = net != 0;
    constexpr int NPAR4 = 3 * BB_COLS / 4;
    const int pj = tid - (2 * BB_THREADS - NPAR4);            // ADAM: >= 0 for the threads that take a parameter float4
    AdamFly4 fw, fp;
    f32x4 pcur;
    AdamPrefetch apf;
    int64_t ofw = 0, ofp = 0;                                 // flat offsets of this thread's two float4 (main network)
    if (ADAM) {
        ofw = (W - ad.theta) + (int64_t)col0 * K + 4 * (tid < wn4 ? tid : 0);
        fw = adam_fly_load4(ad, ofw, tgt);
        const int jc = pj >= 0 ? pj : 0, c4 = 4 * (jc & (BB_COLS / 4 - 1));
        const float* pb = jc < BB_COLS / 4 ? bias : jc < BB_COLS / 2 ? gamma : beta;
        pcur = *(const f32x4*)(pb + po + col0 + c4);
        ofp = (pb - ad.theta) + col0 + c4;
        fp = adam_fly_load4(ad, ofp, tgt);
        apf = adam_prefetch(ad, tid);
    }
    // rows k >= K of the weight tile meet the padding columns of the row tile: zero
    for (int e = tid; e < (KP - K) * BB_COLS; e += BB_THREADS) sWt[K + e / BB_COLS][e % BB_COLS] = 0.f;
    if (mainw) {
#pragma unroll
        for (int i = 0; i < XN; ++i) {
            const int e = tid + BB_THREADS * i;
            const int row = e / K4, q = e - row * K4;
            if (row < BB_ROWS) {
                sXt[4 * q + 0][row] = xv[i][0];
                sXt[4 * q + 1][row] = xv[i][1];
                sXt[4 * q + 2][row] = xv[i][2];
                sXt[4 * q + 3][row] = xv[i][3];
            }
        }
#pragma unroll
        for (int i = 0; i < MN; ++i) {
            const int e = tid + BB_THREADS * i;
            if (e < REC / 4) ((f32x4*)sMom)[e] = mv[i];
        }
    }
    if (ADAM) {
        L1_TL(7);
        adam_derive(ad, apf, &shA, tid);
        L1_TL(8);
        __syncthreads();
        L1_TL(9);
        const AdamScalars sc = shA;
        if (tid < wn4) wv[0] = adam_fly_apply4(ad, sc, fw, wv[0], tgt);
        if (pj >= 0) *(f32x4*)(&sPar[0][0] + 4 * pj) = adam_fly_apply4(ad, sc, fp, pcur, tgt);
        L1_TL(10);
    }
    {
        const unsigned kinv = (65536u + (unsigned)K - 1u) / (unsigned)K;        // i / K for i < 64 K <= 2048: exact (K <= 32)
#pragma unroll
        for (int i = 0; i < (ADAM ? 1 : 2); ++i) {
            const int e = tid + BB_THREADS * i;
            if (e < wn4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned idx = 4u * (unsigned)e + (unsigned)q;
                    const unsigned c = (idx * kinv) >> 16;
                    sWt[idx - c * (unsigned)K][c] = wv[i][q];
                }
            }
        }
    }
    __syncthreads();
    if (tid == 0) hook.parameters_loaded();
    float* oz = out + net * out_net_stride;
    // normalise + ReLU + store of a thread's 4 x 4 piece of the tile (statistics in sStat)
    auto tile_out = [&](const float (&zt)[4][4], int ty_, int tx_) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 y;
            float* yp = (float*)&y;
            f32x4 xh4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = 4 * tx_ + j;
                const float xh = (zt[i][j] - sStat[0][c]) * sStat[1][c];
                const float t = __builtin_fmaf(xh, sStat[2][c], sStat[3][c]);     // (the ReLU decision the backward repeats from xhat)
                xh4[j] = xh;
                yp[j] = t > 0.f ? t : 0.f;
            }
            const unsigned obytes = FULL ? 0x7fffffffu : ((unsigned)(valid - 1) * (unsigned)ldo + BB_COLS) * 4u;
            naf_buf_st_f4(naf_buf(oz + (int64_t)(rb * BB_ROWS) * ldo + col0, obytes), 4u * (unsigned)((4 * ty_ + i) * ldo + 4 * tx_), 0,
                          (f32x4){y.x, y.y, y.z, y.w}, B >= NAF_WT_MIN_B);
            // the main network's xhat too where the backward wants it ready-made (naf_gemm_l1bwd_t.xhat: small batches)
            if (xhat_out && net == 0)
                naf_buf_st_f4(naf_buf(xhat_out + (int64_t)(rb * BB_ROWS) * ldo + col0, obytes), 4u * (unsigned)((4 * ty_ + i) * ldo + 4 * tx_), 0, xh4,
                              B >= NAF_WT_MIN_B);
        }
    };
    if (ADAM && tid >= BB_THREADS) {
        // waves 4 .. 7: the z tile, WHILE waves 0 .. 3 derive the statistics — neither needs the other, only the normalisation
        // needs both (in one sequence in the same four waves: 1.24 + 0.68 us of this kernel's 5.0; side by side: 1.24)
        const int ty2 = ty - BB_THREADS / 16;
        float z[4][4];
        bb_l1_tile<K4>(sXt, sWt, *(const float4*)&sPar[0][4 * tx], ty2, tx, z);
        __syncthreads();                         // the statistics of waves 0 .. 3 are in sStat
        BB_L1_TL_T(3, widx == 0, widx == n_main - 1, BB_THREADS);
        tile_out(z, ty2, tx);
        BB_L1_TL_T(4, widx == 0, widx == n_main - 1, BB_THREADS);
        return;
    }
    if (ADAM) {
        b4v = *(const float4*)&sPar[0][4 * tx];
        bias_cv = sPar[0][cl];
        gmv = sPar[1][cl];
        btv = sPar[2][cl];
    }
    L1_TL(1);
    {
        // statistics of the 64 columns from the moments on MFMA: U = W C (64 x KP; wave w owns columns 16 w .. +15, both
        // 16-wide halves of the KP dimension), then var_c B = U[c] . w_c and (mean_c - b_c) B = w_c . Sx as 16-lane reductions
        // of the accumulator rows. A[m = column][k] = sWt[k][column]; B[k][n] = C[k][n] = C[n][k] (symmetric): one 16-byte
        // read of row n. (On the VALU — 4 threads per column, 144 FMAs each on LDS operands — this was 1.5 of the kernel's
        // 3.7 us: benchmarks/kernel_timeline.py.)
        const float* sC = sMom + KP;
        f32x4 u0 = {0.f, 0.f, 0.f, 0.f}, u1 = u0;
        const bool hi = 16 + mr < KP;                        // KP = 24: rows 24 .. 31 of C do not exist
        const float* c0 = sC + mr * KP, *c1 = sC + (hi ? 16 + mr : 0) * KP;
#pragma unroll
        for (int kk = 0; kk < KP; kk += 16) {
            const bool in = kk + 4 * mg < KP;                // KP = 24: lane groups 2, 3 of the second step are past K
            const int ko = in ? kk + 4 * mg : 0;
            f32x4 b0 = *(const f32x4*)(c0 + ko), b1 = *(const f32x4*)(c1 + ko);
            if (!hi) b1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float a = in ? sWt[ko + q][16 * wave + mr] : 0.f;
                u0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0[q], u0, 0, 0, 0);
                u1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1[q], u1, 0, 0, 0);
            }
        }
        L1_TL(5);
        // lane (mr, mg) holds U[column 16 w + 4 mg + e][n = mr] (u0) and [n = 16 + mr] (u1)
        if (wc_out && net == 0 && rb == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float* dst = wc_out + (int64_t)(col0 + 16 * wave + 4 * mg + e) * KP;
                dst[mr] = u0[e];
                if (hi) dst[16 + mr] = u1[e];
            }
        }
        const float sx0 = sMom[mr], sx1 = hi ? sMom[16 + mr] : 0.f;
        float t[4], md[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 16 * wave + 4 * mg + e;
            const float w0 = sWt[mr][c], w1 = hi ? sWt[16 + mr][c] : 0.f;
            t[e] = u0[e] * w0 + u1[e] * w1;
            md[e] = w0 * sx0 + w1 * sx1;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            t[e] = naf_sum16(t[e]);
            md[e] = naf_sum16(md[e]);
        }
        L1_TL(6);
        if (mr < 4) {                                        // lane mr of the group finishes column 4 mg + mr (= cl)
            const float tt = mr == 0 ? t[0] : mr == 1 ? t[1] : mr == 2 ? t[2] : t[3];
            const float mm = mr == 0 ? md[0] : mr == 1 ? md[1] : mr == 2 ? md[2] : md[3];
            const float mean = bias_cv + mm / (float)B;
            const float var = fmaxf(tt, 0.f) / (float)B;
            const int c = cl, col = col0 + c;
            const float invstd = 1.0f / sqrtf(var + eps);
            sStat[0][c] = mean;
            sStat[1][c] = invstd;
            sStat[2][c] = gmv;
            sStat[3][c] = btv;
            if (rb == 0) {
                hook.before_running_stats();
                const int64_t so = net * stat_net_stride + col;
                const float unbiased = B > 1 ? var * ((float)B / (float)(B - 1)) : var;
                running_mean[so] = (1.0f - momentum) * rm_ + momentum * mean;
                running_var[so] = (1.0f - momentum) * rv_ + momentum * unbiased;
                save_mean[(int64_t)net * H + col] = mean;
                save_invstd[(int64_t)net * H + col] = invstd;
            }
        }
    }
    L1_TL(2);
    if (ADAM) {                                  // (waves 4 .. 7 normalise and store the tile they computed meanwhile)
        __syncthreads();
        return;
    }
    float z[4][4];
    bb_l1_tile<K4>(sXt, sWt, b4v, ty, tx, z);
    __syncthreads();
    L1_TL(3);
    tile_out(z, ty, tx);
    L1_TL(4);
#undef L1_TL
}

