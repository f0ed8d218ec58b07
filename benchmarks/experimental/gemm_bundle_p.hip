// The backward GEMM bundle of learn() for LARGE batches (B >= 1024: BASELINE configs[3], [4]) as PERSISTENT workgroups on
// v_mfma_f32_16x16x4_f32 (gfx950): dA1 = dZ2 W2, dW2 = dZ2^T A1, dWh = dH^T A2 — autograd of naf_neural_network.py:76-87 —
// with the second stage of layer 2's BatchNorm backward as the prologue of the operands that are dZ2 and the batch pass of
// layer 1's backward as the epilogue of the dA1 tiles, exactly as csrc/gemm_bundle.hip does it on 32 x 32 blocks.
//
// Why a second form. With one 32 x 32 block per workgroup (two resident per CU) every block ran load -> LDS -> MFMA -> epilogue
// in sequence and all blocks of a round did so in lock-step: the launch alternated between a phase in which every CU pulled
// its panels out of L2 (48 MB in 3 us at B = 2048: the L2 -> CU rate of the chip) and a phase in which every CU computed — 21 %
// of the f32-MFMA peak at B = 2048, 14.6 us (profiles/r02_timeline_b2048.txt). Here
//   * one workgroup per CU (512 threads, 8 waves, up to 256 registers each) walks a LIST of blocks the host planned
//     (naf_gemm_bundle_p_plan: longest blocks first, least-loaded workgroup of the XCD whose L2 holds the block's operands);
//   * a block is a 64 x 64 output tile over a K range: each staged byte feeds twice the MFMAs of a 32 x 32 block (the L2 -> CU path
//     carries ~70 GB/s per CU; a 32 x 32 x 256 block needs 64-96 KB for 0.85 us of MFMA — more than that path delivers);
//   * K runs in chunks of 128 through TWO LDS buffers (2 x 69.6 KB): in every trip of ONE flat loop over all chunks of all blocks
//     of the workgroup the wave stores chunk u (registers -> buffer u & 1), requests chunk u + 1 from memory (registers), and only
//     then runs the MFMAs of chunk u - 1 out of the other buffer: global latency, LDS stores and MFMAs of three consecutive
//     chunks overlap, ACROSS block boundaries too (the next block's first panel lands while this block computes and finishes);
//     one barrier per chunk.
// Summation order over k is fixed (chunk order, 16-k steps, two accumulators per tile added at the end): bitwise reproducible.
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "common.h"
#include "../../include/naf_hip.h"
#include "bn2bwd_fold.h"

#define GP_THREADS 512
#define GP_BM 64
#define GP_BN 64
#define GP_KC 128
#define GP_LDA (GP_KC + 4)             // k-contiguous A panel [64 rows][132]: 16-B aligned rows, b128 fragment reads spread over the banks
#define GP_LDK (GP_BN + 4)             // k-major panels [128 k][68]: bank = 4 k + column, the two 16-lane groups of a b32 read never collide
#define GP_PANEL (GP_KC * GP_LDK)      // floats per panel (>= 64 * GP_LDA)
#define GP_BUF (2 * GP_PANEL)          // one buffer = A panel + B panel
static_assert(GP_THREADS == GB_THREADS, "the fold / poll helpers of bn2bwd_fold.h run on this workgroup");
static_assert(64 * GP_LDA <= GP_PANEL, "k-contiguous A panel fits a panel slot");

// One product, exactly 64 bytes on a 64-byte boundary inside the kernel arguments: `bundle.d[gi]` with a run-time gi is then ONE
// s_load_dwordx16. (As a 88-byte struct whose fields were fetched where first used — behind the branches of gp_load — a wave
// spent 1.7 us on five dependent scalar round trips before it requested its first panel, and again in every trip of the loop:
// benchmarks/kernel_timeline.py, GP_TL_STARTUP.)
#define GP_F_AK 1
#define GP_F_PRO 2
#define GP_F_EPI 4
struct __attribute__((aligned(64))) GpDesc {
    const float* A;   // a_kmajor ? [K][M] : [M][K]
    const float* B;   // [K][N]
    float* C;         // [M][N] (slab s at C + s * cstride); NULL with the epilogue
    int M, N, lda, ldb, ldc, kper, flags, cstride;    // kper = K / k_split: K range s is [s kper, (s + 1) kper)
    int pad[2];
};
static_assert(sizeof(GpDesc) == 64, "one s_load_dwordx16");
// the desc BY VALUE, all of it requested now (the empty asm pins every field: nothing is fetched later, behind a branch)
__device__ __forceinline__ static GpDesc gp_desc(const GpDesc* d, int gi) {
    GpDesc D = d[__builtin_amdgcn_readfirstlane(gi)];     // (uniform by construction: say so, or the fetch becomes a vector load)
    asm volatile("" ::"s"(D.A), "s"(D.B), "s"(D.C), "s"(D.M), "s"(D.N), "s"(D.lda), "s"(D.ldb), "s"(D.ldc), "s"(D.kper), "s"(D.flags),
                 "s"(D.cstride));
    return D;
}
// a block = one packed word: gi (2 bits) | bm << 2 (8) | bn << 10 (6) | K range << 16 (8); -1: end of the list
#define GP_CODE(gi, bm, bn, s) ((gi) | ((bm) << 2) | ((bn) << 10) | ((s) << 16))
#define GP_MAX_WG_IN_ARGS 256
struct GpBundle {
    GpDesc d[NAF_GEMM_BUNDLE_MAX];
    naf_gemm_bn2bwd_t pro;    // ONE prologue for every product flagged has_pro (they read the same dY2 / Z2)
    naf_gemm_l1bwd_t epi;     // of the product flagged has_epi
    const int* plan;          // [n_wg][NAF_GEMM_P_MAX_BLOCKS] block codes
    int n, n_fold, any_pro;
    // the FIRST block of every workgroup rides in the kernel arguments (n_wg <= 256): its panels are requested two dependent
    // scalar loads after the wave starts instead of three, one of them a miss in L2 (the plan in memory: 1.5 us before the first
    // byte was asked for — benchmarks/kernel_timeline.py)
    int first[GP_MAX_WG_IN_ARGS];
};

struct GpRegs {
    f32x4 a[4], z[4], b[4];
};

// development aid (NAF_BUILD_DEFINES=-DNAF_TIMELINE): 16 wall-clock marks per workgroup — 0 entry, 1 first loads requested,
// 2 constants in LDS, then per trip of the loop (while slots last) 3 + 3 j: chunk stored + next requested, 4 + 3 j: MFMAs (and
// the block's end) done, 5 + 3 j: behind the barrier; 15: exit. naf_timeline_read(2048 + w) copies workgroups w and w + 1.
#define GP_TL_WGS 512
#ifdef NAF_TIMELINE
__device__ long long g_tl_gp[GP_TL_WGS][NAF_TL_SLOTS];
#define GP_TL(slot)                                                                                              \
    do {                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        if (threadIdx.x == 0 && blockIdx.x < GP_TL_WGS && (slot) < NAF_TL_SLOTS) g_tl_gp[blockIdx.x][(slot)] = wall_clock64(); \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
    } while (0)
int naf_tl_read_gp(int w, long long* out) {
    if (w < 0 || w + 2 > GP_TL_WGS) return NAF_ERR_ARG;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tl_gp), 2 * NAF_TL_SLOTS * sizeof(long long),
                                    (size_t)w * NAF_TL_SLOTS * sizeof(long long), hipMemcpyDeviceToHost);
}
#else
#define GP_TL(slot) do { } while (0)
int naf_tl_read_gp(int, long long*) { return NAF_ERR_STATE; }
#endif
struct GpUnit {            // one chunk of one block (all fields wave-uniform)
    int gi, m0, n0, k0, k_hi, slab, first, last;
};

// ---- global -> registers: the chunk's panels, 4 float4 per thread and panel, every address part that can be on the scalar unit
__device__ __forceinline__ static void gp_load(GpRegs& R, const GpDesc& D, const naf_gemm_bn2bwd_t& P, const GpUnit& U, int tid) {
    const bool pro = (D.flags & GP_F_PRO) != 0;
    if (D.flags & GP_F_AK) {      // [K][M]: k = k0 + (tid >> 4) + 32 i, columns m0 + 4 (tid & 15) .. +3; rows past the K range read as 0
        const unsigned ld4 = (unsigned)D.lda * 4u;
        const int c = U.m0 + 4 * (tid & 15);
        const unsigned voff = c < D.M ? (unsigned)(tid >> 4) * ld4 + (unsigned)c * 4u : 0x7f000000u;
        const __amdgpu_buffer_rsrc_t ra = naf_buf(D.A, (unsigned)U.k_hi * ld4);
#pragma unroll
        for (int i = 0; i < 4; ++i) R.a[i] = naf_buf_f4(ra, voff, (unsigned)(U.k0 + 32 * i) * ld4);
        if (pro) {
            const __amdgpu_buffer_rsrc_t rz = naf_buf(P.z, (unsigned)U.k_hi * ld4);
#pragma unroll
            for (int i = 0; i < 4; ++i) R.z[i] = naf_buf_f4(rz, voff, (unsigned)(U.k0 + 32 * i) * ld4);
        }
    } else {               // [M][K]: row m0 + (tid >> 5) + 16 i, k = k0 + 4 (tid & 31) .. +3; rows past M read as 0
        const unsigned ld4 = (unsigned)D.lda * 4u;
        const unsigned voff = (unsigned)(tid >> 5) * ld4 + (unsigned)(tid & 31) * 16u;
        const __amdgpu_buffer_rsrc_t ra = naf_buf(D.A, (unsigned)D.M * ld4);
#pragma unroll
        for (int i = 0; i < 4; ++i) R.a[i] = naf_buf_f4(ra, voff, (unsigned)(U.m0 + 16 * i) * ld4 + (unsigned)U.k0 * 4u);
        if (pro) {
            const __amdgpu_buffer_rsrc_t rz = naf_buf(P.z, (unsigned)D.M * ld4);
#pragma unroll
            for (int i = 0; i < 4; ++i) R.z[i] = naf_buf_f4(rz, voff, (unsigned)(U.m0 + 16 * i) * ld4 + (unsigned)U.k0 * 4u);
        }
    }
    {
        const unsigned ld4 = (unsigned)D.ldb * 4u;
        const int c = U.n0 + 4 * (tid & 15);
        const unsigned voff = c < D.N ? (unsigned)(tid >> 4) * ld4 + (unsigned)c * 4u : 0x7f000000u;
        const __amdgpu_buffer_rsrc_t rb = naf_buf(D.B, (unsigned)U.k_hi * ld4);
#pragma unroll
        for (int i = 0; i < 4; ++i) R.b[i] = naf_buf_f4(rb, voff, (unsigned)(U.k0 + 32 * i) * ld4);
    }
}

// ---- registers -> LDS, dY2 -> dZ2 on the way (cst: [4][256] = mean, k1, k1 c1, invstd k1 c2 per layer-2 feature)
__device__ __forceinline__ static void gp_store(const GpRegs& R, const int flags, const GpUnit& U, float* sA, float* sB,
                                                const float* cst, int tid) {
    const bool a_kmajor = (flags & GP_F_AK) != 0;
    f32x4 va[4] = {R.a[0], R.a[1], R.a[2], R.a[3]};
    if (flags & GP_F_PRO) {
        // the thread's four floats are four consecutive FEATURES: of the block's columns (k-major A) or of the chunk's k (k-contiguous)
        const int ci = a_kmajor ? U.m0 + 4 * (tid & 15) : U.k0 + 4 * (tid & 31);
        const f32x4 mean = *(const f32x4*)(cst + ci), k1 = *(const f32x4*)(cst + 256 + ci), kc1 = *(const f32x4*)(cst + 512 + ci),
                    q = *(const f32x4*)(cst + 768 + ci);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) va[i][j] = __builtin_fmaf(k1[j], va[i][j], -kc1[j]) - (R.z[i][j] - mean[j]) * q[j];
    }
    if (a_kmajor) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *(f32x4*)(sA + ((tid >> 4) + 32 * i) * GP_LDK + 4 * (tid & 15)) = va[i];
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) *(f32x4*)(sA + ((tid >> 5) + 16 * i) * GP_LDA + 4 * (tid & 31)) = va[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) *(f32x4*)(sB + ((tid >> 4) + 32 * i) * GP_LDK + 4 * (tid & 15)) = R.b[i];
}

// ---- one chunk of MFMAs: wave (wm, wn) owns rows 16 wm .. +15 x columns 32 wn .. +31 (two 16-wide tiles), lane (r, g) takes
// k = kk + 4 g .. +3 of row / column r of each operand: one A fragment feeds both tiles. The fragments of step kk + 16 are read
// BEFORE the MFMAs of step kk issue (software pipeline, fully unrolled): with the workgroup's halves in opposite phases a wave is
// alone on its SIMD while it multiplies, and an in-order wave that reads, waits and then multiplies left the matrix pipe idle for
// every LDS round trip (2.3 - 2.7 us per chunk where the MFMAs take 0.9: benchmarks/kernel_timeline.py).
struct GpFrag {
    f32x4 a, b0, b1;
};
template <bool AK>
__device__ __forceinline__ static GpFrag gp_frag(const float* __restrict__ sA, const float* __restrict__ sB, int wm, int wn, int r, int g, int kk) {
    GpFrag f;
    if (AK) {
        const float* q = sA + (kk + 4 * g) * GP_LDK + 16 * wm + r;
        f.a = (f32x4){q[0], q[GP_LDK], q[2 * GP_LDK], q[3 * GP_LDK]};
    } else {
        f.a = *(const f32x4*)(sA + (16 * wm + r) * GP_LDA + kk + 4 * g);
    }
    const float* q = sB + (kk + 4 * g) * GP_LDK + 32 * wn + r;
    f.b0 = (f32x4){q[0], q[GP_LDK], q[2 * GP_LDK], q[3 * GP_LDK]};
    f.b1 = (f32x4){q[16], q[GP_LDK + 16], q[2 * GP_LDK + 16], q[3 * GP_LDK + 16]};
    return f;
}
template <bool AK>
__device__ __forceinline__ static void gp_mfma(const float* __restrict__ sA, const float* __restrict__ sB, int wm, int wn, int r, int g,
                                               f32x4 (&acc)[2][2]) {
    GpFrag cur = gp_frag<AK>(sA, sB, wm, wn, r, g, 0);
#pragma unroll
    for (int kk = 0; kk < GP_KC; kk += 16) {
        GpFrag nxt = cur;
        if (kk + 16 < GP_KC) nxt = gp_frag<AK>(sA, sB, wm, wn, r, g, kk + 16);
        // (scheduling fences: left to itself the scheduler sinks every read to just in front of the two MFMAs that use it —
        //  fewest live registers — and each pair of MFMAs then waits out a whole LDS round trip)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            acc[0][c & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a[c], cur.b0[c], acc[0][c & 1], 0, 0, 0);
            acc[1][c & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a[c], cur.b1[c], acc[1][c & 1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    }
}

// ---- epilogue of a dA1 tile: the batch pass of layer 1's backward on the 64 rows x 64 layer-1 features in registers
// (csrc/gemm_bundle.hip, gemm_l1bwd_epilogue, for a 32 x 32 block): xhat from z = X W1^T recomputed on MFMA, dy = ReLU'(A1) dA1,
// the block sums (sum dy, sum dy xhat) per column -> partials[M/64][N] and the block's share of P = dY^T X -> p_slabs[M/64][N][KP]
// (one 16 x 16 tile of it per wave). What does NOT depend on dA1 happens at the START of the kernel, while the workgroup waits
// for the BatchNorm-backward constants anyway (gp_epi_early): the rows and W1 go to an LDS region of their own, xhat of the
// wave's two tiles and the ReLU mask stay in registers — the plan puts a workgroup's dA1 block first. Behind the tile's last
// MFMA only dy, its sums and the P product are left (two barriers; the first version staged and recomputed there: 2.8 us).
struct GpEpiRegs {
    float xh[2][4];    // xhat at the lane's C/D elements
    float a1[2][4];    // A1 there (the ReLU mask)
};
__device__ static inline void gp_epi_early(const GpDesc& D, const naf_gemm_l1bwd_t& E, int m0, int n0, float* sXW, int tid, int wm, int wn,
                                           int r, int g, GpEpiRegs& R) {
    const int KP = E.kp, XS = KP + 4, q4 = KP >> 2;
    float* sX = sXW;                               // [64 rows][XS]
    float* sW = sX + 64 * XS;                      // [64 columns][XS]
    {
        const int xr = KP == 24 ? tid / 6 : tid >> 3, xq = tid - xr * q4;
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (xr < 64) x = ((const f32x4*)(E.x + (int64_t)(m0 + xr) * E.ldx))[xq];
        float w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + GP_THREADS * i;
            const int c = KP == 24 ? e / 24 : e >> 5, k = e - c * KP;
            w[i] = (c < 64 && k < E.K) ? E.W[(int64_t)(n0 + c) * E.K + k] : 0.f;
        }
        float mean[2], invstd[2], bias[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int col = n0 + 32 * wn + 16 * t + r;
#pragma unroll
            for (int e = 0; e < 4; ++e) R.a1[t][e] = E.a1[(int64_t)(m0 + 16 * wm + 4 * g + e) * E.lda1 + col];
            mean[t] = E.save_mean[col];
            invstd[t] = E.save_invstd[col];
            bias[t] = E.bias[col];
        }
        if (xr < 64) *(f32x4*)(sX + xr * XS + 4 * xq) = x;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + GP_THREADS * i;
            const int c = KP == 24 ? e / 24 : e >> 5, k = e - c * KP;
            if (c < 64) sW[c * XS + k] = w[i];
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 32; kk += 16) {
                if (kk < KP) {                          // KP = 24: lane groups 2, 3 of the second step are past the row: zeros
                    const bool in = kk + 4 * g < KP;
                    const int ko = in ? kk + 4 * g : 0;
                    f32x4 a = *(const f32x4*)(sX + (16 * wm + r) * XS + ko);
                    const f32x4 b = *(const f32x4*)(sW + (32 * wn + 16 * t + r) * XS + ko);
                    if (!in) a = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < 4; ++q) z = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b[q], z, 0, 0, 0);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) R.xh[t][e] = ((z[e] + bias[t]) - mean[t]) * invstd[t];
        }
    }
}
// Split K (the dA1 product cut into two 128-feature halves where the batch gives too few tiles to fill the chip): everything the
// pass produces is LINEAR in dA1 — dy = mask * dA1, the two column sums, P = dY^T X — so each half leaves its own "block" of
// partials / p_slabs (block slab * M/64 + bm) and the finish launch, which adds blocks in index order anyway, adds the halves.
__device__ static inline void gp_epilogue(const GpDesc& D, const naf_gemm_l1bwd_t& E, int m0, int n0, int slab, const f32x4 (&acc)[2],
                                          float* buf, const float* sXW, int tid, int wave, int wm, int wn, int r, int g, const GpEpiRegs& R) {
    const int KP = E.kp, XS = KP + 4;
    const int bm = slab * (D.M >> 6) + (m0 >> 6);
    const float* sX = sXW;
    float* sDY = buf;                              // [64 rows][65]
    float2* sRed = (float2*)(sDY + 64 * 65);       // [4 row tiles][64 columns]
    __syncthreads();                               // every wave is past its last fragment read of this buffer
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float s_dy = 0.f, s_dx = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float dy = R.a1[t][e] > 0.f ? acc[t][e] : 0.f;
            sDY[(16 * wm + 4 * g + e) * 65 + 32 * wn + 16 * t + r] = dy;
            s_dy += dy;
            s_dx += dy * R.xh[t][e];
        }
        s_dy = naf_xor32_add(naf_xor16_add(s_dy));           // the wave's other row groups of the same column
        s_dx = naf_xor32_add(naf_xor16_add(s_dx));
        if (g == 0) sRed[wm * 64 + 32 * wn + 16 * t + r] = make_float2(s_dy, s_dx);
    }
    __syncthreads();
    if (tid < 64) {                                           // the four row tiles in order
        const float2 t0 = sRed[tid], t1 = sRed[64 + tid], t2 = sRed[128 + tid], t3 = sRed[192 + tid];
        ((float2*)E.partials)[(int64_t)bm * D.N + n0 + tid] = make_float2(((t0.x + t1.x) + t2.x) + t3.x, ((t0.y + t1.y) + t2.y) + t3.y);
    }
    {
        // P share: wave w owns features 16 mt .. +15 x k 16 nt .. +15, reduction over the tile's 64 rows.
        // A[m = feature][k = row] = dy[row][feature], B[k = row][n] = x[row][n]
        const int mt = wave & 3, nt = wave >> 2;
        const bool n_on = 16 * nt + r < KP;
        const int xc = n_on ? 16 * nt + r : 0;
        f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = p0;
#pragma unroll
        for (int kk = 0; kk < 64; kk += 16) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = kk + 4 * g + q;
                if (q & 1) p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(sDY[row * 65 + 16 * mt + r], sX[row * XS + xc], p1, 0, 0, 0);
                else p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(sDY[row * 65 + 16 * mt + r], sX[row * XS + xc], p0, 0, 0, 0);
            }
        }
        const f32x4 p = p0 + p1;
        if (n_on) {
#pragma unroll
            for (int e = 0; e < 4; ++e) E.p_slabs[((int64_t)bm * D.N + n0 + 16 * mt + 4 * g + e) * KP + 16 * nt + r] = p[e];
        }
    }
}

// ---- C tile -> memory (rows past M end the resource, a column past N gets an offset past everything: dropped by the hardware)
__device__ __forceinline__ static void gp_store_c(const GpDesc& D, const GpUnit& U, const f32x4 (&acc)[2], int wm, int wn, int r, int g) {
    const unsigned ldc4 = (unsigned)D.ldc * 4u;
    const __amdgpu_buffer_rsrc_t cr = naf_buf(D.C + (int64_t)U.slab * D.cstride, (unsigned)D.M * ldc4);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int cn = U.n0 + 32 * wn + 16 * t + r;
        const unsigned voff = cn < D.N ? (unsigned)(4 * g) * ldc4 + (unsigned)cn * 4u : 0x7f000000u;
#pragma unroll
        for (int e = 0; e < 4; ++e) naf_buf_st_f1(cr, voff, (unsigned)(U.m0 + 16 * wm + e) * ldc4, acc[t][e], true);
    }
}

__device__ __forceinline__ static GpUnit gp_unit(const int kper, const int code, int chunk) {
    GpUnit U;
    U.gi = code & 3;
    U.m0 = ((code >> 2) & 255) * GP_BM;
    U.n0 = ((code >> 10) & 63) * GP_BN;
    U.slab = (code >> 16) & 255;
    const int k_lo = U.slab * kper;
    U.k_hi = k_lo + kper;
    U.k0 = k_lo + chunk * GP_KC;
    U.first = chunk == 0;
    U.last = U.k0 + GP_KC >= U.k_hi;
    return U;
}

#define GP_XW_FLOATS (2 * 64 * 36)      // the epilogue's rows + W1 tile (KP <= 32)
#define GP_LDS_FLOATS (2 * GP_BUF + 4 * 256 + GP_XW_FLOATS)
// one trip's three jobs; the two halves of the workgroup run them in opposite order (see the kernel)
#define GP_STORE_AND_REQUEST()                                                               \
    do {                                                                                     \
        gp_store(R, flags_cur, Ucur, sA, sB, cst, tid);                                      \
        if (have_next) gp_load(R, Dn, bundle.pro, Unext, tid);                               \
    } while (0)
#define GP_COMPUTE(UU, FL, BUFI)                                                             \
    do {                                                                                     \
        const float* pA = smem + (BUFI) * GP_BUF;                                            \
        const float* pB = pA + GP_PANEL;                                                     \
        if ((UU).first) {                                                                    \
            acc[0][0] = acc[0][1] = acc[1][0] = acc[1][1] = (f32x4){0.f, 0.f, 0.f, 0.f};     \
        }                                                                                    \
        if ((FL) & GP_F_AK) gp_mfma<true>(pA, pB, wm, wn, r, g, acc);                        \
        else gp_mfma<false>(pA, pB, wm, wn, r, g, acc);                                      \
        if ((UU).last) {                                                                     \
            const GpDesc Dp = gp_desc(bundle.d, (UU).gi);     /* once per block: where C goes */ \
            const f32x4 c[2] = {acc[0][0] + acc[0][1], acc[1][0] + acc[1][1]};               \
            if (Dp.C) gp_store_c(Dp, (UU), c, wm, wn, r, g);                                 \
            if ((FL) & GP_F_EPI)                                                             \
                gp_epilogue(Dp, bundle.epi, (UU).m0, (UU).n0, (UU).slab, c, smem + (BUFI) * GP_BUF, sXW, tid, wave, wm, wn, r, g, ER); \
        }                                                                                    \
    } while (0)

__global__ __launch_bounds__(GP_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_bundle_p_kernel(const GpBundle bundle) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [2][GP_BUF] panels | [4][256] BatchNorm-backward constants | X, W1
    float* const cst = smem + 2 * GP_BUF;
    float* const sXW = cst + 4 * 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4, wm = wave & 3, wn = wave >> 2;
    const int* __restrict__ plan = bundle.plan + (int64_t)blockIdx.x * NAF_GEMM_P_MAX_BLOCKS;
    GP_TL(0);
    // The launch's first workgroups fold the BatchNorm-backward block sums for everybody, BEFORE anything of their own: every
    // workgroup of the launch waits for their records (buffer 1 as scratch: the first chunk goes to buffer 0)
    if (bundle.n_fold && (int)blockIdx.x < bundle.n_fold) {
        gemm_bn2bwd_fold_block(bundle.pro, (int)blockIdx.x, tid, smem + GP_BUF);
        __syncthreads();
    }
    // the first unit's panels fly before anything else happens
    int bi = 0, chunk = 0;                 // position of the unit whose loads are issued next
    int code = __builtin_amdgcn_readfirstlane((int)blockIdx.x < GP_MAX_WG_IN_ARGS ? bundle.first[blockIdx.x] : plan[0]);
    GpRegs R;
    GpEpiRegs ER;
    const bool have_work = code >= 0;
    GpDesc Dn = gp_desc(bundle.d, have_work ? code & 3 : 0);       // the product of the unit requested next
    GpUnit Unext = gp_unit(Dn.kper, have_work ? code : 0, 0);
    if (have_work) gp_load(R, Dn, bundle.pro, Unext, tid);
    GP_TL(1);
    // a dA1 block leads its workgroup's list: what its epilogue needs besides dA1 itself, now, under the wait for the constants
    if (have_work && (Dn.flags & GP_F_EPI)) gp_epi_early(Dn, bundle.epi, Unext.m0, Unext.n0, sXW, tid, wm, wn, r, g, ER);
    // EVERY workgroup that stages dZ2 takes the 256 columns' constants — they are the same for all of its blocks
    if (bundle.any_pro) {
        gemm_bn2bwd_wait_constants<false>(bundle.pro, 0, tid, cst);
        __syncthreads();
    }
    GP_TL(2);
    if (!have_work) return;

    // One flat loop over the units (chunks) of all blocks of the list. In trip u: unit u goes registers -> buffer u & 1, unit u + 1
    // is requested from memory, unit u - 1 is multiplied out of the other buffer. The two halves of the workgroup take these jobs
    // in OPPOSITE order — waves 0 .. 3 (one per SIMD) store first, waves 4 .. 7 multiply first — so that one wave of every SIMD
    // feeds the matrix pipe while its partner moves 64 KB through the LDS store path (all eight storing at once left the pipe idle
    // for 0.6 - 1.3 us per trip). The buffers differ, so the order inside a wave is free; one barrier per trip.
    f32x4 acc[2][2];
    GpUnit Uprev = Unext;                  // the unit whose MFMAs run in this trip (valid from the second trip on)
    int flags_prev = 0, flags_cur = 0;
    bool have_prev = false;
    const bool store_first = wave < 4;
    int u = 0;
    while (true) {
        const GpUnit Ucur = Unext;         // its panels are in R (in flight)
        flags_cur = Dn.flags;
        float* sA = smem + (u & 1) * GP_BUF;
        float* sB = sA + GP_PANEL;
        // the unit after it: next chunk of the block, or the first chunk of the next block of the list (its product's
        // description is requested here, a whole job before gp_load needs it)
        bool have_next = true;
        if (!Ucur.last) {
            ++chunk;
        } else {
            ++bi;
            chunk = 0;
            code = __builtin_amdgcn_readfirstlane(bi < NAF_GEMM_P_MAX_BLOCKS ? plan[bi] : -1);
            have_next = code >= 0;
            if (have_next) Dn = gp_desc(bundle.d, code & 3);
        }
        if (have_next) Unext = gp_unit(Dn.kper, code, chunk);
#pragma clang loop unroll(disable)
        for (int ph = 0; ph < 2; ++ph) {   // (a loop, not two copies of the jobs: twice the code spilled registers)
            if ((ph == 0) == store_first) {
                GP_STORE_AND_REQUEST();
                GP_TL(3 + 3 * u);
            } else if (have_prev) {
                GP_COMPUTE(Uprev, flags_prev, (u - 1) & 1);
            }
        }
        GP_TL(4 + 3 * u);
        __syncthreads();                   // chunk u is in LDS for everybody; everybody is done with buffer (u - 1) & 1
        GP_TL(5 + 3 * u);
        Uprev = Ucur;
        flags_prev = flags_cur;
        have_prev = true;
        ++u;
        if (!have_next) break;
    }
    GP_COMPUTE(Uprev, flags_prev, (u - 1) & 1);        // the last unit
    GP_TL(15);
}

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
#define GP_LDS_BYTES ((size_t)GP_LDS_FLOATS * sizeof(float))
// more dynamic LDS than the default 64 KB per workgroup: asked for once per process — by the plan call, i.e. when a learner is
// built, never inside a stream capture
static int gp_allow_lds() {
    static bool done = false;
    if (!done) {
        const hipError_t e = hipFuncSetAttribute((const void*)gemm_bundle_p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GP_LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        done = true;
    }
    return NAF_OK;
}
struct GpBlock {
    int gi, bm, bn, k_lo, k_hi, slab, xcd;
    float cost;
    bool epi;
};
// planning: only shapes matter (the epilogue's row pointer is set per minibatch)
static int gp_check(const naf_gemm_desc_t* descs, int n, bool planning) {
    if (!descs || n <= 0 || n > NAF_GEMM_BUNDLE_MAX) return NAF_ERR_ARG;
    int n_epi = 0;
    const naf_gemm_bn2bwd_t* pro = nullptr;
    for (int i = 0; i < n; ++i) {
        const naf_gemm_desc_t& s = descs[i];
        if (!s.A || !s.B || (!s.C && !s.epi) || s.M <= 0 || s.N <= 0 || s.K <= 0 || s.K > 32000) return NAF_ERR_ARG;
        if ((s.M & 15) || (s.N & 15) || (s.K & 15) || !s.b_kmajor || s.sumsq) return NAF_ERR_ARG;
        const int ksn = s.k_split > 0 ? s.k_split : 1;
        if (s.K % ksn || ((s.K / ksn) & 15) || (ksn > 1 && s.C && s.c_split_stride < (int64_t)s.M * s.ldc)) return NAF_ERR_ARG;
        if (s.lda < (s.a_kmajor ? s.M : s.K) || s.ldb < s.N || s.ldc < s.N || (s.lda & 3) || (s.ldb & 3)) return NAF_ERR_ARG;
        if ((((uintptr_t)s.A) & 15) || (((uintptr_t)s.B) & 15)) return NAF_ERR_ARG;
        if (!s.a_kmajor && ((s.K / ksn) % GP_KC)) return NAF_ERR_ARG;     // a k-contiguous A panel is read in whole 128-k chunks
        if (s.M > 256 * GP_BM || s.N > 64 * GP_BN || ksn > 256 || i > 3) return NAF_ERR_ARG;   // (the packed block codes)
        if (s.pro) {
            const naf_gemm_bn2bwd_t& q = *s.pro;
            if (!q.z || !q.partials || !q.gamma || !q.save_mean || !q.save_invstd || !q.d_gamma || !q.d_beta || q.npb < 1 || q.npb > 128 ||
                !q.cst || !q.epoch || ((uintptr_t)q.cst & 15) || q.B <= 0 || q.H != 256 || (s.a_kmajor ? s.M != q.H : s.K != q.H) ||
                ((uintptr_t)q.z & 15) || ((uintptr_t)q.partials & 7))
                return NAF_ERR_ARG;
            if (pro && memcmp(pro, &q, sizeof(q)) != 0) return NAF_ERR_ARG;      // one prologue per launch
            pro = &q;
        }
        if (s.epi) {
            const naf_gemm_l1bwd_t& e = *s.epi;
            if (++n_epi > 1 || (!planning && !e.x) || !e.W || !e.bias || !e.a1 || !e.save_mean || !e.save_invstd || !e.partials || !e.p_slabs || ksn > 2 ||
                (s.M & 63) || (s.N & 63) || e.K <= 0 || (e.kp != 24 && e.kp != 32) || e.K > e.kp || (!planning && (e.ldx < e.kp || (e.ldx & 3))) ||
                e.lda1 < s.N || ((uintptr_t)e.x & 15) || ((uintptr_t)e.partials & 7) || s.a_kmajor)
                return NAF_ERR_ARG;
        }
    }
    return NAF_OK;
}

// The list of blocks every workgroup walks: plan_host[n_wg][NAF_GEMM_P_MAX_BLOCKS] block codes (GP_CODE). Longest blocks first
// (the dA1 tiles with their epilogue lead their workgroup's list: the kernel prefetches the epilogue's operands at its start),
// each to the least-loaded workgroup of the XCD — workgroup w runs on XCD w % 8 under round-robin dispatch; speed only — that
// already pulls the block's operands: a block ROW of dA1 (an eighth of dY2, Z2, A1 and the minibatch rows per XCD, all of W2),
// a K RANGE of the weight gradients (its rows of dY2, Z2, A1 / dH, A2). A workgroup of another XCD takes the block when that
// one would finish more than a chunk later.
extern "C" int naf_gemm_bundle_p_plan(const naf_gemm_desc_t* descs, int n, int n_wg, int32_t* plan_host) {
    const int rc = gp_check(descs, n, true);
    if (rc != NAF_OK) return rc;
    if (!plan_host || n_wg < 8 || n_wg > 4096) return NAF_ERR_ARG;
    std::vector<GpBlock> blocks;
    for (int i = 0; i < n; ++i) {
        const naf_gemm_desc_t& s = descs[i];
        const int ksn = s.k_split > 0 ? s.k_split : 1, kper = s.K / ksn;
        const int tm = (s.M + GP_BM - 1) / GP_BM, tn = (s.N + GP_BN - 1) / GP_BN;
        for (int ks = 0; ks < ksn; ++ks)
            for (int bm = 0; bm < tm; ++bm)
                for (int bn = 0; bn < tn; ++bn) {
                    GpBlock b;
                    b.gi = i; b.bm = bm; b.bn = bn; b.k_lo = ks * kper; b.k_hi = b.k_lo + kper; b.slab = ks;
                    b.epi = s.epi != nullptr;
                    b.cost = (float)((kper + GP_KC - 1) / GP_KC) + (b.epi ? 0.6f : 0.f) + (s.pro ? 0.1f : 0.f);
                    // batch rows the block reads: its K range (weight gradients) or its block row (dA1); XCD = the eighth of the
                    // batch they lie in — where naf_bb_layer2_head wrote them (its chunk map for this bundle: contiguous eighths)
                    const int row0 = s.a_kmajor ? b.k_lo : bm * GP_BM, rows = s.a_kmajor ? s.K : s.M;
                    b.xcd = (int)(((int64_t)row0 * 8) / rows) & 7;
                    blocks.push_back(b);
                }
    }
    std::stable_sort(blocks.begin(), blocks.end(), [](const GpBlock& a, const GpBlock& b) {
        if (a.epi != b.epi) return a.epi;
        return a.cost > b.cost;
    });
    std::vector<float> load(n_wg, 0.f);
    std::vector<int> count(n_wg, 0);
    std::vector<char> has_epi(n_wg, 0);
    for (int i = 0; i < n_wg * NAF_GEMM_P_MAX_BLOCKS; ++i) plan_host[i] = -1;
    for (const GpBlock& b : blocks) {
        int best = -1, best_any = -1;
        for (int w = 0; w < n_wg; ++w) {
            if (count[w] >= NAF_GEMM_P_MAX_BLOCKS || (b.epi && has_epi[w])) continue;
            if (best_any < 0 || load[w] < load[best_any]) best_any = w;
            if ((w & 7) == b.xcd && (best < 0 || load[w] < load[best])) best = w;
        }
        if (best_any < 0) return NAF_ERR_ARG;                        // more blocks than the lists hold
        if (best < 0 || load[best] > load[best_any] + 2.0f) best = best_any;
        plan_host[(size_t)best * NAF_GEMM_P_MAX_BLOCKS + count[best]] = GP_CODE(b.gi, b.bm, b.bn, b.slab);
        load[best] += b.cost;
        ++count[best];
        if (b.epi) has_epi[best] = 1;
    }
    int dev_count = 0;
    if (hipGetDeviceCount(&dev_count) == hipSuccess && dev_count > 0) return gp_allow_lds();
    (void)hipGetLastError();           // (no device: planning is host arithmetic and stays usable)
    return NAF_OK;
}

extern "C" int naf_gemm_bundle_p(const naf_gemm_desc_t* descs, int n, const int32_t* plan_dev, const int32_t* plan_host, int n_wg,
                                 void* stream) {
    const int rc = gp_check(descs, n, false);
    if (rc != NAF_OK) return rc;
    if (!plan_dev || !plan_host || ((uintptr_t)plan_dev & 15) || n_wg < 8 || n_wg > 4096) return NAF_ERR_ARG;
    GpBundle b;
    memset(&b, 0, sizeof(b));
    b.n = n;
    b.plan = (const int*)plan_dev;
    for (int w = 0; w < GP_MAX_WG_IN_ARGS; ++w) b.first[w] = w < n_wg ? plan_host[(size_t)w * NAF_GEMM_P_MAX_BLOCKS] : -1;
    for (int i = 0; i < n; ++i) {
        const naf_gemm_desc_t& s = descs[i];
        GpDesc& d = b.d[i];
        d.A = s.A; d.B = s.B; d.C = s.C; d.M = s.M; d.N = s.N; d.lda = s.lda; d.ldb = s.ldb; d.ldc = s.ldc;
        d.flags = (s.a_kmajor ? GP_F_AK : 0) | (s.pro ? GP_F_PRO : 0) | (s.epi ? GP_F_EPI : 0);
        if (s.c_split_stride > 0x7fffffff) return NAF_ERR_ARG;
        d.cstride = (int)s.c_split_stride;
        d.kper = s.K / (s.k_split > 0 ? s.k_split : 1);
        if (s.pro) {
            b.pro = *s.pro;
            b.any_pro = 1;
            b.n_fold = s.pro->H / GB_FOLD_COLS;
        }
        if (s.epi) b.epi = *s.epi;
    }
    for (int i = n; i < NAF_GEMM_BUNDLE_MAX; ++i) b.d[i] = b.d[0];
    if (b.n_fold > n_wg) return NAF_ERR_ARG;
    const size_t lds = GP_LDS_BYTES;
    const int ra = gp_allow_lds();
    if (ra != NAF_OK) return ra;
    gemm_bundle_p_kernel<<<n_wg, GP_THREADS, lds, (hipStream_t)stream>>>(b);
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}
