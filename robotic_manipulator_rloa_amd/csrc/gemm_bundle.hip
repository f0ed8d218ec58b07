// Several small f32 GEMMs in ONE launch on v_mfma_f32_16x16x4_f32 (gfx950). The backward of a learn() update needs
// three independent products once dZ2 and dH exist — dW2 = dZ2^T A1, dA1 = dZ2 W2, dWh = dH^T A2 (autograd of
// naf_neural_network.py:76-87) — each ~33 MFLOP or less: as three library launches they cost three launch
// boundaries (~3.3 us each, all latency); as one grid of 16x16 output tiles (546 tiles at B=256) they fill half the
// chip once.
//
// C/D map of the MFMA: col = l & 15, row = 4 (l >> 4) + reg. Summation order over k is fixed -> bitwise reproducible.
#include "common.h"
#include "../../include/naf_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GemmDesc {
    const float* A;   // a_kmajor ? [K][M] (ld = lda) : [M][K]
    const float* B;   // b_kmajor ? [K][N] (ld = ldb) : [N][K]
    float* C;         // [M][N], ld = ldc
    float* sumsq;     // nullable: sumsq[block] = sum of C^2 over the block (grad-norm partial)
    int M, N, K, lda, ldb, ldc, a_kmajor, b_kmajor, tile0, tiles_n;
    int k_split, tiles_mn;        // K cut into k_split ranges, one grid of tiles_mn blocks each, slab s at C + s * c_split_stride
    int64_t c_split_stride;
};
struct GemmBundle {
    GemmDesc d[NAF_GEMM_BUNDLE_MAX];
    int n, total_tiles;
};

// ---- LDS-staged form -----------------------------------------------------------------------------------------------
// Workgroup = 512 threads = 8 waves = one 32 x 32 output block (2 x 2 MFMA tiles, two waves per tile: K halves). A K-chunk of 256 of
// both operand panels (32 rows x 256 k each) is staged into LDS with 16-byte global loads and 16-byte LDS stores, then
// every wave reads its fragments: lane (r, g) takes k = 16 j + 4 g .. +3 of row r for BOTH operands, so the four MFMAs
// of macro-step j use each k once. One L2 round trip per 256 k instead of one per
// fragment (the register-fed form of this kernel issued 256 4-byte loads per lane per tile and ran 10.2 us).
#define GB_KC 256                 // k per staged chunk
#define GB_LD (GB_KC + 4)         // [row][k] panels: 16-B aligned rows, b128 fragment reads spread over the banks
#define GB_LDK 36                 // [k][row] panels (k-major operands keep their memory layout): 32 rows + 4 pad
#define GB_PANEL (GB_KC * GB_LDK) // floats per panel buffer (>= 32 * GB_LD)

// Stage a 32-row x kc panel with 16-byte loads AND 16-byte LDS stores, in the operand's own memory order:
//   k-contiguous operand -> LDS [row][k] (stride GB_LD), fragments read as one ds_read_b128 per macro-step
//   k-major operand      -> LDS [k][row] (stride GB_LDK), fragments read as four ds_read_b32 (bank = 4k + row: the two
//                           16-lane groups a b32 read serves per cycle never collide)
// (the first version transposed k-major panels while staging: 4 ds_write_b32 per float4 with a 4-way bank conflict)
// A panel is 32 rows x kc k = kc * 8 float4, GB_KC * 8 / 256 = 8 per thread. ALL of a thread's loads — of both panels —
// are issued before the first LDS store: as a load -> store loop (one load in flight per thread) the staging was 16
// serial memory round trips per block, ~200 cycles each on L2 hits but 545+ on data the previous kernel had just
// written (Infinity Cache): the whole fresh-data penalty of this kernel (benchmarks/chain_probe.py: 1.8 of its 8.5 us).
#ifndef GB_THREADS
#define GB_THREADS 512
#endif
#define GB_KSPLIT (GB_THREADS / 256)   // waves per 16 x 16 tile: each takes 1/GB_KSPLIT of the K chunk
#define GB_PT (GB_KC * 8 / GB_THREADS)   // float4 per thread per panel
// FULL = the chunk is a whole GB_KC (every call but the tail of a K that is not a multiple of 256): row / k indices are
// shifts; the general form divides by a run-time k4n once per element — ~20 integer instructions, 32 times per thread,
// in a kernel whose waves run ~1,100 instructions in all.
template <bool KMAJOR, bool FULL>
__device__ static inline void load_panel(float4 (&v)[GB_PT], const float* __restrict__ p, int ld, int row0,
                                         int rows_total, int k0, int kc, int tid) {
#pragma unroll
    for (int i = 0; i < GB_PT; ++i) {
        const int e = tid + GB_THREADS * i;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (FULL || e < kc * 8) {
            if (KMAJOR) {
                const int k = e >> 3, r4 = (e & 7) * 4;
                if (row0 + r4 < rows_total) v[i] = *(const float4*)(p + (int64_t)(k0 + k) * ld + row0 + r4);
            } else {
                const int k4n = FULL ? GB_KC / 4 : kc >> 2;
                const int row = FULL ? e / (GB_KC / 4) : e / k4n, k4 = (e - row * k4n) * 4;
                if (row0 + row < rows_total) v[i] = *(const float4*)(p + (int64_t)(row0 + row) * ld + k0 + k4);
            }
        }
    }
}

template <bool KMAJOR, bool FULL>
__device__ static inline void store_panel(float* __restrict__ sm, const float4 (&v)[GB_PT], int kc, int tid) {
#pragma unroll
    for (int i = 0; i < GB_PT; ++i) {
        const int e = tid + GB_THREADS * i;
        if (FULL || e < kc * 8) {
            if (KMAJOR) {
                const int k = e >> 3, r4 = (e & 7) * 4;
                *(float4*)(sm + k * GB_LDK + r4) = v[i];
            } else {
                const int k4n = FULL ? GB_KC / 4 : kc >> 2;
                const int row = FULL ? e / (GB_KC / 4) : e / k4n, k4 = (e - row * k4n) * 4;
                *(float4*)(sm + row * GB_LD + k4) = v[i];
            }
        }
    }
}

// fragment of macro-step kk for lane (r, g): elements k = kk + 4 g + c, c = 0..3, of panel row `row`
template <bool KMAJOR>
__device__ static inline float4 read_frag(const float* __restrict__ sm, int row, int g, int kk) {
    if (KMAJOR) {
        const float* q = sm + (kk + 4 * g) * GB_LDK + row;
        return make_float4(q[0], q[GB_LDK], q[2 * GB_LDK], q[3 * GB_LDK]);
    }
    return *(const float4*)(sm + row * GB_LD + kk + 4 * g);
}

template <bool AK, bool BK>
__device__ static inline void gemm_block(const GemmDesc& D, int bm, int bn, int ks, float* sA, float* sB, float* sQ, float* sC) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    // 8 waves: two per 16 x 16 tile of the 32 x 32 block, each taking one half of the K chunk — the per-wave chain of
    // dependent MFMAs (the longest single piece of this kernel: 1.5 of its 5.0 us with 64 of them) is halved; the two
    // halves meet through LDS, lower half first (fixed order)
    const int tile = wave & 3, kh = wave >> 2;
    const int wm = tile >> 1, wn = tile & 1;
    const int m0 = bm * 32, n0 = bn * 32;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // Software pipeline over the K chunks (K = B for the weight gradients: 4 chunks at B = 1024): the global loads of
    // chunk i+1 are issued before the MFMAs of chunk i and land in registers while they run. As a plain
    // load -> store -> barrier -> compute loop every chunk paid its own L2 round trip: 13.3 us per launch at B = 1024.
    // Split K (k_split > 1: the weight gradients at large batches, K = B): this block takes K range ks and writes slab ks;
    // the slabs are added in slab order by the consumer (bb_layer1_bwd_finish's reduce blocks). 64 blocks walking
    // K = 1024 pulled 256 KB each through one CU's L2 port (13.6 us per launch at B = 1024); 256 blocks of K = 256 do not.
    const int kper = D.K / D.k_split, k_lo = ks * kper, k_hi = k_lo + kper;
    float* Cs = D.C + (int64_t)ks * D.c_split_stride;
    float4 va[GB_PT], vb[GB_PT];
    {
        const int kc0 = kper < GB_KC ? kper : GB_KC;
        if (kc0 == GB_KC) {
            load_panel<AK, true>(va, D.A, D.lda, m0, D.M, k_lo, kc0, tid);
            load_panel<BK, true>(vb, D.B, D.ldb, n0, D.N, k_lo, kc0, tid);
        } else {
            load_panel<AK, false>(va, D.A, D.lda, m0, D.M, k_lo, kc0, tid);
            load_panel<BK, false>(vb, D.B, D.ldb, n0, D.N, k_lo, kc0, tid);
        }
    }
    for (int k0 = k_lo; k0 < k_hi; k0 += GB_KC) {
        const int kc = (k_hi - k0) < GB_KC ? (k_hi - k0) : GB_KC;
        if (k0 != k_lo) __syncthreads();                  // previous chunk fully consumed
        if (kc == GB_KC) {
            store_panel<AK, true>(sA, va, kc, tid);
            store_panel<BK, true>(sB, vb, kc, tid);
        } else {
            store_panel<AK, false>(sA, va, kc, tid);
            store_panel<BK, false>(sB, vb, kc, tid);
        }
        const int k1 = k0 + GB_KC;
        if (k1 < k_hi) {                                  // next chunk's loads fly under this chunk's MFMAs
            const int kn = (k_hi - k1) < GB_KC ? (k_hi - k1) : GB_KC;
            if (kn == GB_KC) {
                load_panel<AK, true>(va, D.A, D.lda, m0, D.M, k1, kn, tid);
                load_panel<BK, true>(vb, D.B, D.ldb, n0, D.N, k1, kn, tid);
            } else {
                load_panel<AK, false>(va, D.A, D.lda, m0, D.M, k1, kn, tid);
                load_panel<BK, false>(vb, D.B, D.ldb, n0, D.N, k1, kn, tid);
            }
        }
        __syncthreads();
        const int steps = kc >> 4;                                    // macro-steps of 16 k, dealt in contiguous runs
        const int kbeg = (steps * kh / GB_KSPLIT) << 4, kend = (steps * (kh + 1) / GB_KSPLIT) << 4;
#pragma unroll 4
        for (int kk = kbeg; kk < kend; kk += 16) {
            const float4 a = read_frag<AK>(sA, wm * 16 + r, g, kk);
            const float4 b = read_frag<BK>(sB, wn * 16 + r, g, kk);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc1, 0, 0, 0);
        }
    }
    f32x4 acc = acc0 + acc1;
    if (kh) *(f32x4*)(sC + (((kh - 1) * 4 + tile) * 64 + lane) * 4) = acc;
    __syncthreads();
    float sq = 0.f;
    if (!kh) {
#pragma unroll
        for (int h = 1; h < GB_KSPLIT; ++h) acc = acc + *(const f32x4*)(sC + (((h - 1) * 4 + tile) * 64 + lane) * 4);
        const int cm = m0 + wm * 16 + 4 * g, cn = n0 + wn * 16 + r;
        if (cn < D.N) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (cm + e < D.M) {
                    const float v = acc[e];
                    Cs[(int64_t)(cm + e) * D.ldc + cn] = v;
                    sq += v * v;
                }
        }
    }
    if (D.sumsq) {   // gradient-norm partial of this block (fixed order: shuffles, then the 4 tiles)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
        if (!kh && lane == 0) sQ[tile] = sq;                  // sQ is touched nowhere else: no barrier in front
        __syncthreads();
        if (tid == 0) D.sumsq[bm * D.tiles_n + bn] = sQ[0] + sQ[1] + sQ[2] + sQ[3];
    }
}

__global__ __launch_bounds__(GB_THREADS) void gemm_bundle_kernel(const GemmBundle bundle) {
    __shared__ __attribute__((aligned(16))) float sA[GB_PANEL];
    __shared__ __attribute__((aligned(16))) float sB[GB_PANEL];
    __shared__ float sQ[4];
    __shared__ __attribute__((aligned(16))) float sC[(GB_KSPLIT - 1) * 4 * 64 * 4];
    const int t = blockIdx.x;                             // one 32 x 32 block per workgroup
    int gi = 0;
#pragma unroll
    for (int i = 1; i < NAF_GEMM_BUNDLE_MAX; ++i)
        if (i < bundle.n && t >= bundle.d[i].tile0) gi = i;
    const GemmDesc& D = bundle.d[gi];
    const int ks = (t - D.tile0) / D.tiles_mn;
    const int lt = t - D.tile0 - ks * D.tiles_mn;
    int bm = lt / D.tiles_n, bn = lt - bm * D.tiles_n;
    if (D.tiles_n == 8 && D.M == 256) {
        // 8 x 8 blocks; workgroup t runs on XCD t % 8 (round-robin dispatch) and the column-tile kernels on either side of
        // this launch keep columns 32 x .. 32 x + 31 on XCD x (naf_xcd_tile). k-major A (dW2 = dZ2^T A1): block ROW bm on
        // XCD bm, the one that has just written those dZ2 columns; k-contiguous A (dA1 = dZ2 W2): block COLUMN bn on XCD
        // bn, the one that reads those dA1 columns next. Producer and consumer then share an L2 (+1 % updates/s;
        // placement is speed only, the result does not depend on it).
        const int xcd = lt & 7, slot = lt >> 3;
        if (D.a_kmajor) { bm = xcd; bn = slot; } else { bn = xcd; bm = slot; }
    }
    if (D.a_kmajor) {
        if (D.b_kmajor) gemm_block<true, true>(D, bm, bn, ks, sA, sB, sQ, sC);
        else gemm_block<true, false>(D, bm, bn, ks, sA, sB, sQ, sC);
    } else {
        if (D.b_kmajor) gemm_block<false, true>(D, bm, bn, ks, sA, sB, sQ, sC);
        else gemm_block<false, false>(D, bm, bn, ks, sA, sB, sQ, sC);
    }
}

extern "C" int naf_gemm_bundle(const naf_gemm_desc_t* descs, int n, void* stream) {
    if (!descs || n <= 0 || n > NAF_GEMM_BUNDLE_MAX) return NAF_ERR_ARG;
    GemmBundle b;
    b.n = n;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        const naf_gemm_desc_t& s = descs[i];
        if (!s.A || !s.B || !s.C || s.M <= 0 || s.N <= 0 || s.K <= 0) return NAF_ERR_ARG;
        if ((s.M & 15) || (s.N & 15) || (s.K & 15)) return NAF_ERR_ARG;            // whole 16x16x16 steps only
        const int ksn = s.k_split > 0 ? s.k_split : 1;
        if (ksn > 1 && (s.K % ksn || ((s.K / ksn) & 15) || s.sumsq || s.c_split_stride < (int64_t)s.M * s.ldc)) return NAF_ERR_ARG;
        if (s.lda < (s.a_kmajor ? s.M : s.K) || s.ldb < (s.b_kmajor ? s.N : s.K) || s.ldc < s.N) return NAF_ERR_ARG;
        if ((((uintptr_t)s.A) & 15) || (s.lda & 3) || (((uintptr_t)s.B) & 15) || (s.ldb & 3)) return NAF_ERR_ARG;   // float4 staging
        GemmDesc& d = b.d[i];
        d.A = s.A; d.B = s.B; d.C = s.C; d.sumsq = s.sumsq;
        d.M = s.M; d.N = s.N; d.K = s.K;
        d.lda = s.lda; d.ldb = s.ldb; d.ldc = s.ldc;
        d.a_kmajor = s.a_kmajor; d.b_kmajor = s.b_kmajor;
        d.tile0 = tiles;
        d.tiles_n = (s.N + 31) / 32;
        d.tiles_mn = ((s.M + 31) / 32) * d.tiles_n;
        d.k_split = ksn;
        d.c_split_stride = s.c_split_stride;
        tiles += d.tiles_mn * ksn;
    }
    for (int i = n; i < NAF_GEMM_BUNDLE_MAX; ++i) b.d[i] = b.d[0];
    b.total_tiles = tiles;
    gemm_bundle_kernel<<<tiles, GB_THREADS, 0, (hipStream_t)stream>>>(b);
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}
