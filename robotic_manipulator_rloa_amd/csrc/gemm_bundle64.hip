// The GEMM bundle of the LARGE-batch chain: 64 x 64 output blocks (gemm_bundle.hip's are 32 x 32, tuned for B = 256 where
// there are barely enough blocks to go round). At B >= 1024 the 32 x 32 bundle was bound by L2 -> LDS traffic, not MFMA:
// every block stages (32 + 32) x K floats for 32 x 32 x K multiply-adds, 70 MB per launch at B = 2048 (19.8 us); a 64 x 64
// block stages twice as much for four times the work. Same products (dWh = dH^T A2, dW2 = dZ2^T A1 as split-K slabs,
// dA1 = dZ2 W2 with the layer-1 backward pass as its epilogue), same naf_gemm_desc_t.
//   512 threads = 8 waves; wave w owns C rows 16 (w & 3) .. +15 and columns 32 (w >> 2) .. +31 (two 16 x 16 MFMA tiles), the
//   whole K range of the block. K goes through LDS in chunks of 128 (A panel + B panel = 70 KB: two blocks per CU), the next
//   chunk's loads in flight under the MFMAs (native vector registers: HIP float4 arrays alive across a barrier end in scratch).
//   B operands are k-major ([K][N], every product of the chain); A is k-major (weight gradients) or row-major (dA1).
// C/D map of v_mfma_f32_16x16x4_f32: col = lane & 15, row = 4 (lane >> 4) + reg. Fixed summation order: reproducible.
#include <string.h>
#include "common.h"
#include "../../include/naf_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define G6_T 64                    // tile rows / columns
#define G6_KC 128                  // k per staged chunk
#define G6_THREADS 512
#define G6_LDK (G6_T + 4)          // [k][row] panels (k-major operands keep their memory order)
#define G6_LDR (G6_KC + 4)         // [row][k] panel (row-major A)
#define G6_PANEL (G6_KC * G6_LDK)  // floats per panel buffer (>= G6_T * G6_LDR)
#define G6_PT (G6_KC * G6_T / 4 / G6_THREADS)   // float4 per thread per panel = 4

struct G6Desc {
    const float* A;
    const float* B;
    float* C;
    int M, N, K, lda, ldb, ldc, a_kmajor, tile0, tiles_n, tiles_mn, k_split;
    int64_t c_split_stride;
    naf_gemm_l1bwd_t epi;
};
struct G6Bundle {
    G6Desc d[NAF_GEMM_BUNDLE_MAX];
    int n;
};

template <bool KMAJOR>
__device__ __forceinline__ static void g6_load(f32x4 (&v)[G6_PT], const float* __restrict__ p, int ld, int row0, int rows_total,
                                               int k0, int tid) {
#pragma unroll
    for (int i = 0; i < G6_PT; ++i) {
        const int e = tid + G6_THREADS * i;
        v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (KMAJOR) {
            const int k = e >> 4, r4 = (e & 15) * 4;                 // 16 float4 per k
            if (row0 + r4 < rows_total) v[i] = *(const f32x4*)(p + (int64_t)(k0 + k) * ld + row0 + r4);
        } else {
            const int row = e >> 5, k4 = (e & 31) * 4;               // 32 float4 per row
            if (row0 + row < rows_total) v[i] = *(const f32x4*)(p + (int64_t)(row0 + row) * ld + k0 + k4);
        }
    }
}
template <bool KMAJOR>
__device__ __forceinline__ static void g6_store(float* __restrict__ sm, const f32x4 (&v)[G6_PT], int tid) {
#pragma unroll
    for (int i = 0; i < G6_PT; ++i) {
        const int e = tid + G6_THREADS * i;
        if (KMAJOR) *(f32x4*)(sm + (e >> 4) * G6_LDK + (e & 15) * 4) = v[i];
        else *(f32x4*)(sm + (e >> 5) * G6_LDR + (e & 31) * 4) = v[i];
    }
}
template <bool KMAJOR>
__device__ __forceinline__ static f32x4 g6_frag(const float* __restrict__ sm, int row, int g, int kk) {
    if (KMAJOR) {
        const float* q = sm + (kk + 4 * g) * G6_LDK + row;
        return (f32x4){q[0], q[G6_LDK], q[2 * G6_LDK], q[3 * G6_LDK]};
    }
    return *(const f32x4*)(sm + row * G6_LDR + kk + 4 * g);
}

// what the layer-1 epilogue reads from memory, requested before the K loop: thread = 8 elements of the 64 x 64 tile
// (row = (tid >> 6) + 8 i, column = tid & 63), a float4 of the X tile, three scalars of the W1 tile, one statistic
struct G6Epi {
    f32x4 x;
    float w[4];
    float a1[8];
    float st;
};
__device__ __forceinline__ static void g6_epi_prefetch(const G6Desc& D, int bm, int bn, int tid, G6Epi& R) {
    const naf_gemm_l1bwd_t& E = D.epi;
    const int KP = E.kp, m0 = bm * G6_T, n0 = bn * G6_T;
    const int xr = tid / (KP / 4), xq = tid - xr * (KP / 4);
    R.x = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (xr < G6_T) R.x = ((const f32x4*)(E.x + (int64_t)(m0 + xr) * E.ldx))[xq];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + G6_THREADS * i;
        const int c = e / KP, k = e - c * KP;
        R.w[i] = (c < G6_T && k < E.K) ? E.W[(int64_t)(n0 + c) * E.K + k] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) R.a1[i] = E.a1[(int64_t)(m0 + (tid >> 6) + 8 * i) * E.lda1 + n0 + (tid & 63)];
    R.st = 0.f;
    if (tid < 64) R.st = E.save_mean[n0 + tid];
    else if (tid < 128) R.st = E.save_invstd[n0 + tid - 64];
    else if (tid < 192) R.st = E.bias[n0 + tid - 128];
}

template <bool AK>
__device__ static inline void g6_block(const G6Desc& D, int bm, int bn, int ks, float* sA, float* sB, float* sSt, float2* sRed) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int wm = wave & 3, wn = wave >> 2;
    const int m0 = bm * G6_T, n0 = bn * G6_T;
    G6Epi ep;
    if (D.epi.x) g6_epi_prefetch(D, bm, bn, tid, ep);
    const int kper = D.K / D.k_split, k_lo = ks * kper, k_hi = k_lo + kper;
    f32x4 c00 = {0.f, 0.f, 0.f, 0.f}, c01 = c00, c10 = c00, c11 = c00;     // [n-tile][interleave]
    f32x4 va[G6_PT], vb[G6_PT];
    g6_load<AK>(va, D.A, D.lda, m0, D.M, k_lo, tid);
    g6_load<true>(vb, D.B, D.ldb, n0, D.N, k_lo, tid);
    for (int k0 = k_lo; k0 < k_hi; k0 += G6_KC) {
        if (k0 != k_lo) __syncthreads();                  // previous chunk fully consumed
        g6_store<AK>(sA, va, tid);
        g6_store<true>(sB, vb, tid);
        if (k0 + G6_KC < k_hi) {                          // next chunk's loads fly under this chunk's MFMAs
            g6_load<AK>(va, D.A, D.lda, m0, D.M, k0 + G6_KC, tid);
            g6_load<true>(vb, D.B, D.ldb, n0, D.N, k0 + G6_KC, tid);
        }
        __syncthreads();
#pragma unroll 2
        for (int kk = 0; kk < G6_KC; kk += 16) {
            const f32x4 a = g6_frag<AK>(sA, 16 * wm + r, g, kk);
            const f32x4 b0 = g6_frag<true>(sB, 32 * wn + r, g, kk), b1 = g6_frag<true>(sB, 32 * wn + 16 + r, g, kk);
            c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0.x, c00, 0, 0, 0);
            c10 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b1.x, c10, 0, 0, 0);
            c01 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b0.y, c01, 0, 0, 0);
            c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1.y, c11, 0, 0, 0);
            c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b0.z, c00, 0, 0, 0);
            c10 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b1.z, c10, 0, 0, 0);
            c01 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b0.w, c01, 0, 0, 0);
            c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b1.w, c11, 0, 0, 0);
        }
    }
    const f32x4 t0 = c00 + c01, t1 = c10 + c11;
    if (D.C) {
        float* Cs = D.C + (int64_t)ks * D.c_split_stride;
        const int cm = m0 + 16 * wm + 4 * g, cn = n0 + 32 * wn + r;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (cm + e < D.M) {
                if (cn < D.N) Cs[(int64_t)(cm + e) * D.ldc + cn] = t0[e];
                if (cn + 16 < D.N) Cs[(int64_t)(cm + e) * D.ldc + cn + 16] = t1[e];
            }
    }
    if (!D.epi.x) return;
    // ---- layer-1 backward pass on the 64 x 64 tile of dA1 (see gemm_bundle.hip, gemm_l1bwd_epilogue) ------------------
    const naf_gemm_l1bwd_t& E = D.epi;
    const int KP = E.kp, XS = KP + 4;
    float* sX = sA;                    // [64 rows][XS]
    float* sW = sA + G6_T * XS;        // [64 cols][XS]
    float* sDA = sB;                   // [64][65]: the C tile, then dy in place
    __syncthreads();                   // every wave is past its last fragment read: the panels are free
    {
        const int xr = tid / (KP / 4), xq = tid - xr * (KP / 4);
        if (xr < G6_T) *(f32x4*)(sX + xr * XS + 4 * xq) = ep.x;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + G6_THREADS * i;
            const int c = e / KP, k = e - c * KP;
            if (c < G6_T) sW[c * XS + k] = ep.w[i];
        }
        if (tid < 192) sSt[tid] = ep.st;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sDA[(16 * wm + 4 * g + e) * 65 + 32 * wn + r] = t0[e];
            sDA[(16 * wm + 4 * g + e) * 65 + 32 * wn + 16 + r] = t1[e];
        }
    }
    __syncthreads();
    const int col = tid & 63;
    const float mean = sSt[col], invstd = sSt[64 + col], b = sSt[128 + col];
    f32x4 wv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) wv[q] = q < KP / 4 ? *(const f32x4*)(sW + col * XS + 4 * q) : (f32x4){0.f, 0.f, 0.f, 0.f};
    float s_dy = 0.f, s_dx = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = (tid >> 6) + 8 * i;
        float z = b;                                         // b + sum_k x_k w_k, k ascending: the forward's arithmetic
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q < KP / 4) {
                const f32x4 xv = *(const f32x4*)(sX + row * XS + 4 * q);     // (one row per wave: a broadcast read)
                z = __builtin_fmaf(xv.x, wv[q].x, z);
                z = __builtin_fmaf(xv.y, wv[q].y, z);
                z = __builtin_fmaf(xv.z, wv[q].z, z);
                z = __builtin_fmaf(xv.w, wv[q].w, z);
            }
        const float xh = (z - mean) * invstd;
        const float dy = ep.a1[i] > 0.f ? sDA[row * 65 + col] : 0.f;
        sDA[row * 65 + col] = dy;                            // (this thread's own element)
        s_dy += dy;
        s_dx += dy * xh;
    }
    sRed[wave * 64 + col] = make_float2(s_dy, s_dx);          // 8 rows per wave; the waves meet below
    __syncthreads();
    if (tid < 64) {
        float2 t = sRed[tid];
#pragma unroll
        for (int w = 1; w < 8; ++w) {
            t.x += sRed[w * 64 + tid].x;
            t.y += sRed[w * 64 + tid].y;
        }
        ((float2*)E.partials)[(int64_t)bm * D.N + n0 + tid] = t;
    }
    {   // P share of the block: thread = (column, k lane of 8): k = lane, lane + 8, lane + 16, (lane + 24); row ascending
        const int c = tid >> 3, kq = tid & 7;
        float p[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int row = 0; row < G6_T; ++row) {
            const float dy = sDA[row * 65 + c];
#pragma unroll
            for (int j = 0; j < 4; ++j) p[j] = __builtin_fmaf(dy, sX[row * XS + (kq + 8 * j < KP ? kq + 8 * j : kq)], p[j]);
        }
        float* dst = E.p_slabs + ((int64_t)bm * D.N + n0 + c) * KP;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (kq + 8 * j < KP) dst[kq + 8 * j] = p[j];
    }
}

__global__ __launch_bounds__(G6_THREADS) void gemm_bundle64_kernel(const G6Bundle bundle) {
    __shared__ __attribute__((aligned(16))) float sA[G6_PANEL];
    __shared__ __attribute__((aligned(16))) float sB[G6_PANEL];
    __shared__ float sSt[192];
    __shared__ float2 sRed[8 * 64];
    const int t = blockIdx.x;
    int gi = 0;
#pragma unroll
    for (int i = 1; i < NAF_GEMM_BUNDLE_MAX; ++i)
        if (i < bundle.n && t >= bundle.d[i].tile0) gi = i;
    const G6Desc& D = bundle.d[gi];
    const int ks = (t - D.tile0) / D.tiles_mn;
    const int lt = t - D.tile0 - ks * D.tiles_mn;
    const int bm = lt / D.tiles_n, bn = lt - bm * D.tiles_n;
    if (D.a_kmajor) g6_block<true>(D, bm, bn, ks, sA, sB, sSt, sRed);
    else g6_block<false>(D, bm, bn, ks, sA, sB, sSt, sRed);
}

extern "C" int naf_gemm_bundle64(const naf_gemm_desc_t* descs, int n, void* stream) {
    if (!descs || n <= 0 || n > NAF_GEMM_BUNDLE_MAX) return NAF_ERR_ARG;
    G6Bundle b;
    memset(&b, 0, sizeof(b));
    b.n = n;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        const naf_gemm_desc_t& s = descs[i];
        if (!s.A || !s.B || (!s.C && !s.epi) || s.M <= 0 || s.N <= 0 || s.K <= 0 || !s.b_kmajor || s.sumsq) return NAF_ERR_ARG;
        const int ksn = s.k_split > 0 ? s.k_split : 1;
        if ((s.M & 3) || (s.N & 3) || s.K % ksn || ((s.K / ksn) % G6_KC)) return NAF_ERR_ARG;    // whole 128-k chunks per K range
        if (ksn > 1 && s.c_split_stride < (int64_t)s.M * s.ldc) return NAF_ERR_ARG;
        if (s.lda < (s.a_kmajor ? s.M : s.K) || s.ldb < s.N || (s.C && s.ldc < s.N)) return NAF_ERR_ARG;
        if ((((uintptr_t)s.A) & 15) || (s.lda & 3) || (((uintptr_t)s.B) & 15) || (s.ldb & 3)) return NAF_ERR_ARG;
        G6Desc& d = b.d[i];
        d.A = s.A; d.B = s.B; d.C = s.C;
        d.M = s.M; d.N = s.N; d.K = s.K;
        d.lda = s.lda; d.ldb = s.ldb; d.ldc = s.ldc;
        d.a_kmajor = s.a_kmajor;
        d.tile0 = tiles;
        d.tiles_n = (s.N + G6_T - 1) / G6_T;
        d.tiles_mn = ((s.M + G6_T - 1) / G6_T) * d.tiles_n;
        d.k_split = ksn;
        d.c_split_stride = s.c_split_stride;
        if (s.epi) {
            const naf_gemm_l1bwd_t& e = *s.epi;
            if (!e.x || !e.W || !e.bias || !e.a1 || !e.save_mean || !e.save_invstd || !e.partials || !e.p_slabs || ksn != 1 ||
                (s.M % G6_T) || (s.N % G6_T) || e.K <= 0 || (e.kp != 24 && e.kp != 32) || e.K > e.kp || e.ldx < e.kp || (e.ldx & 3) ||
                e.lda1 < s.N || ((uintptr_t)e.x & 15) || ((uintptr_t)e.partials & 7))
                return NAF_ERR_ARG;
            d.epi = e;
        }
        tiles += d.tiles_mn * ksn;
    }
    for (int i = n; i < NAF_GEMM_BUNDLE_MAX; ++i) b.d[i] = b.d[0];
    gemm_bundle64_kernel<<<tiles, G6_THREADS, 0, (hipStream_t)stream>>>(b);
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}
