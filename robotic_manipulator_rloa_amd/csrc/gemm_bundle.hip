// Several small f32 GEMMs in ONE launch on v_mfma_f32_16x16x4_f32 (gfx950). The backward of a learn() update needs
// three independent products once dZ2 and dH exist — dW2 = dZ2^T A1, dA1 = dZ2 W2, dWh = dH^T A2 (autograd of
// naf_neural_network.py:76-87) — each ~33 MFLOP or less: as three library launches they cost three launch
// boundaries (~3.3 us each, all latency); as one grid of 16x16 output tiles (546 tiles at B=256) they fill half the
// chip once.
//
// One wave per 16x16 output tile, operands straight from L2 (every operand here was just written by the previous
// kernel and is < 300 KB: no LDS staging, no reuse to exploit beyond L2). Lane (r = l & 15, g = l >> 4), macro-step j
// (16 k's), component c: k = 16 j + 4 g + c for BOTH operands, so the four MFMAs of a macro-step use each k once:
//   operand stored k-contiguous ([rows][K], e.g. dZ2 as A of dZ2 @ W2): one float4 per lane per macro-step
//   operand stored k-major     ([K][rows], e.g. dZ2 as A of dZ2^T @ A1): four 4-byte loads, 64 B contiguous per
//                                                                         16-lane group
// C/D map: col = l & 15, row = 4 (l >> 4) + reg. Summation order over k is fixed -> bitwise reproducible.
#include "common.h"
#include "../../include/naf_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GemmDesc {
    const float* A;   // a_kmajor ? [K][M] (ld = lda) : [M][K]
    const float* B;   // b_kmajor ? [K][N] (ld = ldb) : [N][K]
    float* C;         // [M][N], ld = ldc
    int M, N, K, lda, ldb, ldc, a_kmajor, b_kmajor, tile0, tiles_n;
};
struct GemmBundle {
    GemmDesc d[NAF_GEMM_BUNDLE_MAX];
    int n, total_tiles;
};

template <bool KMAJOR>
__device__ static inline void load_frag(const float* __restrict__ p, int ld, int r0, int r, int g, int k0, float* f) {
    if (KMAJOR) {   // element (row = r0 + r, k) at p[k * ld + r0 + r]
        const float* q = p + (int64_t)(k0 + 4 * g) * ld + r0 + r;
        f[0] = q[0];
        f[1] = q[ld];
        f[2] = q[2 * (int64_t)ld];
        f[3] = q[3 * (int64_t)ld];
    } else {        // element (row, k) at p[row * ld + k]
        const float4 v = *(const float4*)(p + (int64_t)(r0 + r) * ld + k0 + 4 * g);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    }
}

// STEPS macro-steps (16 k's each) with EVERY operand fragment requested before the first MFMA: a tile's K loop is a
// dependent chain (load -> MFMA), and with one L2 round trip per macro-step it cost 8.4 us per tile at K = 256;
// with the loads of 16 macro-steps in flight together it is one round trip + 64 MFMAs. Registers are free here
// (one tile per wave, nothing else resident).
template <bool AK, bool BK, int STEPS>
__device__ static inline void gemm_chunk(const GemmDesc& D, int m0, int n0, int r, int g, int k0, f32x4& acc0, f32x4& acc1) {
    float a[STEPS][4], b[STEPS][4];
#pragma unroll
    for (int j = 0; j < STEPS; ++j) {
        load_frag<AK>(D.A, D.lda, m0, r, g, k0 + 16 * j, a[j]);
        load_frag<BK>(D.B, D.ldb, n0, r, g, k0 + 16 * j, b[j]);
    }
#pragma unroll
    for (int j = 0; j < STEPS; ++j) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][0], b[j][0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][1], b[j][1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][2], b[j][2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][3], b[j][3], acc1, 0, 0, 0);
    }
}

template <bool AK, bool BK>
__device__ static inline void gemm_tile(const GemmDesc& D, int tm, int tn, int lane) {
    const int r = lane & 15, g = lane >> 4;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const int m0 = tm * 16, n0 = tn * 16;
    int k0 = 0;
    for (; D.K - k0 >= 256; k0 += 256) gemm_chunk<AK, BK, 16>(D, m0, n0, r, g, k0, acc0, acc1);
    if (D.K - k0 >= 128) { gemm_chunk<AK, BK, 8>(D, m0, n0, r, g, k0, acc0, acc1); k0 += 128; }
    if (D.K - k0 >= 64) { gemm_chunk<AK, BK, 4>(D, m0, n0, r, g, k0, acc0, acc1); k0 += 64; }
    if (D.K - k0 >= 32) { gemm_chunk<AK, BK, 2>(D, m0, n0, r, g, k0, acc0, acc1); k0 += 32; }
    if (D.K - k0 >= 16) { gemm_chunk<AK, BK, 1>(D, m0, n0, r, g, k0, acc0, acc1); k0 += 16; }
    float* c = D.C + (int64_t)(m0 + 4 * g) * D.ldc + n0 + r;
#pragma unroll
    for (int e = 0; e < 4; ++e) c[(int64_t)e * D.ldc] = acc0[e] + acc1[e];
}

__global__ __launch_bounds__(256) void gemm_bundle_kernel(const GemmBundle bundle) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);   // one 16x16 tile per wave
    if (t >= bundle.total_tiles) return;
    int gi = 0;
#pragma unroll
    for (int i = 1; i < NAF_GEMM_BUNDLE_MAX; ++i)
        if (i < bundle.n && t >= bundle.d[i].tile0) gi = i;
    const GemmDesc& D = bundle.d[gi];
    const int lt = t - D.tile0;
    const int tm = lt / D.tiles_n, tn = lt - tm * D.tiles_n;
    if (D.a_kmajor) {
        if (D.b_kmajor) gemm_tile<true, true>(D, tm, tn, lane);
        else gemm_tile<true, false>(D, tm, tn, lane);
    } else {
        if (D.b_kmajor) gemm_tile<false, true>(D, tm, tn, lane);
        else gemm_tile<false, false>(D, tm, tn, lane);
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
        if (s.lda < (s.a_kmajor ? s.M : s.K) || s.ldb < (s.b_kmajor ? s.N : s.K) || s.ldc < s.N) return NAF_ERR_ARG;
        if (!s.a_kmajor && ((((uintptr_t)s.A) & 15) || (s.lda & 3))) return NAF_ERR_ARG;   // float4 fragments
        if (!s.b_kmajor && ((((uintptr_t)s.B) & 15) || (s.ldb & 3))) return NAF_ERR_ARG;
        GemmDesc& d = b.d[i];
        d.A = s.A; d.B = s.B; d.C = s.C;
        d.M = s.M; d.N = s.N; d.K = s.K;
        d.lda = s.lda; d.ldb = s.ldb; d.ldc = s.ldc;
        d.a_kmajor = s.a_kmajor; d.b_kmajor = s.b_kmajor;
        d.tile0 = tiles;
        d.tiles_n = s.N / 16;
        tiles += (s.M / 16) * (s.N / 16);
    }
    for (int i = n; i < NAF_GEMM_BUNDLE_MAX; ++i) b.d[i] = b.d[0];
    b.total_tiles = tiles;
    gemm_bundle_kernel<<<(tiles + 3) / 4, 256, 0, (hipStream_t)stream>>>(b);
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}
