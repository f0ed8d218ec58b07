// Several small f32 GEMMs in ONE launch on v_mfma_f32_16x16x4_f32 (gfx950). The backward of a learn() update needs
// three independent products once dZ2 and dH exist — dW2 = dZ2^T A1, dA1 = dZ2 W2, dWh = dH^T A2 (autograd of
// naf_neural_network.py:76-87) — each ~33 MFLOP or less: as three library launches they cost three launch
// boundaries (~3.3 us each, all latency); as one grid of 16x16 output tiles (546 tiles at B=256) they fill half the
// chip once.
//
// C/D map of the MFMA: col = l & 15, row = 4 (l >> 4) + reg. Summation order over k is fixed -> bitwise reproducible.
#include <string.h>
#include <stdlib.h>
#include "common.h"
#include "../../include/naf_hip.h"
#include "bn2bwd_fold.h"

// naf_gemm_l1bwd_t as the kernel sees it: the ABI struct WITHOUT its `rows` field — that one travels in GemmDesc::c_split_stride
// (free in a descriptor with an epilogue: k_split == 1). The layout of a descriptor is then what it was before round 4 added the
// field: with 8 bytes more per descriptor the launch was 0.3 us slower at B = 256 (A/B on one box, same kernel code otherwise —
// which scalar loads share a cache line in front of the first vector load is all that changed).
struct L1bwdDev {
    const float *x, *W, *bias, *a1, *save_mean, *save_invstd;
    float *partials, *p_slabs;
    int ldx, K, kp, lda1;
    const float *xhat, *gamma, *beta;
};
struct GemmDesc {
    const float* A;   // a_kmajor ? [K][M] (ld = lda) : [M][K]
    const float* B;   // b_kmajor ? [K][N] (ld = ldb) : [N][K]
    float* C;         // [M][N], ld = ldc
    float* sumsq;     // nullable: sumsq[block] = sum of C^2 over the block (grad-norm partial)
    int M, N, K, lda, ldb, ldc, a_kmajor, b_kmajor, tile0, tiles_n;
    int k_split, tiles_mn;        // K cut into k_split ranges, one grid of tiles_mn blocks each, slab s at C + s * c_split_stride
    int64_t c_split_stride;       // (a descriptor with an epilogue: the number of its M rows that are samples, naf_gemm_l1bwd_t.rows)
    L1bwdDev epi;                 // epi.x != NULL: the layer-1 backward pass on the block's C tile (see gemm_l1bwd_epilogue)
    naf_gemm_bn2bwd_t pro;        // pro.z != NULL: A = dY2 becomes dZ2 while it is staged (see gemm_bn2bwd_constants)
};
struct GemmBundle {
    GemmDesc d[NAF_GEMM_BUNDLE_MAX];
    int n, total_tiles;
    int rowmap;                   // XCD-aware placement of the blocks (0: block t is block t; the host sets 2)
    int n_fold, fold_desc;        // n_fold > 0: the first n_fold workgroups fold the block sums of d[fold_desc].pro ONCE for the launch
    naf_gemm_bn2bwd_t fold_pro;   // = d[fold_desc].pro, at an offset the folding workgroups know without reading anything first
};

// ---- LDS-staged form -----------------------------------------------------------------------------------------------
// Workgroup = 512 threads = 8 waves = one 32 x 32 output block (2 x 2 MFMA tiles, two waves per tile: K halves). A K-chunk of 256 of
// both operand panels (32 rows x 256 k each) is staged into LDS with 16-byte global loads and 16-byte LDS stores, then
// every wave reads its fragments: lane (r, g) takes k = 16 j + 4 g .. +3 of row r for BOTH operands, so the four MFMAs
// of macro-step j use each k once. One L2 round trip per 256 k instead of one per
// fragment (the register-fed form of this kernel issued 256 4-byte loads per lane per tile and ran 10.2 us).
// Two forms of the kernel (template parameters T = threads per workgroup, KC = k per staged chunk):
//   <512, 256>  8 waves, two per 16 x 16 tile (K halves), 77.8 KB of LDS: two workgroups per CU. Launches of up to 512 blocks
//               (B <= 1024): one round, and the block's own latency chain is what counts.
//   <256, 128>  4 waves, one per tile, 40 KB of LDS: FOUR workgroups per CU (the same 16 waves), so the 920 blocks of a
//               B = 2048 launch are resident at once instead of in two rounds of 512. Worth 1.5 - 2 %, not more: with every
//               block staging at once the staging phase takes 4.5 us instead of 2.6 — the launch pulls 87 MB of panels through
//               the L2 -> L1 path either way (~19 TB/s over the chip, ~75 GB/s per CU), and its waves spend 43 % of their cycles
//               parked on memory or barriers and 38 % waiting for the MFMA pipe their neighbours hold (SQ_WAIT_ANY /
//               SQ_WAIT_INST_ANY, profiles/r03_pmc_sq_b2048.csv): phases of co-resident blocks coincide instead of interleaving.
// TAIL: some descriptor with an epilogue has rows past the batch in its last row block (a batch that is not whole 32-row blocks):
// a kernel of its own. The launches of every other batch size run code without a trace of it — as a block-uniform branch inside one
// kernel the clamps and the mask cost 0.25 us per update at B = 256 (A/B on one box; the blocks' cold straight-line code got longer).
template <int T, int KC, bool TAIL_ = false, bool HBIG_ = false>
struct GB {
    static constexpr bool TAIL = TAIL_;
    static constexpr bool HBIG = HBIG_;        // layer sizes beyond 256 (round 6): the prologue's column constants refilled per 256 k
    static constexpr int THREADS = T, CHUNK = KC;
    static constexpr int LD = KC + 4;          // [row][k] panels: 16-B aligned rows, b128 fragment reads spread over the banks
    static constexpr int LDK = 36;             // [k][row] panels (k-major operands keep their memory layout): 32 rows + 4 pad
    static constexpr int PANEL = KC * LDK;     // floats per panel buffer (>= 32 * LD)
    static constexpr int KSPLIT = T / 256;     // waves per 16 x 16 tile: each takes 1/KSPLIT of the K chunk
    static constexpr int PT = KC * 8 / T;      // float4 per thread per panel
    static constexpr int RK = KC / 4;          // float4 per row of a k-contiguous panel: T / RK = 8 rows per pass
    static_assert(PT == 4 && T / RK == 8 && 32 * LD <= PANEL && (KSPLIT == 1 || KSPLIT == 2), "panel decomposition");
};
#define GB_PT 4

// Stage a 32-row x kc panel with 16-byte loads AND 16-byte LDS stores, in the operand's own memory order:
//   k-contiguous operand -> LDS [row][k] (stride LD), fragments read as one ds_read_b128 per macro-step
//   k-major operand      -> LDS [k][row] (stride LDK), fragments read as four ds_read_b32 (bank = 4k + row: the two
//                           16-lane groups a b32 read serves per cycle never collide)
// (the first version transposed k-major panels while staging: 4 ds_write_b32 per float4 with a 4-way bank conflict)
// A panel is 32 rows x kc k = kc * 8 float4, KC * 8 / T = 4 per thread. ALL of a thread's loads — of both panels —
// are issued before the first LDS store: as a load -> store loop (one load in flight per thread) the staging was 16
// serial memory round trips per block, ~200 cycles each on L2 hits but 545+ on data the previous kernel had just
// written (Infinity Cache): the whole fresh-data penalty of this kernel (round 2's benchmarks/chain_probe.py: 1.8 of its 8.5 us).
// FULL = the chunk is a whole KC (every call but the tail of a K that is not a multiple of it): row / k indices are
// shifts; the general form divides by a run-time k4n once per element — ~20 integer instructions, 32 times per thread,
// in a kernel whose waves run ~1,100 instructions in all.
template <class G, bool KMAJOR, bool FULL>
__device__ static inline void load_panel(float4 (&v)[GB_PT], const float* __restrict__ p, int ld, int row0,
                                         int rows_total, int k0, int kc, int tid) {
#pragma unroll
    for (int i = 0; i < GB_PT; ++i) {
        const int e = tid + G::THREADS * i;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (FULL || e < kc * 8) {
            if (KMAJOR) {
                const int k = e >> 3, r4 = (e & 7) * 4;
                if (row0 + r4 < rows_total) v[i] = *(const float4*)(p + (int64_t)(k0 + k) * ld + row0 + r4);
            } else {
                const int k4n = FULL ? G::RK : kc >> 2;
                const int row = FULL ? e / G::RK : e / k4n, k4 = (e - row * k4n) * 4;
                if (row0 + row < rows_total) v[i] = *(const float4*)(p + (int64_t)(row0 + row) * ld + k0 + k4);
            }
        }
    }
}

template <class G, bool KMAJOR, bool FULL>
__device__ static inline void store_panel(float* __restrict__ sm, const float4 (&v)[GB_PT], int kc, int tid) {
#pragma unroll
    for (int i = 0; i < GB_PT; ++i) {
        const int e = tid + G::THREADS * i;
        if (FULL || e < kc * 8) {
            if (KMAJOR) {
                const int k = e >> 3, r4 = (e & 7) * 4;
                *(float4*)(sm + k * G::LDK + r4) = v[i];
            } else {
                const int k4n = FULL ? G::RK : kc >> 2;
                const int row = FULL ? e / G::RK : e / k4n, k4 = (e - row * k4n) * 4;
                *(float4*)(sm + row * G::LD + k4) = v[i];
            }
        }
    }
}

// The same panel (whole KC chunks only) through buffer loads (common.h: naf_buf_*): wave-uniform resource and chunk /
// row offsets on the scalar unit, ONE lane offset per panel computed once per block. The flat-addressed form above spends
// ~10 vector instructions per float4 (64-bit multiply-adds, bounds selects); with 8 loads per thread per chunk, 8 waves per
// block and two blocks per CU the blocks of a full launch were bound by VALU issue — every block of a 512-block round took
// 6.3 us where an isolated one takes 3.2 (benchmarks/kernel_timeline.py, the ISA: 250 v_mul_lo_u32 + 170 v_mad_i64_i32 +
// 300 v_lshl_add_u64 in the kernel). Rows / columns past the matrix read as 0: the resource ends at the matrix' last byte
// (row bound), a lane whose float4 lies past the operand's contiguous dimension gets an offset past everything.
struct PanelSrc {
    __amdgpu_buffer_rsrc_t rs;
    unsigned voff, ld4;
};
template <class G, bool KMAJOR>
__device__ __forceinline__ static PanelSrc panel_src(const float* __restrict__ p, int ld, int row0, int rows_total, int k_total, int tid) {
    PanelSrc s;
    s.ld4 = (unsigned)ld * 4u;
    if (KMAJOR) {                                          // [K][rows]: (k, row) at (k * ld + row) * 4; k = (tid >> 3) + (T / 8) i
        s.rs = naf_buf(p, (unsigned)k_total * s.ld4);
        const int r4 = (tid & 7) * 4;
        s.voff = row0 + r4 < rows_total ? (unsigned)(tid >> 3) * s.ld4 + (unsigned)(row0 + r4) * 4u : 0x7f000000u;
    } else {                                               // [rows][K]: row = tid / RK + 8 i, k = 4 (tid % RK); RK = 64: one row per wave
        s.rs = naf_buf(p, (unsigned)rows_total * s.ld4);
        const int lane = tid & 63;
        s.voff = (unsigned)(lane % G::RK) * 16u + (unsigned)(lane / G::RK) * s.ld4;
    }
    return s;
}
template <class G, bool KMAJOR>
__device__ __forceinline__ static void load_panel_buf(float4 (&v)[GB_PT], const PanelSrc& s, int row0, int k0, int wave) {
#pragma unroll
    for (int i = 0; i < GB_PT; ++i) {
        const unsigned soff = KMAJOR ? (unsigned)(k0 + (G::THREADS / 8) * i) * s.ld4
                                     : (unsigned)(row0 + wave * (64 / G::RK) + 8 * i) * s.ld4 + (unsigned)k0 * 4u;
        const naf_f32x4 t = naf_buf_f4(s.rs, s.voff, soff);
        v[i] = make_float4(t.x, t.y, t.z, t.w);
    }
}

// fragment of macro-step kk for lane (r, g): elements k = kk + 4 g + c, c = 0..3, of panel row `row`
template <class G, bool KMAJOR>
__device__ static inline float4 read_frag(const float* __restrict__ sm, int row, int g, int kk) {
    if (KMAJOR) {
        const float* q = sm + (kk + 4 * g) * G::LDK + row;
        return make_float4(q[0], q[G::LDK], q[2 * G::LDK], q[3 * G::LDK]);
    }
    return *(const float4*)(sm + row * G::LD + kk + 4 * g);
}

// Epilogue of the dA1 = dZ2 W2 blocks in the large-batch chain: the C tile (32 batch rows x 32 layer-1 features) is the
// gradient w.r.t. the layer-1 activation, and the whole batch pass of layer 1's backward runs on it without a trip through
// memory (csrc/big_batch.hip, bb_layer1_bwd_kernel does the same as a launch of its own): z recomputed from the minibatch
// rows and W1 (K <= 32), xhat, dy = ReLU'(A1) * dA1, the block sums (sum dy, sum dy*xhat) per column -> partials[M/32][N],
// and the block's share of P = dY^T X -> p_slabs[M/32][N][KP]. dA1 itself is not written (C may be NULL).
// Everything the epilogue reads from memory is requested BEFORE the K loop (gemm_l1bwd_prefetch) and waits in registers.
// Both small products run on MFMA in the accumulator waves' own layout: z = X W1^T (2 x 2 tiles, K = KP) by the four waves
// that hold the dA1 tiles — dy never leaves their registers before it is final — and P = dY^T X (2 x 2 tiles, K = 32 rows)
// by the other four, from dy in LDS. (The first version did both on the VALU, one (row, column) pair per thread with LDS
// operands: 2.1 - 2.8 us of a 7 us block, and two resident blocks per CU take turns at it — benchmarks/kernel_timeline.py.)
template <class G>
struct L1bwdRegs {
    static constexpr int NW = 1024 / G::THREADS;      // scalars of the 32 x KP W1 tile per thread
    f32x4 x;           // one float4 of the X tile (threads < 32 * KP / 4)
    float w[NW];
    float a1[4];       // accumulator waves: A1 at the lane's four C/D elements (rows 4 g + e, column r of the tile)
    float mean, invstd, bias;   // of the lane's column
    int rows_on;       // bit e: the lane's row 4 g + e of the tile is a sample (a batch that is not whole 32-row blocks: E.rows)
};
template <class G>
__device__ static inline void gemm_l1bwd_prefetch(const GemmDesc& D, int bm, int bn, int tid, bool owner, int wm, int wn, int r, int g,
                                                  L1bwdRegs<G>& R) {
    const L1bwdDev& E = D.epi;
    const int KP = E.kp, m0 = bm * 32, n0 = bn * 32;
    const int xr = KP == 24 ? tid / 6 : tid / 8, xq = tid - xr * (KP / 4);       // (KP is 24 or 32: divisions by constants)
    R.x = (f32x4){0.f, 0.f, 0.f, 0.f};
    // (G::TAIL: rows past the batch — D.c_split_stride of the M rows are samples — are read as the last row that exists and carry
    //  dy = 0 through the epilogue)
    const int Mv = G::TAIL ? (int)D.c_split_stride : 0x7fffffff, mlast = Mv - 1;
    R.rows_on = 15;
    if (G::TAIL) {
        const int row0 = m0 + wm * 16 + 4 * g;
        R.rows_on = (row0 < Mv ? 1 : 0) | (row0 + 1 < Mv ? 2 : 0) | (row0 + 2 < Mv ? 4 : 0) | (row0 + 3 < Mv ? 8 : 0);
    }
    if (xr < 32) R.x = ((const f32x4*)(E.x + (int64_t)((G::TAIL && m0 + xr >= Mv) ? mlast : m0 + xr) * E.ldx))[xq];
#pragma unroll
    for (int i = 0; i < L1bwdRegs<G>::NW; ++i) {
        const int e = tid + G::THREADS * i;
        const int c = KP == 24 ? e / 24 : e / 32, k = e - c * KP;
        R.w[i] = (!E.xhat && c < 32 && k < E.K) ? E.W[(int64_t)(n0 + c) * E.K + k] : 0.f;
    }
    const int col = n0 + wn * 16 + r;
    if (E.xhat) {
        // xhat ready-made (naf_bb_layer1_adam kept it): the lane's four C/D elements of it in a1[], gamma / beta of its column in
        // mean / invstd — the ReLU mask is the forward's own fma(xhat, gamma, beta) > 0; nothing of W1 is needed
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = m0 + wm * 16 + 4 * g + e;
            R.a1[e] = owner ? E.xhat[(int64_t)((G::TAIL && row >= Mv) ? mlast : row) * E.lda1 + col] : 0.f;
        }
        R.mean = E.gamma[col];
        R.invstd = E.beta[col];
        R.bias = 0.f;
        return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int row = m0 + wm * 16 + 4 * g + e;
        R.a1[e] = owner ? E.a1[(int64_t)((G::TAIL && row >= Mv) ? mlast : row) * E.lda1 + col] : 0.f;
    }
    R.mean = E.save_mean[col];
    R.invstd = E.save_invstd[col];
    R.bias = E.bias[col];
}

// xhat of the block's layer-1 tile, EARLY: it needs only the minibatch rows and W1 (requested first of all), not the product, so it
// is formed while the panels are in flight and the block waits for the BatchNorm-backward records anyway — the panels' LDS space is
// still free then. Accumulator waves return xhat at their four C/D elements; all waves take the two barriers. (Inside the
// epilogue this was 0.5 of its 0.9 us, behind the product: updates/s A/B on one box 26.7k -> 26.9k at B = 1024, no change at
// 256, where the weight-gradient blocks end the launch.)
template <class G>
__device__ static inline f32x4 gemm_l1bwd_xhat(const GemmDesc& D, bool owner, int wm, int wn, int r, int g, float* sA, int tid,
                                               const L1bwdRegs<G>& R) {
    const L1bwdDev& E = D.epi;
    const int KP = E.kp, XS = KP + 4;
    float* sX = sA;                    // [32 rows][XS]
    float* sW = sA + 32 * XS;          // [32 cols][XS]
    {
        const int xr = KP == 24 ? tid / 6 : tid / 8, xq = tid - xr * (KP / 4);
        if (xr < 32) *(f32x4*)(sX + xr * XS + 4 * xq) = R.x;
#pragma unroll
        for (int i = 0; i < L1bwdRegs<G>::NW; ++i) {
            const int e = tid + G::THREADS * i;
            const int c = KP == 24 ? e / 24 : e / 32, k = e - c * KP;
            if (c < 32) sW[c * XS + k] = R.w[i];
        }
    }
    __syncthreads();
    f32x4 xh = {0.f, 0.f, 0.f, 0.f};
    if (owner) {
        // z tile: rows wm * 16 .. +15 x columns wn * 16 .. +15, K = KP in two steps of 16 (KP = 24: lane groups 2, 3 of the second
        // step are past the row: zeros)
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 32; kk += 16) {
            if (kk < KP) {
                const bool in = kk + 4 * g < KP;
                const int ko = in ? kk + 4 * g : 0;
                f32x4 a = *(const f32x4*)(sX + (wm * 16 + r) * XS + ko), b = *(const f32x4*)(sW + (wn * 16 + r) * XS + ko);
                if (!in) a = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 4; ++q) z = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b[q], z, 0, 0, 0);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) xh[e] = ((z[e] + R.bias) - R.mean) * R.invstd;
    }
    __syncthreads();                   // the panels may be stored now
    return xh;
}

template <class G>
__device__ static inline void gemm_l1bwd_epilogue(const GemmDesc& D, int bm, int bn, const f32x4& acc, const f32x4& xh, bool owner,
                                                  int wm, int wn, int r, int g, float* sA, float* sB, int tid, const L1bwdRegs<G>& R) {
    // KSPLIT = 2: the two products are dealt to the accumulator waves (owner) and to the other four;
    // KSPLIT = 1: every wave holds an accumulator tile and takes its share of P behind the barrier.
    const L1bwdDev& E = D.epi;
    const int KP = E.kp, XS = KP + 4;
    const int n0 = bn * 32;
    float* sX = sA;                    // [32 rows][XS]: the minibatch rows again, for P = dY^T X
    float* sDY = sB;                   // [32 rows][33]
    float2* sRed = (float2*)(sB + 32 * 33);   // [2 row tiles][32 columns]
    // (KSPLIT = 2: the barrier behind the K halves' hand-over has every wave past its last fragment read: the panels are free)
    if (G::KSPLIT == 1) __syncthreads();
    {
        const int xr = KP == 24 ? tid / 6 : tid / 8, xq = tid - xr * (KP / 4);
        if (xr < 32) *(f32x4*)(sX + xr * XS + 4 * xq) = R.x;
    }
    if (owner) {
        float s_dy = 0.f, s_dx = 0.f;
        const bool kept = E.xhat != nullptr;                 // (uniform) a1[] = xhat, mean / invstd = gamma / beta of the column
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xhe = kept ? R.a1[e] : xh[e];
            // (TAIL: a row past the batch contributes nothing. Bitwise, not `&&`: the short-circuit form turned the selects of this
            //  loop into branches — +290 instructions in the kernel, +0.2 us in the epilogue of every dA1 block, also where the
            //  second operand was the constant true)
            const bool on0 = kept ? __builtin_fmaf(R.a1[e], R.mean, R.invstd) > 0.f : R.a1[e] > 0.f;
            const bool on = G::TAIL ? (bool)((int)on0 & (R.rows_on >> e) & 1) : on0;
            const float dy = on ? acc[e] : 0.f;
            sDY[(wm * 16 + 4 * g + e) * 33 + wn * 16 + r] = dy;
            s_dy += dy;
            s_dx += dy * xhe;
        }
        s_dy = naf_xor32_add(naf_xor16_add(s_dy));           // the tile's other row groups of the same column
        s_dx = naf_xor32_add(naf_xor16_add(s_dx));
        if (g == 0) sRed[wm * 32 + wn * 16 + r] = make_float2(s_dy, s_dx);
    }
    __syncthreads();
    if (tid < 32) {
        const float2 t0 = sRed[tid], t1 = sRed[32 + tid];
        ((float2*)E.partials)[(int64_t)bm * D.N + n0 + tid] = make_float2(t0.x + t1.x, t0.y + t1.y);
    }
    if (G::KSPLIT == 1 || !owner) {
        // P share of this block: tile (mt, nt) = columns mt * 16 .. +15 x k nt * 16 .. +15, reduction over the 32 rows.
        // A[m = column][k = row] = dy[row][column], B[k = row][n] = x[row][n]
        const int mt = wm, nt = wn;                          // (KSPLIT = 2: the four non-accumulator waves carry the same (wm, wn) pairs)
        const bool n_on = nt * 16 + r < KP;
        const int xc = n_on ? nt * 16 + r : 0;
        f32x4 pacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 32; kk += 16) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = kk + 4 * g + q;
                pacc = __builtin_amdgcn_mfma_f32_16x16x4f32(sDY[row * 33 + mt * 16 + r], sX[row * XS + xc], pacc, 0, 0, 0);
            }
        }
        if (n_on) {
#pragma unroll
            for (int e = 0; e < 4; ++e) E.p_slabs[((int64_t)bm * D.N + n0 + mt * 16 + 4 * g + e) * KP + nt * 16 + r] = pacc[e];
        }
    }
}

// Prologue of the products that read dZ2 (include/naf_hip.h, naf_gemm_bn2bwd_t): the second stage of layer 2's BatchNorm
// backward folded into the staging of their A panel, so that stage's launch (2.8 us + a 1.3 us boundary at B = 256) and the
// dZ2 round trip through memory disappear. The per-column constants sit in `cst` (4 x 256 floats of LDS: the K-halves buffer,
// free until the MFMAs are over):  dz = k1 dy - k1 c1 - (z - mean) (invstd k1 c2),  k1 = gamma invstd, c1 = sum dy / B, c2 = sum dy xhat / B
//   cst[0][c] = mean, [1] = k1, [2] = k1 c1, [3] = invstd k1 c2
// dy -> dz on a staged A panel. The thread's float4 i covers four consecutive COLUMNS of the operand:
//   k-contiguous A (K = H = 256: always whole chunks): row tid / RK + 8 i, columns k0 + 4 (tid % RK) .. +3;
//   k-major A: k = (tid >> 3) + (T / 8) i, columns 4 (tid & 7) .. +3 of the block — the same columns for every i, in whole chunks (buffer
//   loads) and in the tail chunk of a K range that is not a multiple of KC (load_panel<true, false>: e = tid + T i, column quad e & 7)
template <class G, bool AK>
__device__ __forceinline__ static void gemm_bn2bwd_apply(float4 (&va)[GB_PT], const float4 (&vz)[GB_PT], const float* cst, int tid, int k0) {
    const int ci = AK ? 4 * (tid & 7) : (G::HBIG ? (k0 & 255) : k0) + 4 * (tid & (G::RK - 1));
    const f32x4 mean = *(const f32x4*)(cst + ci), k1 = *(const f32x4*)(cst + 256 + ci), kc1 = *(const f32x4*)(cst + 512 + ci),
                q = *(const f32x4*)(cst + 768 + ci);
#pragma unroll
    for (int i = 0; i < GB_PT; ++i) {
        va[i].x = __builtin_fmaf(k1[0], va[i].x, -kc1[0]) - (vz[i].x - mean[0]) * q[0];
        va[i].y = __builtin_fmaf(k1[1], va[i].y, -kc1[1]) - (vz[i].y - mean[1]) * q[1];
        va[i].z = __builtin_fmaf(k1[2], va[i].z, -kc1[2]) - (vz[i].z - mean[2]) * q[2];
        va[i].w = __builtin_fmaf(k1[3], va[i].w, -kc1[3]) - (vz[i].w - mean[3]) * q[3];
    }
}

NAF_TL_DECL(g_tl_gb);
NAF_TL_READER(naf_tl_read_gb, g_tl_gb)
// every workgroup's entry / exit (is the grid resident at once? in which order do the blocks of the three GEMMs drain?)
#define GB_TL_WGS 4096
#ifdef NAF_TIMELINE
__device__ long long g_tl_gb_wg[2][GB_TL_WGS];
#define GB_TL_WG(which) do { if (threadIdx.x == 0 && blockIdx.x < GB_TL_WGS) g_tl_gb_wg[which][blockIdx.x] = wall_clock64(); } while (0)
int naf_tl_read_gb_wg(int first, long long* out) {
    if (first < 0 || first + NAF_TL_SLOTS > GB_TL_WGS) return NAF_ERR_ARG;
    for (int w = 0; w < 2; ++w) {
        const hipError_t e = hipMemcpyFromSymbol(out + w * NAF_TL_SLOTS, HIP_SYMBOL(g_tl_gb_wg), NAF_TL_SLOTS * sizeof(long long),
                                                 ((size_t)w * GB_TL_WGS + first) * sizeof(long long), hipMemcpyDeviceToHost);
        if (e != hipSuccess) return (int)e;
    }
    return NAF_OK;
}
#else
#define GB_TL_WG(which) do { } while (0)
int naf_tl_read_gb_wg(int, long long*) { return NAF_ERR_STATE; }
#endif
// (timeline marks: the first GEMM block — behind the folding workgroups, if any — and the launch's last)
#define GB_TL(slot) NAF_TL_FL(g_tl_gb, NAF_TL_GEMM_BUNDLE, slot, (int)blockIdx.x == tl_first, blockIdx.x == gridDim.x - 1)
template <class G, bool AK, bool BK>
__device__ static inline void gemm_block(const GemmDesc& D, int bm, int bn, int ks, float* sA, float* sB, float* sC, int tl_first) {
    constexpr int KC = G::CHUNK, KSPLIT = G::KSPLIT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform values on the scalar unit
    const int r = lane & 15, g = lane >> 4;
    // KSPLIT = 2 (8 waves): two per 16 x 16 tile of the 32 x 32 block, each taking one half of the K chunk — the per-wave chain of
    // dependent MFMAs (the longest single piece of this kernel: 1.5 of its 5.0 us with 64 of them) is halved; the two
    // halves meet through LDS, lower half first (fixed order). KSPLIT = 1 (4 waves): one wave per tile, whole chunks.
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
    L1bwdRegs<G> epi_regs;
    // Every field of the product's description that the block uses up to its first MFMA, requested NOW, in one batch. Left to
    // itself the compiler fetches a field where it is first used — behind the branches of the prefetch, of the panel sources and
    // of the prologue — and the kernel arguments are not in the scalar cache when a workgroup starts: ten dependent scalar round
    // trips in front of the first panel load (seen in the ISA; the persistent form of this kernel measured 1.7 us for five of
    // them with benchmarks/kernel_timeline.py).
    asm volatile("" ::"s"(D.A), "s"(D.B), "s"(D.M), "s"(D.N), "s"(D.K), "s"(D.lda), "s"(D.ldb), "s"(D.k_split), "s"(D.epi.x), "s"(D.epi.W),
                 "s"(D.epi.bias), "s"(D.epi.a1), "s"(D.epi.save_mean), "s"(D.epi.save_invstd), "s"(D.epi.ldx), "s"(D.epi.K),
                 "s"(D.epi.kp), "s"(D.epi.lda1), "s"(D.epi.xhat), "s"(D.epi.gamma), "s"(D.epi.beta), "s"(D.pro.z), "s"(D.pro.gamma), "s"(D.pro.save_mean), "s"(D.pro.save_invstd),
                 "s"(D.pro.cst), "s"(D.pro.epoch), "s"(D.pro.errors));
    GB_TL(0);
    GB_TL_WG(0);
    if (D.epi.x) gemm_l1bwd_prefetch<G>(D, bm, bn, tid, !kh, wm, wn, r, g, epi_regs);
    const int kper = D.K / D.k_split, k_lo = ks * kper, k_hi = k_lo + kper;
    const PanelSrc pa = panel_src<G, AK>(D.A, D.lda, m0, D.M, D.K, tid), pb = panel_src<G, BK>(D.B, D.ldb, n0, D.N, D.K, tid);
    const bool pro = D.pro.z != nullptr;                  // (uniform)
    const PanelSrc pz = panel_src<G, AK>(pro ? D.pro.z : D.A, D.lda, m0, D.M, D.K, tid);
    float4 va[GB_PT], vb[GB_PT], vz[GB_PT];
    {
        const int kc0 = kper < KC ? kper : KC;
        if (kc0 == KC) {
            load_panel_buf<G, AK>(va, pa, m0, k_lo, wave);
            if (pro) load_panel_buf<G, AK>(vz, pz, m0, k_lo, wave);
            load_panel_buf<G, BK>(vb, pb, n0, k_lo, wave);
        } else {
            load_panel<G, AK, false>(va, D.A, D.lda, m0, D.M, k_lo, kc0, tid);
            if (pro) load_panel<G, AK, false>(vz, D.pro.z, D.lda, m0, D.M, k_lo, kc0, tid);
            load_panel<G, BK, false>(vb, D.B, D.ldb, n0, D.N, k_lo, kc0, tid);
        }
    }
    f32x4 epi_xh = {0.f, 0.f, 0.f, 0.f};
    // (uniform) recomputed under the panel loads' latency — or nothing to do: the forward pass kept it and the prefetch has it
    if (D.epi.x && !D.epi.xhat) epi_xh = gemm_l1bwd_xhat<G>(D, !kh, wm, wn, r, g, sA, tid, epi_regs);
    if (pro) {                                            // the column constants, under the panel loads' latency too
        gemm_bn2bwd_wait_constants<AK, G::THREADS>(D.pro, m0, tid, sC);
        __syncthreads();
    }
    for (int k0 = k_lo; k0 < k_hi; k0 += KC) {
        const int kc = (k_hi - k0) < KC ? (k_hi - k0) : KC;
        if (k0 != k_lo) __syncthreads();                  // previous chunk fully consumed
        if (G::HBIG && !AK && pro && k0 != k_lo && (k0 & 255) == 0) {
            // (K = H = 512: the next 256 columns' constants; every wave is past the stretch before — the barrier above)
            gemm_bn2bwd_wait_constants<AK, G::THREADS>(D.pro, m0, tid, sC, k0);
            __syncthreads();
        }
        if (pro) gemm_bn2bwd_apply<G, AK>(va, vz, sC, tid, k0);
        if (kc == KC) {
            store_panel<G, AK, true>(sA, va, kc, tid);
            store_panel<G, BK, true>(sB, vb, kc, tid);
        } else {
            store_panel<G, AK, false>(sA, va, kc, tid);
            store_panel<G, BK, false>(sB, vb, kc, tid);
        }
        const int k1 = k0 + KC;
        if (k1 < k_hi) {                                  // next chunk's loads fly under this chunk's MFMAs
            const int kn = (k_hi - k1) < KC ? (k_hi - k1) : KC;
            if (kn == KC) {
                load_panel_buf<G, AK>(va, pa, m0, k1, wave);
                if (pro) load_panel_buf<G, AK>(vz, pz, m0, k1, wave);
                load_panel_buf<G, BK>(vb, pb, n0, k1, wave);
            } else {
                load_panel<G, AK, false>(va, D.A, D.lda, m0, D.M, k1, kn, tid);
                if (pro) load_panel<G, AK, false>(vz, D.pro.z, D.lda, m0, D.M, k1, kn, tid);
                load_panel<G, BK, false>(vb, D.B, D.ldb, n0, D.N, k1, kn, tid);
            }
        }
        __syncthreads();
        if (k0 == k_lo) GB_TL(1);
        const int steps = kc >> 4;                                    // macro-steps of 16 k, dealt in contiguous runs
        const int kbeg = (steps * kh / KSPLIT) << 4, kend = (steps * (kh + 1) / KSPLIT) << 4;
#pragma unroll 4
        for (int kk = kbeg; kk < kend; kk += 16) {
            const float4 a = read_frag<G, AK>(sA, wm * 16 + r, g, kk);
            const float4 b = read_frag<G, BK>(sB, wn * 16 + r, g, kk);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc1, 0, 0, 0);
        }
    }
    f32x4 acc = acc0 + acc1;
    GB_TL(2);
    if (KSPLIT > 1) {
        if (kh) *(f32x4*)(sC + (((kh - 1) * 4 + tile) * 64 + lane) * 4) = acc;
        __syncthreads();
    }
    float sq = 0.f;
    if (!kh) {
#pragma unroll
        for (int h = 1; h < KSPLIT; ++h) acc = acc + *(const f32x4*)(sC + (((h - 1) * 4 + tile) * 64 + lane) * 4);
        if (D.C) {
            // rows past M end the resource, a column past N gets an offset past everything: dropped by the hardware. Those
            // elements are sums over zero panels, so they add nothing to sq either.
            const unsigned ldc4 = (unsigned)D.ldc * 4u;
            const __amdgpu_buffer_rsrc_t cr = naf_buf(D.C + (int64_t)ks * D.c_split_stride, (unsigned)D.M * ldc4);
            const int cn = n0 + wn * 16 + r;
            const unsigned voff = cn < D.N ? (unsigned)(4 * g) * ldc4 + (unsigned)cn * 4u : 0x7f000000u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = acc[e];
                naf_buf_st_f1(cr, voff, (unsigned)(m0 + wm * 16 + e) * ldc4, v, (D.M > D.K ? D.M : D.K) >= NAF_WT_MIN_B);
                sq += v * v;
            }
        }
    }
    GB_TL(3);
    if (D.epi.x) gemm_l1bwd_epilogue<G>(D, bm, bn, acc, epi_xh, !kh, wm, wn, r, g, sA, sB, tid, epi_regs);
    GB_TL(4);
    if (D.sumsq) {   // gradient-norm partial of this block (fixed order: shuffles, then the 4 tiles)
        float* sQ = sA;                                       // (the panels are dead: behind the barrier every wave is past them)
        sq = naf_sum64(sq);
        __syncthreads();
        if (!kh && lane == 0) sQ[tile] = sq;
        __syncthreads();
        if (tid == 0) D.sumsq[bm * D.tiles_n + bn] = sQ[0] + sQ[1] + sQ[2] + sQ[3];
    }
    GB_TL(5);
    GB_TL_WG(1);
}

#ifndef GB_WAVES_PER_EU
#define GB_WAVES_PER_EU 4
#endif
// NFOLD (0 | 256 / GB_FOLD_COLS | 512 / GB_FOLD_COLS): the launch has a BatchNorm-backward prologue and its first NFOLD
// workgroups fold the block sums — one per 32 columns of the layer (H = 256: 8; H = 512, round 6: 16). A template
// parameter, not a kernel argument: everything that waits in this launch waits for those workgroups, and as arguments their way to
// the first load of the partials led through three dependent scalar round trips (n_fold -> fold_desc -> the fields of
// d[fold_desc].pro, ~0.25 us each in front of a cold scalar cache) plus one more in front of the record store; now it is one batch.
template <int T, int KC, int NFOLD, bool TAIL = false>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(GB_WAVES_PER_EU, GB_WAVES_PER_EU))) void gemm_bundle_kernel(const GemmBundle bundle) {
    using G = GB<T, KC, TAIL, (NFOLD > 256 / GB_FOLD_COLS)>;
    constexpr bool FOLD = NFOLD > 0;
    constexpr int GB_FOLD_WGS = NFOLD;
    __shared__ __attribute__((aligned(16))) float sA[G::PANEL];
    __shared__ __attribute__((aligned(16))) float sB[G::PANEL];
    __shared__ __attribute__((aligned(16))) float sC[4 * 64 * 4];   // the column constants of the prologue, then the K halves' hand-over
    int t = blockIdx.x;                                   // one 32 x 32 block per workgroup
    if (FOLD) {
        if (__builtin_expect(t < GB_FOLD_WGS, 0)) {
            const naf_gemm_bn2bwd_t& P = bundle.fold_pro;
            asm volatile("" ::"s"(P.partials), "s"(P.gamma), "s"(P.save_invstd), "s"(P.epoch), "s"(P.npb), "s"(P.B), "s"(P.H), "s"(P.cst),
                         "s"(P.d_gamma), "s"(P.d_beta));
            gemm_bn2bwd_fold_block<T>(P, t, threadIdx.x, sA);
            return;
        }
        t -= GB_FOLD_WGS;
    }
    const int n_fold = FOLD ? GB_FOLD_WGS : 0;
    int gi = 0;
#pragma unroll
    for (int i = 1; i < NAF_GEMM_BUNDLE_MAX; ++i)
        if (i < bundle.n && t >= bundle.d[i].tile0) gi = i;
    const GemmDesc& D = bundle.d[gi];
    // (what the block decode and the dispatch below read, in one batch of scalar loads: see gemm_block)
    asm volatile("" ::"s"(D.tile0), "s"(D.tiles_mn), "s"(D.tiles_n), "s"(D.a_kmajor), "s"(D.b_kmajor), "s"(D.k_split), "s"(D.M), "s"(D.K));
    int ks = (t - D.tile0) / D.tiles_mn;
    const int lt = t - D.tile0 - ks * D.tiles_mn;
    int bm = lt / D.tiles_n, bn = lt - bm * D.tiles_n;
    if (D.tiles_n == 8 && bundle.rowmap) {
        // 8 block columns; workgroup t runs on XCD t % 8 (round-robin dispatch). Placement is speed only, the result does not
        // depend on it. Every XCD has its own L2, so what matters is how much of the operands each of the eight pulls:
        //   k-major A (dW2 = dZ2^T A1): block ROW bm on XCD bm — an eighth of dZ2 (and Z2), all of A1;
        //   k-contiguous A (dA1 = dZ2 W2, with the BatchNorm prologue and the layer-1 epilogue): the block rows of the x-th EIGHTH
        //   of the batch on XCD x, with all 8 block columns of a row — an eighth of dZ2, Z2, A1 and the minibatch rows plus the whole
        //   of W2, instead of all of those and an eighth of W2. (Until the epilogue was fused the dA1 blocks sat by block COLUMN, next
        //   to the kernel that read those dA1 columns; nothing reads them any more, and by rows: 31.7k -> 33.1k updates/s at
        //   B = 256, 28.6k -> 29.4k at 512, 24.0k -> 25.6k at 1024, 17.5k -> 17.9k at 2048, A/B on the same boxes. Contiguous
        //   eighths instead of rows x, x + 8, ...: the rows an XCD pulls for its dA1 blocks are then inside the K range its dW2
        //   blocks walk — +1 % at B = 1024, +0.4 % at 2048, A/B/A/B on one box; bb_layer2_head writes them there.)
        const int xcd = lt & 7, slot = lt >> 3;
        if (D.a_kmajor) {
            const int S = D.k_split;
            if (D.M == 256 && bundle.rowmap >= 2 && (S == 2 || S == 4 || S == 8)) {
                // split K (the weight gradient at large batches): a K RANGE per group of 8 / S XCDs, S block rows each — an L2
                // then pulls its range's rows of dZ2, Z2 (S of 8 column slices) and A1 instead of every range's
                const int idx = t - D.tile0, x = idx & 7, sl = idx >> 3, per = 8 / S;
                ks = x / per;
                bm = S * (x % per) + sl % S;
                bn = sl / S;
            } else if (D.M == 256) { bm = xcd; bn = slot; }
        } else if ((D.tiles_mn & 63) == 0) {              // (whole groups of 8 block rows)
            bm = xcd * (D.tiles_mn >> 6) + (slot >> 3);
            bn = slot & 7;
        }
    }
    if (D.a_kmajor) {
        if (D.b_kmajor) gemm_block<G, true, true>(D, bm, bn, ks, sA, sB, sC, n_fold);
        else gemm_block<G, true, false>(D, bm, bn, ks, sA, sB, sC, n_fold);
    } else {
        if (D.b_kmajor) gemm_block<G, false, true>(D, bm, bn, ks, sA, sB, sC, n_fold);
        else gemm_block<G, false, false>(D, bm, bn, ks, sA, sB, sC, n_fold);
    }
}

// (round 4's second form of this launch — 64 x 32 / 64 x 64 tiles fed by an LDS-DMA ring, measured slower at every size — lives in
//  benchmarks/experimental/gemm_ring.h with its tests and A/B scripts; it is not compiled into the library)
extern "C" int naf_gemm_bundle(const naf_gemm_desc_t* descs, int n, void* stream) {
    if (!descs || n <= 0 || n > NAF_GEMM_BUNDLE_MAX) return NAF_ERR_ARG;
    GemmBundle b;
    b.n = n;
    b.n_fold = b.fold_desc = 0;
    b.rowmap = 2;                 // dA1 blocks by block row on the XCDs, dW2's K ranges by XCD group (the kernel's comment)
    int tiles = 0;
    bool tail = false;           // some epilogue has rows past the batch in its last 32-row block: the TAIL form of the kernel
    for (int i = 0; i < n; ++i) {
        const naf_gemm_desc_t& s = descs[i];
        if (!s.A || !s.B || (!s.C && !s.epi) || s.M <= 0 || s.N <= 0 || s.K <= 0) return NAF_ERR_ARG;
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
        memset(&d.pro, 0, sizeof(d.pro));
        if (s.pro) {
            const naf_gemm_bn2bwd_t& q = *s.pro;
            if (!q.z || !q.partials || !q.gamma || !q.save_mean || !q.save_invstd || !q.d_gamma || !q.d_beta || q.npb < 1 ||
                q.npb > 128 || !q.cst || !q.epoch || ((uintptr_t)q.cst & 15) || q.B <= 0 || (q.H != 256 && q.H != 512) || (s.M & 15) || (s.N & 31) ||
                (s.a_kmajor ? s.M != q.H : s.K != q.H) || ((uintptr_t)q.z & 15) || ((uintptr_t)q.partials & 7))
                return NAF_ERR_ARG;      // (the A operand's columns are the H features: its M when k-major, its K otherwise)
            d.pro = q;
            if (!b.n_fold) {
                b.n_fold = q.H / GB_FOLD_COLS;
                b.fold_desc = i;
            }
        }
        memset(&d.epi, 0, sizeof(d.epi));
        if (s.epi) {
            const naf_gemm_l1bwd_t& e = *s.epi;
            if (!e.x || !e.W || !e.bias || !e.a1 || !e.save_mean || !e.save_invstd || !e.partials || !e.p_slabs || ksn != 1 ||
                (s.M & 15) || (s.N & 31) || e.K <= 0 || (e.kp != 24 && e.kp != 32) || e.K > e.kp || e.ldx < e.kp || (e.ldx & 3) ||
                e.lda1 < s.N || ((uintptr_t)e.x & 15) || ((uintptr_t)e.partials & 7) || (e.xhat && (!e.gamma || !e.beta)) ||
                e.rows < 0 || e.rows > s.M)
                return NAF_ERR_ARG;
            d.epi = L1bwdDev{e.x, e.W, e.bias, e.a1, e.save_mean, e.save_invstd, e.partials, e.p_slabs, e.ldx, e.K, e.kp, e.lda1,
                             e.xhat, e.gamma, e.beta};
            d.c_split_stride = e.rows ? e.rows : s.M;
            if (d.c_split_stride & 31) tail = true;
        }
        tiles += d.tiles_mn * ksn;
    }
    for (int i = n; i < NAF_GEMM_BUNDLE_MAX; ++i) b.d[i] = b.d[0];
    b.total_tiles = tiles;
    // more blocks than the 8-wave form has room for at once (two per CU): the 4-wave form, four workgroups per CU
    // (updates/s, A/B/A/B on one box: B = 1536 20.6k -> 21.0k, B = 2048 20.05k -> 20.35k; B = 1024, 428 blocks: 26.1k -> 25.2k)
    const bool big = tiles + b.n_fold > 512;
    hipStream_t st = (hipStream_t)stream;
    if (b.n_fold == 16) {                                 // H = 512 (round 6)
        b.fold_pro = b.d[b.fold_desc].pro;
        if (tail) {
            if (big) gemm_bundle_kernel<256, 128, 16, true><<<tiles + b.n_fold, 256, 0, st>>>(b);
            else gemm_bundle_kernel<512, 256, 16, true><<<tiles + b.n_fold, 512, 0, st>>>(b);
        } else if (big) gemm_bundle_kernel<256, 128, 16><<<tiles + b.n_fold, 256, 0, st>>>(b);
        else gemm_bundle_kernel<512, 256, 16><<<tiles + b.n_fold, 512, 0, st>>>(b);
    } else if (b.n_fold) {
        b.fold_pro = b.d[b.fold_desc].pro;
        if (tail) {
            if (big) gemm_bundle_kernel<256, 128, 8, true><<<tiles + b.n_fold, 256, 0, st>>>(b);
            else gemm_bundle_kernel<512, 256, 8, true><<<tiles + b.n_fold, 512, 0, st>>>(b);
        } else if (big) gemm_bundle_kernel<256, 128, 8><<<tiles + b.n_fold, 256, 0, st>>>(b);
        else gemm_bundle_kernel<512, 256, 8><<<tiles + b.n_fold, 512, 0, st>>>(b);
    } else {
        memset(&b.fold_pro, 0, sizeof(b.fold_pro));
        if (tail) {
            if (big) gemm_bundle_kernel<256, 128, 0, true><<<tiles, 256, 0, st>>>(b);
            else gemm_bundle_kernel<512, 256, 0, true><<<tiles, 512, 0, st>>>(b);
        } else if (big) gemm_bundle_kernel<256, 128, 0><<<tiles, 256, 0, st>>>(b);
        else gemm_bundle_kernel<512, 256, 0><<<tiles, 512, 0, st>>>(b);
    }
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}
