// The backward GEMMs of a large-batch learn() update (dA1 = dZ2 W2, dW2 = dZ2^T A1, dWh = dH^T A2: autograd of
// naf_neural_network.py:76-87) as ONE launch of 64 x 64 output tiles whose operand panels reach LDS by LDS-DMA
// (`buffer_load_dwordx4 ... lds`: global -> LDS with no VGPR in between) into a RING of K chunks — two chunks in flight under the
// MFMAs of the current one. The second form of naf_gemm_bundle (csrc/gemm_bundle.hip is the first: 32 x 32 blocks, panels staged
// through VGPRs, one chunk in flight), chosen by the host for B >= 1024.
//
// Why (profiles/r03_pmc_sq_b2048.csv, r03_timeline_b2048.txt): the 32 x 32 form stages 87 MB of panels per launch at B = 2048 for
// 0.59 GFLOP (7 FLOP per staged byte), every block's whole K range is ONE chunk (load -> store -> barrier -> 64 MFMAs per wave), so
// nothing inside a block overlaps and co-resident blocks march in phase: 44 % of the wave cycles parked on memory / barriers, 37 %
// queued behind a neighbour's MFMAs, the MFMA pipes busy 30 % of the launch. Here a tile is 64 x 64 (half the staged bytes per
// FLOP), a K range is 8+ chunks of 32 k, and a wave issues the DMAs of chunk c + 2 before it multiplies chunk c.
//
// Workgroup = 512 threads = 4 consumer waves + 4 loader waves (ring_block); consumer w multiplies the WHOLE 64 x 64 tile over k rows
// 8 w .. 8 w + 7 of every 32-k chunk (16 accumulator tiles of 16 x 16 per wave, two k-steps of v_mfma_f32_16x16x4_f32 per tile and
// chunk), and the four partial products meet through LDS at the end, in wave order (fixed order: bitwise reproducible). Register
// blocking is what makes the LDS reads cheap: one 16-byte fragment read per operand feeds four accumulator tiles.
//
// LDS image of a chunk ("slot", 24 KB): A panel | Z panel (the BatchNorm-backward prologue's second operand) | B panel.
//   k-major operand  [K][cols] (every B; A of the weight gradients): 32 k-rows x 64 columns, 256-B rows in memory order. One
//     ds_read_b128 at (k-row, column quad r) gives lane (r, g) its operand for the FOUR tiles that interleave their columns
//     (column = 4 r + t): conflict-free (16 lanes of a read group cover 16 distinct 16-byte slots of a 256-B row).
//   k-contiguous A   [rows][K] (dA1's dY2 / Z2): 64 rows x 32 k, 128-B rows, the eight 16-byte quads of row rho XOR-ed with
//     (rho >> 1) & 7 — on the SOURCE side of the DMA (the destination of a wave-instruction is lane-linear: MI355X guide, LDS-DMA
//     caveat) — so that the ds_read_b64 of 32 lanes (16 rows x 2 k-pairs) touch 32 distinct 8-byte slots.
// C/D map of the MFMA: col = l & 15, row = 4 (l >> 4) + reg; A operand: lane (r, g) holds A[row r][k g].
// (included by gemm_bundle.hip, behind its own kernel: one translation unit, one set of timeline arrays)
#pragma once

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define GR_T 512
#define GR_KC 32
#define GR_PANEL 2048                     // floats of one 64 x 32 panel (8 KB)
#define GR_SLOT (3 * GR_PANEL)            // A | Z | B
#ifndef GR_DEEP
#define GR_DEEP 6                         // slots of the one-workgroup-per-CU form
#endif
#define GR_CST 1024                       // floats: [4][256] per-column constants of the prologue

struct RingDesc {
    const float* A;
    const float* B;
    float* C;
    int M, N, K, lda, ldb, ldc, a_kmajor;
    int tile0, tiles_n, tiles_mn, k_split;
    int nt;                       // column tiles of 16 per block: 4 (64 x 64 blocks) or 2 (64 x 32)
    int64_t c_split_stride;
    naf_gemm_l1bwd_t epi;
    naf_gemm_bn2bwd_t pro;
};
struct RingBundle {
    RingDesc d[NAF_GEMM_BUNDLE_MAX];
    int n, total_tiles, n_fold, wt;
    naf_gemm_bn2bwd_t fold_pro;
};

// one LDS-DMA wave-instruction: 64 lanes x 16 bytes from (resource, lane offset, wave offset) to LDS bytes [lds, lds + 1024)
__device__ __forceinline__ static void gr_dma16(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, unsigned lds_byte) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 3\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rs), "s"(soff), "s"(lds_byte)
                 : "memory", "m0");
}
template <int N>
__device__ __forceinline__ static void gr_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}
// all but the `chunks` youngest chunks of this wave's DMAs have landed (LPC DMAs per chunk and loader wave)
template <int LPC>
__device__ __forceinline__ static void gr_wait_chunks(int chunks) {
    switch (chunks) {
        case 0: gr_wait_vm<0>(); break;
        case 1: gr_wait_vm<LPC>(); break;
        case 2: gr_wait_vm<2 * LPC>(); break;
        case 3: gr_wait_vm<3 * LPC>(); break;
        case 4: gr_wait_vm<4 * LPC>(); break;
        default: gr_wait_vm<5 * LPC>(); break;
    }
}

struct RingSrc {
    __amdgpu_buffer_rsrc_t ra, rz, rb;
    unsigned va, vb, lda4, ldb4;
};

// this loader's share of a chunk: pieces cw and cw + 4 of the A (and Z) panel, and of the B panel pieces cw, cw + 4 (64 columns:
// 4 k-rows x 256 B per piece) or piece cw alone (32 columns: 8 k-rows x 128 B per piece)
template <bool AK, int NT>
__device__ __forceinline__ static void gr_issue_chunk(const RingSrc& S, int m0, int k0, int slot, int cw, bool pro, unsigned lds0) {
    const unsigned base = lds0 + (unsigned)slot * (GR_SLOT * 4);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int p = cw + 4 * u;                              // piece: 4 k-rows (k-major) or 8 rows (k-contiguous)
        const unsigned sa = AK ? (unsigned)(k0 + 4 * p) * S.lda4 : (unsigned)(m0 + 8 * p) * S.lda4 + (unsigned)k0 * 4u;
        gr_dma16(S.ra, S.va, sa, base + (unsigned)p * 1024u);
        if (pro) gr_dma16(S.rz, S.va, sa, base + GR_PANEL * 4 + (unsigned)p * 1024u);
        if (NT == 4) gr_dma16(S.rb, S.vb, (unsigned)(k0 + 4 * p) * S.ldb4, base + 2 * GR_PANEL * 4 + (unsigned)p * 1024u);
    }
    if (NT == 2) gr_dma16(S.rb, S.vb, (unsigned)(k0 + 8 * cw) * S.ldb4, base + 2 * GR_PANEL * 4 + (unsigned)cw * 1024u);
}

// the waiting side of the prologue's hand-off (bn2bwd_fold.h): constants of the block's columns -> cst (LDS, [4][256])
template <bool AK>
__device__ static inline void gr_wait_constants(const naf_gemm_bn2bwd_t& P, int m0, int tid, float* cst) {
    const int ncol = AK ? 64 : 256, col0 = AK ? m0 : 0;
    if (tid < ncol) {
        const int col = col0 + tid;
        const int epoch = *P.epoch;
        const float mean = P.save_mean[col], k1 = P.gamma[col] * P.save_invstd[col];
        f32x4 c;
        if (!gemm_bn2bwd_poll_record(naf_buf(P.cst), col, epoch, &c)) {
            c = gemm_bn2bwd_fold_column<GR_T>(P, col);
            if (P.errors) __hip_atomic_fetch_add((unsigned long long*)P.errors, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        cst[tid] = mean;
        cst[256 + tid] = k1;
        cst[512 + tid] = c[0];
        cst[768 + tid] = c[1];
    }
}

// (timeline marks share the 32 x 32 form's arrays: this header is part of gemm_bundle.hip's translation unit)
#define GR_TL(slot) GB_TL(slot)
#define GR_TL_WG(which) GB_TL_WG(which)

// what the layer-1 epilogue of a dA1 tile reads from memory, requested before the K loop by the wave that will own the final
// elements (csrc/gemm_bundle.hip describes the epilogue; here xhat is always the forward pass' own, naf_gemm_l1bwd_t.xhat)
template <int NB>
struct RingEpiRegs {
    float xh[2][4][NB];  // xhat at the wave's final elements: [row tile a][reg e][column tile b]
    float x[2][4][2];    // minibatch rows as the B operand of P = dY^T X: [a][e][k' tile]
    float gamma[NB], beta[NB];
};

// ---- consumer waves 0 .. 3: the MFMAs ---------------------------------------------------------------------------------------------
template <bool AK, bool PRO, int NS, int NT>
__device__ static inline void ring_consumer(const RingDesc& D, int m0, int k_lo, int nc, float* lds, int cw, int lane, int tl_first) {
    constexpr bool pro = PRO;                  // (a template parameter: no branch on it inside the K loop)
    const int tid = threadIdx.x;
    const int r = lane & 15, g = lane >> 4;
    float* ring = lds;
    float* cst = lds + NS * GR_SLOT;
    f32x4 cm = {0.f, 0.f, 0.f, 0.f}, ck1 = {1.f, 1.f, 1.f, 1.f}, ckc = {0.f, 0.f, 0.f, 0.f}, cq = {0.f, 0.f, 0.f, 0.f};
    if (pro) {
        gr_wait_constants<AK>(D.pro, m0, tid, cst);
        __syncthreads();       // (barrier 0, with the loaders)
        if (AK) {              // the lane's columns m0 + 4 r + t never change: constants in registers
            cm = *(const f32x4*)(cst + 4 * r);
            ck1 = *(const f32x4*)(cst + 256 + 4 * r);
            ckc = *(const f32x4*)(cst + 512 + 4 * r);
            cq = *(const f32x4*)(cst + 768 + 4 * r);
        }
    }
    GR_TL(2);
    f32x4 acc[4][NT];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) acc[mt][tj] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // B fragment of a k-row: the NT interleaved column tiles (column = NT j + t) in one read: 16 bytes (NT = 4) or 8 (NT = 2)
    auto read_b = [&](const float* sB, int krow, float (&b)[NT]) {
        if (NT == 4) {
            const f32x4 v = *(const f32x4*)(sB + krow * 64 + 4 * r);
            b[0] = v[0]; b[1] = v[1]; b[2 % NT] = v[2]; b[3 % NT] = v[3];
        } else {
            const f32x2 v = *(const f32x2*)(sB + krow * 32 + 2 * r);
            b[0] = v[0]; b[1] = v[1];
        }
    };
    // One chunk's operand fragments in registers. Two sets alternate: the LDS reads of chunk c + 1 are issued BEFORE the MFMAs of
    // chunk c and land under them (as a read -> wait -> multiply loop every chunk exposed its LDS latency and its transform: 46
    // cycles per MFMA where the pipe needs 32, in-kernel stamps with the DMAs stubbed out).
    struct Frag {
        f32x4 a4[2], z4[2];        // k-major A: k-step j, the four interleaved row tiles
        f32x2 a2[4], z2[4];        // k-contiguous A: row tile mt, k-steps 0 / 1
        float b[2][NT];
        f32x2 m2, k2, c2, q2;      // k-contiguous A: the prologue's constants of the lane's two k
    };
    auto load = [&](Frag& F, int c) {
        const float* sA = ring + (c % NS) * GR_SLOT;
        const float* sZ = sA + GR_PANEL;
        const float* sB = sA + 2 * GR_PANEL;
        if (AK) {
            // k-step j: k-row 8 cw + 4 j + g; one b128 per operand feeds the four interleaved tiles
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int krow = 8 * cw + 4 * j + g, off = krow * 64 + 4 * r;
                F.a4[j] = *(const f32x4*)(sA + off);
                if (pro) F.z4[j] = *(const f32x4*)(sZ + off);
                read_b(sB, krow, F.b[j]);
            }
        } else {
            // lane (r, g) supplies k = 8 cw + 2 g + j (j = k-step): an 8-byte read per row tile, a 16- (8-) byte read of B's k-row
            const int kl = 8 * cw + 2 * g;                   // within the chunk
            const int quad = kl >> 2, half = kl & 2;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int rho = mt * 16 + r;
                const int off = rho * 32 + ((quad ^ ((rho >> 1) & 7)) << 2) + half;
                F.a2[mt] = *(const f32x2*)(sA + off);
                if (pro) F.z2[mt] = *(const f32x2*)(sZ + off);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) read_b(sB, kl + j, F.b[j]);
            if (pro) {
                const int kg = k_lo + GR_KC * c + kl;          // the feature (column of dY2 / Z2) of k-step 0
                F.m2 = *(const f32x2*)(cst + kg);
                F.k2 = *(const f32x2*)(cst + 256 + kg);
                F.c2 = *(const f32x2*)(cst + 512 + kg);
                F.q2 = *(const f32x2*)(cst + 768 + kg);
            }
        }
    };
    auto multiply = [&](Frag& F) {
#if defined(GR_STUB) && GR_STUB == 2            /* GR_STUB = 2: the consumers multiply nothing (what the DMAs take alone; wrong results) */
        return;
#endif
        if (AK) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (pro) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) F.a4[j][t] = __builtin_fmaf(ck1[t], F.a4[j][t], -ckc[t]) - (F.z4[j][t] - cm[t]) * cq[t];
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj)
                        acc[mt][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(F.a4[j][mt], F.b[j][tj], acc[mt][tj], 0, 0, 0);
            }
        } else {
            if (pro) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        F.a2[mt][j] = __builtin_fmaf(F.k2[j], F.a2[mt][j], -F.c2[j]) - (F.z2[mt][j] - F.m2[j]) * F.q2[j];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj)
                        acc[mt][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(F.a2[mt][j], F.b[j][tj], acc[mt][tj], 0, 0, 0);
        }
    };
    // Barrier j (j = 0 .. nc): chunk j has landed (the loaders waited for it) and every consumer has READ chunk j - 1 — whose slot the
    // loaders refill behind it. The reads of a chunk are waited for before the next barrier (they were issued a whole chunk of MFMAs
    // earlier), so "read" is literal.
    Frag F0, F1;
    __builtin_amdgcn_s_barrier();
    GR_TL(5);
    load(F0, 0);
    for (int c = 0; c < nc; c += 2) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (c + 1 < nc) load(F1, c + 1);
        multiply(F0);
        if (c + 1 < nc) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (c + 2 < nc) load(F0, c + 2);
            multiply(F1);
        }
    }
    GR_TL(3);
    // (the loop's last barrier is the one behind which the ring is free: every consumer has read the last chunk)
    // the K quarters meet: every consumer leaves its 16 partial tiles in LDS ([wave][tile][lane]); the loaders add them up
    f32x4* sM = (f32x4*)ring;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) sM[(cw * 4 * NT + mt * NT + tj) * 64 + lane] = acc[mt][tj];
    __syncthreads();                           // (barrier B)
    GR_TL(4);
}

// ---- loader waves 4 .. 7: the LDS-DMAs, then the block's tail (their registers are free; a consumer's hold 16 accumulator tiles) -----
template <bool AK, bool PRO, int NS, int NT>
__device__ static inline void ring_loader(const RingDesc& D, int bm, int bn, int ks, int k_lo, int nc, float* lds, int cw, int lane,
                                          bool wt) {
    constexpr bool pro = PRO;
    constexpr int AHEAD = NS - 1;
    constexpr int LPC = (PRO ? 4 : 2) + (NT == 4 ? 2 : 1);      // DMAs per chunk and loader wave
    constexpr int NB = NT / 2;                                 // column tiles per owner of dA1's final elements
    constexpr int TW = 4 * NT;                                 // tiles per consumer wave
    const int r = lane & 15, g = lane >> 4;
    const int m0 = bm * 64, n0 = bn * 16 * NT;
    float* ring = lds;
    const unsigned lds0 = (unsigned)(uintptr_t)lds;
    // the epilogue's operands first: plain loads, older than every DMA (so a counted vmcnt never waits for them). Loader w owns the
    // final elements of row tiles 2 h, 2 h + 1 and column tiles 2 v, 2 v + 1 (h = w >> 1, v = w & 1): 32 rows x 32 interleaved columns
    RingEpiRegs<NB> R;
    const int eh = cw >> 1, ev = cw & 1;
    if (!AK && D.epi.x) {
        const naf_gemm_l1bwd_t& E = D.epi;
        const int col = n0 + NT * r + NB * ev;                 // (column = n0 + NT j + tile: the owner's tiles NB v .. NB v + NB - 1)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = m0 + 32 * eh + 16 * a + 4 * g + e;
#pragma unroll
                for (int b = 0; b < NB; ++b) R.xh[a][e][b] = E.xhat[(int64_t)row * E.lda1 + col + b];
                R.x[a][e][0] = E.x[(int64_t)row * E.ldx + r];
                R.x[a][e][1] = E.x[(int64_t)row * E.ldx + 16 + r];
            }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            R.gamma[b] = E.gamma[col + b];
            R.beta[b] = E.beta[col + b];
        }
    }
    // DMA sources: wave-uniform resources and chunk offsets, one lane offset per operand
    RingSrc S;
    S.lda4 = (unsigned)D.lda * 4u;
    S.ldb4 = (unsigned)D.ldb * 4u;
    const unsigned a_bytes = (unsigned)(AK ? D.K : D.M) * S.lda4;
    S.ra = naf_buf(D.A, a_bytes);
    S.rz = naf_buf(pro ? D.pro.z : D.A, a_bytes);
    S.rb = naf_buf(D.B, (unsigned)D.K * S.ldb4);
    if (AK) {
        S.va = (unsigned)(lane >> 4) * S.lda4 + (unsigned)(m0 + 4 * (lane & 15)) * 4u;
    } else {
        // piece p = cw + 4 u holds rows 8 p .. 8 p + 7; lane (i = lane >> 3, s = lane & 7) fills slot s of row rho = 8 p + i with
        // the row's quad s ^ ((rho >> 1) & 7); (rho >> 1) & 7 = (4 p + (i >> 1)) & 7 depends on p only through its parity = cw & 1
        const int i = lane >> 3, s = lane & 7;
        const int q = s ^ ((4 * (cw & 1) + (i >> 1)) & 7);
        S.va = (unsigned)i * S.lda4 + (unsigned)q * 16u;
    }
    S.vb = NT == 4 ? (unsigned)(lane >> 4) * S.ldb4 + (unsigned)(n0 + 4 * (lane & 15)) * 4u
                   : (unsigned)(lane >> 3) * S.ldb4 + (unsigned)(n0 + 4 * (lane & 7)) * 4u;
#pragma unroll
    for (int c = 0; c < AHEAD; ++c)
        if (c < nc) gr_issue_chunk<AK, NT>(S, m0, k_lo + GR_KC * c, c, cw, pro, lds0);
    if (pro) __builtin_amdgcn_s_barrier();     // (barrier 0: the consumers' constants; nothing here waits for it or for the DMAs)
    for (int c = 0; c < nc; ++c) {
        const int ahead = nc - 1 - c < AHEAD - 1 ? nc - 1 - c : AHEAD - 1;     // chunks issued behind chunk c
        gr_wait_chunks<LPC>(ahead);
        __builtin_amdgcn_s_barrier();
#if !defined(GR_STUB) || GR_STUB != 1       /* GR_STUB = 1: no DMAs inside the loop (what the consumers take alone; wrong results) */
        if (c + AHEAD < nc) gr_issue_chunk<AK, NT>(S, m0, k_lo + GR_KC * (c + AHEAD), (c + AHEAD) % NS, cw, pro, lds0);
#endif
    }
    __builtin_amdgcn_s_barrier();              // (barrier A)
    __builtin_amdgcn_s_barrier();              // (barrier B: the consumers' partial tiles are in LDS — their stores were waited for)
    asm volatile("" ::: "memory");
    const f32x4* sM = (const f32x4*)ring;
    if (AK) {
        // loader w takes A tile index w (rows m0 + 4 i + w) with all NT column tiles: whole 16- (8-) byte stores
        if (D.C) {
            f32x4 fin[NT];
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) {
                f32x4 sum = sM[(0 * TW + cw * NT + tj) * 64 + lane];
#pragma unroll
                for (int ww = 1; ww < 4; ++ww) sum = sum + sM[(ww * TW + cw * NT + tj) * 64 + lane];
                fin[tj] = sum;
            }
            float* C = D.C + (int64_t)ks * D.c_split_stride;
            const int n = n0 + NT * r;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = m0 + 4 * (4 * g + e) + cw;
                if (row < D.M && n < D.N) {
                    if (NT == 4) {
                        const naf_f32x4 v = {fin[0][e], fin[1][e], fin[2 % NT][e], fin[3 % NT][e]};
                        naf_buf_st_f4(naf_buf(C), (unsigned)(row * D.ldc + n) * 4u, 0, v, wt);
                    } else {
                        *(f32x2*)(C + (int64_t)row * D.ldc + n) = (f32x2){fin[0][e], fin[1][e]};
                    }
                }
            }
        }
        return;
    }
    f32x4 fin[2][NB];                                          // [row tile a][column tile b]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int t = (2 * eh + a) * NT + NB * ev + b;
            f32x4 sum = sM[(0 * TW + t) * 64 + lane];
#pragma unroll
            for (int ww = 1; ww < 4; ++ww) sum = sum + sM[(ww * TW + t) * 64 + lane];
            fin[a][b] = sum;
        }
    if (D.C) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = m0 + (2 * eh + a) * 16 + 4 * g + e, n = n0 + NT * r + NB * ev;
                if (row < D.M && n < D.N) {
#pragma unroll
                    for (int b = 0; b < NB; ++b) D.C[(int64_t)row * D.ldc + n + b] = fin[a][b][e];
                }
            }
    }
    if (!D.epi.x) return;
    // layer 1's backward batch pass on the final elements: dy = ReLU'(.) dA1, block sums, P share
    const naf_gemm_l1bwd_t& E = D.epi;
    const int bm32 = 2 * bm + eh;
    float sdy[NB], sdx[NB];
    f32x4 dy[2][NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) sdy[b] = sdx[b] = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = R.xh[a][e][b];
                const bool on = __builtin_fmaf(xh, R.gamma[b], R.beta[b]) > 0.f;     // the forward's own ReLU decision
                const float v = on ? fin[a][b][e] : 0.f;
                dy[a][b][e] = v;
                sdy[b] += v;
                sdx[b] += v * xh;
            }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        sdy[b] = naf_xor32_add(naf_xor16_add(sdy[b]));
        sdx[b] = naf_xor32_add(naf_xor16_add(sdx[b]));
    }
    if (g == 0) {
        float* pp = E.partials + ((int64_t)bm32 * D.N + n0 + NT * r + NB * ev) * 2;
#pragma unroll
        for (int b = 0; b < NB; ++b) *(f32x2*)(pp + 2 * b) = (f32x2){sdy[b], sdx[b]};
    }
    // P = dY^T X of the 32 rows: the accumulator layout of dy IS the A operand layout with k = (g, e) <-> row 4 g + e
    const int KP = E.kp;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 p = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int e = 0; e < 4; ++e) p = __builtin_amdgcn_mfma_f32_16x16x4f32(dy[a][b][e], R.x[a][e][t], p, 0, 0, 0);
            const int kq = 16 * t + r;
            if (kq < KP) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = n0 + NT * (4 * g + e) + NB * ev + b;
                    E.p_slabs[((int64_t)bm32 * D.N + n) * KP + kq] = p[e];
                }
            }
        }
}

// Workgroup = 512 threads: waves 0 .. 3 multiply (consumers), waves 4 .. 7 issue the LDS-DMAs (loaders). An LDS-DMA costs the
// wave that issues it 60 - 180 cycles (MI355X guide), and a wave issues in order: dealt between a consumer's MFMAs, six DMAs per
// chunk left its MFMA pipe idle for a third of every chunk (in-kernel stamps: 0.6 - 0.8 us per chunk against 0.43 of MFMAs). A
// loader wave shares its SIMD with a consumer and costs it nothing. One s_barrier per chunk joins the two sides: behind barrier c
// chunk c has landed (the loaders waited for it: counted vmcnt) and every consumer is done with chunk c - 1, whose slot the loaders
// refill with chunk c + NS - 1. Two code paths with the same number of barriers; neither keeps the other's registers alive.
template <bool AK, bool PRO, int NS, int NT>
__device__ static inline void ring_block(const RingDesc& D, int bm, int bn, int ks, float* lds, bool wt, int tl_first) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    asm volatile("" ::"s"(D.A), "s"(D.B), "s"(D.C), "s"(D.M), "s"(D.N), "s"(D.K), "s"(D.lda), "s"(D.ldb), "s"(D.k_split), "s"(D.epi.x),
                 "s"(D.epi.xhat), "s"(D.epi.gamma), "s"(D.epi.beta), "s"(D.epi.ldx), "s"(D.epi.lda1), "s"(D.epi.kp), "s"(D.pro.z), "s"(D.pro.gamma),
                 "s"(D.pro.save_mean), "s"(D.pro.save_invstd), "s"(D.pro.cst), "s"(D.pro.epoch), "s"(D.pro.errors));
    GR_TL(0);
    GR_TL_WG(0);
    const int kper = D.K / D.k_split, k_lo = ks * kper, nc = kper / GR_KC;
    if (wave >= 4) ring_loader<AK, PRO, NS, NT>(D, bm, bn, ks, k_lo, nc, lds, wave & 3, lane, wt);
    else ring_consumer<AK, PRO, NS, NT>(D, bm * 64, k_lo, nc, lds, wave, lane, tl_first);
    GR_TL_WG(1);
}

#define GR_FOLD_WGS (256 / GB_FOLD_COLS)
template <bool FOLD, int NS, int NT>
__global__ __launch_bounds__(GR_T) __attribute__((amdgpu_waves_per_eu(NS == 3 ? 4 : 2, NS == 3 ? 4 : 2))) void gemm_ring_kernel(const RingBundle bundle) {
    __shared__ __attribute__((aligned(16))) float lds[NS * GR_SLOT + GR_CST];
    static_assert(NS * GR_SLOT >= 4 * 16 * 64 * 4, "the ring doubles as the K quarters' hand-over area (64 KB)");
    int t = blockIdx.x;
    if (FOLD) {
        if (__builtin_expect(t < GR_FOLD_WGS, 0)) {
            const naf_gemm_bn2bwd_t& P = bundle.fold_pro;
            asm volatile("" ::"s"(P.partials), "s"(P.gamma), "s"(P.save_invstd), "s"(P.epoch), "s"(P.npb), "s"(P.B), "s"(P.H), "s"(P.cst),
                         "s"(P.d_gamma), "s"(P.d_beta));
            gemm_bn2bwd_fold_block<GR_T>(P, t, threadIdx.x, lds);
            return;
        }
        t -= GR_FOLD_WGS;
    }
    const int n_fold = FOLD ? GR_FOLD_WGS : 0;
    int gi = 0;
#pragma unroll
    for (int i = 1; i < NAF_GEMM_BUNDLE_MAX; ++i)
        if (i < bundle.n && t >= bundle.d[i].tile0) gi = i;
    const RingDesc& D = bundle.d[gi];
    asm volatile("" ::"s"(D.tile0), "s"(D.tiles_mn), "s"(D.tiles_n), "s"(D.a_kmajor), "s"(D.k_split), "s"(D.M), "s"(D.K));
    const int idx = t - D.tile0;
    int ks = idx / D.tiles_mn;
    const int lt = idx - ks * D.tiles_mn;
    int bm = lt / D.tiles_n, bn = lt - bm * D.tiles_n;
    // Placement (speed only; workgroup t runs on XCD t % 8): every launch of the chain works by eighths of the batch — XCD x holds
    // the rows of the x-th eighth in its L2 (csrc/big_batch.hip, bb_place_rows). dA1: the row tiles of eighth x with all their column
    // tiles on XCD x. Weight gradients cut into S K ranges: range s = rows of eighths 8 s / S ..: on the XCDs that hold them.
    const int tiles_m = D.tiles_mn / D.tiles_n;
    if (!D.a_kmajor) {
        if ((tiles_m & 7) == 0 && (D.tile0 & 7) == 0) {
            const int x = lt & 7, sl = lt >> 3;
            bm = x * (tiles_m >> 3) + sl / D.tiles_n;
            bn = sl % D.tiles_n;
        }
    } else {
        const int S = D.k_split;
        if ((S == 2 || S == 4 || S == 8) && (D.tile0 & 7) == 0 && ((D.tiles_mn * S) & 7) == 0) {
            const int x = idx & 7, sl = idx >> 3, per = 8 / S;
            const int tile = sl * per + (x % per);             // sl < tiles_mn / per
            ks = x / per;
            bm = tile / D.tiles_n;
            bn = tile - bm * D.tiles_n;
        }
    }
    const bool wt = bundle.wt != 0;
#define GR_GO(AK_, PRO_) ring_block<AK_, PRO_, NS, NT>(D, bm, bn, ks, lds, wt, n_fold)
    if (D.a_kmajor) {
        if (D.pro.z) GR_GO(true, true);
        else GR_GO(true, false);
    } else {
        if (D.pro.z) GR_GO(false, true);
        else GR_GO(false, false);
    }
#undef GR_GO
}

// 1: the bundle does not fit this form (the caller runs the 32 x 32 form); NAF_OK: launched; anything else: an error
#ifndef GR_NT
#define GR_NT 2
#endif
static int naf_gemm_ring_launch(const naf_gemm_desc_t* descs, int n, hipStream_t st) {
    // 64 x 32 blocks: twice the blocks of 64 x 64 at half the MFMAs each. The launch is bound by MFMA issue at the clock the chip
    // holds under load (~1.6 GHz: in-kernel stubs, DESIGN), not by staging — the DMAs alone take a fifth of the K loop — so what
    // counts is how evenly the blocks cover the 1024 SIMDs and how much of a block's fixed cost another block's MFMAs hide
    const int nt = GR_NT;
    RingBundle b;
    memset(&b, 0, sizeof(b));
    b.n = n;
    int tiles = 0, max_dim = 0;
    for (int i = 0; i < n; ++i) {
        const naf_gemm_desc_t& s = descs[i];
        if (!s.b_kmajor || s.sumsq) return 1;                              // B is [K][N] in every product of the chain
        const int ksn = s.k_split > 0 ? s.k_split : 1;
        if (s.K % ksn || (s.K / ksn) % GR_KC || (s.lda & 3) || (s.ldb & 3) || (s.ldc & 3)) return 1;
        if (s.a_kmajor) {
            if (!s.C || (s.N & 3)) return 1;
        } else {
            if ((s.M & 63) || (s.N & (16 * nt - 1)) || ksn != 1) return 1;
            if (s.epi && (!s.epi->xhat || !s.epi->gamma || !s.epi->beta || s.epi->ldx < 32)) return 1;   // the epilogue reads the kept xhat
            if (!s.epi && !s.C) return 1;
        }
        if ((uint64_t)(s.a_kmajor ? s.K : s.M) * s.lda * 4 >= 0x7fffffffull || (uint64_t)s.K * s.ldb * 4 >= 0x7fffffffull) return 1;
        RingDesc& d = b.d[i];
        d.A = s.A; d.B = s.B; d.C = s.C;
        d.M = s.M; d.N = s.N; d.K = s.K;
        d.lda = s.lda; d.ldb = s.ldb; d.ldc = s.ldc;
        d.a_kmajor = s.a_kmajor;
        d.tile0 = tiles;
        d.nt = nt;
        d.tiles_n = (s.N + 16 * nt - 1) / (16 * nt);
        d.tiles_mn = ((s.M + 63) / 64) * d.tiles_n;
        d.k_split = ksn;
        d.c_split_stride = s.c_split_stride;
        if (s.pro) {
            const naf_gemm_bn2bwd_t& q = *s.pro;
            if (q.H != 256 || (s.a_kmajor ? s.M != 256 : s.K != 256) || q.npb < 1 || q.npb > 128) return 1;
            d.pro = q;
            if (!b.n_fold) {
                b.n_fold = q.H / GB_FOLD_COLS;
                b.fold_pro = q;
            }
        }
        if (s.epi) d.epi = *s.epi;
        tiles += d.tiles_mn * ksn;
        if (s.M > max_dim) max_dim = s.M;
        if (s.K > max_dim) max_dim = s.K;
    }
    for (int i = n; i < NAF_GEMM_BUNDLE_MAX; ++i) b.d[i] = b.d[0];
    b.total_tiles = tiles;
    b.wt = max_dim >= NAF_WT_MIN_B;
    // Two shapes of the launch. More blocks than CUs: 64 x 32 blocks on a 3-slot ring (76 KB, 128 registers: two workgroups per CU,
    // whose fixed costs hide under each other's MFMAs). Otherwise one workgroup per CU on a deep ring (GR_DEEP slots).
    const bool deep = tiles + b.n_fold <= 256 && GR_DEEP != 3;
    const int grid = tiles + b.n_fold;
#define GR_LAUNCH(F_) do { if (deep) { if (nt == 2) gemm_ring_kernel<F_, GR_DEEP, 2><<<grid, GR_T, 0, st>>>(b); \
                                       else gemm_ring_kernel<F_, GR_DEEP, 4><<<grid, GR_T, 0, st>>>(b); } \
                           else gemm_ring_kernel<F_, 3, 2><<<grid, GR_T, 0, st>>>(b); } while (0)
    if (b.n_fold) GR_LAUNCH(true);
    else GR_LAUNCH(false);
#undef GR_LAUNCH
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}
