// NAF head math shared by naf_head.hip (heads rows staged from memory) and fused_layers.hip (heads rows produced
// in LDS by an MFMA GEMM). One sample per 8-lane group (16-, 32- or 64-lane group: G below), lane i owns row i of L. See naf_head.hip for the mapping.
#pragma once
#include "common.h"
#include "../../include/naf_hip.h"

#define HEAD_SPB 32          // samples per workgroup of the MFMA-fused kernel (fused_layers.hip)
#define HEAD_THREADS 256
#define NH_SPB 8             // samples per workgroup of the stand-alone head kernels (naf_head.hip): one wave, so
#define NH_THREADS 64        // B=256 spreads over 32 CUs (3.6 us per launch vs 4.1 us with 32 samples per workgroup)
#define HEAD_MAX_LDH 48      // A=8: 8+36+1 = 45 -> 48
#define LT_STRIDE 9          // 8x8 L tile padded to 9 columns: column reads hit distinct banks

__device__ static inline float group8_sum(float x) {
    return naf_sum8(x);                                  // (DPP: common.h; bitwise the xor tree 1, 2, 4)
}
// all-reduce over an aligned group of G lanes (16 | 32 | 64: the wider sample groups of arms with more than 8 joints)
template <int G>
__device__ __forceinline__ static float naf_group_sum(float x) {
    return G == 16 ? naf_sum16(x) : (G == 32 ? naf_xor16_add(naf_sum16(x)) : naf_sum64(x));
}

// sh_in : HEAD_SPB heads rows (stride ldh) already in LDS and visible (caller synchronised)
// sh_out: HEAD_SPB x ldh floats, zero-filled by the caller when MODE != 0; receives d_heads rows
// sh_L  : HEAD_SPB*G*(G+1) floats (matmul mode only); sh_red: HEAD_THREADS/64 floats
// Ends with a __syncthreads() when MODE != 0, after which sh_out is complete. MODE 0 returns early per lane.
// MODE: 0 = forward only (q, optional mu); 1 = backward given dq; 2 = fused TD target + MSE + backward
// u_val: this lane's action component (lane i of the sample's G-lane group, 0 beyond A); r_val / vnext_val / dq_val:
// the sample's reward, V'(s') and dLoss/dQ, needed on lane 0 of the group only. The caller fetches them BEFORE the
// barrier that publishes sh_in, so their latency overlaps the staging instead of following it.
// G: lanes per sample — 8 (A <= 8: every BASELINE config), 16 (9 .. 11 joints inside the row-split chain's fused layer-2 launch,
// csrc/big_batch.hip; the arithmetic per sample is the same in the same order, as in naf_head_wide.hip), 32 | 64 (17 .. 64 joints:
// naf_head_any_kernel, csrc/naf_head_wide.hip). With G > 8 only the samples that exist (s_loc < ns) touch sh_L: the caller sizes it
// for its live rows, not for every lane group of the workgroup.
template <int PMODE, int MODE, int NTHREADS = HEAD_THREADS, int G = 8>
__device__ static inline void naf_head_body(const float* sh_in, float* sh_out, float* sh_L, float* sh_red, int ldh,
                                            float u_val, float r_val, float vnext_val, float dq_val, float gamma,
                                            float* __restrict__ q_out, float* __restrict__ mu_out,
                                            float* __restrict__ loss_partials, int B, int A, int64_t s0, int ns) {
    static_assert(G == 8 || G == 16 || G == 32 || G == 64, "lanes per sample");
    constexpr int LTS = G + 1;    // G x G L tile padded to G + 1 columns: column reads hit distinct banks (LT_STRIDE at G = 8)
    const int T = A * (A + 1) / 2;
    const int tid = threadIdx.x;
    const int s_loc = tid / G;    // sample within the workgroup
    const int i = tid & (G - 1);  // row of L owned by this lane
    const int64_t s = s0 + s_loc;
    const bool live = s_loc < ns;
    const bool row_on = live && i < A;
    const bool lt_on = G == 8 || live;

    const float* hrow = sh_in + s_loc * ldh;
    float mu = 0.f, d = 0.f, Vv = 0.f;
    float t_row[G], L_row[G];
#pragma unroll
    for (int j = 0; j < G; ++j) { t_row[j] = 0.f; L_row[j] = 0.f; }
    float tii = 0.f, Lii = 0.f;   // Hadamard mode: the diagonal entry of this lane's row is all P = L (*) L^T keeps
    if (row_on) {
        mu = tanhf(hrow[i]);
        d = u_val - mu;
        const int rbase = A + i * (i + 1) / 2;
        if (PMODE == NAF_P_HADAMARD) {
            // one tanh per lane instead of eight: the wave executes every j of the loop below whatever its mask, and the
            // kernel is one wave whose run time is its instruction count
            tii = tanhf(hrow[rbase + i]);
            Lii = expf(tii);
        } else {
#pragma unroll
            for (int j = 0; j < G; ++j) {
                if (j <= i) {
                    float t = tanhf(hrow[rbase + j]);
                    t_row[j] = t;
                    L_row[j] = (j == i) ? expf(t) : t;
                }
            }
        }
    }
    if (live) Vv = hrow[A + T];

    // ---- quadratic form ---------------------------------------------------------------------------
    float quad_part = 0.f;
    float w = 0.f;  // matmul mode: w_i = (L^T d)_i, lane i holds component i
    float Pii = 0.f;
    if (PMODE == NAF_P_HADAMARD) {
        // P = L (*) L^T = diag(L_ii^2): off-diagonal entries of L multiply structural zeros of L^T
        Pii = Lii * Lii;
        quad_part = Pii * d * d;
    } else {
        float* Lt = sh_L + s_loc * G * LTS;
        if (lt_on) {
#pragma unroll
            for (int j = 0; j < G; ++j) Lt[i * LTS + j] = L_row[j];
        }
        __syncthreads();  // reached by every lane: no early exit above
        const int gb = (tid & 63) & ~(G - 1);  // first lane of this sample's group inside the wave
#pragma unroll
        for (int k = 0; k < G; ++k) {
            float dk = __shfl(d, gb + k);
            if (k >= i && lt_on) w += Lt[k * LTS + i] * dk;  // column i of L
        }
        quad_part = w * w;
    }
    const float quad = G == 8 ? group8_sum(quad_part) : naf_group_sum<G>(quad_part);
    const float Q = Vv - 0.5f * quad;

    if (MODE == 0) {
        if (live && i == 0) q_out[s] = Q;
        if (mu_out && row_on) mu_out[s * A + i] = mu;
        return;
    }

    // ---- dLoss/dQ ---------------------------------------------------------------------------------
    float dq = 0.f;
    float sq_err = 0.f;
    if (MODE == 1) {
        if (live) dq = __shfl(dq_val, (tid & 63) & ~(G - 1));
    } else {
        // lane 0 of the group fetches r and V'(s'); the group shares them by shuffle (uniform control flow)
        float y = 0.f;
        if (live && i == 0) y = r_val + gamma * vnext_val;
        y = __shfl(y, (tid & 63) & ~(G - 1));
        if (live) {
            float e = Q - y;
            dq = 2.0f * e / (float)B;
            if (i == 0) {
                sq_err = e * e / (float)B;
                if (q_out) q_out[s] = Q;
            }
        }
    }

    // ---- backward ---------------------------------------------------------------------------------
    float* orow = sh_out + s_loc * ldh;
    if (PMODE == NAF_P_HADAMARD) {
        if (row_on) {
            // dQ/dmu_i = P_ii d_i ; dQ/dl_ii = -P_ii d_i^2 (through L_ii = exp(t)); off-diagonals: exactly 0
            orow[i] = dq * (Pii * d) * (1.0f - mu * mu);
            orow[A + i * (i + 1) / 2 + i] = dq * (-(Pii * d * d)) * (1.0f - tii * tii);
        }
    } else {
        const int gb = (tid & 63) & ~(G - 1);
        float Lw = 0.f;
        float wj[G];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            wj[j] = __shfl(w, gb + j);
            Lw += L_row[j] * wj[j];  // L_row[j] = 0 for j > i
        }
        if (row_on) {
            // dQ/dmu = L w ; dQ/dL_ij = -d_i w_j (j <= i); diagonal chains through exp
            orow[i] = dq * Lw * (1.0f - mu * mu);
            const int rbase = A + i * (i + 1) / 2;
#pragma unroll
            for (int j = 0; j < G; ++j) {
                if (j <= i) {
                    float dL = -d * wj[j];
                    float dt = (j == i) ? dL * L_row[j] : dL;
                    orow[rbase + j] = dq * dt * (1.0f - t_row[j] * t_row[j]);
                }
            }
        }
    }
    if (live && i == 0) orow[A + T] = dq;  // dQ/dV = 1

    if (MODE == 2) {
        // workgroup sum of squared TD errors (one per sample, on lane 0 of its group), fixed order -> bitwise reproducible
        float x = sq_err;
        x = G == 8 ? naf_xor32_add(naf_xor16_add(naf_xor8_add(x))) : (G == 16 ? naf_xor32_add(naf_xor16_add(x)) : (G == 32 ? naf_xor32_add(x) : x));
        if ((tid & 63) == 0) sh_red[tid >> 6] = x;
    }
    __syncthreads();
    if (MODE == 2 && tid == 0 && loss_partials) {
        float x = 0.f;
        for (int k = 0; k < NTHREADS / 64; ++k) x += sh_red[k];
        loss_partials[blockIdx.x] = x;
    }
}

// Exploration noise for ONE sample per G-lane group (lane i = tid & (G - 1) owns component i; G = 8, or 16 for 9 .. 16 joints):
// action = clamp(mu + noise_scale * P^{-1/2} z, -1, 1), z ~ N(0, I) from Philox keyed by (ctr, sample, lane, seed). Hadamard:
// P = diag(L_ii^2), so the covariance inverse(P) is diag(exp(-2 tanh l_ii)); matmul: cov = (L L^T)^-1 = L^-T L^-1, x solves
// L^T x = z by back substitution over the group. hrow: the sample's heads row (global or LDS); Lt: G * (G + 1) floats of LDS
// (matmul mode). Contains a __syncthreads() in matmul mode: call it from every thread of the workgroup (live = false
// for lanes without a sample).
// the draw itself: z ~ N(0, 1) of (seed, stream position, sample, row) by Philox + Box-Muller — it depends on nothing else, so a
// caller may take it ahead of the heads (naf_act_noise_body_z)
__device__ static inline float naf_act_noise_z(uint64_t seed, uint64_t ctr, int64_t s, int i, bool row_on) {
    float z = 0.f;
    if (row_on) {
        Philox4 p = philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)s, (uint32_t)i, (uint32_t)seed,
                                  (uint32_t)(seed >> 32));
        float u1 = naf_u01(p.v[0]), u2 = naf_u01(p.v[1]);
        z = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
    }
    return z;
}
template <int PMODE, int G = 8>
__device__ static inline void naf_act_noise_body_z(const float* hrow, float* Lt, float* __restrict__ action_out, float z,
                                                   float noise_scale, int64_t s, bool live, int A, int tid) {
    static_assert(G == 8 || G == 16 || G == 32 || G == 64, "lanes per sample");
    constexpr int LTS = G + 1;
    const int i = tid & (G - 1);
    const bool row_on = live && i < A;
    float mu = 0.f, Lii = 1.f;
    float L_row[G];
#pragma unroll
    for (int j = 0; j < G; ++j) L_row[j] = 0.f;
    if (row_on) {
        mu = tanhf(hrow[i]);
        const int rbase = A + i * (i + 1) / 2;
        if (PMODE == NAF_P_HADAMARD) {
            Lii = expf(tanhf(hrow[rbase + i]));
        } else {
#pragma unroll
            for (int j = 0; j < G; ++j) {
                if (j <= i) {
                    float t = tanhf(hrow[rbase + j]);
                    L_row[j] = (j == i) ? expf(t) : t;
                }
            }
        }
    }
    float x = 0.f;
    if (PMODE == NAF_P_HADAMARD) {
        x = z / Lii;
    } else {
        if (live) {
#pragma unroll
            for (int j = 0; j < G; ++j) Lt[i * LTS + j] = L_row[j];
        }
        __syncthreads();
        const int gb = (tid & 63) & ~(G - 1);
        // solve L^T x = z: x_j = (z_j - sum_{k>j} L_kj x_k) / L_jj, j = A-1 .. 0 (lane j owns x_j)
        float acc = z;
        for (int k = G - 1; k >= 0; --k) {
            float Lkk = Lt[k * LTS + k];
            float xk_mine = (k < A) ? acc / (Lkk == 0.f ? 1.f : Lkk) : 0.f;
            float xk = __shfl(xk_mine, gb + k);  // lane k's value is the finished x_k
            if (i == k) x = xk;
            if (i < k) acc -= Lt[k * LTS + i] * xk;
        }
    }
    if (row_on) {
        float a = mu + noise_scale * x;
        a = fminf(1.0f, fmaxf(-1.0f, a));
        action_out[s * A + i] = a;
    }
}
template <int PMODE, int G = 8>
__device__ static inline void naf_act_noise_body(const float* hrow, float* Lt, float* __restrict__ action_out,
                                                 uint64_t seed, uint64_t ctr, float noise_scale, int64_t s, bool live,
                                                 int A, int tid) {
    const float z = naf_act_noise_z(seed, ctr, s, tid & (G - 1), live && (tid & (G - 1)) < A);
    naf_act_noise_body_z<PMODE, G>(hrow, Lt, action_out, z, noise_scale, s, live, A, tid);
}
