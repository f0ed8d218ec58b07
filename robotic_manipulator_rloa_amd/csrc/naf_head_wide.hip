// The NAF head for 9 <= A <= 16 joints (gfx950): the kernels of naf_head.hip hold one sample per 8-lane group (A <= 8: every
// BASELINE config); the reference builds `matrix_entries` for ANY action size (naf_neural_network.py:53-54, A (A + 1) / 2
// outputs), so a 9-joint arm must train too. Here: one sample per 16-lane group, lane i owns row i of L, four samples per wave,
// sixteen per workgroup; the same arithmetic in the same order per sample as head_body.h (tanh(mu), tanh(l), exp on the diagonal,
// P = L (*) L^T or L L^T, Q = V - 1/2 d^T P d, the TD target, the MSE and the whole backward), group sums over 16 lanes on DPP.
// The heads rows (up to 16 + 136 + 1 = 153 -> 160 floats) are read where they lie; d_heads rows are assembled in LDS and leave as
// whole rows. Replaces naf_neural_network.py:81-115 (+ autograd) and naf_algorithm.py:199-208 for those action sizes; reached
// through the entry points of naf_head.hip (naf_head_fwd / _bwd / _fwd_bwd_mse, naf_act_noise), which dispatch on A.
// 17 <= A <= 64 (second half of the file): one sample per 32- / 64-lane group through the shared body of head_body.h.
#include "common.h"
#include "../../include/naf_hip.h"
#include "head_body.h"

#define HW_G 16                  // lanes per sample
#define HW_THREADS 256
#define HW_SPB (HW_THREADS / HW_G)
#define HW_MAX_LDH 160
#define HW_LT 17                 // 16 x 16 L tile padded to 17 columns

// MODE: 0 = forward only (q, optional mu); 1 = backward given dq; 2 = fused TD target + MSE + backward
template <int PMODE, int MODE>
__global__ __launch_bounds__(HW_THREADS) void naf_head_wide_kernel(const float* __restrict__ heads, int ldh,
                                                                   const float* __restrict__ u, int ldu,
                                                                   const float* __restrict__ r, int ldr,
                                                                   const float* __restrict__ v_next, int ldv,
                                                                   const float* __restrict__ dq_in, float gamma,
                                                                   float* __restrict__ q_out, float* __restrict__ mu_out,
                                                                   float* __restrict__ d_heads, float* __restrict__ loss_partials,
                                                                   int B, int A) {
    __shared__ __attribute__((aligned(16))) float sh_out[MODE == 0 ? 4 : HW_SPB * HW_MAX_LDH];
    __shared__ float sh_L[PMODE == NAF_P_MATMUL ? HW_SPB * HW_G * HW_LT : 1];
    __shared__ float sh_red[HW_THREADS / 64];
    const int tid = threadIdx.x, T = A * (A + 1) / 2;
    const int s_loc = tid / HW_G, i = tid & (HW_G - 1);
    const int64_t s = (int64_t)blockIdx.x * HW_SPB + s_loc;
    const bool live = s < B, row_on = live && i < A;
    const int gb = (tid & 63) & ~(HW_G - 1);            // first lane of this sample's group inside the wave
    if (MODE != 0)
        for (int k = tid; k < HW_SPB * ldh; k += HW_THREADS) sh_out[(k / ldh) * HW_MAX_LDH + k % ldh] = 0.f;

    const float* hrow = heads + (live ? s : 0) * ldh;
    float mu = 0.f, d = 0.f, Vv = 0.f, tii = 0.f, Lii = 0.f;
    float t_row[HW_G], L_row[HW_G];
#pragma unroll
    for (int j = 0; j < HW_G; ++j) { t_row[j] = 0.f; L_row[j] = 0.f; }
    if (row_on) {
        mu = tanhf(hrow[i]);
        d = u[s * ldu + i] - mu;
        const int rbase = A + i * (i + 1) / 2;
        if (PMODE == NAF_P_HADAMARD) {
            tii = tanhf(hrow[rbase + i]);
            Lii = expf(tii);
        } else {
#pragma unroll
            for (int j = 0; j < HW_G; ++j) {
                if (j <= i) {
                    const float t = tanhf(hrow[rbase + j]);
                    t_row[j] = t;
                    L_row[j] = (j == i) ? expf(t) : t;
                }
            }
        }
    }
    if (live) Vv = hrow[A + T];

    float quad_part = 0.f, w = 0.f, Pii = 0.f;
    if (PMODE == NAF_P_HADAMARD) {
        Pii = Lii * Lii;                                 // P = L (*) L^T keeps the squared diagonal only
        quad_part = Pii * d * d;
        if (MODE != 0) __syncthreads();                  // (the zero fill of sh_out)
    } else {
        float* Lt = sh_L + s_loc * HW_G * HW_LT;
#pragma unroll
        for (int j = 0; j < HW_G; ++j) Lt[i * HW_LT + j] = L_row[j];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < HW_G; ++k) {
            const float dk = __shfl(d, gb + k);
            if (k >= i) w += Lt[k * HW_LT + i] * dk;      // column i of L: w = L^T d
        }
        quad_part = w * w;
    }
    const float quad = naf_sum16(quad_part);
    const float Q = Vv - 0.5f * quad;
    if (MODE == 0) {
        if (live && i == 0) q_out[s] = Q;
        if (mu_out && row_on) mu_out[s * A + i] = mu;
        return;
    }

    float dq = 0.f, sq_err = 0.f;
    if (MODE == 1) {
        float v = (live && i == 0) ? dq_in[s] : 0.f;
        dq = __shfl(v, gb);
    } else {
        float y = 0.f;
        if (live && i == 0) y = r[s * ldr] + gamma * v_next[s * ldv];
        y = __shfl(y, gb);
        if (live) {
            const float e = Q - y;
            dq = 2.0f * e / (float)B;
            if (i == 0) {
                sq_err = e * e / (float)B;
                if (q_out) q_out[s] = Q;
            }
        }
    }
    float* orow = sh_out + s_loc * HW_MAX_LDH;
    if (PMODE == NAF_P_HADAMARD) {
        if (row_on) {
            orow[i] = dq * (Pii * d) * (1.0f - mu * mu);
            orow[A + i * (i + 1) / 2 + i] = dq * (-(Pii * d * d)) * (1.0f - tii * tii);
        }
    } else {
        float Lw = 0.f, wj[HW_G];
#pragma unroll
        for (int j = 0; j < HW_G; ++j) {
            wj[j] = __shfl(w, gb + j);
            Lw += L_row[j] * wj[j];                       // L_row[j] = 0 for j > i
        }
        if (row_on) {
            orow[i] = dq * Lw * (1.0f - mu * mu);
            const int rbase = A + i * (i + 1) / 2;
#pragma unroll
            for (int j = 0; j < HW_G; ++j) {
                if (j <= i) {
                    const float dL = -d * wj[j];
                    const float dt = (j == i) ? dL * L_row[j] : dL;
                    orow[rbase + j] = dq * dt * (1.0f - t_row[j] * t_row[j]);
                }
            }
        }
    }
    if (live && i == 0) orow[A + T] = dq;                  // dQ/dV = 1
    if (MODE == 2) {
        // the workgroup's squared TD errors (one per sample, on lane 0 of its group), fixed order
        float x = naf_sum64(sq_err);
        if ((tid & 63) == 0) sh_red[tid >> 6] = x;
    }
    __syncthreads();
    if (MODE == 2 && tid == 0 && loss_partials) {
        float x = 0.f;
        for (int k = 0; k < HW_THREADS / 64; ++k) x += sh_red[k];
        loss_partials[blockIdx.x] = x;
    }
    const int64_t s0 = (int64_t)blockIdx.x * HW_SPB;
    const int ns = (B - s0) < HW_SPB ? (int)(B - s0) : HW_SPB;
    for (int k = tid; k < ns * ldh; k += HW_THREADS) d_heads[s0 * ldh + k] = sh_out[(k / ldh) * HW_MAX_LDH + k % ldh];
}

// ------------------------------------------------------------------------------------------------------------------------------
// 17 <= A <= 64 (round 6: the reference builds its head for ANY action size; beyond 16 joints NetLayout raised): one sample per 32- or
// 64-lane group through the shared body (head_body.h, G = 32 | 64) — the heads rows (up to 64 + 2080 + 1 = 2145 -> 2160 floats) are
// staged into LDS, d_heads rows assembled there and stored as whole rows. 256 threads: 8 | 4 samples per workgroup. Not a fast path
// (nothing about a 40-joint arm is): the same arithmetic, checked against the f64 oracle like the others.
// ------------------------------------------------------------------------------------------------------------------------------
template <int PMODE, int MODE, int G>
__global__ __launch_bounds__(HW_THREADS) void naf_head_any_kernel(const float* __restrict__ heads, int ldh,
                                                                  const float* __restrict__ u, int ldu,
                                                                  const float* __restrict__ r, int ldr,
                                                                  const float* __restrict__ v_next, int ldv,
                                                                  const float* __restrict__ dq_in, float gamma,
                                                                  float* __restrict__ q_out, float* __restrict__ mu_out,
                                                                  float* __restrict__ d_heads, float* __restrict__ loss_partials,
                                                                  int B, int A) {
    constexpr int SPB = HW_THREADS / G, LDH_MAX = G == 32 ? 576 : 2160;
    __shared__ __attribute__((aligned(16))) float sh_in[SPB * LDH_MAX];
    __shared__ __attribute__((aligned(16))) float sh_out[MODE == 0 ? 4 : SPB * LDH_MAX];
    __shared__ float sh_L[PMODE == NAF_P_MATMUL ? SPB * G * (G + 1) : 1];
    __shared__ float sh_red[HW_THREADS / 64];
    const int tid = threadIdx.x, s_loc = tid / G, i = tid & (G - 1);
    const int64_t s0 = (int64_t)blockIdx.x * SPB;
    const int ns = (B - s0) < SPB ? (int)(B - s0) : SPB;
    for (int k = tid; k < ns * ldh; k += HW_THREADS) sh_in[k] = heads[s0 * ldh + k];
    if (MODE != 0)
        for (int k = tid; k < SPB * ldh; k += HW_THREADS) sh_out[k] = 0.f;
    const bool live = s_loc < ns;
    const int64_t s = s0 + s_loc;
    const float u_val = (live && i < A) ? u[s * ldu + i] : 0.f;
    float r_val = 0.f, vn_val = 0.f, dq_val = 0.f;
    if (live && i == 0) {
        if (MODE == 2) { r_val = r[s * ldr]; vn_val = v_next[s * ldv]; }
        if (MODE == 1) dq_val = dq_in[s];
    }
    __syncthreads();
    naf_head_body<PMODE, MODE, HW_THREADS, G>(sh_in, sh_out, sh_L, sh_red, ldh, u_val, r_val, vn_val, dq_val, gamma, q_out, mu_out,
                                              loss_partials, B, A, s0, ns);
    if (MODE == 0) return;                                 // (the body returned per lane; nothing to store)
    for (int k = tid; k < ns * ldh; k += HW_THREADS) d_heads[s0 * ldh + k] = sh_out[k];
}

// called by the entry points of naf_head.hip for A > 8 (arguments already checked there)
int naf_head_wide_launch(int mode, const float* heads, int ldh, const float* u, int ldu, const float* r, int ldr, const float* v_next,
                         int ldv, const float* dq, float gamma, float* q_out, float* mu_out, float* d_heads, float* loss_partials,
                         int B, int A, int p_mode, hipStream_t st) {
    if (A > HW_G) {
        if (A > NAF_MAX_A_WIDE || ldh > (A <= 32 ? 576 : 2160)) return NAF_ERR_ARG;
        const int g = A <= 32 ? 32 : 64;
        const int blocks_any = (B + HW_THREADS / g - 1) / (HW_THREADS / g);
#define HA_GO(PM, MD, GV) naf_head_any_kernel<PM, MD, GV><<<blocks_any, HW_THREADS, 0, st>>>(heads, ldh, u, ldu, r, ldr, v_next, ldv, dq, gamma, \
                                                                                       q_out, mu_out, d_heads, loss_partials, B, A)
#define HA_MD(PM, GV)                          \
    do {                                       \
        if (mode == 0) HA_GO(PM, 0, GV);       \
        else if (mode == 1) HA_GO(PM, 1, GV);  \
        else HA_GO(PM, 2, GV);                 \
    } while (0)
        if (p_mode == NAF_P_HADAMARD) { if (g == 32) HA_MD(NAF_P_HADAMARD, 32); else HA_MD(NAF_P_HADAMARD, 64); }
        else { if (g == 32) HA_MD(NAF_P_MATMUL, 32); else HA_MD(NAF_P_MATMUL, 64); }
#undef HA_MD
#undef HA_GO
        NAF_CHECK_LAUNCH();
        return NAF_OK;
    }
    if (A <= 8 || ldh > HW_MAX_LDH) return NAF_ERR_ARG;
    const int blocks = (B + HW_SPB - 1) / HW_SPB;
#define HW_GO(PM, MD) naf_head_wide_kernel<PM, MD><<<blocks, HW_THREADS, 0, st>>>(heads, ldh, u, ldu, r, ldr, v_next, ldv, dq, gamma, \
                                                                              q_out, mu_out, d_heads, loss_partials, B, A)
    if (p_mode == NAF_P_HADAMARD) {
        if (mode == 0) HW_GO(NAF_P_HADAMARD, 0);
        else if (mode == 1) HW_GO(NAF_P_HADAMARD, 1);
        else HW_GO(NAF_P_HADAMARD, 2);
    } else {
        if (mode == 0) HW_GO(NAF_P_MATMUL, 0);
        else if (mode == 1) HW_GO(NAF_P_MATMUL, 1);
        else HW_GO(NAF_P_MATMUL, 2);
    }
#undef HW_GO
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}

// exploration noise for 9 <= A <= 16: action = clamp(mu + noise_scale * P^{-1/2} z, -1, 1), one sample per 16-lane group; the same
// Philox stream (seed, position, sample, row) and the same arithmetic as head_body.h's naf_act_noise_body
template <int PMODE>
__global__ __launch_bounds__(HW_THREADS) void naf_act_noise_wide_kernel(const float* __restrict__ heads, int ldh,
                                                                        float* __restrict__ action_out, uint64_t seed,
                                                                        const uint64_t* __restrict__ counter_dev, uint64_t counter_off,
                                                                        float noise_scale, int E, int A) {
    __shared__ float sh_L[PMODE == NAF_P_MATMUL ? HW_SPB * HW_G * HW_LT : 1];
    const int tid = threadIdx.x, s_loc = tid / HW_G, i = tid & (HW_G - 1);
    const int64_t s = (int64_t)blockIdx.x * HW_SPB + s_loc;
    const uint64_t ctr = (counter_dev ? *counter_dev : 0ull) + counter_off;
    const bool live = s < E, row_on = live && i < A;
    const float* hrow = heads + (live ? s : 0) * ldh;
    float mu = 0.f, z = 0.f, Lii = 1.f, L_row[HW_G];
#pragma unroll
    for (int j = 0; j < HW_G; ++j) L_row[j] = 0.f;
    if (row_on) {
        mu = tanhf(hrow[i]);
        const int rbase = A + i * (i + 1) / 2;
        if (PMODE == NAF_P_HADAMARD) {
            Lii = expf(tanhf(hrow[rbase + i]));
        } else {
#pragma unroll
            for (int j = 0; j < HW_G; ++j) {
                if (j <= i) {
                    const float t = tanhf(hrow[rbase + j]);
                    L_row[j] = (j == i) ? expf(t) : t;
                }
            }
        }
        const Philox4 p = philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)s, (uint32_t)i, (uint32_t)seed,
                                        (uint32_t)(seed >> 32));
        const float u1 = naf_u01(p.v[0]), u2 = naf_u01(p.v[1]);
        z = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
    }
    float x = 0.f;
    if (PMODE == NAF_P_HADAMARD) {
        x = z / Lii;
    } else {
        float* Lt = sh_L + s_loc * HW_G * HW_LT;
#pragma unroll
        for (int j = 0; j < HW_G; ++j) Lt[i * HW_LT + j] = L_row[j];
        __syncthreads();
        const int gb = (tid & 63) & ~(HW_G - 1);
        float acc = z;                                     // solve L^T x = z by back substitution (lane j owns x_j)
        for (int k = HW_G - 1; k >= 0; --k) {
            const float Lkk = Lt[k * HW_LT + k];
            const float xk_mine = (k < A) ? acc / (Lkk == 0.f ? 1.f : Lkk) : 0.f;
            const float xk = __shfl(xk_mine, gb + k);
            if (i == k) x = xk;
            if (i < k) acc -= Lt[k * HW_LT + i] * xk;
        }
    }
    if (row_on) {
        float a = mu + noise_scale * x;
        a = fminf(1.0f, fmaxf(-1.0f, a));
        action_out[s * A + i] = a;
    }
}

// 17 <= A <= 64: the shared noise body on a 32- / 64-lane group (head_body.h), the heads row read where it lies
template <int PMODE, int G>
__global__ __launch_bounds__(HW_THREADS) void naf_act_noise_any_kernel(const float* __restrict__ heads, int ldh,
                                                                       float* __restrict__ action_out, uint64_t seed,
                                                                       const uint64_t* __restrict__ counter_dev, uint64_t counter_off,
                                                                       float noise_scale, int E, int A) {
    constexpr int SPB = HW_THREADS / G;
    __shared__ float sh_L[PMODE == NAF_P_MATMUL ? SPB * G * (G + 1) : 1];
    const int tid = threadIdx.x, s_loc = tid / G;
    const int64_t s = (int64_t)blockIdx.x * SPB + s_loc;
    const uint64_t ctr = (counter_dev ? *counter_dev : 0ull) + counter_off;
    const bool live = s < E;
    naf_act_noise_body<PMODE, G>(heads + (live ? s : 0) * ldh, sh_L + s_loc * G * (G + 1), action_out, seed, ctr, noise_scale, s, live,
                                 A, tid);
}

int naf_act_noise_wide_launch(const float* heads, int ldh, float* action_out, uint64_t seed, const uint64_t* counter_dev,
                              uint64_t counter_off, float noise_scale, int E, int A, int p_mode, hipStream_t st) {
    if (A > HW_G) {
        if (A > NAF_MAX_A_WIDE) return NAF_ERR_ARG;
        const int g = A <= 32 ? 32 : 64;
        const int nb = (E + HW_THREADS / g - 1) / (HW_THREADS / g);
#define NA_GO(PM, GV) naf_act_noise_any_kernel<PM, GV><<<nb, HW_THREADS, 0, st>>>(heads, ldh, action_out, seed, counter_dev, counter_off, \
                                                                               noise_scale, E, A)
        if (p_mode == NAF_P_HADAMARD) { if (g == 32) NA_GO(NAF_P_HADAMARD, 32); else NA_GO(NAF_P_HADAMARD, 64); }
        else { if (g == 32) NA_GO(NAF_P_MATMUL, 32); else NA_GO(NAF_P_MATMUL, 64); }
#undef NA_GO
        NAF_CHECK_LAUNCH();
        return NAF_OK;
    }
    if (A <= 8) return NAF_ERR_ARG;
    const int blocks = (E + HW_SPB - 1) / HW_SPB;
    if (p_mode == NAF_P_HADAMARD)
        naf_act_noise_wide_kernel<NAF_P_HADAMARD><<<blocks, HW_THREADS, 0, st>>>(heads, ldh, action_out, seed, counter_dev, counter_off,
                                                                                 noise_scale, E, A);
    else
        naf_act_noise_wide_kernel<NAF_P_MATMUL><<<blocks, HW_THREADS, 0, st>>>(heads, ldh, action_out, seed, counter_dev, counter_off,
                                                                               noise_scale, E, A);
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}
