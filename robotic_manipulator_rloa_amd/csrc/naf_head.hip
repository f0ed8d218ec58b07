// NAF head for gfx950: tanh(mu), tanh(l), tril unpack, exp on the diagonal, P, the quadratic advantage,
// Q, the MSE/TD epilogue and the whole backward, fused. Replaces naf_neural_network.py:81-115 (+ its
// autograd) and naf_algorithm.py:199-208 of the reference.
//
// Mapping: one sample per 8-lane group (A <= 8), lane i owns row i of L; 8 samples per 64-lane wave = one
// workgroup (NH_SPB). The heads rows of a workgroup are one contiguous 32*ldh*4-byte span: staged into LDS
// with 16-B/lane loads, and d_heads leaves the same way. The A x A contraction is per sample — not a GEMM,
// so no MFMA: row reductions are 8-lane xor shuffles, column accesses of L go through a padded LDS tile.
#include "head_body.h"

// MODE: 0 = forward only (q, optional mu); 1 = backward given dq; 2 = fused TD target + MSE + backward
template <int PMODE, int MODE, int NSLAB = 0>
__global__ __launch_bounds__(NH_THREADS) void naf_head_kernel(const float* __restrict__ heads, int ldh,
                                                                const float* __restrict__ u, int ldu,
                                                                const float* __restrict__ r, int ldr,
                                                                const float* __restrict__ v_next, int ldv,
                                                                const float* __restrict__ dq_in, float gamma,
                                                                float* __restrict__ q_out, float* __restrict__ mu_out,
                                                                float* __restrict__ d_heads,
                                                                float* __restrict__ loss_partials, int B, int A,
                                                                int n_slabs, int64_t slab_stride,
                                                                int64_t vn_slab_stride) {
    // NSLAB > 0: `heads` (and `v_next`) are split-K partial results, NSLAB slabs slab_stride (vn_slab_stride) floats
    // apart, produced by naf_bn_relu_fwd_heads_partial; they are added in index order while being staged. NSLAB is a
    // compile-time constant on purpose: one wave runs this kernel, so its length in INSTRUCTIONS is its run time, and
    // run-time slab counts (clamped indices, predicated adds) cost 3.4 us per launch against 0.6 us for the loads
    constexpr bool SPLITK = NSLAB > 0;
    __shared__ __attribute__((aligned(16))) float sh_in[NH_SPB * HEAD_MAX_LDH];
    __shared__ __attribute__((aligned(16))) float sh_out[MODE == 0 ? 4 : NH_SPB * HEAD_MAX_LDH];
    __shared__ float sh_L[PMODE == NAF_P_MATMUL ? NH_SPB * 8 * LT_STRIDE : 1];
    __shared__ float sh_red[NH_THREADS / 64];
    const int tid = threadIdx.x;
    const int64_t s0 = (int64_t)blockIdx.x * NH_SPB;
    const int ns = (B - s0) < NH_SPB ? (int)(B - s0) : NH_SPB;

    // per-sample scalars first: their loads fly while the heads rows are staged
    const int s_loc_ = tid >> 3, i_ = tid & 7;
    const bool live_ = s_loc_ < ns;
    const int64_t s_ = s0 + s_loc_;
    const float u_val = (live_ && i_ < A) ? u[s_ * ldu + i_] : 0.f;
    float r_val = 0.f, vnext_val = 0.f, dq_val = 0.f;
    if (live_ && i_ == 0) {
        if (MODE == 2) {
            r_val = r[s_ * ldr];
            if (!SPLITK) vnext_val = v_next[s_ * ldv];
        }
        if (MODE == 1) dq_val = dq_in[s_];
    }
    // ---- stage this workgroup's heads rows (contiguous span) ------------------------------------
    {
        const float4* src = (const float4*)(heads + s0 * ldh);
        const int n4 = ns * ldh / 4;  // ldh % 4 == 0 (checked on the host)
        if (!SPLITK) {
            for (int k = tid; k < n4; k += NH_THREADS) ((float4*)sh_in)[k] = src[k];
        } else {
            // split-K input: every slab piece this thread needs — its float4 of the heads rows and its sample's partial
            // V'(s') — is requested before the first use, then added in slab order
            const int64_t sv = (live_ ? s_ : s0) * ldv;
            constexpr int NS = SPLITK ? NSLAB : 1;
            // V'(s') pieces first, then the rows' pieces; the scheduling barrier keeps every load in front of the first
            // add (left alone, the scheduler kept ~12 loads in flight to save registers: 3 round trips instead of 1)
            float pv[NS];
#pragma unroll
            for (int j = 0; j < NS; ++j) pv[j] = v_next[sv + j * vn_slab_stride];
            for (int k0 = 0; k0 < n4; k0 += NH_THREADS) {
                const int k = (k0 + tid < n4) ? k0 + tid : 0;
                float4 p[NS];
#pragma unroll
                for (int j = 0; j < NS; ++j) p[j] = ((const float4*)(heads + s0 * ldh + j * slab_stride))[k];
                __builtin_amdgcn_sched_barrier(0);
                float4 acc = p[0];
#pragma unroll
                for (int j = 1; j < NS; ++j) { acc.x += p[j].x; acc.y += p[j].y; acc.z += p[j].z; acc.w += p[j].w; }
                if (k0 + tid < n4) ((float4*)sh_in)[k0 + tid] = acc;
            }
            float v = pv[0];
#pragma unroll
            for (int j = 1; j < NS; ++j) v += pv[j];
            if (MODE == 2 && live_ && i_ == 0) vnext_val = v;
        }
        if (MODE != 0) {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k = tid; k < n4; k += NH_THREADS) ((float4*)sh_out)[k] = z;
        }
    }
    __syncthreads();
    naf_head_body<PMODE, MODE, NH_THREADS>(sh_in, sh_out, sh_L, sh_red, ldh, u_val, r_val, vnext_val, dq_val, gamma, q_out, mu_out,
                               loss_partials, B, A, s0, ns);
    if (MODE == 0) return;
    {
        float4* dst = (float4*)(d_heads + s0 * ldh);
        const int n4 = ns * ldh / 4;
        for (int k = tid; k < n4; k += NH_THREADS) dst[k] = ((const float4*)sh_out)[k];
    }
}

// 9 <= A <= 64: one sample per 16- / 32- / 64-lane group (csrc/naf_head_wide.hip)
int naf_head_wide_launch(int mode, const float* heads, int ldh, const float* u, int ldu, const float* r, int ldr, const float* v_next,
                         int ldv, const float* dq, float gamma, float* q_out, float* mu_out, float* d_heads, float* loss_partials,
                         int B, int A, int p_mode, hipStream_t st);
int naf_act_noise_wide_launch(const float* heads, int ldh, float* action_out, uint64_t seed, const uint64_t* counter_dev,
                              uint64_t counter_off, float noise_scale, int E, int A, int p_mode, hipStream_t st);
// (heads rows of A + A (A + 1) / 2 + 1 floats rounded up to 16: 160 at 16 joints, 576 at 32, 2160 at 64)
static int head_wide_max_ldh(int A) { return A <= 16 ? 160 : (A <= 32 ? 576 : 2160); }

static int head_args_ok(const float* heads, int ldh, const float* u, int ldu, int B, int A, int p_mode) {
    if (!heads || !u || B <= 0 || A <= 0 || A > NAF_MAX_A_WIDE) return 0;
    if (p_mode != NAF_P_HADAMARD && p_mode != NAF_P_MATMUL) return 0;
    if (ldh < A + A * (A + 1) / 2 + 1 || ldh > (A > NAF_MAX_A ? head_wide_max_ldh(A) : HEAD_MAX_LDH) || (ldh & 3) != 0) return 0;
    if (((uintptr_t)heads & 15) != 0 || ldu < A) return 0;
    return 1;
}

#define HEAD_LAUNCH(PM, MD, ...)                                                                       \
    do {                                                                                               \
        if ((PM) == NAF_P_HADAMARD)                                                                    \
            naf_head_kernel<NAF_P_HADAMARD, MD><<<blocks, NH_THREADS, 0, st>>>(__VA_ARGS__);         \
        else                                                                                           \
            naf_head_kernel<NAF_P_MATMUL, MD><<<blocks, NH_THREADS, 0, st>>>(__VA_ARGS__);           \
    } while (0)

extern "C" int naf_head_fwd(const float* heads_pre, int ldh, const float* u, int ldu, float* q, float* mu_out, int B,
                            int A, int p_mode, void* stream) {
    if (!head_args_ok(heads_pre, ldh, u, ldu, B, A, p_mode) || !q) return NAF_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (A > NAF_MAX_A)
        return naf_head_wide_launch(0, heads_pre, ldh, u, ldu, nullptr, 0, nullptr, 0, nullptr, 0.f, q, mu_out, nullptr, nullptr, B, A, p_mode, st);
    int blocks = (B + NH_SPB - 1) / NH_SPB;
    HEAD_LAUNCH(p_mode, 0, heads_pre, ldh, u, ldu, nullptr, 0, nullptr, 0, nullptr, 0.f, q, mu_out, nullptr, nullptr, B, A,
                1, 0, 0);
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}

extern "C" int naf_head_bwd(const float* heads_pre, int ldh, const float* u, int ldu, const float* dq, float* d_heads,
                            int B, int A, int p_mode, void* stream) {
    if (!head_args_ok(heads_pre, ldh, u, ldu, B, A, p_mode) || !dq || !d_heads || ((uintptr_t)d_heads & 15) != 0)
        return NAF_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (A > NAF_MAX_A)
        return naf_head_wide_launch(1, heads_pre, ldh, u, ldu, nullptr, 0, nullptr, 0, dq, 0.f, nullptr, nullptr, d_heads, nullptr, B, A, p_mode, st);
    int blocks = (B + NH_SPB - 1) / NH_SPB;
    HEAD_LAUNCH(p_mode, 1, heads_pre, ldh, u, ldu, nullptr, 0, nullptr, 0, dq, 0.f, nullptr, nullptr, d_heads, nullptr, B, A,
                1, 0, 0);
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}

extern "C" int naf_head_fwd_bwd_mse(const float* heads_pre, int ldh, const float* u, int ldu, const float* r, int ldr,
                                    const float* v_next, int ldv, float gamma, float* q_out, float* d_heads,
                                    float* loss_partials, int B, int A, int p_mode, void* stream) {
    if (!head_args_ok(heads_pre, ldh, u, ldu, B, A, p_mode) || !r || !v_next || !d_heads ||
        ((uintptr_t)d_heads & 15) != 0 || ldr < 1 || ldv < 1)
        return NAF_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (A > NAF_MAX_A)
        return naf_head_wide_launch(2, heads_pre, ldh, u, ldu, r, ldr, v_next, ldv, nullptr, gamma, q_out, nullptr, d_heads, loss_partials, B,
                                    A, p_mode, st);
    int blocks = (B + NH_SPB - 1) / NH_SPB;
    HEAD_LAUNCH(p_mode, 2, heads_pre, ldh, u, ldu, r, ldr, v_next, ldv, nullptr, gamma, q_out, nullptr, d_heads,
                loss_partials, B, A, 1, 0, 0);
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}

extern "C" int naf_head_fwd_bwd_mse_splitk(const float* heads_partial, int64_t slab_stride, const float* vnext_partial,
                                           int n_slabs, int ldh,
                                           const float* u, int ldu, const float* r, int ldr, float gamma, float* q_out,
                                           float* d_heads, float* loss_partials, int B, int A, int p_mode,
                                           void* stream) {
    if (!head_args_ok(heads_partial, ldh, u, ldu, B, A, p_mode) || A > NAF_MAX_A || !r || !vnext_partial || !d_heads ||
        ((uintptr_t)d_heads & 15) != 0 || ldr < 1 || (n_slabs != 32 && n_slabs != 16 && n_slabs != 4) || (slab_stride & 3) != 0 ||
        slab_stride < (int64_t)B * ldh)
        return NAF_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    int blocks = (B + NH_SPB - 1) / NH_SPB;
#define HEAD_SPLITK(PM, NS)                                                                                       \
    naf_head_kernel<PM, 2, NS><<<blocks, NH_THREADS, 0, st>>>(heads_partial, ldh, u, ldu, r, ldr, vnext_partial, 1,    \
                                                               nullptr, gamma, q_out, nullptr, d_heads, loss_partials, \
                                                               B, A, n_slabs, slab_stride, (int64_t)B)
    if (p_mode == NAF_P_HADAMARD) {
        if (n_slabs == 32) HEAD_SPLITK(NAF_P_HADAMARD, 32);
        else if (n_slabs == 16) HEAD_SPLITK(NAF_P_HADAMARD, 16);
        else HEAD_SPLITK(NAF_P_HADAMARD, 4);
    } else {
        if (n_slabs == 32) HEAD_SPLITK(NAF_P_MATMUL, 32);
        else if (n_slabs == 16) HEAD_SPLITK(NAF_P_MATMUL, 16);
        else HEAD_SPLITK(NAF_P_MATMUL, 4);
    }
#undef HEAD_SPLITK
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}

// ------------------------------------------------------------------------------------------------
// exploration noise: action = clamp(mu + noise_scale * P^{-1/2} z). Hadamard: P = diag(L_ii^2) so the
// covariance inverse(P) is diag(exp(-2 tanh l_ii)); matmul: cov = (L L^T)^-1 = L^-T L^-1, sample x solves
// L^T x = z by back substitution over the 8-lane group.
// ------------------------------------------------------------------------------------------------
template <int PMODE>
__global__ __launch_bounds__(HEAD_THREADS) void naf_act_noise_kernel(const float* __restrict__ heads, int ldh,
                                                                     float* __restrict__ action_out, uint64_t seed,
                                                                     const uint64_t* __restrict__ counter_dev,
                                                                     uint64_t counter_off, float noise_scale, int E,
                                                                     int A) {
    __shared__ float sh_L[PMODE == NAF_P_MATMUL ? HEAD_SPB * 8 * LT_STRIDE : 1];
    const int tid = threadIdx.x;
    const int64_t s = (int64_t)blockIdx.x * HEAD_SPB + (tid >> 3);
    const uint64_t ctr = (counter_dev ? *counter_dev : 0ull) + counter_off;
    naf_act_noise_body<PMODE>(heads + s * ldh, sh_L + (tid >> 3) * 8 * LT_STRIDE, action_out, seed, ctr, noise_scale, s,
                              s < E, A, tid);
}

extern "C" int naf_act_noise(const float* heads_pre, int ldh, float* action_out, uint64_t seed,
                             const uint64_t* counter_dev, uint64_t counter_off, float noise_scale, int E, int A,
                             int p_mode, void* stream) {
    if (!heads_pre || !action_out || E <= 0 || A <= 0 || A > NAF_MAX_A_WIDE) return NAF_ERR_ARG;
    if (p_mode != NAF_P_HADAMARD && p_mode != NAF_P_MATMUL) return NAF_ERR_ARG;
    if (ldh < A + A * (A + 1) / 2 + 1) return NAF_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (A > NAF_MAX_A)
        return naf_act_noise_wide_launch(heads_pre, ldh, action_out, seed, counter_dev, counter_off, noise_scale, E, A, p_mode, st);
    int blocks = (E + HEAD_SPB - 1) / HEAD_SPB;
    if (p_mode == NAF_P_HADAMARD)
        naf_act_noise_kernel<NAF_P_HADAMARD><<<blocks, HEAD_THREADS, 0, st>>>(heads_pre, ldh, action_out, seed, counter_dev,
                                                                             counter_off, noise_scale, E, A);
    else
        naf_act_noise_kernel<NAF_P_MATMUL><<<blocks, HEAD_THREADS, 0, st>>>(heads_pre, ldh, action_out, seed, counter_dev,
                                                                           counter_off, noise_scale, E, A);
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}
