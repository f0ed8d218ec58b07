// clip_grad_norm_ + Adam + Polyak soft update over ONE flat f32 parameter buffer, gfx950.
// Replaces naf_algorithm.py:209-210 (clip_grad_norm_(params, 1); optimizer.step()) and :217-226
// (soft_update: 14 x `target.copy_(tau*main + (1-tau)*target)`): ~60 torch dispatches -> 2 launches.
// HBM/L2-bound streaming: 16 B per lane per access, 36 B/param algorithmic (read theta,g,m,v,theta';
// write theta,m,v,theta'), 12 B/param for the standalone Polyak. No host sync: the clip factor and the
// step count are read from device memory.
#include "common.h"
#include "../../include/naf_hip.h"

// FP contraction is switched off for the update formulas so that tau*a + (1-tau)*b rounds like the
// reference's two multiplies and one add.
#pragma clang fp contract(off)

#define OPT_THREADS 256

__global__ __launch_bounds__(OPT_THREADS) void grad_norm_partials_kernel(const float* __restrict__ g, size_t n,
                                                                         float* __restrict__ partials,
                                                                         int32_t* step_dev) {
    // (the optimizer step count is advanced here, one launch ahead of its only reader)
    __shared__ float red[OPT_THREADS / 64];
    const size_t base = (size_t)blockIdx.x * NAF_NORM_CHUNK;
    float acc = 0.f;
    // NAF_NORM_CHUNK / (256*4) = 4 float4 per thread, all issued before use
    float4 v[NAF_NORM_CHUNK / (OPT_THREADS * 4)];
#pragma unroll
    for (int k = 0; k < NAF_NORM_CHUNK / (OPT_THREADS * 4); ++k) {
        size_t e = base + ((size_t)k * OPT_THREADS + threadIdx.x) * 4;
        if (e + 3 < n) v[k] = *(const float4*)(g + e);
        else {
            v[k].x = e + 0 < n ? g[e + 0] : 0.f;
            v[k].y = e + 1 < n ? g[e + 1] : 0.f;
            v[k].z = e + 2 < n ? g[e + 2] : 0.f;
            v[k].w = 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < NAF_NORM_CHUNK / (OPT_THREADS * 4); ++k)
        acc += v[k].x * v[k].x + v[k].y * v[k].y + v[k].z * v[k].z + v[k].w * v[k].w;
    acc = naf_sum64(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int k = 0; k < OPT_THREADS / 64; ++k) s += red[k];
        partials[blockIdx.x] = s;
        if (blockIdx.x == 0 && step_dev) *step_dev += 1;  // nobody reads the step count in this launch
    }
}

extern "C" int naf_grad_norm_partials(const float* g, size_t n, float* partials, int32_t* step_dev, void* stream) {
    if (!g || !partials || n == 0 || ((uintptr_t)g & 15) != 0) return NAF_ERR_ARG;
    int blocks = (int)((n + NAF_NORM_CHUNK - 1) / NAF_NORM_CHUNK);
    grad_norm_partials_kernel<<<blocks, OPT_THREADS, 0, (hipStream_t)stream>>>(g, n, partials, step_dev);
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}

// b^t for integer t >= 0 by square-and-multiply in double: ~2 log2(t) multiplies instead of the libm pow() call
__device__ static inline double ipow(double b, int t) {
    double r = 1.0;
    while (t > 0) {
        if (t & 1) r *= b;
        b *= b;
        t >>= 1;
    }
    return r;
}

struct AdamScalars {
    float clip_scale;   // inv_world * min(1, max_norm / (total_norm + 1e-6))
    float step_size;    // lr / (1 - beta1^t)
    float inv_bc2_sqrt; // 1 / sqrt(1 - beta2^t)
    int skip;           // the norm partials carry the poison of a timed-out gradient exchange: leave every buffer as it is
};

__device__ static inline void adam_one(float& th, float gr, float& m, float& v, float* tg, const AdamScalars& sc,
                                       float beta1, float beta2, float eps, float tau, float one_minus_tau) {
    const float gs = gr * sc.clip_scale;
    m = m + (gs - m) * (1.0f - beta1);                 // exp_avg.lerp_(grad, 1 - beta1)
    v = v * beta2 + ((1.0f - beta2) * gs) * gs;        // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    const float denom = sqrtf(v) * sc.inv_bc2_sqrt + eps;
    th = th - sc.step_size * (m / denom);              // param.addcdiv_(exp_avg, denom, value=-step_size)
    if (tg) *tg = tau * th + one_minus_tau * (*tg);    // soft_update with the freshly stepped main weights
}

NAF_TL_DECL(g_tl_opt);
NAF_TL_READER(naf_tl_read_opt, g_tl_opt)
__global__ __launch_bounds__(OPT_THREADS) void adam_polyak_kernel(float* __restrict__ theta, const float* __restrict__ g,
                                                                  float* __restrict__ m, float* __restrict__ v,
                                                                  float* __restrict__ target,
                                                                  const float* __restrict__ partials, int n_partials,
                                                                  float max_norm, float lr, float beta1, float beta2,
                                                                  float eps, float tau, float one_minus_tau,
                                                                  const int32_t* __restrict__ step_dev, float inv_world,
                                                                  size_t n) {
    __shared__ AdamScalars sh;
    NAF_TL(g_tl_opt, NAF_TL_ADAM, 0);
    const size_t n4 = n / 4;
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // first trip's operands AND the norm partials are requested up front, branch-free (indices clamped, results masked
    // later): the partial sum, the sqrt and the two double-precision powers of thread 0 then run under the latency of
    // these loads instead of in front of it. (With the loads inside `if`s the compiler waited for the operand loads at
    // the merge before it even issued the partials' loads: two serial round trips.)
    const size_t i0c = i0 < n4 ? i0 : (n4 ? n4 - 1 : 0);
    float4 th0 = ((float4*)theta)[i0c];
    float4 gr0 = ((const float4*)g)[i0c];
    float4 mm0 = ((float4*)m)[i0c];
    float4 vv0 = ((float4*)v)[i0c];
    float4 tg0 = ((float4*)(target ? target : theta))[i0c];
    const int t = *step_dev;                 // (uniform: a scalar load, in flight with the rest)
    float pr[NAF_MAX_NORM_PARTIALS / 64];
#pragma unroll
    for (int j = 0; j < NAF_MAX_NORM_PARTIALS / 64; ++j) {
        const int k = (int)(threadIdx.x & 63) + 64 * j;
        pr[j] = partials[k < n_partials ? k : 0];
    }
    if (threadIdx.x < 64) {
        // every workgroup re-derives the same scalars from the same partials in the same order: the first wave takes
        // the partials 64 at a time, folds them with xor shuffles
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NAF_MAX_NORM_PARTIALS / 64; ++j) s += ((int)threadIdx.x + 64 * j < n_partials) ? pr[j] : 0.f;
        // more partials than were prefetched (flat buffers beyond 1M parameters): the rest in the same lane-major order
        for (int k = (int)threadIdx.x + NAF_MAX_NORM_PARTIALS; k < n_partials; k += 64) s += partials[k];
        s = naf_sum64(s);
        if (threadIdx.x == 0) {
            // a sum of squares is never negative: -inf is what xgmi_allreduce_kernel leaves when a peer's contribution
            // did not arrive in time (csrc/xgmi_reduce.hip) — the update is then skipped on this rank, whole
            sh.skip = s < 0.f;
            const float total_norm = sqrtf(s) * inv_world;
            float clip = max_norm / (total_norm + 1e-6f);
            clip = clip > 1.0f ? 1.0f : clip;
            sh.clip_scale = clip * inv_world;
        }
    } else if (threadIdx.x == 64) {
        // the bias corrections (double precision, as torch computes them on the host) do not depend on the partials:
        // the second wave works them out while the first one folds the norm (0.5 us when one thread did both in turn)
        const double bc1 = 1.0 - ipow((double)beta1, t);
        const double bc2 = 1.0 - ipow((double)beta2, t);
        sh.step_size = (float)((double)lr / bc1);
        sh.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    }
    __syncthreads();
    NAF_TL(g_tl_opt, NAF_TL_ADAM, 1);
    const AdamScalars sc = sh;
    if (sc.skip) return;
    for (size_t i = i0; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 th, gr, mm, vv, tg = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i == i0) {
            th = th0; gr = gr0; mm = mm0; vv = vv0; tg = tg0;
        } else {
            th = ((float4*)theta)[i];
            gr = ((const float4*)g)[i];
            mm = ((float4*)m)[i];
            vv = ((float4*)v)[i];
            if (target) tg = ((float4*)target)[i];
        }
        adam_one(th.x, gr.x, mm.x, vv.x, target ? &tg.x : nullptr, sc, beta1, beta2, eps, tau, one_minus_tau);
        adam_one(th.y, gr.y, mm.y, vv.y, target ? &tg.y : nullptr, sc, beta1, beta2, eps, tau, one_minus_tau);
        adam_one(th.z, gr.z, mm.z, vv.z, target ? &tg.z : nullptr, sc, beta1, beta2, eps, tau, one_minus_tau);
        adam_one(th.w, gr.w, mm.w, vv.w, target ? &tg.w : nullptr, sc, beta1, beta2, eps, tau, one_minus_tau);
        ((float4*)theta)[i] = th;
        ((float4*)m)[i] = mm;
        ((float4*)v)[i] = vv;
        if (target) ((float4*)target)[i] = tg;
    }
    NAF_TL(g_tl_opt, NAF_TL_ADAM, 2);
    // tail (n % 4 elements)
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        size_t e = n4 * 4 + threadIdx.x;
        float th = theta[e], mm = m[e], vv = v[e];
        float tg = target ? target[e] : 0.f;
        adam_one(th, g[e], mm, vv, target ? &tg : nullptr, sc, beta1, beta2, eps, tau, one_minus_tau);
        theta[e] = th; m[e] = mm; v[e] = vv;
        if (target) target[e] = tg;
    }
}

extern "C" int naf_adam_polyak_fused(float* theta, const float* g, float* m, float* v, float* theta_target,
                                     const float* partials, int n_partials, float max_norm, float lr, float beta1,
                                     float beta2, float eps, float tau, float one_minus_tau, const int32_t* step_dev,
                                     float inv_world, size_t n, void* stream) {
    if (!theta || !g || !m || !v || !partials || !step_dev || n < 4 || n_partials <= 0) return NAF_ERR_ARG;
    if ((((uintptr_t)theta | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)theta_target) & 15) != 0)
        return NAF_ERR_ARG;
    size_t n4 = (n + 3) / 4;
    int blocks = (int)((n4 + OPT_THREADS - 1) / OPT_THREADS);
    if (blocks > 2048) blocks = 2048;
    adam_polyak_kernel<<<blocks, OPT_THREADS, 0, (hipStream_t)stream>>>(theta, g, m, v, theta_target, partials,
                                                                        n_partials, max_norm, lr, beta1, beta2, eps,
                                                                        tau, one_minus_tau, step_dev, inv_world, n);
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}

__global__ __launch_bounds__(OPT_THREADS) void polyak_kernel(float* __restrict__ target, const float* __restrict__ main_,
                                                             float tau, float one_minus_tau, size_t n) {
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 t = ((float4*)target)[i];
        const float4 p = ((const float4*)main_)[i];
        t.x = tau * p.x + one_minus_tau * t.x;
        t.y = tau * p.y + one_minus_tau * t.y;
        t.z = tau * p.z + one_minus_tau * t.z;
        t.w = tau * p.w + one_minus_tau * t.w;
        ((float4*)target)[i] = t;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        size_t e = n4 * 4 + threadIdx.x;
        target[e] = tau * main_[e] + one_minus_tau * target[e];
    }
}

extern "C" int naf_polyak_update(float* target, const float* main_, float tau, float one_minus_tau, size_t n,
                                 void* stream) {
    if (!target || !main_ || n == 0 || (((uintptr_t)target | (uintptr_t)main_) & 15) != 0) return NAF_ERR_ARG;
    size_t n4 = (n + 3) / 4;
    int blocks = (int)((n4 + OPT_THREADS - 1) / OPT_THREADS);
    if (blocks > 2048) blocks = 2048;
    polyak_kernel<<<blocks, OPT_THREADS, 0, (hipStream_t)stream>>>(target, main_, tau, one_minus_tau, n);
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}
