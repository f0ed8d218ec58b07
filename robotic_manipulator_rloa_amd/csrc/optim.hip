// clip_grad_norm_ + Adam + Polyak soft update over ONE flat f32 parameter buffer, gfx950.
// Replaces naf_algorithm.py:209-210 (clip_grad_norm_(params, 1); optimizer.step()) and :217-226
// (soft_update: 14 x `target.copy_(tau*main + (1-tau)*target)`): ~60 torch dispatches -> 2 launches.
// HBM/L2-bound streaming: 16 B per lane per access, 36 B/param algorithmic (read theta,g,m,v,theta';
// write theta,m,v,theta'), 12 B/param for the standalone Polyak. No host sync: the clip factor and the
// step count are read from device memory.
#include "common.h"
#include "../../include/naf_hip.h"
#include "adam_body.h"

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

NAF_TL_DECL(g_tl_opt);
NAF_TL_READER(naf_tl_read_opt, g_tl_opt)
// the update as a launch of its own: adam_block (adam_body.h) over the whole buffer
__global__ __launch_bounds__(OPT_THREADS) void adam_polyak_kernel(const AdamArgs A, size_t n) {
    __shared__ AdamScalars sh;
    NAF_TL(g_tl_opt, NAF_TL_ADAM, 0);
    const size_t n4 = n / 4;
    if (!adam_block<OPT_THREADS>(A, 0, n4, blockIdx.x, gridDim.x, &sh, threadIdx.x, false)) return;
    NAF_TL(g_tl_opt, NAF_TL_ADAM, 1);
    // tail (n % 4 elements)
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const AdamScalars sc = sh;
        size_t e = n4 * 4 + threadIdx.x;
        float th = A.theta[e], mm = A.m[e], vv = A.v[e];
        float tg = A.target ? A.target[e] : 0.f;
        adam_one(th, A.g[e], mm, vv, tg, A.target != nullptr, sc, A.beta1, A.beta2, A.eps, A.tau, A.one_minus_tau);
        A.theta[e] = th; A.m[e] = mm; A.v[e] = vv;
        if (A.target) A.target[e] = tg;
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
    AdamArgs A;
    A.theta = theta; A.g = g; A.m = m; A.v = v; A.target = theta_target; A.partials = partials; A.n_partials = n_partials;
    A.max_norm = max_norm; A.lr = lr; A.beta1 = beta1; A.beta2 = beta2; A.eps = eps; A.tau = tau; A.one_minus_tau = one_minus_tau;
    A.step_dev = step_dev; A.inv_world = inv_world;
    A.bc = nullptr;
    adam_polyak_kernel<<<blocks, OPT_THREADS, 0, (hipStream_t)stream>>>(A, n);
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
