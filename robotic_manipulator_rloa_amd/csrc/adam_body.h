// clip + Adam + Polyak on the flat parameter buffers as DEVICE code shared by three launches (gfx950):
//   adam_polyak_kernel (optim.hip)           the update as a launch of its own (naf_adam_polyak_fused)
//   bb_layer1_kernel<.., ADAM> (big_batch.hip)  the DEFERRED update of the previous learn(): extra workgroups of the NEXT
//                                            update's first launch step everything behind the layer-1 segment, while the layer-1
//                                            workgroups evaluate the layer-1 parameters they read AS THE UPDATE WILL LEAVE THEM
//                                            (adam_fly_*: nothing of that segment is written in that launch — its row blocks
//                                            and the two nets' workgroups all read the same old values)
//   bb_linear_stats*_kernel<ADAM>            extra workgroups of the second launch then step the layer-1 segment in place
// One code path (adam_one) for the stored and for the on-the-fly value, so both are the same bits. Replaces
// naf_algorithm.py:209-210 (clip_grad_norm_(params, 1); optimizer.step()) and :217-226 (soft_update).
#pragma once
#include "common.h"
#include "../../include/naf_hip.h"

#define ADAM_THREADS 256

struct AdamArgs {
    float* theta;              // main parameters (flat)
    const float* g;
    float* m;
    float* v;
    float* target;             // nullable: no Polyak
    const float* partials;     // sums of squares covering every gradient element once
    int n_partials;
    float max_norm, lr, beta1, beta2, eps, tau, one_minus_tau;
    const int32_t* step_dev;   // optimizer step count t = *step_dev (already advanced by the launch that finalised the gradient)
    float inv_world;
    // nullable: 2 x 4 floats of device scratch {step_size, inv_bc2_sqrt, t (as bits), 0}, slot = t & 1. The bias corrections are two double-precision
    // powers, a division and a square root on ONE lane — 0.3 us on the critical path of every workgroup that derives the step's
    // scalars. The workgroup that steps the layer-1 segment from the second launch (adam_block with lo4 == 0, off every critical
    // path) leaves the NEXT step's here; a reader takes them when the tag is its step number and computes them itself otherwise.
    float* bc;
};

struct AdamScalars {
    float clip_scale;   // inv_world * min(1, max_norm / (total_norm + 1e-6))
    float step_size;    // lr / (1 - beta1^t)
    float inv_bc2_sqrt; // 1 / sqrt(1 - beta2^t)
    int skip;           // the norm partials carry the poison of a timed-out gradient exchange: leave every buffer as it is
};

// b^t for integer t >= 0 by square-and-multiply in double: ~2 log2(t) multiplies instead of the libm pow() call
__device__ static inline double adam_ipow(double b, int t) {
    double r = 1.0;
    while (t > 0) {
        if (t & 1) r *= b;
        b *= b;
        t >>= 1;
    }
    return r;
}

// FP contraction is switched off for the update formulas so that tau*a + (1-tau)*b rounds like the reference's two
// multiplies and one add.
// (tg by reference + a flag, not a nullable pointer: the address of a register value sent the kernel's locals to scratch)
__device__ __forceinline__ static void adam_one(float& th, float gr, float& m, float& v, float& tg, bool has_tg, const AdamScalars& sc,
                                                float beta1, float beta2, float eps, float tau, float one_minus_tau) {
#pragma clang fp contract(off)
    const float gs = gr * sc.clip_scale;
    m = m + (gs - m) * (1.0f - beta1);                 // exp_avg.lerp_(grad, 1 - beta1)
    v = v * beta2 + ((1.0f - beta2) * gs) * gs;        // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    const float denom = sqrtf(v) * sc.inv_bc2_sqrt + eps;
    th = th - sc.step_size * (m / denom);              // param.addcdiv_(exp_avg, denom, value=-step_size)
    if (has_tg) tg = tau * th + one_minus_tau * tg;    // soft_update with the freshly stepped main weights
}

// The norm partials (NAF_MAX_NORM_PARTIALS of them, 64 per trip of wave 0) and the step count, requested by every thread
// branch-free so that they fly with whatever else the caller has in flight.
struct AdamPrefetch {
    float pr[NAF_MAX_NORM_PARTIALS / 64];
    int t;
};
__device__ __forceinline__ static AdamPrefetch adam_prefetch(const AdamArgs& A, int tid) {
    AdamPrefetch p;
    p.t = *A.step_dev;                       // (uniform: a scalar load)
#pragma unroll
    for (int j = 0; j < NAF_MAX_NORM_PARTIALS / 64; ++j) {
        const int k = (tid & 63) + 64 * j;
        p.pr[j] = A.partials[k < A.n_partials ? k : 0];
    }
    return p;
}
__device__ static inline void adam_bias_corrections(const AdamArgs& A, int t, float* step_size, float* inv_bc2_sqrt) {
    const double bc1 = 1.0 - adam_ipow((double)A.beta1, t);
    const double bc2 = 1.0 - adam_ipow((double)A.beta2, t);
    *step_size = (float)((double)A.lr / bc1);
    *inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
}
// Every workgroup re-derives the same scalars from the same partials in the same order: the first wave takes the partials 64
// at a time and folds them with the fixed-order lane sums; the second wave's first lane works out the bias corrections (double
// precision, as torch computes them on the host) meanwhile. The caller puts a workgroup barrier behind it (>= 128 threads).
__device__ __forceinline__ static void adam_derive(const AdamArgs& A, const AdamPrefetch& p, AdamScalars* sh, int tid) {
#pragma clang fp contract(off)
    if (tid < 64) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NAF_MAX_NORM_PARTIALS / 64; ++j) s += (tid + 64 * j < A.n_partials) ? p.pr[j] : 0.f;
        // more partials than were prefetched (flat buffers beyond 1M parameters): the rest in the same lane-major order
        for (int k = tid + NAF_MAX_NORM_PARTIALS; k < A.n_partials; k += 64) s += A.partials[k];
        s = naf_sum64(s);
        if (tid == 0) {
            // a sum of squares is never negative: -inf is what xgmi_allreduce_kernel leaves when a peer's contribution
            // did not arrive in time (csrc/xgmi_reduce.hip) — the update is then skipped on this rank, whole
            sh->skip = s < 0.f;
            const float total_norm = sqrtf(s) * A.inv_world;
            float clip = A.max_norm / (total_norm + 1e-6f);
            clip = clip > 1.0f ? 1.0f : clip;
            sh->clip_scale = clip * A.inv_world;
        }
    } else if (tid == 64) {
        bool have = false;
        if (A.bc) {                          // (two slots by the parity of t: the writer of t + 1 never touches what readers of t read)
            const float* b = A.bc + 4 * (p.t & 1);
            const float tagf = b[2];
            if (__builtin_bit_cast(int, tagf) == p.t) {
                sh->step_size = b[0];
                sh->inv_bc2_sqrt = b[1];
                have = true;
            }
        }
        if (!have) adam_bias_corrections(A, p.t, &sh->step_size, &sh->inv_bc2_sqrt);
    }
}

// One workgroup's share of the update of float4 range [lo4, hi4) of the flat buffers: workgroup `wg` of `nwg`, NT threads.
// The first trip's operands AND the norm partials are requested up front, branch-free (indices clamped, results masked
// later): the partial sum, the sqrt and the two double-precision powers then run under the latency of these loads instead of
// in front of it. Returns false when the update is skipped (poisoned norm).
// wt: write-through stores (sc0 sc1) — for the workgroups that ride on another kernel's launch and are done long before it
// ends: their 1.2 MB are then clean at the kernel boundary instead of waiting there for the write-back.
template <int NT>
__device__ static inline bool adam_block(const AdamArgs& A, size_t lo4, size_t hi4, int wg, int nwg, AdamScalars* sh, int tid,
                                         bool wt) {
    const size_t i0 = lo4 + (size_t)wg * NT + tid;
    const size_t i0c = i0 < hi4 ? i0 : (hi4 > lo4 ? hi4 - 1 : lo4);
    float4 th0 = ((float4*)A.theta)[i0c];
    float4 gr0 = ((const float4*)A.g)[i0c];
    float4 mm0 = ((float4*)A.m)[i0c];
    float4 vv0 = ((float4*)A.v)[i0c];
    float4 tg0 = ((float4*)(A.target ? A.target : A.theta))[i0c];
    const AdamPrefetch pf = adam_prefetch(A, tid);
    adam_derive(A, pf, sh, tid);
    __syncthreads();
    const AdamScalars sc = *sh;
    if (A.bc && lo4 == 0 && wg == 0 && tid == NT - 1) {      // the next step's bias corrections, for the next launch's readers
        float ss, ib;
        adam_bias_corrections(A, pf.t + 1, &ss, &ib);
        float* b = A.bc + 4 * ((pf.t + 1) & 1);
        b[0] = ss;
        b[1] = ib;
        b[2] = __builtin_bit_cast(float, pf.t + 1);
    }
    if (sc.skip) return false;
    float* target = A.target;
    for (size_t i = i0; i < hi4; i += (size_t)nwg * NT) {
        float4 th, gr, mm, vv, tg = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i == i0) {
            th = th0; gr = gr0; mm = mm0; vv = vv0; tg = tg0;
        } else {
            th = ((float4*)A.theta)[i];
            gr = ((const float4*)A.g)[i];
            mm = ((float4*)A.m)[i];
            vv = ((float4*)A.v)[i];
            if (target) tg = ((float4*)target)[i];
        }
        adam_one(th.x, gr.x, mm.x, vv.x, tg.x, target != nullptr, sc, A.beta1, A.beta2, A.eps, A.tau, A.one_minus_tau);
        adam_one(th.y, gr.y, mm.y, vv.y, tg.y, target != nullptr, sc, A.beta1, A.beta2, A.eps, A.tau, A.one_minus_tau);
        adam_one(th.z, gr.z, mm.z, vv.z, tg.z, target != nullptr, sc, A.beta1, A.beta2, A.eps, A.tau, A.one_minus_tau);
        adam_one(th.w, gr.w, mm.w, vv.w, tg.w, target != nullptr, sc, A.beta1, A.beta2, A.eps, A.tau, A.one_minus_tau);
        if (wt) {                               // (buffers of a learner are far below 2 GiB: 32-bit byte offsets)
            const unsigned off = (unsigned)(i * 16);
            naf_buf_st_f4(naf_buf(A.theta), off, 0, (naf_f32x4){th.x, th.y, th.z, th.w}, true);
            naf_buf_st_f4(naf_buf(A.m), off, 0, (naf_f32x4){mm.x, mm.y, mm.z, mm.w}, true);
            naf_buf_st_f4(naf_buf(A.v), off, 0, (naf_f32x4){vv.x, vv.y, vv.z, vv.w}, true);
            if (target) naf_buf_st_f4(naf_buf(target), off, 0, (naf_f32x4){tg.x, tg.y, tg.z, tg.w}, true);
        } else {
            ((float4*)A.theta)[i] = th;
            ((float4*)A.m)[i] = mm;
            ((float4*)A.v)[i] = vv;
            if (target) ((float4*)target)[i] = tg;
        }
    }
    return true;
}

// ---- the value a parameter WILL have after the pending update, without writing anything ---------------------------------
// o = float offset of the parameter inside the flat buffers; `cur` = what the reading network holds there now: the main
// network's theta for net 0, the target's for net 1 (which also needs the main network's old value to step it first).
struct AdamFly4 {
    naf_f32x4 g, m, v, thm;
};
struct AdamFly1 {
    float g, m, v, thm;
};
__device__ __forceinline__ static AdamFly4 adam_fly_load4(const AdamArgs& A, int64_t o, bool is_target) {
    AdamFly4 p;
    p.g = *(const naf_f32x4*)(A.g + o);
    p.m = *(const naf_f32x4*)(A.m + o);
    p.v = *(const naf_f32x4*)(A.v + o);
    p.thm = *(const naf_f32x4*)(A.theta + (is_target ? o : 0));     // (net 0 has the value already: any valid address)
    return p;
}
__device__ __forceinline__ static AdamFly1 adam_fly_load1(const AdamArgs& A, int64_t o, bool is_target) {
    AdamFly1 p;
    p.g = A.g[o];
    p.m = A.m[o];
    p.v = A.v[o];
    p.thm = A.theta[is_target ? o : 0];
    return p;
}
__device__ __forceinline__ static float adam_fly_apply1(const AdamArgs& A, const AdamScalars& sc, const AdamFly1& p, float cur,
                                                        bool is_target) {
    if (sc.skip) return cur;
    float th = is_target ? p.thm : cur, m = p.m, v = p.v, tg = cur;
    adam_one(th, p.g, m, v, tg, is_target, sc, A.beta1, A.beta2, A.eps, A.tau, A.one_minus_tau);
    return is_target ? tg : th;
}
__device__ __forceinline__ static naf_f32x4 adam_fly_apply4(const AdamArgs& A, const AdamScalars& sc, const AdamFly4& p, naf_f32x4 cur,
                                                            bool is_target) {
    naf_f32x4 out;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const AdamFly1 s = {p.g[q], p.m[q], p.v[q], p.thm[q]};
        out[q] = adam_fly_apply1(A, sc, s, cur[q], is_target);
    }
    return out;
}

__host__ static inline bool adam_args_from(const naf_adam_args_t& s, AdamArgs& a) {
    if (!s.theta || !s.grad || !s.m || !s.v || !s.partials || !s.step_dev || s.n_partials <= 0) return false;
    if ((((uintptr_t)s.theta | (uintptr_t)s.grad | (uintptr_t)s.m | (uintptr_t)s.v | (uintptr_t)s.theta_target) & 15) != 0) return false;
    a.theta = s.theta; a.g = s.grad; a.m = s.m; a.v = s.v; a.target = s.theta_target;
    a.partials = s.partials; a.n_partials = s.n_partials;
    a.max_norm = s.max_norm; a.lr = s.lr; a.beta1 = s.beta1; a.beta2 = s.beta2; a.eps = s.eps;
    a.tau = s.tau; a.one_minus_tau = s.one_minus_tau; a.step_dev = s.step_dev; a.inv_world = s.inv_world;
    a.bc = s.bc;
    return true;
}
