// The arithmetic of NAFAgent.act's eval-mode forward (naf_algorithm.py:158-178 with naf_neural_network.py:76-87), shared by the
// two launches that run it — policy_act_kernel (csrc/policy_act.hip: one workgroup per state streams the weights) and
// adam_act_kernel (csrc/step_path.hip: the workgroups of the optimizer step multiply the weights they have just stepped) — with
// every multiply-add spelled out, so that both produce the same bits whatever the optimizer makes of the code around them.
#pragma once
#include "common.h"

#define ACT_MAX_S 32

typedef float act_f4 __attribute__((ext_vector_type(4)));

// BatchNorm1d in eval mode (running statistics) + ReLU on one pre-activation
__device__ __forceinline__ static float act_bn_relu(float z, float rm, float rv, float g, float be, float eps) {
#pragma clang fp contract(off)
    const float y = (z - rm) * (1.0f / sqrtf(rv + eps)) * g + be;
    return y > 0.f ? y : 0.f;
}
// layer 1, one output row: z = b + sum_k w[k] x[k], k ascending (w[k] = 0 beyond the state size)
__device__ __forceinline__ static float act_layer1_row(const float (&w)[ACT_MAX_S], const float* x, float b, float g, float be,
                                                       float rm, float rv, float eps) {
    float z = b;
#pragma unroll
    for (int k = 0; k < ACT_MAX_S; ++k) z = __builtin_fmaf(w[k], x[k], z);
    return act_bn_relu(z, rm, rv, g, be, eps);
}
// a lane's share of a 256-wide row: four consecutive inputs
__device__ __forceinline__ static float act_dot4(act_f4 w, act_f4 x) {
    return __builtin_fmaf(w.w, x.w, __builtin_fmaf(w.z, x.z, __builtin_fmaf(w.y, x.y, w.x * x.x)));
}
// ... and its share of the second 256 inputs of a 512-wide row (round 6: layer sizes up to 512), continuing the same chain
__device__ __forceinline__ static float act_dot4_acc(act_f4 w, act_f4 x, float acc) {
    return __builtin_fmaf(w.w, x.w, __builtin_fmaf(w.z, x.z, __builtin_fmaf(w.y, x.y, __builtin_fmaf(w.x, x.x, acc))));
}
// the 64 lanes' shares of ONE row: the xor tree over levels 1, 2, 4, 8, 16, 32 — the tree pa_fold32 (policy_act.hip) walks for 32
// rows at once — on DPP modifiers and permlane swaps (common.h, naf_sum64: bitwise the xor-tree result), not six ds_bpermute
// round trips (~100 cycles each: 0.3 us per row, and the heads are four to six rows per wave in a row)
__device__ __forceinline__ static float act_sum64(float p) { return naf_sum64(p); }
