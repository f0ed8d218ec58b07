// NAFAgent.act() for E states in ONE launch (gfx950): the eval-mode forward of the main network — Linear, BatchNorm
// with running statistics, ReLU, twice; the three head Linears — and the exploration noise, one workgroup per state.
// Replaces naf_algorithm.py:158-178 / naf_neural_network.py:76-87,119-121 for a batch of states: seven launches before
// (3 GEMMs, 2 BN kernels, noise, counter), and at one state per call (the reference's own loop) nearly all of act()'s
// device time was their launch boundaries.
//
// One state is a chain of three matrix-VECTOR products (330 KB of weights, L2-resident): they are streamed with
// coalesced 16-byte loads, a wave per group of output rows, every lane holding four consecutive inputs; the per-row
// partial sums of the 64 lanes are folded by a recursive-halving exchange (a lane keeps half of its values and sends
// the other half at each xor level), 32 values in 32 cross-lane moves instead of 192.
// The hidden width is the framework's fixed 256 (rl_framework.py:452): H is a compile-time constant here.
#include "act_body.h"
#include "head_body.h"

#define PA_H 256
#define PA_THREADS 512
#define PA_WAVES (PA_THREADS / 64)
#define PA_ROWS_PER_WAVE (PA_H / PA_WAVES)   // 32 output rows of layer 2 per wave
#define PA_MAX_S ACT_MAX_S

typedef float pa_f4 __attribute__((ext_vector_type(4)));

// v[0..31]: this lane's partial sums of 32 rows. Returns the sum over the 64 lanes of row `*row_out` (every lane ends
// with one finished row; lanes l and l ^ 32 hold the same one).
__device__ static inline float pa_fold32(float (&v)[32], int lane, int* row_out) {
    float a[16], b[8], c[4], d[2];
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8, b4 = lane & 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = (b0 ? v[16 + i] : v[i]) + __shfl_xor(b0 ? v[i] : v[16 + i], 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = (b1 ? a[8 + i] : a[i]) + __shfl_xor(b1 ? a[i] : a[8 + i], 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = (b2 ? b[4 + i] : b[i]) + __shfl_xor(b2 ? b[i] : b[4 + i], 4);
#pragma unroll
    for (int i = 0; i < 2; ++i) d[i] = (b3 ? c[2 + i] : c[i]) + __shfl_xor(b3 ? c[i] : c[2 + i], 8);
    float e = (b4 ? d[1] : d[0]) + __shfl_xor(b4 ? d[0] : d[1], 16);
    e += __shfl_xor(e, 32);
    *row_out = (b0 ? 16 : 0) + (b1 ? 8 : 0) + (b2 ? 4 : 0) + (b3 ? 2 : 0) + (b4 ? 1 : 0);
    return e;
}

// G: lanes of the state's group in the noise body — 8, or 16 for 9 .. 11 joints (round 6: NH up to 78 rows of Wh; the heads loop walks
// NH as it finds it, so the wider kernel differs in its LDS arrays and in the noise body's lane map only)
#define PA_MAX_NH_WIDE 80
#define PA_MAX_A_WIDE 11
template <int PMODE, int G>
__global__ __launch_bounds__(PA_THREADS) void policy_act_kernel(
    const float* __restrict__ obs, int ldobs, int S, const float* __restrict__ W1, const float* __restrict__ b1,
    const float* __restrict__ g1, const float* __restrict__ be1, const float* __restrict__ W2,
    const float* __restrict__ b2, const float* __restrict__ g2, const float* __restrict__ be2,
    const float* __restrict__ Wh, int ldw, int NH, const float* __restrict__ rm1, const float* __restrict__ rv1,
    const float* __restrict__ rm2, const float* __restrict__ rv2, float eps, float* __restrict__ heads_out, int ldh,
    float* __restrict__ action_out, uint64_t seed, uint64_t* __restrict__ counter_dev, uint32_t* __restrict__ ticket,
    float noise_scale, int E, int A) {
    __shared__ float sObs[PA_MAX_S];
    __shared__ __attribute__((aligned(16))) float sA1[PA_H];
    __shared__ __attribute__((aligned(16))) float sA2[PA_H];
    __shared__ float sHeads[G == 8 ? HEAD_MAX_LDH : PA_MAX_NH_WIDE];
    __shared__ float sL[PMODE == NAF_P_MATMUL ? G * (G + 1) : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t s = blockIdx.x;
    const uint64_t ctr = *counter_dev;                      // advanced by the LAST workgroup to finish, below

    // ---- layer 1: thread j owns output j (K = S <= 32: a row of W1 is 84 B at S = 21) --------------------------------
    if (tid < PA_MAX_S) sObs[tid] = tid < S ? obs[s * ldobs + tid] : 0.f;
    float w1[PA_MAX_S];
    float p1 = 0.f, q1 = 0.f, r1 = 0.f, t1 = 0.f, u1 = 0.f;
    if (tid < PA_H) {
#pragma unroll
        for (int k = 0; k < PA_MAX_S; ++k) w1[k] = k < S ? W1[(int64_t)tid * S + k] : 0.f;
        p1 = b1[tid]; q1 = g1[tid]; r1 = be1[tid]; t1 = rm1[tid]; u1 = rv1[tid];
    }
    // layer-2 rows of this wave and the BatchNorm parameters of the row this lane will finish: requested now, they do
    // not depend on layer 1
    const int rloc = ((lane & 1) ? 16 : 0) + ((lane & 2) ? 8 : 0) + ((lane & 4) ? 4 : 0) + ((lane & 8) ? 2 : 0) +
                     ((lane & 16) ? 1 : 0);
    const int row2 = wave * PA_ROWS_PER_WAVE + rloc;
    const float p2 = b2[row2], q2 = g2[row2], r2 = be2[row2], t2 = rm2[row2], u2 = rv2[row2];
    pa_f4 w2[PA_ROWS_PER_WAVE];
#pragma unroll
    for (int r = 0; r < PA_ROWS_PER_WAVE; ++r)
        w2[r] = *(const pa_f4*)(W2 + (int64_t)(wave * PA_ROWS_PER_WAVE + r) * PA_H + 4 * lane);
    __syncthreads();
    if (tid < PA_H) sA1[tid] = act_layer1_row(w1, sObs, p1, q1, r1, t1, u1, eps);
    __syncthreads();

    // ---- layer 2: wave w owns rows 32 w .. 32 w + 31, lane l the inputs 4 l .. 4 l + 3 --------------------------------
    {
        const pa_f4 x = *(const pa_f4*)(sA1 + 4 * lane);
        float part[PA_ROWS_PER_WAVE];
#pragma unroll
        for (int r = 0; r < PA_ROWS_PER_WAVE; ++r) part[r] = act_dot4(w2[r], x);
        int rfold;
        const float z = pa_fold32(part, lane, &rfold);      // rfold == rloc
        if (lane < 32) sA2[row2] = act_bn_relu(z + p2, t2, u2, q2, r2, eps);
    }
    __syncthreads();

    // ---- heads: NH <= 45 (G = 16: 78) rows of Wh[NHP][ldw]; column PA_H of Wh is the bias (the activations' constant-1 column) ------
    {
        const pa_f4 x = *(const pa_f4*)(sA2 + 4 * lane);
        for (int h = wave; h < NH; h += PA_WAVES) {
            const pa_f4 w = *(const pa_f4*)(Wh + (int64_t)h * ldw + 4 * lane);
            float p = act_sum64(act_dot4(w, x));
            if (lane == 0) {
                p += Wh[(int64_t)h * ldw + PA_H];
                sHeads[h] = p;
                if (heads_out) heads_out[s * ldh + h] = p;
            }
        }
    }
    __syncthreads();

    // ---- mu, exploration noise, clamp: the first G lanes are this state's group ------------------------------------------
    naf_act_noise_body<PMODE, G>(sHeads, sL, action_out, seed, ctr, noise_scale, s, tid < G && s < E, A, tid);

    // the noise stream moves on once every workgroup has read the counter: the last one to get here advances it
    if (tid == 0) {
        if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
            *ticket = 0;
            *counter_dev = ctr + 1;
        }
    }
}

// ---- layer size 512 (round 6: widths in (256, 512] are stored as 512, learner.NetLayout) --------------------------------------------
// The same launch on 512 hidden units: thread j owns output j of layer 1 (512 threads), a wave owns 64 rows of layer 2 — two rounds
// of the 32-row fold, each row's 512 inputs as two float4 per lane (inputs 4 l .. 4 l + 3 and 256 + 4 l .. 256 + 4 l + 3, one fmaf
// chain through both: act_dot4 -> act_dot4_acc) — and the heads rows likewise. adam_act_kernel<.., 512> (csrc/step_path.hip) runs the
// same arithmetic in the same order.
#define PA_H2 512
template <int PMODE, int G>
__global__ __launch_bounds__(PA_THREADS) void policy_act_512_kernel(
    const float* __restrict__ obs, int ldobs, int S, const float* __restrict__ W1, const float* __restrict__ b1,
    const float* __restrict__ g1, const float* __restrict__ be1, const float* __restrict__ W2,
    const float* __restrict__ b2, const float* __restrict__ g2, const float* __restrict__ be2,
    const float* __restrict__ Wh, int ldw, int NH, const float* __restrict__ rm1, const float* __restrict__ rv1,
    const float* __restrict__ rm2, const float* __restrict__ rv2, float eps, float* __restrict__ heads_out, int ldh,
    float* __restrict__ action_out, uint64_t seed, uint64_t* __restrict__ counter_dev, uint32_t* __restrict__ ticket,
    float noise_scale, int E, int A) {
    constexpr int H = PA_H2, RPW = H / PA_WAVES;             // 64 rows of layer 2 per wave
    __shared__ float sObs[PA_MAX_S];
    __shared__ __attribute__((aligned(16))) float sA1[H];
    __shared__ __attribute__((aligned(16))) float sA2[H];
    __shared__ float sHeads[G == 8 ? HEAD_MAX_LDH : PA_MAX_NH_WIDE];
    __shared__ float sL[PMODE == NAF_P_MATMUL ? G * (G + 1) : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t s = blockIdx.x;
    const uint64_t ctr = *counter_dev;
    if (tid < PA_MAX_S) sObs[tid] = tid < S ? obs[s * ldobs + tid] : 0.f;
    float w1[PA_MAX_S];
#pragma unroll
    for (int k = 0; k < PA_MAX_S; ++k) w1[k] = k < S ? W1[(int64_t)tid * S + k] : 0.f;
    const float p1 = b1[tid], q1 = g1[tid], r1 = be1[tid], t1 = rm1[tid], u1 = rv1[tid];
    __syncthreads();
    sA1[tid] = act_layer1_row(w1, sObs, p1, q1, r1, t1, u1, eps);
    __syncthreads();
    // ---- layer 2 ---------------------------------------------------------------------------------------------------------------
    {
        const int rloc = ((lane & 1) ? 16 : 0) + ((lane & 2) ? 8 : 0) + ((lane & 4) ? 4 : 0) + ((lane & 8) ? 2 : 0) + ((lane & 16) ? 1 : 0);
        const pa_f4 x0 = *(const pa_f4*)(sA1 + 4 * lane), x1 = *(const pa_f4*)(sA1 + 256 + 4 * lane);
        for (int rnd = 0; rnd < RPW / 32; ++rnd) {
            const int row0 = wave * RPW + 32 * rnd, row2 = row0 + rloc;
            const float p2 = b2[row2], q2 = g2[row2], r2 = be2[row2], t2 = rm2[row2], u2 = rv2[row2];
            float part[32];
            pa_f4 w[32];
#pragma unroll
            for (int r = 0; r < 32; ++r) w[r] = *(const pa_f4*)(W2 + (int64_t)(row0 + r) * H + 4 * lane);
#pragma unroll
            for (int r = 0; r < 32; ++r) part[r] = act_dot4(w[r], x0);
#pragma unroll
            for (int r = 0; r < 32; ++r) w[r] = *(const pa_f4*)(W2 + (int64_t)(row0 + r) * H + 256 + 4 * lane);
#pragma unroll
            for (int r = 0; r < 32; ++r) part[r] = act_dot4_acc(w[r], x1, part[r]);
            int rfold;
            const float z = pa_fold32(part, lane, &rfold);  // rfold == rloc
            if (lane < 32) sA2[row2] = act_bn_relu(z + p2, t2, u2, q2, r2, eps);
        }
    }
    __syncthreads();
    // ---- heads ------------------------------------------------------------------------------------------------------------------
    {
        const pa_f4 x0 = *(const pa_f4*)(sA2 + 4 * lane), x1 = *(const pa_f4*)(sA2 + 256 + 4 * lane);
        for (int h = wave; h < NH; h += PA_WAVES) {
            const pa_f4 w0 = *(const pa_f4*)(Wh + (int64_t)h * ldw + 4 * lane), w1h = *(const pa_f4*)(Wh + (int64_t)h * ldw + 256 + 4 * lane);
            float p = act_sum64(act_dot4_acc(w1h, x1, act_dot4(w0, x0)));
            if (lane == 0) {
                p += Wh[(int64_t)h * ldw + H];
                sHeads[h] = p;
                if (heads_out) heads_out[s * ldh + h] = p;
            }
        }
    }
    __syncthreads();
    naf_act_noise_body<PMODE, G>(sHeads, sL, action_out, seed, ctr, noise_scale, s, tid < G && s < E, A, tid);
    if (tid == 0) {
        if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
            *ticket = 0;
            *counter_dev = ctr + 1;
        }
    }
}

extern "C" int naf_policy_act(const float* obs, int ldobs, int S, const float* W1, const float* b1, const float* g1,
                              const float* be1, const float* W2, const float* b2, const float* g2, const float* be2,
                              const float* Wh, int ldw, int NH, const float* running_mean1, const float* running_var1,
                              const float* running_mean2, const float* running_var2, float eps, int H, float* heads_out,
                              int ldh, float* action_out, uint64_t seed, uint64_t* counter_dev, uint32_t* ticket,
                              float noise_scale, int E, int A, int p_mode, void* stream) {
    if (!obs || !W1 || !b1 || !g1 || !be1 || !W2 || !b2 || !g2 || !be2 || !Wh || !running_mean1 || !running_var1 ||
        !running_mean2 || !running_var2 || !action_out || !counter_dev || !ticket)
        return NAF_ERR_ARG;
    if ((H != PA_H && H != PA_H2) || S <= 0 || S > PA_MAX_S || ldobs < S || E <= 0 || A <= 0 || A > PA_MAX_A_WIDE) return NAF_ERR_ARG;
    if (NH != A + A * (A + 1) / 2 + 1 || NH > (A > NAF_MAX_A ? PA_MAX_NH_WIDE : HEAD_MAX_LDH) || ldw <= H || (ldw & 3) != 0) return NAF_ERR_ARG;
    if ((((uintptr_t)W2 | (uintptr_t)Wh) & 15) != 0 || (heads_out && ldh < NH)) return NAF_ERR_ARG;
    if (p_mode != NAF_P_HADAMARD && p_mode != NAF_P_MATMUL) return NAF_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
#define PA_GO(PM, GV)                                                                                                         \
    policy_act_kernel<PM, GV><<<E, PA_THREADS, 0, st>>>(obs, ldobs, S, W1, b1, g1, be1, W2, b2, g2, be2, Wh, ldw, NH, running_mean1, \
                                                       running_var1, running_mean2, running_var2, eps, heads_out, ldh, action_out,  \
                                                       seed, counter_dev, ticket, noise_scale, E, A)
#define PA_GO2(PM, GV)                                                                                                        \
    policy_act_512_kernel<PM, GV><<<E, PA_THREADS, 0, st>>>(obs, ldobs, S, W1, b1, g1, be1, W2, b2, g2, be2, Wh, ldw, NH, running_mean1, \
                                                           running_var1, running_mean2, running_var2, eps, heads_out, ldh, action_out,  \
                                                           seed, counter_dev, ticket, noise_scale, E, A)
    if (H == PA_H2) {
        if (p_mode == NAF_P_HADAMARD) {
            if (A > NAF_MAX_A) PA_GO2(NAF_P_HADAMARD, 16);
            else PA_GO2(NAF_P_HADAMARD, 8);
        } else {
            if (A > NAF_MAX_A) PA_GO2(NAF_P_MATMUL, 16);
            else PA_GO2(NAF_P_MATMUL, 8);
        }
    } else if (p_mode == NAF_P_HADAMARD) {
        if (A > NAF_MAX_A) PA_GO(NAF_P_HADAMARD, 16);
        else PA_GO(NAF_P_HADAMARD, 8);
    } else {
        if (A > NAF_MAX_A) PA_GO(NAF_P_MATMUL, 16);
        else PA_GO(NAF_P_MATMUL, 8);
    }
#undef PA_GO2
#undef PA_GO
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}
