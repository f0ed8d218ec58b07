// BatchNorm1d (+ Linear bias, + ReLU) around the trunk GEMMs, gfx950. Replaces
// `torch.relu(self.bnK(self.linear(x)))` minus the GEMM itself (naf_neural_network.py:76-78) in training
// mode (batch statistics, running-stat update: the reference never puts either net in eval() inside
// learn(), naf_algorithm.py:194-202), its autograd, and the eval-mode form used by act() (:170-173).
//
// Tile: a workgroup owns 32 feature columns x ALL B rows, so batch statistics never leave the workgroup
// (no atomics, no second launch, fixed summation order). 32 x 32 threads: lane tx = column, ty = row
// phase; a wave covers two rows x 32 columns = two 128-B lines per load instruction. Each thread keeps its
// rows in registers (RPT = ceil(B/32) values), so the matrix is read once. The matrices here are
// <= 2 MB and L2-resident between the producing GEMM and this kernel; the kernel is latency-bound, the
// design goal is one pass and one launch for both networks.
#include "bn_tile.h"
#include "../../include/naf_hip.h"

template <int RPT, int BN_TX, int BN_TY>
__global__ __launch_bounds__(BN_TX* BN_TY) void bn_relu_fwd_train_kernel(
    const float* __restrict__ g, int64_t g_net_stride, int ldg, const float* __restrict__ bias,
    const float* __restrict__ gamma, const float* __restrict__ beta, int64_t param_net_stride,
    float* __restrict__ running_mean, float* __restrict__ running_var, int64_t stat_net_stride, float* __restrict__ out,
    int64_t out_net_stride, int ldo, float* __restrict__ save_mean, float* __restrict__ save_invstd, int B, int H,
    float momentum, float eps) {
    __shared__ float red[BN_TX * BN_TY / 64][BN_TX + 1];
    __shared__ float redv[BN_TX * BN_TY / 64][BN_TX + 1];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int col = naf_xcd_tile(blockIdx.x, gridDim.x) * BN_TX + tx;
    const int net = blockIdx.y;
    const bool col_on = col < H;
    const float* gz = g + net * g_net_stride;
    float* oz = out + net * out_net_stride;
    float x[RPT];
    float sum = 0.f;
    // unconditional loads (clamped at the edges, masked afterwards): `on ? load + b : 0` made the compiler branch around
    // every row's load and wait at each merge — RPT serial round trips instead of one
    const int colc = col_on ? col : H - 1;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int row = ty + k * BN_TY;
        x[k] = gz[(int64_t)(row < B ? row : B - 1) * ldg + colc];
    }
    const int64_t po = net * param_net_stride;
    const float b = (bias && col_on) ? bias[po + col] : 0.f;
    // every per-column scalar is requested up front: its latency hides under the matrix loads below
    const float gm = col_on ? gamma[po + col] : 0.f;
    const float bt = col_on ? beta[po + col] : 0.f;
    const int64_t so = net * stat_net_stride + col;
    const float rm_old = (ty == 0 && col_on) ? running_mean[so] : 0.f;
    const float rv_old = (ty == 0 && col_on) ? running_var[so] : 0.f;

#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        x[k] = (col_on && (ty + k * BN_TY) < B) ? x[k] + b : 0.f;
        sum += x[k];
    }
    const float mean = bn_col_reduce<BN_TX, BN_TY, true>(sum, red, tx, ty) / (float)B;
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        int row = ty + k * BN_TY;
        float dlt = (row < B) ? x[k] - mean : 0.f;
        ss += dlt * dlt;
    }
    const float var = bn_col_reduce<BN_TX, BN_TY, true>(ss, redv, tx, ty) / (float)B;  // biased: what normalises
    const float invstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        int row = ty + k * BN_TY;
        if (col_on && row < B) {
            float y = (x[k] - mean) * invstd * gm + bt;
            oz[(int64_t)row * ldo + col] = y > 0.f ? y : 0.f;
        }
    }
    if (ty == 0 && col_on) {
        const float unbiased = B > 1 ? var * ((float)B / (float)(B - 1)) : var;
        running_mean[so] = (1.0f - momentum) * rm_old + momentum * mean;
        running_var[so] = (1.0f - momentum) * rv_old + momentum * unbiased;
        save_mean[(int64_t)net * H + col] = mean;
        save_invstd[(int64_t)net * H + col] = invstd;
    }
}

template <int RPT, int BN_TX, int BN_TY>
__global__ __launch_bounds__(BN_TX* BN_TY) void bn_relu_bwd_kernel(
    const float* __restrict__ d_out, int ld_dout, const float* __restrict__ g, int ldg, const float* __restrict__ bias,
    const float* __restrict__ out, int ldo, const float* __restrict__ gamma, const float* __restrict__ save_mean,
    const float* __restrict__ save_invstd, float* __restrict__ d_z, int ldd, float* __restrict__ d_gamma,
    float* __restrict__ d_beta, float* __restrict__ d_bias, int B, int H) {
    __shared__ float red[BN_TX * BN_TY / 64][BN_TX + 1];
    __shared__ float red2[BN_TX * BN_TY / 64][BN_TX + 1];
    __shared__ float red3[BN_TX * BN_TY / 64][BN_TX + 1];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int col = naf_xcd_tile(blockIdx.x, gridDim.x) * BN_TX + tx;
    const bool col_on = col < H;
    const float b = (bias && col_on) ? bias[col] : 0.f;
    const float mean = col_on ? save_mean[col] : 0.f;
    const float invstd = col_on ? save_invstd[col] : 0.f;
    const float gm = col_on ? gamma[col] : 0.f;

    float xh[RPT], dy[RPT];
    float s_dy = 0.f, s_dyxh = 0.f;
    float zl[RPT], ol[RPT], dl[RPT];
    const int colc = col_on ? col : H - 1;       // unconditional loads first (see the forward kernel)
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int row = ty + k * BN_TY;
        const int64_t rowc = row < B ? row : B - 1;
        zl[k] = g[rowc * ldg + colc];
        ol[k] = out[rowc * ldo + colc];
        dl[k] = d_out[rowc * ld_dout + colc];
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        int row = ty + k * BN_TY;
        bool on = col_on && row < B;
        float z = on ? zl[k] + b : 0.f;
        float o = on ? ol[k] : 0.f;
        float dd = on ? dl[k] : 0.f;
        xh[k] = on ? (z - mean) * invstd : 0.f;
        dy[k] = o > 0.f ? dd : 0.f;  // ReLU mask from the forward's own output
        s_dy += dy[k];
        s_dyxh += dy[k] * xh[k];
    }
    float dbeta, dgamma;
    bn_col_reduce2<BN_TX, BN_TY, true>(s_dy, s_dyxh, red, red2, tx, ty, &dbeta, &dgamma);
    const float invB = 1.0f / (float)B;
    const float k1 = gm * invstd;
    float s_dz = 0.f;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        int row = ty + k * BN_TY;
        if (col_on && row < B) {
            float dz = k1 * (dy[k] - dbeta * invB - xh[k] * (dgamma * invB));
            d_z[(int64_t)row * ldd + col] = dz;
            s_dz += dz;
        }
    }
    const float dbias = bn_col_reduce<BN_TX, BN_TY, true>(s_dz, red3, tx, ty);  // Linear bias under a train-mode BN: ~0 up to rounding
    if (ty == 0 && col_on) {
        d_gamma[col] = dgamma;
        d_beta[col] = dbeta;
        if (d_bias) d_bias[col] = dbias;
    }
}

// ---- any batch size: the same tile with the rows STREAMED instead of held (B > 2048) ------------------------------------------------
// The reference takes any positive batch_size (rl_framework.py:186-189). Beyond 16 rows per thread the register-resident kernels
// above would spill, so these read the matrix once per pass — column sums; squared deviations from the mean (the two-pass variance
// of the kernels above, same order of the partial sums per thread); normalise — from L2 / the Infinity Cache (a 4096 x 256 f32
// matrix is 4 MB). Slower per row (three reads instead of one) and only ever used where nothing else fits.
template <int BN_TX, int BN_TY>
__global__ __launch_bounds__(BN_TX* BN_TY) void bn_relu_fwd_train_stream_kernel(
    const float* __restrict__ g, int64_t g_net_stride, int ldg, const float* __restrict__ bias,
    const float* __restrict__ gamma, const float* __restrict__ beta, int64_t param_net_stride,
    float* __restrict__ running_mean, float* __restrict__ running_var, int64_t stat_net_stride, float* __restrict__ out,
    int64_t out_net_stride, int ldo, float* __restrict__ save_mean, float* __restrict__ save_invstd, int B, int H,
    float momentum, float eps) {
    __shared__ float red[BN_TX * BN_TY / 64][BN_TX + 1];
    __shared__ float redv[BN_TX * BN_TY / 64][BN_TX + 1];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int col = naf_xcd_tile(blockIdx.x, gridDim.x) * BN_TX + tx;
    const int net = blockIdx.y;
    const bool col_on = col < H;
    const int colc = col_on ? col : H - 1;
    const float* gz = g + net * g_net_stride;
    float* oz = out + net * out_net_stride;
    const int64_t po = net * param_net_stride;
    const float b = (bias && col_on) ? bias[po + col] : 0.f;
    const float gm = col_on ? gamma[po + col] : 0.f;
    const float bt = col_on ? beta[po + col] : 0.f;
    const int64_t so = net * stat_net_stride + col;
    const float rm_old = (ty == 0 && col_on) ? running_mean[so] : 0.f;
    const float rv_old = (ty == 0 && col_on) ? running_var[so] : 0.f;
    float sum = 0.f;
    for (int row = ty; row < B; row += BN_TY) sum += col_on ? gz[(int64_t)row * ldg + colc] + b : 0.f;
    const float mean = bn_col_reduce<BN_TX, BN_TY, true>(sum, red, tx, ty) / (float)B;
    float ss = 0.f;
    for (int row = ty; row < B; row += BN_TY) {
        const float dlt = col_on ? (gz[(int64_t)row * ldg + colc] + b) - mean : 0.f;
        ss += dlt * dlt;
    }
    const float var = bn_col_reduce<BN_TX, BN_TY, true>(ss, redv, tx, ty) / (float)B;
    const float invstd = 1.0f / sqrtf(var + eps);
    if (col_on)
        for (int row = ty; row < B; row += BN_TY) {
            const float y = ((gz[(int64_t)row * ldg + col] + b) - mean) * invstd * gm + bt;
            oz[(int64_t)row * ldo + col] = y > 0.f ? y : 0.f;
        }
    if (ty == 0 && col_on) {
        const float unbiased = B > 1 ? var * ((float)B / (float)(B - 1)) : var;
        running_mean[so] = (1.0f - momentum) * rm_old + momentum * mean;
        running_var[so] = (1.0f - momentum) * rv_old + momentum * unbiased;
        save_mean[(int64_t)net * H + col] = mean;
        save_invstd[(int64_t)net * H + col] = invstd;
    }
}

template <int BN_TX, int BN_TY>
__global__ __launch_bounds__(BN_TX* BN_TY) void bn_relu_bwd_stream_kernel(
    const float* __restrict__ d_out, int ld_dout, const float* __restrict__ g, int ldg, const float* __restrict__ bias,
    const float* __restrict__ out, int ldo, const float* __restrict__ gamma, const float* __restrict__ save_mean,
    const float* __restrict__ save_invstd, float* __restrict__ d_z, int ldd, float* __restrict__ d_gamma,
    float* __restrict__ d_beta, float* __restrict__ d_bias, int B, int H) {
    __shared__ float red[BN_TX * BN_TY / 64][BN_TX + 1];
    __shared__ float red2[BN_TX * BN_TY / 64][BN_TX + 1];
    __shared__ float red3[BN_TX * BN_TY / 64][BN_TX + 1];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int col = naf_xcd_tile(blockIdx.x, gridDim.x) * BN_TX + tx;
    const bool col_on = col < H;
    const int colc = col_on ? col : H - 1;
    const float b = (bias && col_on) ? bias[col] : 0.f;
    const float mean = col_on ? save_mean[col] : 0.f;
    const float invstd = col_on ? save_invstd[col] : 0.f;
    const float gm = col_on ? gamma[col] : 0.f;
    float s_dy = 0.f, s_dyxh = 0.f;
    for (int row = ty; row < B; row += BN_TY) {
        const float z = g[(int64_t)row * ldg + colc] + b, o = out[(int64_t)row * ldo + colc], dd = d_out[(int64_t)row * ld_dout + colc];
        const float dy = (col_on && o > 0.f) ? dd : 0.f;
        s_dy += dy;
        s_dyxh += dy * ((z - mean) * invstd);
    }
    float dbeta, dgamma;
    bn_col_reduce2<BN_TX, BN_TY, true>(s_dy, s_dyxh, red, red2, tx, ty, &dbeta, &dgamma);
    const float invB = 1.0f / (float)B, k1 = gm * invstd;
    float s_dz = 0.f;
    if (col_on)
        for (int row = ty; row < B; row += BN_TY) {
            const float z = g[(int64_t)row * ldg + col] + b, o = out[(int64_t)row * ldo + col], dd = d_out[(int64_t)row * ld_dout + col];
            const float dy = o > 0.f ? dd : 0.f, xh = (z - mean) * invstd;
            const float dz = k1 * (dy - dbeta * invB - xh * (dgamma * invB));
            d_z[(int64_t)row * ldd + col] = dz;
            s_dz += dz;
        }
    const float dbias = bn_col_reduce<BN_TX, BN_TY, true>(s_dz, red3, tx, ty);
    if (ty == 0 && col_on) {
        d_gamma[col] = dgamma;
        d_beta[col] = dbeta;
        if (d_bias) d_bias[col] = dbias;
    }
}

__global__ __launch_bounds__(256) void bn_relu_fwd_eval_kernel(const float* __restrict__ g, int ldg,
                                                               const float* __restrict__ bias,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta,
                                                               const float* __restrict__ running_mean,
                                                               const float* __restrict__ running_var,
                                                               float* __restrict__ out, int ldo, int B, int H, float eps) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < (int64_t)B * H;
         e += (int64_t)gridDim.x * blockDim.x) {
        int row = (int)(e / H), col = (int)(e - (int64_t)row * H);
        float z = g[(int64_t)row * ldg + col] + (bias ? bias[col] : 0.f);
        float invstd = 1.0f / sqrtf(running_var[col] + eps);
        float y = (z - running_mean[col]) * invstd * gamma[col] + beta[col];
        out[(int64_t)row * ldo + col] = y > 0.f ? y : 0.f;
    }
}

// Tile: 8 feature columns x 64 row phases up to B = 512 (fwd 3.3 us / bwd 3.7 us per launch at B = 256; a 32 x 32 tile: 4.5 /
// 5.6 — picked with round 2's benchmarks/kernel_probe.py among twelve shapes), 8 x 128 beyond (B = 2048: 8.3 / 10.1 us vs 10.7 / 15.4 for
// 8 x 64). At most 16 rows per thread (B <= 2048): more rows per thread spill to scratch.
#define BN_LAUNCH_RPT(KERNEL, TXv, TYv, ...)                                                               \
    do {                                                                                                   \
        dim3 grid((H + TXv - 1) / TXv, nets_), block(TXv, TYv);                                            \
        int rpt = (B + TYv - 1) / TYv;                                                                     \
        if (rpt <= 2) KERNEL<2, TXv, TYv><<<grid, block, 0, st>>>(__VA_ARGS__);                            \
        else if (rpt <= 4) KERNEL<4, TXv, TYv><<<grid, block, 0, st>>>(__VA_ARGS__);                       \
        else if (rpt <= 8) KERNEL<8, TXv, TYv><<<grid, block, 0, st>>>(__VA_ARGS__);                       \
        else if (rpt <= 16) KERNEL<16, TXv, TYv><<<grid, block, 0, st>>>(__VA_ARGS__);                     \
        else return NAF_ERR_ARG;                                                                           \
    } while (0)

#define BN_DISPATCH(KERNEL, ...)                                                                           \
    do {                                                                                                   \
        if (B <= 512) BN_LAUNCH_RPT(KERNEL, 8, 64, __VA_ARGS__);                                           \
        else BN_LAUNCH_RPT(KERNEL, 8, 128, __VA_ARGS__);                                                   \
    } while (0)

#define BN_MAX_B (16 * 128)  // 128 row phases x up to 16 rows per thread

extern "C" int naf_bn_relu_fwd_train(const float* g, int64_t g_net_stride, int ldg, const float* bias,
                                     const float* gamma, const float* beta, int64_t param_net_stride,
                                     float* running_mean, float* running_var, int64_t stat_net_stride, float* out,
                                     int64_t out_net_stride, int ldo, float* save_mean, float* save_invstd, int B, int H,
                                     int nets, float momentum, float eps, void* stream) {
    if (!g || !gamma || !beta || !running_mean || !running_var || !out || !save_mean || !save_invstd) return NAF_ERR_ARG;
    if (B <= 0 || H <= 0 || nets <= 0 || ldg < H || ldo < H) return NAF_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int nets_ = nets;
    if (B > BN_MAX_B) {                         // any batch size: the streamed form
        dim3 grid((H + 7) / 8, nets), block(8, 128);
        bn_relu_fwd_train_stream_kernel<8, 128><<<grid, block, 0, st>>>(g, g_net_stride, ldg, bias, gamma, beta, param_net_stride,
                                                                          running_mean, running_var, stat_net_stride, out, out_net_stride,
                                                                          ldo, save_mean, save_invstd, B, H, momentum, eps);
        NAF_CHECK_LAUNCH();
        return NAF_OK;
    }
    BN_DISPATCH(bn_relu_fwd_train_kernel, g, g_net_stride, ldg, bias, gamma, beta, param_net_stride, running_mean,
                running_var, stat_net_stride, out, out_net_stride, ldo, save_mean, save_invstd, B, H, momentum, eps);
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}

extern "C" int naf_bn_relu_bwd(const float* d_out, int ld_dout, const float* g, int ldg, const float* bias,
                               const float* out, int ldo, const float* gamma, const float* save_mean,
                               const float* save_invstd, float* d_z, int ldd, float* d_gamma, float* d_beta,
                               float* d_bias, int B, int H, void* stream) {
    if (!d_out || !g || !out || !gamma || !save_mean || !save_invstd || !d_z || !d_gamma || !d_beta) return NAF_ERR_ARG;
    if (B <= 0 || H <= 0 || ld_dout < H || ldg < H || ldo < H || ldd < H) return NAF_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int nets_ = 1;
    if (B > BN_MAX_B) {
        dim3 grid((H + 7) / 8, 1), block(8, 128);
        bn_relu_bwd_stream_kernel<8, 128><<<grid, block, 0, st>>>(d_out, ld_dout, g, ldg, bias, out, ldo, gamma, save_mean, save_invstd,
                                                                   d_z, ldd, d_gamma, d_beta, d_bias, B, H);
        NAF_CHECK_LAUNCH();
        return NAF_OK;
    }
    BN_DISPATCH(bn_relu_bwd_kernel, d_out, ld_dout, g, ldg, bias, out, ldo, gamma, save_mean, save_invstd, d_z, ldd,
                d_gamma, d_beta, d_bias, B, H);
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}

extern "C" int naf_bn_relu_fwd_eval(const float* g, int ldg, const float* bias, const float* gamma, const float* beta,
                                    const float* running_mean, const float* running_var, float* out, int ldo, int B,
                                    int H, float eps, void* stream) {
    if (!g || !gamma || !beta || !running_mean || !running_var || !out) return NAF_ERR_ARG;
    if (B <= 0 || H <= 0 || ldg < H || ldo < H) return NAF_ERR_ARG;
    int64_t total = (int64_t)B * H;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    bn_relu_fwd_eval_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(g, ldg, bias, gamma, beta, running_mean,
                                                                      running_var, out, ldo, B, H, eps);
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}
