/*
 * naf_hip.h — C ABI of libnaf_hip.so: the MI355X (gfx950) NAF learn()/replay hot path.
 *
 * The reference (JavierMtz5/robotic_manipulator_rloa) is pure Python and has NO FFI/plugin
 * interface; its hot path is a chain of stock torch ops. Each entry point below replaces one such
 * chain, cited as file:line relative to the reference checkout. The Python host
 * (robotic_manipulator_rloa_amd/_lib.py) binds these with ctypes; INTEGRATION.md shows the stub a
 * reference maintainer would add.
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - Every data pointer is a DEVICE pointer owned by the caller (e.g. torch.Tensor.data_ptr()),
 *     f32 unless stated otherwise. Nothing is allocated after naf_replay_create().
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream). Every call only enqueues
 *     work on `stream` and returns; no call synchronises, so all of them are HIP-graph capturable.
 *   - Return value: 0 = NAF_OK, <0 = argument/state error (NAF_ERR_*), >0 = hipError_t.
 *   - A = action size, 1 <= A <= 8 for the fused kernels (every BASELINE config; up to 11 in naf_bb_layer2_head, naf_policy_act,
 *     naf_adam_polyak_act and the naf_step_* launches: one sample per 16-lane group), up to 64 for the replay ring and the stand-alone
 *     head / noise entry points (naf_head_*, naf_act_noise: one sample per 16- / 32- / 64-lane group beyond 8 / 16 / 32).  T = A(A+1)/2.  "heads row" = [mu_pre(A) | l_pre(T) | V(1)],
 *     row stride ldh >= A+T+1 floats; l_pre is the row-major lower triangle
 *     (0,0),(1,0),(1,1),(2,0)... exactly as torch.tril_indices orders it
 *     (naf_components/naf_neural_network.py:98-100).
 *   - "transition row" = [state(S) | action(A) | reward | 0-pad | next_state(S) | done | 0-pad]: next_state starts
 *     at naf_replay_row_off_next_state(S,A) = round_up(S+A+1, 4) floats (16-byte aligned, = 28 at S=21/A=6, 32 at
 *     S=23/A=7), done follows it; rows are padded to naf_replay_row_floats(S,A) floats (64 for A<=8: 256 B = two
 *     128-B lines) in the HBM ring; gathered minibatches keep the same offsets inside a row and drop the tail padding
 *     (row stride naf_replay_batch_row_floats(S,A) or anything up to the ring's).
 */
#ifndef NAF_HIP_H
#define NAF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NAF_OK 0
#define NAF_ERR_ARG (-1)
#define NAF_ERR_STATE (-2)

/* P = L (*) L^T elementwise — what the reference computes (naf_neural_network.py:104) */
#define NAF_P_HADAMARD 0
/* P = L @ L^T — textbook NAF (Gu et al. 2016) */
#define NAF_P_MATMUL 1

/* stored action is truncated toward zero on gather: the reference's `.long()` (utils/replay_buffer.py:60) */
#define NAF_ACTION_TRUNC_INT 0
#define NAF_ACTION_FLOAT 1

typedef struct naf_replay naf_replay_t;

#define NAF_XGMI_MAX_WORLD 8
#define NAF_XGMI_HANDLE_BYTES 64
/* what a kernel outside xgmi_reduce.hip needs to push part of the gradient early (naf_bn_relu_bwd_wgrad_push; the
 * one-shot gradient all-reduce is described with the naf_xgmi_* entry points below) */
typedef struct {
    void* peer_base[NAF_XGMI_MAX_WORLD]; /* every rank's slab as mapped in this process */
    uint64_t* ctrl;
    uint64_t data_off, n_pad;
    int rank, world;
    long long timeout_ticks;             /* bound of a wait on a peer's flag, 100-MHz ticks (naf_xgmi_set_timeout) */
    uint64_t* host_timeouts;             /* pinned host word a timed-out wait bumps (naf_xgmi_timeouts_nowait) */
} naf_xgmi_push_t;

/* ---- library ------------------------------------------------------------------------------ */
/* Bumped whenever a signature or a struct in this header changes; the ctypes host compares the library's answer with
 * the value in this header and refuses a mismatch (a stale .so would otherwise be called with wrong argument lists). */
#define NAF_HIP_ABI_VERSION 34
int naf_hip_abi_version(void);
/* "gfx950" — the only architecture this library carries code objects for */
const char* naf_hip_arch(void);
/* development aid: phase marks of the large-batch chain's kernels (csrc/common.h, NAF_TL_*; kernel_id 0..6 = bb_layer1,
 * bb_linear_stats, bb_layer2_head, bb_bn_bwd_stage2, gemm_bundle, bb_layer1_bwd_finish, adam_polyak, 7 = step_prep, 8 = adam_act). Copies
 * out[2][16] = 100 MHz wall-clock values left by the first and by the last workgroup of the kernel's most recent launch
 * (synchronises the device). kernel_id 1024 + i: entry (out[0][..]) and exit (out[1][..]) clocks of workgroups 16 i .. 16 i + 15
 * of the last gemm_bundle launch. NAF_ERR_STATE unless the library was built with -DNAF_TIMELINE (NAF_BUILD_DEFINES);
 * benchmarks/kernel_timeline.py is the reader. Not part of the data path. */
int naf_timeline_read(int kernel_id, long long* out);

/* Host-side half of a hand-over through DEVICE memory (no stream, no copy engine): memcpy of `bytes` bytes from host memory
 * straight into device memory — every device allocation is mapped for the CPU on this platform (large BAR) — followed by a store
 * fence, so the bytes have left the CPU before the caller launches the kernel that reads them. That kernel must load them with
 * system scope. naf_step_prep's src_row / n_word may point to such memory: its first dependent load is then a local-memory
 * latency instead of a PCIe round trip to pinned host memory. */
int naf_host_store_supported(int device); /* 1: the CPU can store into this device's allocations (large BAR); 0: not; ask first */
int naf_host_publish(void* dst_device, const void* src_host, size_t bytes);
/* the same followed by hipGraphLaunch(graph_exec, stream) — one call per timestep of the per-timestep path (bytes == 0: launch
 * only). graph_exec: a hipGraphExec_t (e.g. torch.cuda.CUDAGraph.raw_cuda_graph_exec()). */
int naf_host_publish_launch(void* dst_device, const void* src_host, size_t bytes, void* graph_exec, void* stream);
/* A device allocation of the library's own for such hand-overs (plain hipMalloc: CPU-mapped on a large-BAR device whatever the
 * framework allocator does — torch's expandable segments and memory pools need not be), and the proof that the hand-over works
 * on this machine: naf_host_store_selftest stores n distinct patterns of 65 words through the path's own memcpy + fence and has a
 * kernel read each back with the path's own loads (system scope) before the next is stored. Returns the number of patterns that
 * came back wrong (0: use the hand-over; > 0: do not — the caller falls back to pinned host memory and says so), < 0 on an error.
 * The hand-over carries NAFAgent.step()'s transition (naf_algorithm.py:129-142) to the graph of the timestep. */
int naf_host_store_alloc(size_t bytes, void** out_device_ptr);
int naf_host_store_free(void* device_ptr);
int naf_host_store_selftest(void* dst_device, int n_patterns, void* stream);

/* ---- replay buffer: HBM ring of transition rows ------------------------------------------ */
/* replaces ReplayBuffer.__init__ (utils/replay_buffer.py:16-30): deque(maxlen=buffer_size) */
int naf_replay_row_floats(int S, int A);
int naf_replay_row_off_next_state(int S, int A);
/* `rows` : capacity * row_floats f32 (caller-owned device memory)
 * `meta` : 8 x uint64 device words {head, size, total_added, sample_counter, 0,0,0, bad_index_count},
 *          zero-initialised by the caller. */
int naf_replay_create(uint64_t capacity, int S, int A, float* rows, uint64_t* meta, naf_replay_t** out);
int naf_replay_destroy(naf_replay_t* h);
/* replaces ReplayBuffer.__len__ (replay_buffer.py:69-75): current fill, read back from the device. The ONLY entry point
 * that synchronises `stream` (the Python host keeps its own count and never calls it on the hot path). */
int naf_replay_size(naf_replay_t* h, uint64_t* size_out /* host */, void* stream);
/* replaces ReplayBuffer.add (replay_buffer.py:32-45) for n transitions at once: FIFO append with
 * eviction of the oldest when full. `src_rows`: n packed transition rows on the device. n <= capacity. */
int naf_replay_add_batch(naf_replay_t* h, const float* src_rows, int n, void* stream);
/* The same for a launch that is captured ONCE and replayed on timesteps with and without a new transition (the head
 * node of NAFAgent.step's update graph, naf_algorithm.py:144 `self.memory.add(...)` followed by :147-156): the row count
 * is read by the kernel from `n_word` (device-visible memory, e.g. pinned host; clamped to [0, n_max]); 0 appends nothing
 * and leaves {head, size, total} alone. n_max rows must fit one workgroup (n_max * row_floats <= 4096). */
int naf_replay_add_counted(naf_replay_t* h, const float* src_rows, const int32_t* n_word, int n_max, void* stream);
/* replaces `random.sample(self.memory, k)` (replay_buffer.py:55): n_batches independent minibatches of
 * B deque positions (0 = oldest), uniform over the current size, without replacement inside a
 * minibatch when `without_replacement` != 0 (and size >= B). Philox4x32-10 keyed by `seed`; stream position
 * = *counter_dev + counter_off + minibatch (counter_dev may be NULL). idx: n_batches*B int32. */
int naf_replay_sample_indices(naf_replay_t* h, uint64_t seed, const uint64_t* counter_dev, uint64_t counter_off,
                              int32_t* idx, int B, int n_batches, int without_replacement, void* stream);
/* the same draw for minibatches beyond 4096 (the reference takes any positive batch_size: rl_framework.py:186-189): the duplicate
 * check's table lives in device memory — `scratch`: n_batches * naf_replay_sample_scratch_ints(B) int32, owned by the caller, no
 * initialisation needed — instead of one workgroup's LDS. Same rule, same indices (bit for bit what the LDS form would draw).
 * 1 <= B <= 1,048,576 (round 6: the table slot of an element has a word of its own; 16384 before). */
int naf_replay_sample_scratch_ints(int B);
int naf_replay_sample_indices_big(naf_replay_t* h, uint64_t seed, const uint64_t* counter_dev, uint64_t counter_off, int32_t* idx,
                                  int B, int n_batches, int without_replacement, int32_t* scratch, void* stream);
/* replaces the np.stack/vstack + from_numpy + .to(device) chain (replay_buffer.py:57-65):
 * out_rows[n][out_ld] = the leading out_ld floats of ring[deque position idx[i]], actions truncated when
 * action_mode == TRUNC_INT. out_ld: a multiple of 4 between round_up(used floats, 4) and naf_replay_row_floats;
 * naf_replay_batch_row_floats(S, A) is the minibatch row the learner kernels expect at least (52 floats at S=21/A=6,
 * 56 at S=23/A=7: no padding to whole 128-B lines on the output side). */
int naf_replay_batch_row_floats(int S, int A);
int naf_replay_gather_rows(naf_replay_t* h, const int32_t* idx, float* out_rows, int n, int out_ld, int action_mode,
                           void* stream);
/* same, to the reference's five separate tensors (states[n,S], actions[n,A], rewards[n], next_states[n,S],
 * dones[n]) — the ReplayBuffer.sample() return contract (replay_buffer.py:67). */
int naf_replay_gather_soa(naf_replay_t* h, const int32_t* idx, float* s, float* u, float* r, float* s2,
                          float* d, int n, int action_mode, void* stream);
/* *p += inc on the device (1 thread): advances sampler / noise stream counters inside a graph */
int naf_counter_add(uint64_t* p, uint64_t inc, void* stream);

/* ---- NAF head ------------------------------------------------------------------------------ */
/* replaces naf_neural_network.py:81-115 after the three head Linears: tanh(mu), tanh(l), tril scatter,
 * exp on the diagonal, P, Q = V - 1/2 d^T P d with d = u - mu.  u rows: u + i*ldu. q: B floats. */
int naf_head_fwd(const float* heads_pre, int ldh, const float* u, int ldu, float* q, float* mu_out /*nullable, B*A*/,
                 int B, int A, int p_mode, void* stream);
/* autograd of the above (implicit in loss.backward(), naf_algorithm.py:208): given dq[B] = dLoss/dQ,
 * d_heads[B][ldh] = dLoss/d(heads_pre) (pad columns written as 0). */
int naf_head_bwd(const float* heads_pre, int ldh, const float* u, int ldu, const float* dq, float* d_heads,
                 int B, int A, int p_mode, void* stream);
/* fused learn() middle (naf_algorithm.py:199-208): y = r + gamma * v_next; Q as above;
 * loss = mean((Q - y)^2); d_heads = dLoss/d(heads_pre).  r: r[i*ldr], v_next: v_next[i*ldv].
 * loss_partials[ceil(B/NAF_HEAD_SPB)]: per-workgroup sums of (Q-y)^2 / B, summed by the reader in index order
 * (bitwise reproducible). q_out nullable. */
#define NAF_HEAD_SPB 8
int naf_head_fwd_bwd_mse(const float* heads_pre, int ldh, const float* u, int ldu, const float* r, int ldr,
                         const float* v_next, int ldv, float gamma, float* q_out, float* d_heads,
                         float* loss_partials, int B, int A, int p_mode, void* stream);
/* the same, fed by the split-K partial heads of naf_bn_relu_fwd_heads_partial: heads_partial[n_slabs][B][ldh] (main
 * net; slabs slab_stride >= B*ldh floats apart — keep it off a power of two, the n_slabs pieces of one row are read
 * together and would otherwise share an L2 channel and set
 * net) and vnext_partial[n_slabs][B] (target net's V column) are summed in slab order while being staged; n_slabs in {4, 16, 32}. */
int naf_head_fwd_bwd_mse_splitk(const float* heads_partial, int64_t slab_stride, const float* vnext_partial, int n_slabs,
                                int ldh,
                                const float* u, int ldu, const float* r, int ldr, float gamma, float* q_out,
                                float* d_heads, float* loss_partials, int B, int A, int p_mode, void* stream);
/* replaces MultivariateNormal(mu, inverse(P)).sample() + clamp (naf_neural_network.py:119-121) for E
 * states: action = clamp(mu + noise_scale * P^{-1/2} z, -1, 1), z ~ N(0, I) from Philox
 * (seed, *counter_dev + counter_off, sample, lane). action_out: E x A. */
int naf_act_noise(const float* heads_pre, int ldh, float* action_out, uint64_t seed, const uint64_t* counter_dev,
                  uint64_t counter_off, float noise_scale, int E, int A, int p_mode, void* stream);

/* NAFAgent.act() for E states in ONE launch (naf_algorithm.py:158-178 with naf_neural_network.py:76-87,119-121):
 * eval-mode forward of the main net (Linear -> BatchNorm with running statistics -> ReLU, twice; the NH = A+T+1 head
 * rows of Wh[.][ldw], whose column H is the bias) for obs[E][ldobs], then naf_act_noise's mu / noise / clamp with the
 * same Philox stream (seed, *counter_dev, state, lane). *counter_dev is advanced by one when the launch is over (by
 * the last workgroup to finish: `ticket` is a zero-initialised uint32 the library uses for that). H must be 256 or 512,
 * S <= 32, A <= 11 (beyond 8 joints the state's group in the noise body is 16 lanes wide). heads_out (nullable): [E][ldh]
 * pre-activations. One workgroup per state. */
int naf_policy_act(const float* obs, int ldobs, int S, const float* W1, const float* b1, const float* g1, const float* be1,
                   const float* W2, const float* b2, const float* g2, const float* be2, const float* Wh, int ldw, int NH,
                   const float* running_mean1, const float* running_var1, const float* running_mean2,
                   const float* running_var2, float eps, int H, float* heads_out, int ldh, float* action_out,
                   uint64_t seed, uint64_t* counter_dev, uint32_t* ticket, float noise_scale, int E, int A, int p_mode,
                   void* stream);

/* ---- BatchNorm1d + ReLU around the trunk GEMMs ---------------------------------------------- */
/* replaces `torch.relu(self.bnK(linear(x)))` minus the GEMM (naf_neural_network.py:76-78) in TRAINING mode
 * for `nets` networks in one launch (net n uses pointer + n*stride): z = g + bias; batch mean / biased var;
 * y = gamma*(z-mean)*invstd + beta; out = max(y,0); running stats updated with momentum and the unbiased
 * variance (torch BatchNorm1d defaults, naf_neural_network.py:42,46). save_mean/save_invstd: [nets][H]. */
int naf_bn_relu_fwd_train(const float* g, int64_t g_net_stride, int ldg, const float* bias, const float* gamma,
                          const float* beta, int64_t param_net_stride, float* running_mean, float* running_var,
                          int64_t stat_net_stride, float* out, int64_t out_net_stride, int ldo, float* save_mean,
                          float* save_invstd, int B, int H, int nets, float momentum, float eps, void* stream);
/* eval mode (NAFAgent.act, naf_algorithm.py:170-173): running statistics, no update */
int naf_bn_relu_fwd_eval(const float* g, int ldg, const float* bias, const float* gamma, const float* beta,
                         const float* running_mean, const float* running_var, float* out, int ldo, int B, int H,
                         float eps, void* stream);
/* backward of bn+relu in training mode: given d_out (grad w.r.t. the ReLU output), the forward's g/bias,
 * out (ReLU mask), saved mean/invstd: d_z[B][ldd], d_gamma[H], d_beta[H], d_bias[H] (= column sums of d_z). */
int naf_bn_relu_bwd(const float* d_out, int ld_dout, const float* g, int ldg, const float* bias, const float* out,
                    int ldo, const float* gamma, const float* save_mean, const float* save_invstd, float* d_z,
                    int ldd, float* d_gamma, float* d_beta, float* d_bias, int B, int H, void* stream);

/* ---- trunk layers with their small GEMMs folded in (csrc/fused_layers.hip) --------------------------------- */
/* Linear(K <= 32) + BatchNorm1d(train) + ReLU for `nets` networks in one launch: replaces
 * `torch.relu(self.bn1(self.input_layer(input_)))` INCLUDING the GEMM (naf_neural_network.py:76).
 * x rows (x + net*x_net_stride + row*ldx) must be 16-B aligned with ldx % 4 == 0 and at least 24 (K <= 24) or 32
 * readable floats per row; W is row-major [H][K] (torch Linear.weight), net n at + n*param_net_stride like bias/
 * gamma/beta. */
int naf_linear_bn_relu_fwd_train(const float* x, int64_t x_net_stride, int ldx, int K, const float* W, const float* bias,
                                 const float* gamma, const float* beta, int64_t param_net_stride, float* running_mean,
                                 float* running_var, int64_t stat_net_stride, float* out, int64_t out_net_stride, int ldo,
                                 float* save_mean, float* save_invstd, int B, int H, int nets, float momentum, float eps,
                                 void* stream);
/* backward of the above for one network: d_gamma, d_beta, d_bias and d_W[H][K] = dZ^T X (dZ never leaves registers) */
/* sumsq_partials (nullable): [ceil(H/8)] per-workgroup sums of squares of every gradient this launch writes — the
 * first half of clip_grad_norm_ folded in (see naf_adam_polyak_fused); step_dev (nullable): *step_dev += 1, as
 * naf_grad_norm_partials does. */
int naf_bn_relu_bwd_wgrad(const float* d_out, int ld_dout, const float* x, int ldx, int K, const float* W,
                          const float* bias, const float* out, int ldo, const float* gamma, const float* save_mean,
                          const float* save_invstd, float* d_gamma, float* d_beta, float* d_bias, float* d_W,
                          float* sumsq_partials, int32_t* step_dev, int B, int H, void* stream);
/* naf_bn_relu_bwd_wgrad with, in the same launch, extra workgroups that push grad[push_lo, push_hi) — gradient segments
 * finished by EARLIER launches — to the data-parallel peers (NULL push = exactly naf_bn_relu_bwd_wgrad). */
int naf_bn_relu_bwd_wgrad_push(const float* d_out, int ld_dout, const float* x, int ldx, int K, const float* W,
                               const float* bias, const float* out, int ldo, const float* gamma, const float* save_mean,
                               const float* save_invstd, float* d_gamma, float* d_beta, float* d_bias, float* d_W,
                               float* sumsq_partials, int32_t* step_dev, int B, int H, const naf_xgmi_push_t* push,
                               const float* grad, size_t push_lo, size_t push_hi, void* stream);
/* feature columns per workgroup of the three kernels above and below (= entries per H in their sumsq_partials) */
int naf_fused_tile_cols(void);
/* d_out = d_heads[B][ldh] @ Wh[ldh][ldw] computed on the fly (ldh in {16,32,48}, pad columns zero), then the
 * ReLU/BN backward of naf_bn_relu_bwd: replaces the dA2 GEMM + bn_relu_bwd pair */
int naf_heads_bwd_bn_relu_bwd(const float* d_heads, int ldh, const float* Wh, int ldw, const float* g, int ldg,
                              const float* bias, const float* out, int ldo, const float* gamma, const float* save_mean,
                              const float* save_invstd, float* d_z, int ldd, float* d_gamma, float* d_beta, float* d_bias,
                              float* sumsq_partials /* nullable, [ceil(H / naf_fused_tile_cols())] */, int B, int H, void* stream);
/* layer 2 of BOTH nets (net 0 = main, net 1 = target; pointer + net*stride): bias + BatchNorm1d(train) + ReLU exactly
 * as naf_bn_relu_fwd_train, plus the heads Linears (naf_neural_network.py:81-87) split over K: workgroup w owns 8
 * feature columns and writes heads_partial[w][row][NHP] = A2[row][8w..8w+8) . Wh[:, 8w..8w+8)^T (+ column H of Wh, the
 * bias, in w = 0) for the main net (slab w starts at heads_partial + w*slab_stride floats), vnext_partial[w][row] = the
 * same for head column v_col of the target net.
 * n_slabs = H/8. H % 8 == 0, B <= 512, NHP in {16,32,48}. */
int naf_bn_relu_fwd_heads_partial(const float* g, int64_t g_net_stride, int ldg, const float* bias, const float* gamma,
                                  const float* beta, int64_t param_net_stride, float* running_mean, float* running_var,
                                  int64_t stat_net_stride, float* out, int64_t out_net_stride, int ldo, float* save_mean,
                                  float* save_invstd, const float* Wh, int64_t wh_net_stride, int ldw, int NHP, int v_col,
                                  float* heads_partial, int64_t slab_stride, float* vnext_partial, int B, int H,
                                  float momentum, float eps, void* stream);

/* ---- learn() at large batches: row-split kernels, two-stage batch statistics (csrc/big_batch.hip) --------------
 * For B > 512 (BASELINE configs[3]: B = 1024, configs[4]: B = 2048) the batch is cut into NAF_BB_ROWS-row blocks that
 * spread over the whole chip. BatchNorm1d's training-mode statistics (naf_neural_network.py:76-78 via torch) become two
 * stages: the producer of a pre-activation tile writes per block and column (sum, sum of squared deviations from the
 * BLOCK mean) into partials[net][B/64][H] (float2), every consumer folds the B/64 partials of its columns in block order
 * (Chan's formula; fixed order, no atomics). B % 64 == 0, 64 <= B <= 2048, H % 64 == 0 everywhere below. */
#define NAF_BB_ROWS 64
/* Layer 1 is linear in the minibatch rows, so its batch statistics (forward AND backward) follow from the first two
 * moments of the input columns: record (naf_bb_moments_floats(K) f32) = [Sx(KP) | C(KP x KP)], Sx = column sums, C = centred
 * second moments sum_r (x_j - m_j)(x_k - m_k), KP = 24 (K <= 24) or 32; accumulated in double. One launch for n_batches
 * minibatches (x + b*batch_stride) x nets inputs (+ net*x_net_stride): mom[(b*nets + net) * record]. */
int naf_bb_moments_floats(int K);
int naf_bb_moments(const float* x, int64_t batch_stride, int64_t x_net_stride, int ldx, int K, float* mom, int B, int n_batches,
                   int nets, void* stream);
/* The DEFERRED optimizer step (round 2): clip + Adam + Polyak of the PREVIOUS learn() (naf_algorithm.py:209-213) carried by the
 * first two launches of the NEXT one instead of a launch of its own — one launch and one launch boundary less per update in
 * a chain of updates. All fields as the arguments of naf_adam_polyak_fused; the flat buffers hold n floats, the layer-1
 * parameters [W1 | b1 | g1 | be1] of the main network are floats [0, l1_floats) of them (l1_floats % 4 == 0).
 *   naf_bb_layer1_adam:       `adam` != NULL: extra workgroups of the launch step floats [l1_floats, n) in place; the layer-1
 *                             workgroups read W / bias / gamma / beta (which must lie inside [theta, theta + l1_floats), the
 *                             target's param_net_stride floats behind in theta_target) and evaluate them as the step WILL
 *                             leave them — same code, same bits — without writing them.
 *   naf_bb_linear_stats_adam: `adam` != NULL: extra workgroups step floats [0, l1_floats) in place.
 * Both calls of one update get the same struct; gradient, partials and step count must stay untouched until both have run.
 * The caller ends a chain of updates with naf_adam_polyak_fused (learner.py: TrainChunk). adam == NULL: no step rides along. */
typedef struct {
    float* theta;
    const float* grad;
    float* m;
    float* v;
    float* theta_target;    /* nullable */
    const float* partials;
    int n_partials;
    float max_norm, lr, beta1, beta2, eps, tau, one_minus_tau;
    const int32_t* step_dev;
    float inv_world;
    int64_t n, l1_floats;
    float* bc;   /* nullable: 8 floats of device scratch (two slots by step parity); the workgroups that step the layer-1 segment (naf_bb_linear_stats_adam) leave the
                    NEXT step's bias corrections there, tagged with its number, for the next launch's readers */
} naf_adam_args_t;
/* layer 1 for `nets` networks, K = state size <= 32 (rows and W as in naf_linear_bn_relu_fwd_train): mean_c = b_c + w_c . m,
 * var_c = w_c^T C w_c / B from mom[net] (this minibatch's records), out = ReLU(BN(x W^T + b)), running statistics (by the
 * block-0 workgroups), save_mean / save_invstd [nets][H]; wc_out (nullable): [H][KP], row c = w_c C of net 0, for
 * naf_bb_layer1_bwd_finish. Replaces `self.bn1(self.input_layer(x))` + ReLU (naf_neural_network.py:76-77) for both networks. */
int naf_bb_layer1_adam(const float* x, int64_t x_net_stride, int ldx, int K, const float* W, const float* bias,
                       const float* gamma, const float* beta, int64_t param_net_stride, const float* mom, float* running_mean,
                       float* running_var, int64_t stat_net_stride, float* out, int64_t out_net_stride, int ldo,
                       float* save_mean, float* save_invstd, float* wc_out,
                       float* xhat_out /* nullable: [B][ldo], xhat of net 0 — for naf_gemm_l1bwd_t.xhat */, int B, int H, int nets,
                       float momentum, float eps, const naf_adam_args_t* adam /* nullable (HOST pointer, copied into the launch) */,
                       void* stream);
/* z[net] = a[net] W[net]^T + bias[net] (torch Linear, K = 256, N % 64 == 0) on f32 MFMA, 64 x 32 tiles (64 x 16 up to B = 512),
 * with the column statistics partials of every 64-row block written by the epilogue: replaces `self.hidden_layer(x)`
 * (naf_neural_network.py:78) and the statistics pass of bn2 for both networks. */
int naf_bb_linear_stats_adam(const float* a, int64_t a_net_stride, int lda, const float* W, const float* bias,
                             int64_t param_net_stride, float* z, int64_t z_net_stride, int ldz, float* partials, int B, int N,
                             int K, int nets, const naf_adam_args_t* adam /* nullable (HOST pointer) */, void* stream);
/* rows per workgroup of naf_bb_layer2_head (16) = rows per block of its partials_bw: size partials_bw for B / rows blocks and
 * tell the bundle's BatchNorm-backward prologue (naf_gemm_bn2bwd_t.npb) that many */
int naf_bb_layer2_head_rows(int B);
/* layer 2 from its pre-activations on: fold the statistics partials, normalise, ReLU (A2), the three head Linears, the NAF head
 * (naf_head_fwd_bwd_mse: Q, y = r + gamma V'(s'), MSE, d_heads) and the first half of layer 2's backward (dA2 = d_heads Wh, ReLU
 * mask, dY2, block sums) in ONE launch for H = 256: a workgroup owns 16 batch rows across all features of both nets, heads and
 * dA2 on f32 MFMA. Outputs: a2_out (main net's A2, ldo >= H; the target's is not needed again), running statistics, save_mean /
 * save_invstd [2][H], q_out[B], d_heads[B][NHP], loss_partials[B/rows], dy_out (dY2), partials_bw[B/rows][H] (float2: sum dy,
 * sum dy*xhat), rows = naf_bb_layer2_head_rows(B). u / r: the action and reward columns of the minibatch rows. Replaces
 * naf_neural_network.py:78-115 + naf_algorithm.py:199-208 and the first half of layer 2's BatchNorm backward. */
/* optional: the statistics of layer 2 folded ONCE per launch instead of by every workgroup. With more than 8 statistics blocks
 * (B > 512) the launch's first 16 workgroups fold 32 (net, column) pairs each and publish one 16-byte record per pair — (mean,
 * invstd, *epoch, biased variance) — to `records` ([2 H] x 4 floats, 16-B aligned, device scratch nothing else touches); the other
 * workgroups poll the records until they carry *epoch (same protocol, and the same way out of a long wait — the thread folds its pair
 * itself — as naf_gemm_bn2bwd_t below: *epoch must differ
 * from launch to launch — naf_bb_layer1_bwd_finish(fold_epoch) advances it once per update, so ONE launch per update may use a given
 * records buffer; `errors`: nullable pinned host word that counts those fallbacks). At smaller batches the argument is ignored. */
typedef struct naf_bb_stats_once {
    float* records;
    const int* epoch;
    uint64_t* errors;
    /* round 6 (ABI 31): H = 512 runs TWO workgroups per row block, one per 256-column half; their partial heads meet in `exchange`
     * (device scratch owned by the caller, 16-byte aligned, zero-initialised: naf_bb_layer2_head_exchange_floats(B, NHP) floats),
     * as records tagged with *epoch. Required (with records / epoch) at H = 512, ignored at H = 256. */
    float* exchange;
} naf_bb_stats_once_t;
int naf_bb_layer2_head_exchange_floats(int B, int NHP);
int naf_bb_layer2_head(const float* z, int64_t z_net_stride, int ldz, const float* gamma, const float* beta,
                       int64_t param_net_stride, const float* partials, float* running_mean, float* running_var,
                       int64_t stat_net_stride, float* a2_out, int ldo, float* save_mean, float* save_invstd, const float* Wh,
                       int64_t wh_net_stride, int ldw, int NHP, const float* u, int ldu, const float* r, int ldr, float gamma_td,
                       float* q_out, float* d_heads, float* loss_partials, float* dy_out, int ldd, float* partials_bw, int B,
                       int H, int A, int p_mode, float momentum, float eps, const naf_bb_stats_once_t* once /* nullable */,
                       void* stream);
/* backward of layer 1, row-split: the batch pass is the epilogue of the bundle's dA1 blocks (naf_gemm_l1bwd_t below), this is
 * the finish launch. With dz = k1 (dy - c1 - xhat c2): dW[c][k] = k1_c (P[c][k] - c1_c Sx[k] - c2_c invstd_c (w_c C)[k]),
 * P = dY^T X — the xhat term comes from the moments record, so the pass only produces dy = ReLU'(out) * d_out, its block sums
 * partials1[nb1][H] (float2: sum dy, sum dy*xhat) and the block shares p_slabs[nb1][H][KP] of P (KP = naf_bb_layer1_bwd_kp(K)).
 * finish: folds them in block order -> d_W[H][K], d_gamma, d_beta, d_bias = 0 (sum_r dz vanishes identically; the reference's
 * value is rounding noise that the train-mode BatchNorm cancels), d_bias2 (layer 2: nb = 0 writes the 0 that sum is identically —
 * the chain's stage 2 runs inside the bundle, naf_gemm_bn2bwd_t; nb > 0: dz2_col_partials[nb][H] block sums are added). K <= 32.
 * mom: the MAIN net's moments record (Sx); wc: [H][KP], row c = w_c C, left by the forward pass (naf_bb_layer1_adam, wc_out).
 * sumsq_partials (nullable): sums of squares of everything written here plus d_gamma2 / d_beta2 (then required: read, not
 * written), one entry per workgroup: naf_bb_layer1_bwd_finish_blocks(H) finish blocks, then the slab segments' (below);
 * step_dev (nullable): *step_dev += 1. */
int naf_bb_layer1_bwd_kp(int K);
/* segs (HOST array, n_segs <= 2, may be 0): split-K slabs of the bundle's weight gradients (naf_gemm_desc_t.k_split), added in
 * slab order by extra workgroups of the same launch: dst[i] = sum_s src[s * stride + i], i < n (n % 4 == 0, n_slabs <= 8);
 * their sums of squares follow the naf_bb_layer1_bwd_finish_blocks(H) entries of sumsq_partials, ceil(n / 1024) entries per
 * segment. */
typedef struct naf_bb_slab_seg {
    const float* src;
    float* dst;
    int64_t stride;
    int n, n_slabs;
} naf_bb_slab_seg_t;
int naf_bb_layer1_bwd_finish_blocks(int H);
int naf_bb_layer1_bwd_finish(const float* p_slabs, int K, const float* partials1, int nb1 /* blocks of partials1 / p_slabs:
                             M / 32 from the bundle's epilogue */, const float* dz2_col_partials, int nb,
                             const float* mom, const float* wc, const float* gamma, const float* save_invstd, float* d_W,
                             float* d_gamma, float* d_beta, float* d_bias, float* d_bias2, const float* d_gamma2,
                             const float* d_beta2, float* sumsq_partials, int32_t* step_dev, int B, int H,
                             const naf_bb_slab_seg_t* segs, int n_segs,
                             int* fold_epoch /* nullable: *fold_epoch += 1 (naf_gemm_bn2bwd_t.epoch of the bundle launch in front) */,
                             const naf_xgmi_push_t* push /* nullable (HOST pointer): data parallel over peer memory — the workgroups that add
                             the slab segments also store them into this rank's slot on every peer (naf_xgmi_push_desc); the all-reduce that
                             follows is then told those ranges went ahead (naf_xgmi_allreduce_sum_from2) */,
                             const float* grad_base /* with push: the flat gradient the segments' dst lie in */,
                             size_t merge_total /* 0: push only. n (the flat gradient's length; the two segments must be its last two
                             pieces): the WHOLE exchange happens in this launch — its last-arriving workgroup pushes what the segments do
                             not cover and raises the epoch flags, every workgroup that holds gradient elements waits for the peers' flags
                             (bounded: naf_xgmi_set_timeout) and leaves the rank-ordered sum over ranks in the flat gradient;
                             sumsq_partials (required) then hold the REDUCED gradient's sums of squares, one entry per workgroup PLUS ONE
                             (the ranges the segments do not cover, at index [workgroups]), -inf where a wait timed out; *step_dev += 1. No naf_xgmi_allreduce_* call follows (where clip_grad_norm_ sits in
                             the reference, naf_algorithm.py:207-210) */,
                             void* stream);

/* ---- several small f32 GEMMs in one launch (csrc/gemm_bundle.hip) ------------------------------------------- */
/* C[M][N] = op(A) op(B): A is [M][K] row-major (a_kmajor = 0) or stored transposed [K][M] (a_kmajor = 1), B is
 * [N][K] row-major (b_kmajor = 0; i.e. C = A B^T like torch Linear) or [K][N] (b_kmajor = 1). M, N, K multiples of 16;
 * A and B 16-byte aligned with lda, ldb multiples of 4 (panels are staged into LDS with 16-byte loads).
 * Replaces the dW2 / dA1 / dWh GEMMs of the backward pass (autograd of naf_neural_network.py:76-87) with one grid of
 * 32 x 32 output blocks computed on v_mfma_f32_16x16x4_f32. `descs` is a HOST array (copied into the launch). */
#define NAF_GEMM_BUNDLE_MAX 4
/* optional epilogue of a product whose C = dA1 (gradient w.r.t. the layer-1 activations, M = batch rows, N = layer-1 features;
 * M, N multiples of 32): the batch pass of layer 1's backward (see naf_bb_layer1_bwd_finish) on every 32 x 32 C block while it is in registers — partials
 * [M/32][N] (float2) and p_slabs [M/32][N][kp]; C may then be NULL (dA1 is not stored). */
typedef struct naf_gemm_l1bwd {
    const float* x;          /* minibatch rows of the net (state columns), ldx floats apart */
    const float* W;          /* W1 [N][K] */
    const float* bias;       /* b1 [N] */
    const float* a1;         /* layer-1 activations [M][lda1]: the ReLU mask */
    const float* save_mean;  /* [N] */
    const float* save_invstd;
    float* partials;
    float* p_slabs;
    int ldx, K, kp, lda1;
    /* nullable: xhat of layer 1 as naf_bb_layer1_adam left it ([M][lda1], with gamma / beta [N]): the epilogue then reads its tile
     * of it (the ReLU mask is fma(xhat, gamma, beta) > 0, the forward's own expression) instead of recomputing z = x W^T + b and
     * xhat from W, bias, a1, save_mean and save_invstd — on the critical path of small batches, where the launch is one round of
     * blocks, that recomputation was 0.6 us of an update; at large batches the extra B x N floats each way cost more than it. */
    const float* xhat;
    const float* gamma;
    const float* beta;
    /* rows of the M that are samples (0: all M). A batch that is not whole 16-row groups runs with M = the next multiple of 16 over
     * buffers of that many rows; the rows past `rows` (0 < rows <= M) carry nothing into partials / p_slabs, and x is read up
     * to row rows - 1 only. */
    int rows;
} naf_gemm_l1bwd_t;
/* optional prologue on the A operand of a product: A = dY2 (the ReLU-masked gradient w.r.t. layer 2's BatchNorm output, written by
 * naf_bb_layer2_head) is turned into dZ2 = k1 (dy - c1 - xhat c2) WHILE the panel is staged — the second stage of layer 2's
 * BatchNorm backward without a launch of its own and without dZ2 in memory. The block sums are folded ONCE per launch: the
 * launch's first H / 32 workgroups fold 32 columns each (npb <= 128) and publish one 16-byte record per column to cst; the k-major
 * fold also writes d_gamma / d_beta. The bias gradient of the Linear in front (sum_r dz) is identically zero under a train-mode
 * BatchNorm and is not produced: pass nb = 0 to naf_bb_layer1_bwd_finish, which then writes d_bias2 = 0.
 * Restrictions: H = 256, M and N multiples of 32, npb <= 128. */
typedef struct naf_gemm_bn2bwd {
    const float* z;          /* Z2, same shape and leading dimension as A */
    const float* partials;   /* float2 [npb][H]: (sum dy, sum dy*xhat) per row block (naf_bb_layer2_head's partials_bw) */
    const float* gamma;      /* [H] */
    const float* save_mean;
    const float* save_invstd;
    float* d_gamma;          /* [H] out */
    float* d_beta;
    int npb, B, H;
    /* cst: [H] x 4 floats, 16-B aligned, device scratch that nothing else touches — one record per column (k1 c1, invstd k1 c2,
     * *epoch, 0); the GEMM blocks poll the records of their columns until they carry *epoch. *epoch (device word) must differ from
     * launch to launch: naf_bb_layer1_bwd_finish(fold_epoch) advances it. */
    float* cst;
    const int* epoch;
    /* nullable: pinned HOST word (device-visible), a diagnostic. A poll lasts 20 us at most; a thread whose record has not come by
     * then (several processes share the GPU and the folding workgroup of its XCD is still queued) folds its column itself — the
     * same sums in the same order, so the result is the same bits either way — and adds 1 here, where the host sees it without
     * synchronising. No wait can expire into a wrong number. */
    uint64_t* errors;
} naf_gemm_bn2bwd_t;
typedef struct naf_gemm_desc {
    const float* A;
    const float* B;
    float* C;
    float* sumsq; /* nullable: [ceil(M/32)*ceil(N/32)] per-block sums of C^2 (gradient-norm partials) */
    int M, N, K, lda, ldb, ldc, a_kmajor, b_kmajor;
    int k_split;            /* 0 / 1: the whole K in one block. > 1: K cut into k_split ranges ((K / k_split) % 16 == 0), range s */
    int64_t c_split_stride; /* writing its partial product to C + s * c_split_stride floats (sumsq must be NULL); the consumer adds
                               the slabs in index order (naf_bb_layer1_bwd_finish) */
    const naf_gemm_l1bwd_t* epi; /* nullable (HOST pointer, copied into the launch) */
    const naf_gemm_bn2bwd_t* pro; /* nullable (HOST pointer, copied into the launch) */
} naf_gemm_desc_t;
int naf_gemm_bundle(const naf_gemm_desc_t* descs, int n, void* stream);

/* ---- clip + Adam + Polyak over one flat parameter buffer -------------------------------------- */
/* first half of clip_grad_norm_(params, 1) (naf_algorithm.py:209): partials[i] = sum of g^2 over chunk i of
 * NAF_NORM_CHUNK floats; n_partials = ceil(n / NAF_NORM_CHUNK). If step_dev != NULL also does *step_dev += 1
 * (the optimizer step count the following naf_adam_polyak_fused reads). */
#define NAF_NORM_CHUNK 4096
#define NAF_MAX_NORM_PARTIALS 256 /* naf_adam_polyak_fused prefetches this many partials; more are accepted */
int naf_grad_norm_partials(const float* g, size_t n, float* partials, int32_t* step_dev, void* stream);
/* second half of clip_grad_norm_ + Adam.step() (naf_algorithm.py:209-210) + soft_update (:217-226) in one
 * pass over {theta, g, m, v, theta_target}: 36 B/param. `partials`: sums of squares covering every gradient element once
 * (from naf_grad_norm_partials, or the sumsq outputs of the gradient-producing kernels). grad = g * inv_world * clip, clip =
 * min(1, max_norm / (inv_world*sqrt(sum partials) + 1e-6)); Adam with torch defaults' formulas and bias
 * correction at t = *step_dev; theta_target = tau*theta_new + one_minus_tau*theta_target.
 * theta_target may be NULL (no Polyak). */
int naf_adam_polyak_fused(float* theta, const float* g, float* m, float* v, float* theta_target,
                          const float* partials, int n_partials, float max_norm, float lr, float beta1, float beta2,
                          float eps, float tau, float one_minus_tau, const int32_t* step_dev, float inv_world, size_t n,
                          void* stream);
/* NAFAgent.soft_update (naf_algorithm.py:217-226): target = tau*main + one_minus_tau*target, 12 B/param */
int naf_polyak_update(float* target, const float* main, float tau, float one_minus_tau, size_t n, void* stream);

/* ---- the per-timestep path: one environment, one transition, one minibatch, one update per timestep (csrc/step_path.hip) ------
 * What NAFAgent.step() does between two act() calls of the reference's loop (naf_algorithm.py:144-156, :249-261), in two
 * launches around the five of the row-split chain instead of seven.
 *
 * naf_step_prep: `self.memory.add(...)` of the timestep's transition (naf_algorithm.py:144 / utils/replay_buffer.py:32-45) +
 * `random.sample(self.memory, k)` (:55) + the stacking of the minibatch (:57-65) + the moments record of layer 1's inputs
 * (naf_bb_moments) in ONE launch of one workgroup:
 *   src_row / n_word (both or neither): one transition row (device-visible memory, e.g. pinned host) and the word that says
 *     whether this tick brings it (0 / 1, read where the kernel runs, as naf_replay_add_counted); NULL: nothing is appended;
 *     row_out (nullable, device, naf_replay_row_floats floats): a copy of the row as read, whatever the count says — the
 *     launch that ends the timestep (naf_adam_polyak_act) takes the policy's next observation from its next_state columns
 *     instead of paying for a second read of host memory on its critical path;
 *   the draw is naf_replay_sample_indices' (Philox stream position *counter_dev, advanced by one HERE; same indices bit for bit),
 *     taken on the ring as the append leaves it (add, then sample: the reference's order); idx_out (nullable): the B positions;
 *   out_rows [B][out_ld], action_mode: naf_replay_gather_rows' output; mom [2][naf_bb_moments_floats(S)]: naf_bb_moments'
 *     records of the state (net 0) and next-state (net 1) columns, bit for bit. B <= 4096, S <= 32;
 *   spec_rec / idx_spec (nullable, device: NAF_STEP_SPEC_INTS int32, 16-byte aligned, zero-initialised by the caller / B int32): the
 *     record and the indices of a PREFETCH of this timestep's minibatch that the previous timestep's last launch left in out_rows /
 *     mom (naf_adam_polyak_act with `prefetch`). The record says what that launch assumed — one row to append to the {head, size} it
 *     found, the stream position — and whether the row to come was among the positions drawn; if all of it holds HERE, this launch
 *     appends the row, advances the counters, copies idx_spec to idx_out and is done (1 us instead of 10); otherwise — and always
 *     without a record — it does everything itself. The same minibatch, moments and indices either way. The record is cleared. */
#define NAF_STEP_SPEC_INTS 12 /* [8], [9]: timesteps that took the prefetched minibatch / that drew for themselves */
/*   copies (nullable, HOST pointer): up to three ranges of 4-byte words (<= 4096 each) copied src -> dst by the launch before anything
 *     else — the pipelined form of the path keeps a WORKING copy of what a learn() chain changes besides the gradient (BatchNorm
 *     running statistics, optimizer step count, loss partials) next to the public one; a launch that starts a timestep over resets
 *     working from public here, naf_adam_polyak_act commits working to public (its `prefetch->copies`). */
typedef struct naf_step_copies {
    const void* src[3];
    void* dst[3];
    int n_words[3];            /* 0: unused */
} naf_step_copies_t;
int naf_step_prep(naf_replay_t* h, const float* src_row, const int32_t* n_word, float* row_out, uint64_t seed,
                  uint64_t* counter_dev, int32_t* idx_out, float* out_rows, int out_ld, int action_mode, float* mom, int B,
                  int without_replacement, int32_t* spec_rec, const int32_t* idx_spec, const naf_step_copies_t* copies, void* stream);
/* naf_adam_polyak_act: naf_adam_polyak_fused (clip_grad_norm_ + Adam.step + soft_update, naf_algorithm.py:209-213, :217-226) AND
 * naf_policy_act for ONE state (NAFAgent.act, :158-178: the loop's next `self.act(state)`, :249) in one launch: the workgroups
 * that step a slice of the main network's parameters keep the new values in registers and multiply them with the policy's
 * activations, which cross workgroups as (value, epoch) records. Same parameters, Adam state and target as naf_adam_polyak_fused
 * leaves them, same action as naf_policy_act computes from them (bit for bit, both: tests/test_kernels_gpu.py).
 *   adam: as naf_bb_layer1_adam's (l1_floats and bc are ignored); the flat buffers must be exactly
 *     [W1 (H x S) | b1 | g1 | be1 | W2 (H x H) | b2 | g2 | be2 | Wh (NHP x HP)] at the offsets `net` names (floats), H = 256 | 512;
 *   obs [S] and action_out [A] may be pinned host memory; heads_out (nullable) [A + T + 1] pre-activations;
 *   seed / counter_dev / noise_scale / p_mode: naf_policy_act's noise stream (*counter_dev advanced by one);
 *   sync: naf_adam_polyak_act_sync_ints() int32 of device scratch, zero-initialised ONCE by the caller, 16-byte aligned, used by
 *     this entry point only (one launch in flight per buffer);
 *   host_errors (nullable): pinned host word that counts polls whose 2-ms bound ran out (the action is then NaN: an error);
 *   host_seq (nullable): pinned host uint32 that receives the launch's ordinal (1, 2, ...: the number of calls on `sync`) behind
 *     the action's words — for readers that synchronise the stream (stores to host memory may pass one another on the way);
 *   action_rec (nullable): pinned host memory, 64 bytes, 16-byte aligned: three (A > 8: four) chunks {a[3j], a[3j + 1], a[3j + 2], ordinal}, ONE
 *     16-byte store each — a host that polls the chunks it needs (j < ceil(A / 3)) until they carry the ordinal it expects reads
 *     the action without synchronising the stream, whatever order the stores arrive in;
 *   prefetch (nullable, HOST pointer): one more workgroup of the launch draws, gathers and takes the moments of the NEXT
 *     timestep's minibatch — naf_step_prep's arguments of the same names, on the ring as ONE more append will leave it — beside the
 *     launch's own work and behind the action's announcement to the host (what a timestep draws depends on the row it appends only
 *     through the fill level, and through the row itself if the draw picks it). mode 1: nothing is committed — the ring's counters
 *     and *counter_dev stay as they are; spec_rec / idx_spec receive the record and the indices the next naf_step_prep checks.
 *     mode 2 (the PIPELINED timestep: this launch is the first of its graph, the learn() chain on the prefetched minibatch follows
 *     it and computes the NEXT update's gradient while the host steps the environment): the workgroup first does what naf_step_prep
 *     does when the record holds — reads [src_row | n_word], appends, advances the counters, hands idx_spec to idx_out — and then
 *     prefetches on the ring as it has become. The host launches this form only after it has read `valid` from host_spec; a
 *     record that does not hold here is counted in *pipe_errors (pinned host; the caller raises). host_spec (nullable, pinned host
 *     uint32[2], 8-byte aligned, written by ONE 8-byte store): [0] = the launch's ordinal (as host_seq), [1] = whether the prefetch
 *     holds (the row to come was not among the positions drawn). copies: committed working -> public state, see naf_step_copies_t.
 *   obs_system_scope: obs lies in device memory the host stored into (naf_host_publish): read with system-scope loads. */
typedef struct naf_step_prefetch {
    naf_replay_t* replay;
    uint64_t seed;
    uint64_t* counter_dev;     /* the SAMPLER's stream position (mode 1: read only) */
    int32_t* idx_spec;         /* [B] */
    float* out_rows;           /* [B][out_ld] */
    int out_ld, action_mode;
    float* mom;
    int B, without_replacement;
    int32_t* spec_rec;         /* NAF_STEP_SPEC_INTS int32 */
    int mode;                  /* 1: prefetch only; 2: this timestep's append, then the prefetch (naf_step_prefetch only);
                                * 0 (naf_adam_polyak_act only): no prefetching workgroup, the launch commits `copies` */
    const float* src_row;      /* mode 2 */
    const int32_t* n_word;     /* mode 2 */
    float* row_out;            /* mode 2, nullable */
    int32_t* idx_out;          /* mode 2, nullable [B] */
    uint32_t* host_spec;       /* nullable */
    uint64_t* pipe_errors;     /* nullable */
    naf_step_copies_t copies;
    /* round 6 (ABI 30) */
    int depth;                 /* 0 / 1: the NEXT timestep's minibatch; 2: the one behind it (two appends and one draw ahead) */
    int32_t* spec_rec_in;      /* mode 2: the record / indices of the minibatch THIS timestep consumes; NULL: spec_rec / idx_spec */
    int32_t* idx_spec_in;
    uint32_t* pf_seq;          /* device word owned by the prefetching workgroup, zero-initialised by the caller: the ordinal its
                                * verdicts carry (+ 1 per prefetch); required with host_spec */
} naf_step_prefetch_t;
/* The prefetch as a launch of its own (one workgroup; csrc/step_path.hip, step_prefetch_kernel) — what the reference does at
 * utils/replay_buffer.py:55-65 for a LATER timestep of the loop of naf_algorithm.py:249-261, taken off that loop's critical path.
 *   mode 1: draw, gather and take the moments of the minibatch `depth` appends ahead of the ring as found; nothing is committed.
 *   mode 2: first this timestep's part — read [src_row | n_word], check the record spec_rec_in against the ring and the stream
 *     position found (a mismatch is counted in *pipe_errors), append, advance the counters, hand idx_spec_in to idx_out — then
 *     prefetch `depth` appends ahead of the ring as that leaves it.
 *   depth 2: the minibatch is void if EITHER of the two rows to come is among the positions drawn; the record names the state
 *     the consuming launch must find before its append (one append and one draw further on than the state this launch left).
 *   The verdict {ordinal, valid} goes to host_spec in ONE 8-byte store behind an agent-scope release of everything the workgroup
 *   wrote: a host that has read it may launch, on any stream, work that reads the minibatch, the ring or the counters.
 * naf_step_launch: naf_host_publish (bytes == 0: nothing to publish) + hipGraphLaunch(graph_exec, stream) + naf_step_prefetch(
 *   prefetch, side_stream) (prefetch == NULL: none) — one timestep of NAFAgent.step() in one trip through the FFI.
 *   prefetch_first != 0: the prefetch is launched BEFORE the graph (neither reads what the other writes): its verdict reaches the
 *   host 8 us sooner — the time hipGraphLaunch keeps the calling thread — and the action as much later. */
int naf_step_prefetch(const naf_step_prefetch_t* prefetch, void* stream);
int naf_step_launch(void* dst_device, const void* src_host, size_t bytes, void* graph_exec, void* stream,
                    const naf_step_prefetch_t* prefetch, void* side_stream, int prefetch_first);
typedef struct naf_act_net {
    int S, A, H, NHP, HP;
    int64_t off_W1, off_b1, off_g1, off_be1, off_W2, off_b2, off_g2, off_be2, off_Wh;
    const float *running_mean1, *running_var1, *running_mean2, *running_var2;
    float eps;
} naf_act_net_t;
int naf_adam_polyak_act_sync_ints(void);
int naf_adam_polyak_act(const naf_adam_args_t* adam, const naf_act_net_t* net, const float* obs, float* heads_out,
                        float* action_out, uint64_t seed, uint64_t* counter_dev, float noise_scale, int p_mode, int32_t* sync,
                        uint64_t* host_errors, uint32_t* host_seq, uint32_t* action_rec, const naf_step_prefetch_t* prefetch,
                        int obs_system_scope, void* stream);
/* naf_adam_polyak_act_layer1 (round 6): the same launch with LAYER 1 of the NEXT update's chain riding on it — what naf_bb_layer1_adam
 * does as a launch of its own (both networks' Linear 1 + BatchNorm(train) + ReLU from the minibatch's moments record,
 * naf_neural_network.py:76), in B/64 x H/64 x 2 extra workgroups that start once the launch's optimizer step has written the layer-1
 * segment of both networks and the commit (`prefetch->copies`, mode 0) has copied the working BatchNorm statistics they advance.
 * The chain behind this launch then starts at GEMM 2 (naf_bb_linear_stats_adam with adam = NULL): it no longer waits for the act()
 * tail of this launch nor for one launch boundary — 40 -> 35 us per timestep of the reference's loop at B = 64. The same body, the same
 * bits as the launch of its own (csrc/layer1_body.h). `layer1`: the arguments of naf_bb_layer1_adam of the same names (nets must be
 * 2; W / bias / gamma / beta: the MAIN network's, the target's param_net_stride floats behind — the buffers this launch steps);
 * `prefetch`: NULL or mode 0 (the commit only). Everything else as naf_adam_polyak_act. */
typedef struct {
    const float* x;              /* the minibatch rows: net 0 reads `state`, net 1 `next_state` x_net_stride floats further on */
    int64_t x_net_stride;
    int ldx, K;
    const float *W, *bias, *gamma, *beta;
    int64_t param_net_stride;
    const float* mom;            /* [2][naf_bb_moments_floats(K)] */
    float *running_mean, *running_var;
    int64_t stat_net_stride;
    float* out;                  /* A1 [2][..][ldo] */
    int64_t out_net_stride;
    int ldo;
    float *save_mean, *save_invstd, *wc_out, *xhat_out;
    int B, H, nets;
    float momentum, eps;
} naf_bb_layer1_t;
int naf_adam_polyak_act_layer1(const naf_adam_args_t* adam, const naf_act_net_t* net, const float* obs, float* heads_out,
                               float* action_out, uint64_t seed, uint64_t* counter_dev, float noise_scale, int p_mode, int32_t* sync,
                               uint64_t* host_errors, uint32_t* host_seq, uint32_t* action_rec, const naf_step_prefetch_t* prefetch,
                               int obs_system_scope, const naf_bb_layer1_t* layer1, void* stream);

/* ---- synthetic manipulator environment (stand-in for the PyBullet Environment) ---------------- */
/* One step of E independent kinematic-chain arms on the device, emitting transition rows
 * (environment/environment.py:431-485 state layout / reward constants); see csrc/synth_env.hip. */
/* Per-env episode bookkeeping of the many-env training / evaluation loops — what NAFAgent.run keeps per episode
 * (`score += reward`, `frame`; naf_algorithm.py:246-270) and what test_trained_model reads at an episode's end
 * (`done`, `reward == 250`; rl_framework.py:341-355). One record per (vector step, env), frames == 0 when no episode of
 * that env ended in that step. 32 bytes. */
typedef struct naf_episode_record {
    double score;        /* sum of the episode's rewards, accumulated in double in step order (Python's float sum) */
    int32_t frames;      /* steps the episode took (the reference's last `frame`); 0 = no episode ended here */
    int32_t done;        /* 1: terminal state (target reached / collision), 0: the frame budget ran out */
    float last_reward;   /* reward of the episode's last step (+250 = target reached) */
    int32_t episode;     /* 1-based ordinal of the episode among this env's episodes */
    uint32_t step_lo;    /* low 32 bits of the vector-step counter the record was written at */
    uint32_t env;        /* env index */
} naf_episode_record_t;
/* records (nullable): [record_slots][E] ring of episode records; the step writes slot (*counter_dev % record_slots) for
 * every env on EVERY call, so a host that copies the ring once per record_slots steps sees every finished episode in
 * (step, env) order without atomics or a per-step synchronisation. Needs counter_dev. */
int naf_synth_env_step(float* env_state, const float* actions, float* out_rows, float* obs_next, int E, int A,
                       uint64_t seed, const uint64_t* counter_dev, int max_frames, naf_episode_record_t* records,
                       int record_slots, void* stream);
/* preset_host (nullable, HOST pointer, preset_floats = 15 or NAF_SYNTH_PRESET_FLOATS = 23 floats): [initial joint
 * positions(8) | target xyz | obstacle xyz | obstacle_jitter | variation(8)]; NULL = the reference's KUKA demo preset.
 * obstacle_jitter > 0: per-env obstacle position, uniform in a cube of that half-width around the preset (seeded by
 * (seed, env)). variation[k]: half-width of the uniform range joint k's initial position is drawn from at every reset
 * (initial_positions_variation_range, environment.py:284-293); 0.1 on every joint when only 15 floats are given. */
#define NAF_SYNTH_PRESET_FLOATS 23
int naf_synth_env_reset(float* env_state, float* obs, int E, int A, uint64_t seed, uint64_t counter,
                        const float* preset_host, int preset_floats, void* stream);
int naf_synth_env_state_floats(int A);

/* ---- one-shot gradient all-reduce over peer-mapped memory (no reference counterpart) ----------
 * The data-parallel exchange that follows loss.backward() (naf_algorithm.py:207-210 on every rank): sum of the flat
 * gradient over the W <= 8 GPUs of one node, each rank pushing its gradient into a receive slot on every peer over
 * xGMI and summing the W contributions in rank order (bit-identical on all ranks). See csrc/xgmi_reduce.hip.
 *   create  : allocates this rank's receive slab (uncached device memory) for n_floats-long gradients (n_floats % 4
 *             == 0); timeout_s bounds every wait on a peer (a time-out is counted, the kernel then proceeds)
 *   export  : writes the slab's 64-byte hipIpc handle; the host exchanges the W handles (torch.distributed)
 *   connect : all_handle_bytes = W x 64 bytes in rank order; maps every peer slab (hipIpcOpenMemHandle) after enabling
 *             peer access to the devices the peers run on
 *   allreduce_sum : one launch on `stream`: grad_out[i] = sum over ranks of grad_in[i] (in place allowed);
 *             sumsq_partials (nullable) receives ceil(n_floats / naf_xgmi_chunk_floats()) partial sums of
 *             grad_out^2 for naf_adam_polyak_fused; step_dev (nullable) is advanced by one. Capturable.
 *   status  : blocking read of the epoch (all-reduces done) and of the number of timed-out waits (must stay 0). */
int naf_xgmi_chunk_floats(void);
int naf_xgmi_create(int rank, int world, size_t n_floats, double timeout_s, void** handle);
int naf_xgmi_set_timeout(void* handle, double timeout_s); /* for launches enqueued (or captured) after this call */
int naf_xgmi_mem_kind(void* handle); /* 2 = uncached, 1 = fine-grained */
int naf_xgmi_export(void* handle, void* out_handle_bytes);
int naf_xgmi_connect(void* handle, const void* all_handle_bytes, const int* peer_devices /* nullable, W device indices */);
/* the same for communicators that live in ONE process (no hipIpc: each rank's slab is the process's own memory) — a rehearsal of
 * world sizes a one-GPU box cannot host as processes; all_handles[world] = the communicators of ranks 0 .. world - 1. */
int naf_xgmi_connect_local(void* handle, void* const* all_handles);
int naf_xgmi_allreduce_sum(void* handle, const float* grad_in, float* grad_out, float* sumsq_partials,
                           int32_t* step_dev, void* stream);
/* Early push: grad[lo, hi) (multiples of 4, hi <= n_floats) goes to the peers ahead of the all-reduce proper, either from
 * a launch of its own (push_early: used by the self-test) or from extra workgroups of naf_bn_relu_bwd_wgrad_push. The
 * all-reduce that follows must then be naf_xgmi_allreduce_sum_from with the same `lo`: it pushes only [0, lo). */
int naf_xgmi_push_desc(void* handle, naf_xgmi_push_t* out);
int naf_xgmi_push_early(void* handle, const float* grad_in, size_t lo, size_t hi, void* stream);
int naf_xgmi_allreduce_sum_from(void* handle, const float* grad_in, float* grad_out, float* sumsq_partials,
                                int32_t* step_dev, size_t pushed_lo, void* stream);
/* the same with a SECOND range that went ahead: grad_in[skip_lo, skip_hi) (multiples of 4; skip_lo == skip_hi: none). The
 * row-split chain's finish launch pushes the W2 and Wh segments, which are not adjacent in the flat buffer. */
int naf_xgmi_allreduce_sum_from2(void* handle, const float* grad_in, float* grad_out, float* sumsq_partials,
                                 int32_t* step_dev, size_t pushed_lo, size_t skip_lo, size_t skip_hi, void* stream);
int naf_xgmi_status(void* handle, uint64_t* epoch, uint64_t* timeouts);
/* Time-outs so far, read from a pinned host word the kernel bumps: never synchronises, so the training loop polls it
 * after every chunk. A timed-out all-reduce leaves -inf in its sumsq partial, which makes naf_adam_polyak_fused skip
 * that update on this rank (no wrong step reaches the weights); the host is expected to stop on a non-zero count. */
int naf_xgmi_timeouts_nowait(void* handle, uint64_t* timeouts);
/* teardown in two collective halves: disconnect = wait for this device, then unmap every peer slab; destroy = (disconnect and)
 * release this rank's own slab. The host runs a barrier between the two, so no slab is released while a peer still maps it. */
int naf_xgmi_disconnect(void* handle);
int naf_xgmi_destroy(void* handle);

#ifdef __cplusplus
}
#endif
#endif /* NAF_HIP_H */
