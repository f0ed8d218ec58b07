// HBM-resident uniform replay buffer for gfx950: ring of 256-byte transition rows, device-side
// Philox index generation, coalesced row gather. Replaces the reference's deque + random.sample +
// numpy stacking (utils/replay_buffer.py:16-75). HBM-bound byte moving: no LDS reuse to exploit, the
// rules that matter are 16-B/lane accesses, whole 128-B lines per row and enough loads in flight.
#include "common.h"
#include "../../include/naf_hip.h"
#include <new>

struct naf_replay {
    uint64_t capacity;
    int S, A, row_floats;
    float* rows;
    uint64_t* meta;  // {head, size, total_added, sample_counter, -, -, -, bad_index_count}
    uint32_t magic;
};
#define NAF_REPLAY_MAGIC 0x4e414652u

enum { META_HEAD = 0, META_SIZE = 1, META_TOTAL = 2, META_SAMPLE_CTR = 3, META_BAD_IDX = 7 };

extern "C" int naf_replay_row_floats(int S, int A) {
    if (S <= 0 || A <= 0) return NAF_ERR_ARG;
    // whole 128-B lines, and a power-of-two number of float4 per row so a row maps onto 2^k lanes
    int need = naf_row_off_done(S, A) + 1;
    int rf = 32;
    while (rf < need) rf *= 2;
    return rf;
}

extern "C" int naf_replay_row_off_next_state(int S, int A) {
    if (S <= 0 || A <= 0) return NAF_ERR_ARG;
    return naf_row_off_s2(S, A);
}

extern "C" int naf_replay_create(uint64_t capacity, int S, int A, float* rows, uint64_t* meta, naf_replay_t** out) {
    if (!out || !rows || !meta || capacity == 0 || S <= 0 || A <= 0 || A > NAF_MAX_A) return NAF_ERR_ARG;
    if (capacity > 0x7fffffffull) return NAF_ERR_ARG;  // deque positions are int32
    if (((uintptr_t)rows & 15) != 0) return NAF_ERR_ARG;
    naf_replay* h = new (std::nothrow) naf_replay;
    if (!h) return NAF_ERR_STATE;
    h->capacity = capacity;
    h->S = S;
    h->A = A;
    h->row_floats = naf_replay_row_floats(S, A);
    h->rows = rows;
    h->meta = meta;
    h->magic = NAF_REPLAY_MAGIC;
    *out = h;
    return NAF_OK;
}

extern "C" int naf_replay_destroy(naf_replay_t* h) {
    if (!h || h->magic != NAF_REPLAY_MAGIC) return NAF_ERR_STATE;
    h->magic = 0;
    delete h;
    return NAF_OK;
}

extern "C" int naf_replay_size(naf_replay_t* h, uint64_t* size_out, void* stream) {
    if (!h || h->magic != NAF_REPLAY_MAGIC) return NAF_ERR_STATE;
    if (!size_out) return NAF_ERR_ARG;
    // the one entry point that waits: the fill level lives on the device (appends are asynchronous)
    hipError_t e = hipMemcpyAsync(size_out, &h->meta[META_SIZE], sizeof(uint64_t), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    return e == hipSuccess ? NAF_OK : (int)e;
}

// ------------------------------------------------------------------------------------------------
// add: scatter n packed rows to ring positions (head + i) mod capacity, then advance {head,size,total}
// ------------------------------------------------------------------------------------------------
__global__ void replay_add_rows_kernel(float4* __restrict__ ring, const uint64_t* __restrict__ meta,
                                       const float4* __restrict__ src, int n, uint64_t cap, int rf4_shift) {
    const uint64_t head = meta[META_HEAD];
    const int rf4 = 1 << rf4_shift;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ((int64_t)n << rf4_shift);
         g += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = g >> rf4_shift;
        int c = (int)(g & (rf4 - 1));
        uint64_t phys = (head + (uint64_t)r) % cap;
        ring[(phys << rf4_shift) + c] = src[g];
    }
}

__global__ void replay_advance_kernel(uint64_t* meta, uint64_t n, uint64_t cap) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        uint64_t head = meta[META_HEAD], size = meta[META_SIZE];
        meta[META_HEAD] = (head + n) % cap;
        size += n;
        meta[META_SIZE] = size > cap ? cap : size;
        meta[META_TOTAL] += n;
    }
}

// n*rf4 <= 1024: one workgroup does the scatter AND the advance (one launch per vector-env step)
__global__ __launch_bounds__(1024) void replay_add_small_kernel(float4* __restrict__ ring, uint64_t* meta,
                                                                const float4* __restrict__ src, int n,
                                                                uint64_t cap, int rf4_shift) {
    const uint64_t head = meta[META_HEAD];
    const uint64_t size = meta[META_SIZE];
    const int rf4 = 1 << rf4_shift;
    __syncthreads();  // every lane has read {head,size} before lane 0 rewrites them
    int g = threadIdx.x;
    if (g < (n << rf4_shift)) {
        int r = g >> rf4_shift;
        int c = g & (rf4 - 1);
        uint64_t phys = (head + (uint64_t)r) % cap;
        ring[(phys << rf4_shift) + c] = src[g];
    }
    if (threadIdx.x == 0) {
        meta[META_HEAD] = (head + (uint64_t)n) % cap;
        uint64_t s2 = size + (uint64_t)n;
        meta[META_SIZE] = s2 > cap ? cap : s2;
        meta[META_TOTAL] += (uint64_t)n;
    }
}

static inline int ilog2_exact(int x) {
    int s = 0;
    while ((1 << s) < x) ++s;
    return s;
}

extern "C" int naf_replay_add_batch(naf_replay_t* h, const float* src_rows, int n, void* stream) {
    if (!h || h->magic != NAF_REPLAY_MAGIC) return NAF_ERR_STATE;
    if (!src_rows || n < 0 || (uint64_t)n > h->capacity || ((uintptr_t)src_rows & 15) != 0) return NAF_ERR_ARG;
    if (n == 0) return NAF_OK;
    hipStream_t st = (hipStream_t)stream;
    const int rf4 = h->row_floats / 4;
    const int sh = ilog2_exact(rf4);
    if ((int64_t)n * rf4 <= 1024) {
        int threads = naf_round_up(n * rf4, 64);
        replay_add_small_kernel<<<1, threads, 0, st>>>((float4*)h->rows, h->meta, (const float4*)src_rows, n,
                                                        h->capacity, sh);
        NAF_CHECK_LAUNCH();
        return NAF_OK;
    }
    int64_t total = (int64_t)n * rf4;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    replay_add_rows_kernel<<<blocks, 256, 0, st>>>((float4*)h->rows, h->meta, (const float4*)src_rows, n,
                                                    h->capacity, sh);
    NAF_CHECK_LAUNCH();
    replay_advance_kernel<<<1, 64, 0, st>>>(h->meta, (uint64_t)n, h->capacity);
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}

// ------------------------------------------------------------------------------------------------
// sample: one workgroup per minibatch. Draw t of attempt a = mulhi64(philox(ctr, t, a), size).
// Without replacement: element t redraws while an element j < t holds the same value; rounds repeat
// until no duplicate is left (expected number of redraws ~ B^2 / 2N; population >= 4B so a redraw collides with
// probability <= 1/4 and NAF_SAMPLE_MAX_ROUNDS rounds always suffice in practice). Deterministic in
// (seed, counter, size): the numpy restatement (oracle.replay_sample_indices) reproduces it bit for bit.
// ------------------------------------------------------------------------------------------------
#define NAF_SAMPLE_MAX_ROUNDS 64

__device__ static inline int sample_draw(uint64_t ctr, uint32_t t, uint32_t attempt, uint64_t seed, uint64_t size) {
    Philox4 p = philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), t, attempt, (uint32_t)seed, (uint32_t)(seed >> 32));
    uint64_t r = ((uint64_t)p.v[0] << 32) | (uint64_t)p.v[1];
    return (int)__umul64hi(r, size);
}

__global__ __launch_bounds__(1024) void replay_sample_kernel(const uint64_t* __restrict__ meta, uint64_t seed,
                                                             const uint64_t* __restrict__ counter_dev,
                                                             uint64_t counter_off, int32_t* __restrict__ idx, int B,
                                                             int without_replacement) {
    extern __shared__ __attribute__((aligned(16))) int vals[];  // 4*B ints: B sampled values, or the whole population in the dense regime
    const uint64_t size = meta[META_SIZE];
    const uint64_t ctr = (counter_dev ? *counter_dev : 0ull) + counter_off + (uint64_t)blockIdx.x;
    int32_t* out = idx + (int64_t)blockIdx.x * B;
    if (size == 0) {  // nothing to sample from: emit position 0 (the gather flags it as a bad index)
        for (int t = threadIdx.x; t < B; t += blockDim.x) out[t] = 0;
        return;
    }
    if (without_replacement && size >= (uint64_t)B && size < 4ull * (uint64_t)B) {
        // dense regime (population < 4B, i.e. the first learn() calls after the `len > batch_size` gate,
        // naf_algorithm.py:150): rejection would need ~size rounds, so do a partial Fisher-Yates over the
        // population held in LDS (size < 4B ints fits the 4B-int allocation). Sequential, rare, short.
        int* perm = vals;
        for (int t = threadIdx.x; t < (int)size; t += blockDim.x) perm[t] = t;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int t = 0; t < B; ++t) {
                int j = t + sample_draw(ctr, (uint32_t)t, 0xFFFFFFFFu, seed, size - (uint64_t)t);
                int a = perm[t];
                perm[t] = perm[j];
                perm[j] = a;
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < B; t += blockDim.x) out[t] = perm[t];
        return;
    }
    const bool dedupe = without_replacement && size >= (uint64_t)B;
    // elements owned by this thread: t = threadIdx.x + k*blockDim.x, k < 4 (B <= 4096)
    uint32_t attempt[4] = {0, 0, 0, 0};
    for (int k = 0, t = threadIdx.x; t < B; t += blockDim.x, ++k) vals[t] = sample_draw(ctr, (uint32_t)t, 0u, seed, size);
    __syncthreads();
    if (dedupe) {
        for (int round = 0; round < NAF_SAMPLE_MAX_ROUNDS; ++round) {
            int dupmask = 0;
            for (int k = 0, t = threadIdx.x; t < B; t += blockDim.x, ++k) {
                // "some earlier element holds the same value": four candidates per LDS read, eight reads in flight (the
                // one-int-per-iteration form was a chain of ~B/2 dependent LDS round trips: 20 us per launch at B = 256)
                const int mine = vals[t];
                // The scan is VALU-bound (B candidates per element). Wave-uniform trip counts keep the loops unrolled
                // into independent LDS reads (with a per-lane bound they ran as one dependent read per iteration):
                // candidates below the wave's first element need no position mask, only the wave's own 64 positions do.
                // (A sort-based variant, bitonic in LDS, measured slower: 8.9 vs 7.1 us at B = 256.)
                const int wave_first = t & ~63;                   // uniform inside a wave
                const int4* v4 = (const int4*)vals;
                bool dup = false;
                const int full4 = wave_first >> 2;
#pragma unroll 8
                for (int j4 = 0; j4 < full4; ++j4) {
                    const int4 q = v4[j4];
                    dup |= (q.x == mine) | (q.y == mine) | (q.z == mine) | (q.w == mine);
                }
                const int end4 = (wave_first + 64 < B ? wave_first + 64 : B + 3) >> 2;
#pragma unroll 8
                for (int j4 = full4; j4 < end4; ++j4) {
                    const int4 q = v4[j4];
                    const int j = 4 * j4;
                    dup |= ((q.x == mine) & (j + 0 < t)) | ((q.y == mine) & (j + 1 < t)) | ((q.z == mine) & (j + 2 < t)) |
                           ((q.w == mine) & (j + 3 < t));
                }
                if (dup) dupmask |= (1 << k);
            }
            int any = __syncthreads_or(dupmask);  // also orders the reads above before the writes below
            if (!any) break;
            for (int k = 0, t = threadIdx.x; t < B; t += blockDim.x, ++k) {
                if (dupmask & (1 << k)) {
                    attempt[k] += 1u;
                    vals[t] = sample_draw(ctr, (uint32_t)t, attempt[k], seed, size);
                }
            }
            __syncthreads();
        }
    }
    for (int t = threadIdx.x; t < B; t += blockDim.x) out[t] = vals[t];
}

extern "C" int naf_replay_sample_indices(naf_replay_t* h, uint64_t seed, const uint64_t* counter_dev,
                                         uint64_t counter_off, int32_t* idx, int B, int n_batches,
                                         int without_replacement, void* stream) {
    if (!h || h->magic != NAF_REPLAY_MAGIC) return NAF_ERR_STATE;
    if (!idx || B <= 0 || B > 4096 || n_batches <= 0) return NAF_ERR_ARG;
    int threads = B >= 1024 ? 1024 : naf_round_up(B, 64);
    replay_sample_kernel<<<n_batches, threads, (size_t)B * 4 * sizeof(int), (hipStream_t)stream>>>(
        h->meta, seed, counter_dev, counter_off, idx, B, without_replacement);
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}

// ------------------------------------------------------------------------------------------------
// gather (row layout): 2^k lanes x float4 per 256-B row, ROWS_PER_THREAD independent rows per lane so
// several index->row dependent loads are in flight per lane. Algorithmic traffic: 4*(2S+A+2) B read +
// the same written per row (400 B at S=21/A=6); physical: 2 x row_floats*4 B.
// ------------------------------------------------------------------------------------------------
typedef float nt_f4 __attribute__((ext_vector_type(4)));   // native vector type the nontemporal builtins accept

template <int RPT, int NT /* bit 0: nontemporal loads, bit 1: nontemporal stores */>
__global__ __launch_bounds__(256) void replay_gather_rows_kernel(const float4* __restrict__ ring,
                                                                 uint64_t* __restrict__ meta,
                                                                 const int32_t* __restrict__ idx,
                                                                 float4* __restrict__ out, int n, uint64_t cap,
                                                                 int rf4_shift, int trunc_lo, int trunc_hi) {
    const uint64_t head = meta[META_HEAD];
    const uint64_t size = meta[META_SIZE];
    const uint64_t base = head + cap - size;  // physical position of deque element 0 (oldest)
    const int rf4 = 1 << rf4_shift;
    const int lanes_per_block_rows = blockDim.x >> rf4_shift;  // rows handled per block per pass
    const int c = threadIdx.x & (rf4 - 1);
    const int rl = threadIdx.x >> rf4_shift;
    const int64_t row0 = ((int64_t)blockIdx.x * RPT) * lanes_per_block_rows + rl;
    int64_t pos[RPT];
    bool ok[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        int64_t r = row0 + (int64_t)k * lanes_per_block_rows;
        ok[k] = r < n;
        int64_t i = ok[k] ? (int64_t)idx[r] : 0;
        if (ok[k] && (i < 0 || (uint64_t)i >= size)) {
            if (c == 0) atomicAdd((unsigned long long*)&meta[META_BAD_IDX], 1ull);
            i = 0;
        }
        pos[k] = (int64_t)((base + (uint64_t)i) % cap);
    }
    float4 v[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k)
        if (ok[k]) {
            // bulk launches (RPT > 1) stream: every ring row is touched once, nothing is re-read by this kernel
            if (NT & 1) {
                const nt_f4 t = __builtin_nontemporal_load((const nt_f4*)&ring[(pos[k] << rf4_shift) + c]);
                v[k] = make_float4(t.x, t.y, t.z, t.w);
            } else {
                v[k] = ring[(pos[k] << rf4_shift) + c];
            }
        }
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        if (!ok[k]) continue;
        // `.long()` of the reference: truncate the action columns toward zero
        int f0 = c * 4;
        if (f0 + 3 >= trunc_lo && f0 < trunc_hi) {
            if (f0 + 0 >= trunc_lo && f0 + 0 < trunc_hi) v[k].x = truncf(v[k].x);
            if (f0 + 1 >= trunc_lo && f0 + 1 < trunc_hi) v[k].y = truncf(v[k].y);
            if (f0 + 2 >= trunc_lo && f0 + 2 < trunc_hi) v[k].z = truncf(v[k].z);
            if (f0 + 3 >= trunc_lo && f0 + 3 < trunc_hi) v[k].w = truncf(v[k].w);
        }
        int64_t r = row0 + (int64_t)k * lanes_per_block_rows;
        if (NT & 2) {
            nt_f4 t = {v[k].x, v[k].y, v[k].z, v[k].w};
            __builtin_nontemporal_store(t, (nt_f4*)&out[(r << rf4_shift) + c]);
        } else {
            out[(r << rf4_shift) + c] = v[k];        // minibatch-sized launches: the learner reads these rows next
        }
    }
}

// bulk launches: cache policy of the streamed rows (bit 0 nontemporal loads, bit 1 nontemporal stores)
int g_gather_nt = -1;   // -1 = default (nontemporal stores on bulk launches)

extern "C" int naf_replay_gather_rows(naf_replay_t* h, const int32_t* idx, float* out_rows, int n, int action_mode,
                                      void* stream) {
    if (!h || h->magic != NAF_REPLAY_MAGIC) return NAF_ERR_STATE;
    if (!idx || !out_rows || n < 0 || ((uintptr_t)out_rows & 15) != 0) return NAF_ERR_ARG;
    if (action_mode != NAF_ACTION_TRUNC_INT && action_mode != NAF_ACTION_FLOAT) return NAF_ERR_ARG;
    if (n == 0) return NAF_OK;
    const int rf4 = h->row_floats / 4;
    const int sh = ilog2_exact(rf4);
    const int rows_per_pass = 256 / rf4;
    int lo = h->S, hi = h->S + h->A;
    if (action_mode == NAF_ACTION_FLOAT) lo = hi = 0x7fffffff;
    hipStream_t st = (hipStream_t)stream;
    // small launches: 1 row per lane-group so that every CU gets a workgroup; bulk launches: 4 rows in flight
    if ((int64_t)n <= 256 * 8 * rows_per_pass) {
        int blocks = (n + rows_per_pass - 1) / rows_per_pass;
        replay_gather_rows_kernel<1, 0><<<blocks, 256, 0, st>>>((const float4*)h->rows, h->meta, idx, (float4*)out_rows, n,
                                                             h->capacity, sh, lo, hi);
    } else {
        int per_block = rows_per_pass * 4;
        int blocks = (n + per_block - 1) / per_block;
#define GATHER_BULK(NTV)                                                                                              \
    replay_gather_rows_kernel<4, NTV><<<blocks, 256, 0, st>>>((const float4*)h->rows, h->meta, idx, (float4*)out_rows, n, \
                                                              h->capacity, sh, lo, hi)
        // The gathered rows are written once and not re-read by this kernel: nontemporal STORES keep them from evicting
        // ring lines (measured, 4 Mi rows per launch, interleaved A/B: 977 MiB ring 0.393 -> 0.340 ms, 244 MiB ring
        // 0.346 -> 0.300 ms). Nontemporal LOADS of the ring do not help (0.388 ms) and cost 10 % on a ring that fits the
        // 256 MiB Infinity Cache, so loads stay temporal. naf_debug_set(1, mode) overrides for A/B timing.
        int nt = g_gather_nt < 0 ? 2 : g_gather_nt;
        switch (nt) {
            case 1: GATHER_BULK(1); break;
            case 2: GATHER_BULK(2); break;
            case 3: GATHER_BULK(3); break;
            default: GATHER_BULK(0); break;
        }
    }
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}

// ------------------------------------------------------------------------------------------------
// gather to the reference's five tensors (API path of ReplayBuffer.sample()): one thread per element
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void replay_gather_soa_kernel(const float* __restrict__ ring,
                                                                uint64_t* __restrict__ meta,
                                                                const int32_t* __restrict__ idx, float* __restrict__ s,
                                                                float* __restrict__ u, float* __restrict__ r,
                                                                float* __restrict__ s2, float* __restrict__ d, int n,
                                                                uint64_t cap, int row_floats, int S, int A,
                                                                int trunc) {
    const uint64_t head = meta[META_HEAD];
    const uint64_t size = meta[META_SIZE];
    const uint64_t base = head + cap - size;
    const int off_s2 = naf_row_off_s2(S, A), off_d = naf_row_off_done(S, A);
    const int used = off_d + 1;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < (int64_t)n * used;
         g += (int64_t)gridDim.x * blockDim.x) {
        int64_t row = g / used;
        int c = (int)(g - row * used);
        int64_t i = idx[row];
        if (i < 0 || (uint64_t)i >= size) {
            if (c == 0) atomicAdd((unsigned long long*)&meta[META_BAD_IDX], 1ull);
            i = 0;
        }
        uint64_t phys = (base + (uint64_t)i) % cap;
        float v = ring[phys * (uint64_t)row_floats + c];
        if (c < S) s[row * S + c] = v;
        else if (c < S + A) u[row * A + (c - S)] = trunc ? truncf(v) : v;
        else if (c == S + A) r[row] = v;
        else if (c < off_s2) continue;                       // alignment pad
        else if (c < off_d) s2[row * S + (c - off_s2)] = v;
        else d[row] = v;
    }
}

extern "C" int naf_replay_gather_soa(naf_replay_t* h, const int32_t* idx, float* s, float* u, float* r, float* s2,
                                     float* d, int n, int action_mode, void* stream) {
    if (!h || h->magic != NAF_REPLAY_MAGIC) return NAF_ERR_STATE;
    if (!idx || !s || !u || !r || !s2 || !d || n < 0) return NAF_ERR_ARG;
    if (action_mode != NAF_ACTION_TRUNC_INT && action_mode != NAF_ACTION_FLOAT) return NAF_ERR_ARG;
    if (n == 0) return NAF_OK;
    int used = naf_row_off_done(h->S, h->A) + 1;
    int64_t total = (int64_t)n * used;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    replay_gather_soa_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(h->rows, h->meta, idx, s, u, r, s2, d, n,
                                                                       h->capacity, h->row_floats, h->S, h->A,
                                                                       action_mode == NAF_ACTION_TRUNC_INT);
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}

__global__ void counter_add_kernel(uint64_t* p, uint64_t inc) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *p += inc;
}

extern "C" int naf_counter_add(uint64_t* p, uint64_t inc, void* stream) {
    if (!p) return NAF_ERR_ARG;
    counter_add_kernel<<<1, 64, 0, (hipStream_t)stream>>>(p, inc);
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}
