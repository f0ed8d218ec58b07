// HBM-resident uniform replay buffer for gfx950: ring of 256-byte transition rows, device-side
// Philox index generation, coalesced row gather. Replaces the reference's deque + random.sample +
// numpy stacking (utils/replay_buffer.py:16-75). HBM-bound byte moving: no LDS reuse to exploit, the
// rules that matter are 16-B/lane accesses, whole 128-B lines per row and enough loads in flight.
#include <stdlib.h>
#include "common.h"
#include "sample_body.h"
#include "replay_dev.h"
#include "../../include/naf_hip.h"
#include <new>

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
    if (!out || !rows || !meta || capacity == 0 || S <= 0 || A <= 0 || A > NAF_MAX_A_WIDE) return NAF_ERR_ARG;
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

// the same with the row count read where the kernel runs: a captured graph whose FIRST node is "append this timestep's
// transition" is replayed on ticks that have no new transition too (idle ticks of a data-parallel run(), a step() that went
// through the staging area) — the host sets the word to 0 for those and the node appends nothing
__global__ __launch_bounds__(1024) void replay_add_counted_kernel(float4* __restrict__ ring, uint64_t* meta,
                                                                  const float4* __restrict__ src, const int32_t* n_word,
                                                                  int n_max, uint64_t cap, int rf4_shift) {
    int n = __builtin_nontemporal_load(n_word);
    n = n < 0 ? 0 : (n > n_max ? n_max : n);
    const uint64_t head = meta[META_HEAD];
    const uint64_t size = meta[META_SIZE];
    const int rf4 = 1 << rf4_shift;
    __syncthreads();
    int g = threadIdx.x;
    if (g < (n << rf4_shift)) {
        int r = g >> rf4_shift;
        int c = g & (rf4 - 1);
        uint64_t phys = (head + (uint64_t)r) % cap;
        ring[(phys << rf4_shift) + c] = src[g];
    }
    if (threadIdx.x == 0 && n > 0) {
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

extern "C" int naf_replay_add_counted(naf_replay_t* h, const float* src_rows, const int32_t* n_word, int n_max, void* stream) {
    if (!h || h->magic != NAF_REPLAY_MAGIC) return NAF_ERR_STATE;
    const int rf4 = h->row_floats / 4;
    if (!src_rows || !n_word || n_max < 1 || (uint64_t)n_max > h->capacity || (int64_t)n_max * rf4 > 1024 ||
        ((uintptr_t)src_rows & 15) != 0 || ((uintptr_t)n_word & 3) != 0)
        return NAF_ERR_ARG;
    replay_add_counted_kernel<<<1, naf_round_up(n_max * rf4, 64), 0, (hipStream_t)stream>>>(
        (float4*)h->rows, h->meta, (const float4*)src_rows, n_word, n_max, h->capacity, ilog2_exact(rf4));
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}

// ------------------------------------------------------------------------------------------------
// sample: one workgroup per minibatch; the draw itself is csrc/sample_body.h (shared with the per-timestep launch of
// csrc/step_path.hip).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void replay_sample_kernel(const uint64_t* __restrict__ meta, uint64_t seed,
                                                             const uint64_t* __restrict__ counter_dev,
                                                             uint64_t counter_off, int32_t* __restrict__ idx, int B,
                                                             int without_replacement, int hash_bits) {
    // B sampled values (+ the hash table of the duplicate check: 2 x 2^hash_bits ints), or the whole population (< 4 B
    // ints) in the dense regime
    extern __shared__ __attribute__((aligned(16))) int vals[];
    const uint64_t size = meta[META_SIZE];
    const uint64_t ctr = (counter_dev ? *counter_dev : 0ull) + counter_off + (uint64_t)blockIdx.x;
    int32_t* out = idx + (int64_t)blockIdx.x * B;
    replay_sample_body(vals, threadIdx.x, blockDim.x, size, ctr, seed, B, without_replacement, hash_bits);
    for (int t = threadIdx.x; t < B; t += blockDim.x) out[t] = vals[t];
}

extern "C" int naf_replay_sample_indices(naf_replay_t* h, uint64_t seed, const uint64_t* counter_dev,
                                         uint64_t counter_off, int32_t* idx, int B, int n_batches,
                                         int without_replacement, void* stream) {
    if (!h || h->magic != NAF_REPLAY_MAGIC) return NAF_ERR_STATE;
    if (!idx || B <= 0 || B > 4096 || n_batches <= 0) return NAF_ERR_ARG;
    int threads = B >= 1024 ? 1024 : naf_round_up(B, 64);
    // duplicate check through a hash table of M = 2^bits >= 2 B slots: B + 2 M ints of LDS — up to 80 KB at B = 4096, more than
    // the 64 KB a workgroup gets without asking (gfx950 has 160 KB per CU), so the kernel's limit is raised once per process.
    // (Until round 4 the larger batches fell back to the O(B^2) scan: 392 us per launch at B = 4096 against 12 at 2048.)
    int bits = sample_hash_bits(B);
    size_t lds_ints = sample_lds_ints(B, bits);
    if (lds_ints * sizeof(int) > 64 * 1024) {
        static int raised_dev[64];   // per device: 0 = not tried, 1 = raised, -1 = refused (then: the scan)
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        int& raised = raised_dev[dev];
        if (!raised)
            raised = hipFuncSetAttribute((const void*)replay_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) ==
                             hipSuccess ? 1 : -1;
        if (raised < 0 || lds_ints * sizeof(int) > 96 * 1024) bits = 0;
        lds_ints = sample_lds_ints(B, bits);         // (5 B ints = 80 KB at B = 4096: needs the raised limit too)
        if (raised < 0 && lds_ints * sizeof(int) > 64 * 1024) return NAF_ERR_STATE;
    }
    replay_sample_kernel<<<n_batches, threads, lds_ints * sizeof(int), (hipStream_t)stream>>>(
        h->meta, seed, counter_dev, counter_off, idx, B, without_replacement, bits);
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}

// ------------------------------------------------------------------------------------------------
// sample, minibatches beyond 4096: the reference takes any positive batch_size (rl_framework.py:186-189 -> random.sample,
// utils/replay_buffer.py:55). The draw above lives in one workgroup's LDS (B values + a hash table of >= 4 B ints: 80 KB at
// B = 4096); beyond that the SAME rule — element t redraws while an earlier element holds its value — runs on a table in device
// memory: one workgroup per minibatch again, the values in the output array itself, attempts / slots / keys / smallest-holder in a
// scratch area of naf_replay_sample_scratch_ints(B) ints per minibatch (round 6: the slot has a word of its own instead of the
// attempt word's upper half, which capped the table at 32768 entries and B at 16384 — now B <= NAF_SAMPLE_BIG_MAX = 2^20). Table reads go through agent-scope atomic loads (a plain load could
// be served by a stale line of this CU's L1 from the round before). Tens of microseconds per round instead of two — at batch
// sizes whose update takes a millisecond. Same indices as the LDS form would give, bit for bit (oracle.replay_sample_indices).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void replay_sample_big_kernel(const uint64_t* __restrict__ meta, uint64_t seed,
                                                                 const uint64_t* __restrict__ counter_dev, uint64_t counter_off,
                                                                 int32_t* idx, int B, int without_replacement, int hash_bits,
                                                                 int* scratch, int64_t scratch_stride) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const uint64_t size = meta[META_SIZE];
    const uint64_t ctr = (counter_dev ? *counter_dev : 0ull) + counter_off + (uint64_t)blockIdx.x;
    int* vals = idx + (int64_t)blockIdx.x * B;
    int* attempt = scratch + (int64_t)blockIdx.x * scratch_stride;       // [B]
    int* slot = attempt + B;                                             // [B]
    int* keys = slot + B;                                                // [M]
    int* tmin = keys + (1 << hash_bits);                                 // [M]
    if (size == 0) {
        for (int t = tid; t < B; t += nt) vals[t] = 0;
        return;
    }
    if (without_replacement && size >= (uint64_t)B && size < 4ull * (uint64_t)B) {
        // dense regime: partial Fisher-Yates over the population (< 4 B ints: the table's room, 2 M >= 4 B), swap targets in vals
        int* perm = keys;
        for (int t = tid; t < B; t += nt) vals[t] = t + sample_draw(ctr, (uint32_t)t, 0xFFFFFFFFu, seed, size - (uint64_t)t);
        for (int t = tid; t < (int)size; t += nt) perm[t] = t;
        __syncthreads();
        if (tid == 0) {
            for (int t = 0; t < B; ++t) {
                const int j = __hip_atomic_load(&vals[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int a = __hip_atomic_load(&perm[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int b = __hip_atomic_load(&perm[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&perm[t], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&perm[j], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        for (int t = tid; t < B; t += nt) vals[t] = __hip_atomic_load(&perm[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    for (int t = tid; t < B; t += nt) {
        vals[t] = sample_draw(ctr, (uint32_t)t, 0u, seed, size);
        attempt[t] = 0;
    }
    if (!(without_replacement && size >= (uint64_t)B)) return;
    const int M = 1 << hash_bits;
    for (int round = 0; round < NAF_SAMPLE_MAX_ROUNDS; ++round) {
        for (int e = tid; e < M; e += nt) {
            keys[e] = -1;
            tmin[e] = 0x7fffffff;
        }
        __syncthreads();
        for (int t = tid; t < B; t += nt) {
            const int mine = vals[t];                  // (written by this thread)
            unsigned h = ((unsigned)mine * 2654435761u) >> (32 - hash_bits);
            for (int probe = 0; probe < M; ++probe) {
                const int was = atomicCAS(&keys[h], -1, mine);
                if (was == -1 || was == mine) break;
                h = (h + 1) & (unsigned)(M - 1);
            }
            atomicMin(&tmin[h], t);
            slot[t] = (int)h;
        }
        __syncthreads();
        int dup_any = 0;
        for (int t = tid; t < B; t += nt) {
            const int a = attempt[t];
            const unsigned h = (unsigned)slot[t];
            const bool dup = __hip_atomic_load(&tmin[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < t;
            attempt[t] = (a & 0x7fff) | (dup ? (1 << 15) : 0);           // (bit 15: redraw)
            dup_any |= dup;
        }
        const int any = __syncthreads_or(dup_any);
        if (!any) break;
        for (int t = tid; t < B; t += nt) {
            const int a = attempt[t];
            if (a & (1 << 15)) {
                const int att = (a & 0x7fff) + 1;
                attempt[t] = att;
                vals[t] = sample_draw(ctr, (uint32_t)t, (uint32_t)att, seed, size);
            }
        }
        __syncthreads();
    }
}

#define NAF_SAMPLE_BIG_MAX (1 << 20)
extern "C" int naf_replay_sample_scratch_ints(int B) {
    if (B <= 0 || B > NAF_SAMPLE_BIG_MAX) return NAF_ERR_ARG;
    int bits = 1;
    while ((1 << bits) < 2 * B) ++bits;
    return 2 * B + 2 * (1 << bits);
}

extern "C" int naf_replay_sample_indices_big(naf_replay_t* h, uint64_t seed, const uint64_t* counter_dev, uint64_t counter_off,
                                             int32_t* idx, int B, int n_batches, int without_replacement, int32_t* scratch,
                                             void* stream) {
    if (!h || h->magic != NAF_REPLAY_MAGIC) return NAF_ERR_STATE;
    if (!idx || !scratch || B <= 0 || B > NAF_SAMPLE_BIG_MAX || n_batches <= 0) return NAF_ERR_ARG;
    int bits = 1;
    while ((1 << bits) < 2 * B) ++bits;
    replay_sample_big_kernel<<<n_batches, 1024, 0, (hipStream_t)stream>>>(h->meta, seed, counter_dev, counter_off, idx, B,
                                                                          without_replacement, bits, (int*)scratch,
                                                                          (int64_t)naf_replay_sample_scratch_ints(B));
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}

// ------------------------------------------------------------------------------------------------
// gather (row layout). The OUTPUT is a dense stream of float4: minibatch rows are `out_ld` floats apart (out_ld =
// naf_replay_batch_row_floats: 52 at S=21/A=6, 56 at S=23/A=7 — the row without the ring's padding to two whole
// 128-B lines), so output float4 j belongs to row j / W4, column j % W4 (W4 = out_ld / 4). One lane per output
// float4, RPT of them in flight per lane: stores are perfectly coalesced (consecutive lanes, consecutive 16 B), the
// W4 lanes of a row read the leading W4 * 16 B of its 256-B ring row. Nothing is written that the learner does not
// read: the first version copied whole padded rows (physical traffic 512 B/row against 400 algorithmic = 1.28x); now
// 256 B read (the row's two lines) + 208 B written = 1.15x.
// Algorithmic traffic: 4*(2S+A+2) B read + the same written per row (400 B at S=21/A=6) + 4 B of index.
// ------------------------------------------------------------------------------------------------
typedef float nt_f4 __attribute__((ext_vector_type(4)));   // native vector type the nontemporal builtins accept

template <int RPT, int NT /* bit 0: nontemporal loads, bit 1: nontemporal stores */, int W4C /* 0 = run-time width */>
__global__ __launch_bounds__(256) void replay_gather_rows_kernel(const float4* __restrict__ ring,
                                                                 uint64_t* __restrict__ meta,
                                                                 const int32_t* __restrict__ idx,
                                                                 float4* __restrict__ out, int n, uint64_t cap,
                                                                 int rf4_shift, int w4_rt, int trunc_lo, int trunc_hi) {
    const uint64_t head = meta[META_HEAD];
    const uint64_t size = meta[META_SIZE];
    const uint64_t base = head + cap - size;  // physical position of deque element 0 (oldest)
    const unsigned w4 = W4C ? (unsigned)W4C : (unsigned)w4_rt;
    const int64_t total = (int64_t)n * w4;    // float4 to produce
    const int64_t j0 = (int64_t)blockIdx.x * (256 * RPT) + threadIdx.x;
    int64_t src[RPT];
    int col[RPT];
    bool ok[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int64_t j = j0 + (int64_t)k * 256;
        ok[k] = j < total;
        const unsigned jj = ok[k] ? (unsigned)j : 0u;       // total < 2^31 (checked on the host)
        const unsigned r = jj / w4;
        col[k] = (int)(jj - r * w4);
        int64_t i = (int64_t)idx[r];
        if (i < 0 || (uint64_t)i >= size) {       // EVERY lane: the tail lanes load row idx[0] unconditionally below
            if (ok[k] && col[k] == 0) atomicAdd((unsigned long long*)&meta[META_BAD_IDX], 1ull);
            i = 0;
        }
        // one wrap at most: base < 2 cap and i < size <= cap (a 64-bit modulo here is ~40 instructions per lane)
        uint64_t pos = base + (uint64_t)i;
        pos = pos >= cap ? pos - cap : pos;
        pos = pos >= cap ? pos - cap : pos;
        src[k] = (int64_t)((pos << rf4_shift) + (uint64_t)col[k]);
    }
    float4 v[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        // unconditional (lanes beyond the end re-read row idx[0]): a branch around each load would serialise them
        if (NT & 1) {
            const nt_f4 t = __builtin_nontemporal_load((const nt_f4*)&ring[src[k]]);
            v[k] = make_float4(t.x, t.y, t.z, t.w);
        } else {
            v[k] = ring[src[k]];
        }
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        if (!ok[k]) continue;
        // `.long()` of the reference: truncate the action columns toward zero
        const int f0 = col[k] * 4;
        if (f0 + 3 >= trunc_lo && f0 < trunc_hi) {
            if (f0 + 0 >= trunc_lo && f0 + 0 < trunc_hi) v[k].x = truncf(v[k].x);
            if (f0 + 1 >= trunc_lo && f0 + 1 < trunc_hi) v[k].y = truncf(v[k].y);
            if (f0 + 2 >= trunc_lo && f0 + 2 < trunc_hi) v[k].z = truncf(v[k].z);
            if (f0 + 3 >= trunc_lo && f0 + 3 < trunc_hi) v[k].w = truncf(v[k].w);
        }
        const int64_t j = j0 + (int64_t)k * 256;
        if (NT & 2) {
            nt_f4 t = {v[k].x, v[k].y, v[k].z, v[k].w};
            __builtin_nontemporal_store(t, (nt_f4*)&out[j]);
        } else {
            out[j] = v[k];                            // minibatch-sized launches: the learner reads these rows next
        }
    }
}

extern "C" int naf_replay_batch_row_floats(int S, int A) {
    if (S <= 0 || A <= 0) return NAF_ERR_ARG;
    // [state | action | reward | pad | next_state | done] rounded up to whole float4, and wide enough for the layer-1
    // kernels, which read next_state as 6 or 8 float4 (fused_layers.hip: K4 = 6 up to S = 24, 8 up to S = 32)
    const int k4 = (S + 3) / 4;
    const int k4d = k4 <= 6 ? 6 : (k4 <= 8 ? 8 : k4);
    int w = naf_row_off_done(S, A) + 1;
    const int w2 = naf_row_off_s2(S, A) + 4 * k4d;
    w = naf_round_up(w > w2 ? w : w2, 4);
    const int rf = naf_replay_row_floats(S, A);
    return w < rf ? w : rf;
}

extern "C" int naf_replay_gather_rows(naf_replay_t* h, const int32_t* idx, float* out_rows, int n, int out_ld,
                                      int action_mode, void* stream) {
    if (!h || h->magic != NAF_REPLAY_MAGIC) return NAF_ERR_STATE;
    if (!idx || !out_rows || n < 0 || ((uintptr_t)out_rows & 15) != 0) return NAF_ERR_ARG;
    if (action_mode != NAF_ACTION_TRUNC_INT && action_mode != NAF_ACTION_FLOAT) return NAF_ERR_ARG;
    // out_ld: row stride of the output = number of leading floats copied per row; whole float4, at least the used part
    // of a row, at most the ring's row
    if ((out_ld & 3) != 0 || out_ld < naf_round_up(naf_row_off_done(h->S, h->A) + 1, 4) || out_ld > h->row_floats)
        return NAF_ERR_ARG;
    if (n == 0) return NAF_OK;
    const int w4 = out_ld / 4;
    if ((int64_t)n * w4 >= 0x7fffffffll) return NAF_ERR_ARG;
    const int sh = ilog2_exact(h->row_floats / 4);
    int lo = h->S, hi = h->S + h->A;
    if (action_mode == NAF_ACTION_FLOAT) lo = hi = 0x7fffffff;
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = (int64_t)n * w4;
#define GATHER_LAUNCH(RPTV, NTV, W4V)                                                                                  \
    replay_gather_rows_kernel<RPTV, NTV, W4V><<<blocks, 256, 0, st>>>((const float4*)h->rows, h->meta, idx,            \
                                                                      (float4*)out_rows, n, h->capacity, sh, w4, lo, hi)
#define GATHER_W4(RPTV, NTV)                                                                                           \
    do {                                                                                                               \
        if (w4 == 13) GATHER_LAUNCH(RPTV, NTV, 13);                                                                    \
        else if (w4 == 14) GATHER_LAUNCH(RPTV, NTV, 14);                                                               \
        else if (w4 == 16) GATHER_LAUNCH(RPTV, NTV, 16);                                                               \
        else GATHER_LAUNCH(RPTV, NTV, 0);                                                                              \
    } while (0)
    // small launches: one float4 per lane so that every CU gets a workgroup; bulk launches: 4 in flight per lane
    if (total <= 256 * 8 * 256) {
        const int blocks = (int)((total + 255) / 256);
        GATHER_W4(1, 0);
    } else {
        const int blocks = (int)((total + 1023) / 1024);
        // The gathered rows are written once and not re-read by this kernel: nontemporal STORES keep them from evicting
        // ring lines (measured, 4 Mi rows per launch, interleaved A/B: 977 MiB ring 0.393 -> 0.340 ms, 244 MiB ring
        // 0.346 -> 0.300 ms). Nontemporal LOADS of the ring do not help (0.388 ms) and cost 10 % on a ring that fits the
        // 256 MiB Infinity Cache, so loads stay temporal (template argument NT: bit 0 nontemporal loads, bit 1 stores).
        GATHER_W4(4, 2);
    }
#undef GATHER_W4
#undef GATHER_LAUNCH
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
