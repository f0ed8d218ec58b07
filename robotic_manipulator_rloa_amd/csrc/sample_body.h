// The minibatch draw of the replay sampler (csrc/replay.hip, replay_sample_kernel; csrc/step_path.hip, step_prep_kernel): ONE body
// for the launch that draws the minibatches of a chunk and for the per-timestep launch that appends, draws, gathers and takes the
// moments of one minibatch — the same Philox draws and the same redraw rule, so the indices are the same bits from either.
// Replaces `random.sample(self.memory, k=self.batch_size)` (utils/replay_buffer.py:55).
#pragma once
#include "common.h"

// Draw t of attempt a = mulhi64(philox(ctr, t, a), size). Without replacement: element t redraws while an element j < t holds
// the same value; rounds repeat until no duplicate is left (expected number of redraws ~ B^2 / 2N; population >= 4B so a redraw
// collides with probability <= 1/4 and NAF_SAMPLE_MAX_ROUNDS rounds always suffice in practice). Deterministic in
// (seed, counter, size): the numpy restatement (oracle.replay_sample_indices) reproduces it bit for bit.
#define NAF_SAMPLE_MAX_ROUNDS 64
#ifndef SB_MARK
#define SB_MARK(slot) do { } while (0)    // (timeline hook of an including kernel)
#endif

__device__ static inline int sample_draw(uint64_t ctr, uint32_t t, uint32_t attempt, uint64_t seed, uint64_t size) {
    Philox4 p = philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), t, attempt, (uint32_t)seed, (uint32_t)(seed >> 32));
    uint64_t r = ((uint64_t)p.v[0] << 32) | (uint64_t)p.v[1];
    return (int)__umul64hi(r, size);
}

// LDS the draw of one minibatch of B needs, in ints: the B values + the hash table of the duplicate check (2 x 2^hash_bits),
// at least 4 B (the dense regime holds the whole population, < 4 B). hash_bits = 0: the O(B^2) scan instead of the table.
// Which duplicate check: the hash table (hash_bits > 0) wherever its LDS fits, the O(B^2) scan otherwise (hash_bits = 0). (Tried
// inside the per-timestep launch for the small batches and measured no faster there, rocprofv3 A/B/A/B at B = 64, 11.3 us per launch
// either way: the scan for B <= 128, and a register form for one-wave minibatches — 63 v_readlane compares per round, no LDS, no
// barrier. The launch is a chain of cold-code phases of sixteen waves; the duplicate check is not what bounds it.)
__host__ __device__ static inline int sample_hash_bits(int B) {
    int bits = 1;
    while ((1 << bits) < 2 * B) ++bits;
    return bits;
}
// ints of LDS a draw needs: the B values (+ the table), and never less than 5 B — the dense regime holds the whole population
// (< 4 B) and the B swap targets behind it
__host__ __device__ static inline size_t sample_lds_ints(int B, int hash_bits) {
    size_t n = hash_bits > 0 ? (size_t)B + 2 * ((size_t)1 << hash_bits) : 0;
    return n < (size_t)B * 5 ? (size_t)B * 5 : n;
}

// One minibatch by the `nt` threads of a workgroup (tid = 0 .. nt - 1, nt a multiple of 64, B <= 4 nt): on return vals[0 .. B)
// hold the B deque positions (0 = oldest) and every thread has passed a workgroup barrier behind the last write.
// size == 0: every position is 0 (the gather flags it as a bad index).
__device__ __forceinline__ static void replay_sample_body(int* vals, const int tid, const int nt, const uint64_t size, const uint64_t ctr,
                                                 const uint64_t seed, const int B, const int without_replacement, const int hash_bits) {
    if (size == 0) {
        for (int t = tid; t < B; t += nt) vals[t] = 0;
        __syncthreads();
        return;
    }
    if (without_replacement && size >= (uint64_t)B && size < 4ull * (uint64_t)B) {
        // dense regime (population < 4B, i.e. the first learn() calls after the `len > batch_size` gate,
        // naf_algorithm.py:150): rejection would need ~size rounds, so do a partial Fisher-Yates over the
        // population held in LDS (size < 4B ints fits the 4B-int allocation). Sequential, rare, short.
        // (the B draws do not depend on the permutation: every thread works out its own — ten Philox rounds each — and only the
        //  swaps are sequential; as one thread drawing and swapping in turn this was 45 us per launch at B = 256. The swap targets
        //  sit behind the population: the caller provides >= 5 B ints.)
        int* perm = vals;
        int* tgt = vals + 4 * B;
        for (int t = tid; t < B; t += nt) tgt[t] = t + sample_draw(ctr, (uint32_t)t, 0xFFFFFFFFu, seed, size - (uint64_t)t);
        for (int t = tid; t < (int)size; t += nt) perm[t] = t;
        __syncthreads();
        if (tid == 0) {
            for (int t = 0; t < B; ++t) {
                const int j = tgt[t];
                const int a = perm[t];
                perm[t] = perm[j];
                perm[j] = a;
            }
        }
        __syncthreads();
        return;
    }
    const bool dedupe = without_replacement && size >= (uint64_t)B;
    // elements owned by this thread: t = tid + k * nt, k < 4 (B <= 4096 at nt = 1024)
    uint32_t attempt[4] = {0, 0, 0, 0};
    for (int k = 0, t = tid; t < B; t += nt, ++k) vals[t] = sample_draw(ctr, (uint32_t)t, 0u, seed, size);
    __syncthreads();
    SB_MARK(13);
    if (dedupe && hash_bits > 0) {
        // "some earlier element holds the same value" through a hash table in LDS instead of a scan: O(B) per round where
        // the scan is O(B^2) — 92 us per launch at B = 2048, 30 at B = 1024, on the critical path of every vector step.
        // Open addressing over M = 2^hash_bits >= 2 B slots: keys[] (the value, claimed by compare-and-swap) and tmin[] (the
        // smallest element index holding it, atomic min): element t is a duplicate iff tmin of its value < t. The same rule,
        // so the same indices bit for bit (oracle.replay_sample_indices), whatever order the lanes insert in. Rebuilt every
        // round (a redrawn element's old value must not linger).
        const int M = 1 << hash_bits;
        int* keys = vals + B;
        int* tmin = keys + M;
        for (int round = 0; round < NAF_SAMPLE_MAX_ROUNDS; ++round) {
            for (int e = tid; e < M; e += nt) {
                keys[e] = -1;
                tmin[e] = 0x7fffffff;
            }
            __syncthreads();
            int slot[4] = {0, 0, 0, 0};
            for (int k = 0, t = tid; t < B; t += nt, ++k) {
                const int mine = vals[t];
                unsigned h = ((unsigned)mine * 2654435761u) >> (32 - hash_bits);
                for (int probe = 0; probe < M; ++probe) {          // (load factor <= 1/2: a free or matching slot exists)
                    const int was = atomicCAS(&keys[h], -1, mine);
                    if (was == -1 || was == mine) break;
                    h = (h + 1) & (unsigned)(M - 1);
                }
                atomicMin(&tmin[h], t);
                slot[k] = (int)h;
            }
            __syncthreads();
            SB_MARK(14);
            int dupmask = 0;
            for (int k = 0, t = tid; t < B; t += nt, ++k)
                if (tmin[slot[k]] < t) dupmask |= (1 << k);
            int any = __syncthreads_or(dupmask);  // also orders the reads above before the writes below
            SB_MARK(15);
            if (!any) break;
            for (int k = 0, t = tid; t < B; t += nt, ++k) {
                if (dupmask & (1 << k)) {
                    attempt[k] += 1u;
                    vals[t] = sample_draw(ctr, (uint32_t)t, attempt[k], seed, size);
                }
            }
            __syncthreads();
        }
    } else if (dedupe) {
        for (int round = 0; round < NAF_SAMPLE_MAX_ROUNDS; ++round) {
            int dupmask = 0;
            for (int k = 0, t = tid; t < B; t += nt, ++k) {
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
            for (int k = 0, t = tid; t < B; t += nt, ++k) {
                if (dupmask & (1 << k)) {
                    attempt[k] += 1u;
                    vals[t] = sample_draw(ctr, (uint32_t)t, attempt[k], seed, size);
                }
            }
            __syncthreads();
        }
    }
    __syncthreads();
}
