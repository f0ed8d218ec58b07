// The per-timestep path of the reference's training loop (NAFAgent.step -> act, naf_algorithm.py:129-178, :249-261) as the
// reference runs it — ONE environment, one transition, one minibatch, one update per timestep — for gfx950.
//
// With every launch of a chunk of updates serving one update only, the timestep was 12 launches (counted append, sample, counter,
// gather, moments, the five of the row-split chain, the optimizer step, act()) and the launch boundaries between them were a third
// of its device time. Two launches take seven of those:
//
//   step_prep_kernel      ReplayBuffer.add of the timestep's transition (utils/replay_buffer.py:32-45; read from pinned host memory,
//                         counted by a pinned word: 0 = this tick brings no row) + random.sample (:55) + the stacking of the
//                         minibatch (:57-65) + the moments record of layer 1's inputs (csrc/moments_body.h) — ONE workgroup of 1024
//                         threads: the draw and the duplicate check live in LDS anyway, the rows of a minibatch are 13 - 52 KB,
//                         and nothing here is bound by anything but dependent latencies. The draw and the moments are the bodies
//                         the chunked launches run (sample_body.h, moments_body.h): same indices, same record, bit for bit.
//   adam_act_kernel       clip_grad_norm_ + Adam.step + soft_update (naf_algorithm.py:209-213, :217-226) of the timestep's update
//                         AND NAFAgent.act (:158-178) for the next timestep's state in one launch: the workgroups that step a
//                         slice of the parameters HOLD THE NEW WEIGHTS IN REGISTERS, so they also multiply them with the policy's
//                         activations — the eval-mode forward costs no second pass over the 330 KB of weights (as a launch of its
//                         own, one workgroup streaming them fresh from another XCD's write-back, act() was the longest kernel of
//                         the timestep: 8.6 - 10.7 us). The layers' activations (256 + 256 floats) cross workgroups as
//                         self-validating (value, epoch) records written and polled with sc1 accesses, the protocol of
//                         csrc/bn2bwd_fold.h; every poll is bounded by wall clock and a bound that runs out is counted where the
//                         host sees it and turns the action into NaN (the host raises) — no wave can wait forever.
//
// ... and one more workgroup of adam_act_kernel runs step_prep's body for the NEXT timestep (step_prep_body, MODE 1 / 2): what a
// timestep draws depends on its own transition only through the ring's fill level and — if the draw picks it — the row, so the
// minibatch can be there before the transition is. With it there, so can the gradient: on one GPU the timestep's graph is
// adam_act_kernel (append, the waiting gradient's optimizer step, act(), prefetch) followed by the chain for the next update, six
// launches of which the host waits for the first (engine.TrainChunk).
#include <string.h>
#include "common.h"
#include "../../include/naf_hip.h"
NAF_TL_DECL(g_tl_sp);
#ifdef NAF_TIMELINE
// (the stand-alone launch leaves both rows of marks; inside adam_act_kernel the prefetching workgroup is the launch's LAST one and
//  leaves the second row — benchmarks/step_timeline.py knows which is which)
#define BM_MARK(slot) NAF_TL(g_tl_sp, NAF_TL_STEP_PREP, slot)
#define SB_MARK(slot) NAF_TL(g_tl_sp, NAF_TL_STEP_PREP, slot)
#endif
#include "act_body.h"
#include "adam_body.h"
#include "head_body.h"
#ifdef NAF_TIMELINE
// (layer 1 riding on adam_act_kernel leaves its marks in this file's table, row NAF_TL_BB_LAYER1: naf_timeline_read(2048 + 0))
#define BB_L1_TL(slot, is_first, is_last) NAF_TL_FL(g_tl_sp, NAF_TL_BB_LAYER1, slot, is_first, is_last)
#define BB_L1_TL_T(slot, is_first, is_last, thread) NAF_TL_FL_T(g_tl_sp, NAF_TL_BB_LAYER1, slot, is_first, is_last, thread)
#endif
#include "layer1_body.h"      // bb_layer1_impl: layer 1 of the row-split chain, riding on adam_act_kernel (round 6)
#include "moments_body.h"
#include "replay_dev.h"
#include "sample_body.h"

NAF_TL_READER(naf_tl_read_sp, g_tl_sp)

// =====================================================================================================================
// step_prep_kernel
// =====================================================================================================================
#define SP_THREADS 1024
#define SP_RPT 4
#define SP_CACHE_ROWS 256
#define SP_CACHE_BYTES 65536   // ... and at most 64 KB of them (ring rows of 128 floats — state sizes beyond 26, 9 .. 11 joints — gather
                               //     72 .. 88 floats per row: 256 of those would not fit beside the draw's table)
#define SP_MAX_RF4 32          // a ring row is at most 128 floats here
static inline bool sp_cached(int B, int out_ld) { return B <= SP_CACHE_ROWS && (size_t)B * out_ld * sizeof(float) <= SP_CACHE_BYTES; }

struct StepPrepArgs {
    float4* ring;
    uint64_t* meta;
    uint64_t cap;
    int rf4_shift;
    const float4* src_row;     // nullable: pinned host (or device) row of this timestep's transition
    const int32_t* n_word;     // nullable: 0 / 1 rows to append (read where the kernel runs)
    float4* row_out;           // nullable: the row as read, in device memory (whatever the count says)
    uint64_t seed;
    uint64_t* counter;         // the sampler's stream position: read, advanced by one
    int32_t* idx_out;          // nullable: the B deque positions drawn
    float4* out_rows;          // the minibatch, w4 float4 per row
    int w4, trunc_lo, trunc_hi, off_s2_4;
    float* mom;                // [2][KP + KP * KP]
    int B, without_replacement, hash_bits;
    // the prefetch of the NEXT timestep's minibatch (see step_prep_body): its record and the positions it drew
    int32_t* spec_rec;         // nullable: SP_REC_INTS ints — MODE 0: the record to check; MODE 1 / 2: the record this prefetch LEAVES
    int32_t* idx_spec;         // nullable [B] — ... and the positions it drew
    int32_t* spec_rec_in;      // MODE 2: the record / positions of the minibatch THIS timestep consumes (at depth 1 the same buffers
    int32_t* idx_spec_in;      //         as the ones it leaves; at depth 2 the ones left two launches ago); MODE 0: == spec_rec / idx_spec
    int depth;                 // MODE 1 / 2: how many appends lie between the ring as found (MODE 2: as left) and the draw: 1 | 2
    uint32_t* pf_seq;          // nullable device word owned by the prefetching workgroup: the ordinal its verdicts carry
    // the pipelined form (step_prep_body, MODE 2) and the learner's working / public state (`copies`)
    uint32_t* host_spec;       // nullable pinned host words {ordinal, valid}: the prefetch's verdict where the host reads it
    uint64_t* pipe_errors;     // nullable pinned host word: MODE 2 launches whose record did not hold (a host-side logic error)
    const unsigned* cp_src[3];
    unsigned* cp_dst[3];
    int cp_n[3];
};
#define SP_COPY_WPT 4          // words per thread and copy: ranges of up to 4096 words (the BatchNorm statistics of both nets at layer size 512)
// the prefetch's record: what it assumed — the ring as ONE more append leaves the state it found, the sampler's stream position —
// and whether the minibatch it left in out_rows / mom / idx_spec can stand for the one the next timestep's launch would draw
// (+ two counters for the host: timesteps that took the prefetched minibatch / that drew for themselves)
enum { SP_REC_VALID = 0, SP_REC_B, SP_REC_CTR, SP_REC_HEAD = 4, SP_REC_SIZE = 6, SP_REC_TAKEN = 8, SP_REC_DRAWN = 9, SP_REC_INTS = 12 };
static_assert(SP_REC_INTS == NAF_STEP_SPEC_INTS, "include/naf_hip.h");

// the leading floats of a ring row as the learner sees them: `.long()` of the reference on the action columns
__device__ __forceinline__ static float4 sp_trunc(float4 v, int f0, int lo, int hi) {
    if (f0 + 3 >= lo && f0 < hi) {
        if (f0 + 0 >= lo && f0 + 0 < hi) v.x = truncf(v.x);
        if (f0 + 1 >= lo && f0 + 1 < hi) v.y = truncf(v.y);
        if (f0 + 2 >= lo && f0 + 2 < hi) v.z = truncf(v.z);
        if (f0 + 3 >= lo && f0 + 3 < hi) v.w = truncf(v.w);
    }
    return v;
}

// The body of the launch, and of the PREFETCH that adam_act_kernel's extra workgroup runs for the next timestep (SPEC):
//
// What a timestep draws depends on the transition it appends in two ways only — the ring's fill level, and the row itself IF the
// draw picks it (probability B / fill: 0.06 % at B = 64 in a ring of 1e5 rows). So the last launch of timestep t can already
// draw, gather and take the moments of timestep t + 1's minibatch from the ring "as one more append will leave it", beside its
// own work and off the host's critical path (the launch announces its action to the host before this workgroup is done, and the
// host then steps the environment). The prefetch commits nothing — not the ring's counters, not the sampler's stream position
// — and leaves a record of what it assumed; timestep t + 1's launch checks the record against what it finds (a row to append,
// the same {head, size}, the same stream position, the new row not among the positions drawn) and, if it holds, appends the row,
// advances the counters and is done: 1 us instead of 10. If not — the first timestep, an idle tick of a data-parallel run, the new
// row drawn, the ring touched in between — it does everything itself as before. Either way the minibatch, its moments and the
// indices are the bits the chunked launches produce: the same bodies on the same state.
//
// CACHE: B <= SP_CACHE_ROWS (a kernel of its own: with the choice made at run time every load of the moments' staging sat behind
// a branch with both forms' code around it, and a cold instruction stream is what these few microseconds are made of)
// MODE 0: the timestep's own launch (append, then draw — or take what the prefetch left).
// MODE 1: the prefetch alone (SPEC): the ring as ONE more append will leave it; commits nothing.
// MODE 2: the timestep's append AND the prefetch for the next one in the same workgroup — the PIPELINED form of the path
//         (engine.TrainChunk): the host has read the previous prefetch's verdict and launches a graph that starts with
//         adam_act_kernel; this workgroup appends the row (its minibatch is in place, and so is the gradient the chain has taken
//         from it while the host stepped the environment), hands the prefetched indices over, and prefetches again. It checks the
//         record all the same: a mismatch here is a host-side logic error and is counted where the host raises.
// DEPTH (P.depth, MODE 1 / 2; round 6): how many appends lie between the ring as the prefetch finds (MODE 2: leaves) it and the draw.
//   1: the next timestep's minibatch, as above. 2: the one BEHIND it — the ring as two more appends will leave it, the stream
//   position one draw further on, void if either of the two rows to come is among the positions drawn (2 B / fill). With the
//   minibatch of timestep t + 1 in place before timestep t's graph starts, that graph's chain does not wait for this workgroup any
//   more: the depth-2 prefetch is a launch of its OWN (step_prefetch_kernel) on a second stream, beside the graph (a fork and a join
//   inside a hipGraph cost 30 us on this runtime: benchmarks/probe/graph_branch.hip). Three sets of {minibatch, moments, indices,
//   record} rotate: the one timestep t consumes (spec_rec_in / idx_spec_in), the one the chain of its graph reads, the one this
//   launch fills. Ordering across the two streams is by construction: the host launches the graph of timestep t + 1 only after it has
//   read this launch's verdict, and the verdict is stored behind a release of everything the workgroup wrote.
// `copies`: up to three word ranges copied at the start (MODE 0: public -> working state of the learner, the reset that discards
// what a chain run on a prefetch that did not hold has left; MODE 1 / 2: working -> public, the commit of the update the launch's
// other workgroups apply). 4-byte words, ranges that no other workgroup of the launch writes.
template <int K4, bool CACHE, int MODE>
__device__ __forceinline__ static void step_prep_body(const StepPrepArgs& P, unsigned char* sp_smem, float4* sNew, int* sHit) {
    constexpr bool SPEC = MODE != 0;                    // the draw is for the NEXT timestep
    constexpr bool APPEND = MODE != 1;                  // reads [row | count] and appends for THIS timestep
    constexpr int KP = 4 * K4, REC = KP + KP * KP;
    BmShared* S = (BmShared*)sp_smem;                                  // [2]: one per net; the draw's table lives here first
    int* sPos = (int*)(sp_smem + 2 * sizeof(BmShared));                // [B]: physical ring rows of the minibatch
    // up to SP_CACHE_ROWS rows the gathered minibatch also stays in LDS, where the moments take it from (behind sPos, 16-B aligned)
    float4* sRows = (float4*)(sp_smem + 2 * sizeof(BmShared) + (size_t)naf_round_up(P.B, 4) * sizeof(int));
    // Small batches (the per-timestep shapes): everything the launch leaves in memory is stored in one go at its end, from LDS; the
    // appended row is taken from LDS by whoever draws it. (Tried for speed — on the theory that a store in flight holds up the
    // barriers behind it — and measured neutral: hipcc's workgroup barrier on gfx950 waits for LDS traffic only, s_waitcnt
    // lgkmcnt(0). Kept: the phases between the draw and the moments then touch memory for loads only.)
    constexpr bool cache = CACHE;
    int* vals = (int*)sp_smem;
    const int tid = threadIdx.x, B = P.B;
    const int rf4 = 1 << P.rf4_shift;
    // the ordinal this workgroup's verdict will carry: a word nobody else writes (round 5 derived it from adam_act_kernel's epoch,
    // which the launch's last workgroup rewrites without waiting for this one)
    const unsigned pf_epoch = (SPEC && P.pf_seq) ? *P.pf_seq + 1u : 0u;
    NAF_TL(g_tl_sp, NAF_TL_STEP_PREP, 0);

    // ---- ReplayBuffer.add: 0 or 1 rows, from (pinned host) memory -------------------------------------------------------
    int n = 0;
    float4 row4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool has_row = APPEND && P.n_word && P.src_row;
    if (has_row) {
        // system-scope loads: pinned host memory, or device memory the HOST has stored into (naf_host_publish) — this XCD's L2
        // may hold what the previous launch read there
        n = (int)__builtin_amdgcn_raw_buffer_load_b32(naf_buf(P.n_word, 4), 0, 0, 17);
        n = n < 0 ? 0 : (n > 1 ? 1 : n);
        if (tid < rf4) {                               // (unconditional: flies beside the count)
            const naf_u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(naf_buf(P.src_row), 16u * (unsigned)tid, 0, 17);
            const unsigned r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];      // (scalar copies: a bit_cast of a vector ELEMENT reads element 0)
            row4 = make_float4(__builtin_bit_cast(float, r0), __builtin_bit_cast(float, r1), __builtin_bit_cast(float, r2),
                               __builtin_bit_cast(float, r3));
        }
    }
    uint64_t head = P.meta[META_HEAD], size = P.meta[META_SIZE], total = P.meta[META_TOTAL];
    uint64_t ctr = *P.counter;
    // (the copies' words: requested with everything else)
    unsigned cpw[3][SP_COPY_WPT];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < SP_COPY_WPT; ++k) {
            const int w = tid + SP_THREADS * k;
            cpw[c][k] = (P.cp_src[c] && w < P.cp_n[c]) ? P.cp_src[c][w] : 0u;
        }
    // the prefetch's record, as the previous timestep's last launch left it (a launch boundary ago: plain loads)
    bool take = false;
    int ispec[4] = {0, 0, 0, 0};                        // (its indices: requested beside the record, whether they will be wanted or not)
    if (APPEND && P.spec_rec_in) {
        if (P.idx_out && P.idx_spec_in) {
#pragma unroll
            for (int k = 0; k < 4; ++k) ispec[k] = P.idx_spec_in[tid + SP_THREADS * k < B ? tid + SP_THREADS * k : 0];
        }
        const int4 ra = ((const int4*)P.spec_rec_in)[0], rb = ((const int4*)P.spec_rec_in)[1];
        const uint64_t r_ctr = (uint64_t)(uint32_t)ra.z | ((uint64_t)(uint32_t)ra.w << 32);
        const uint64_t r_head = (uint64_t)(uint32_t)rb.x | ((uint64_t)(uint32_t)rb.y << 32);
        const uint64_t r_size = (uint64_t)(uint32_t)rb.z | ((uint64_t)(uint32_t)rb.w << 32);
        take = n == 1 && ra.x == 1 && ra.y == B && r_ctr == ctr && r_head == head && r_size == size;      // (uniform)
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < SP_COPY_WPT; ++k) {
            const int w = tid + SP_THREADS * k;
            if (P.cp_dst[c] && w < P.cp_n[c]) P.cp_dst[c][w] = cpw[c][k];
        }
    uint64_t head2 = n ? (head + 1 == P.cap ? 0 : head + 1) : head;
    uint64_t size2 = size + (uint64_t)n;
    size2 = size2 > P.cap ? P.cap : size2;
    int newpos = n ? (int)head : -1;                    // physical row the append fills
    if (APPEND && tid < rf4) sNew[tid] = row4;
    auto store_row_and_counters = [&]() {
        if (n && tid < rf4) P.ring[(head << P.rf4_shift) + tid] = row4;
        if (P.row_out && has_row && tid < rf4) P.row_out[tid] = row4;
        if (tid == 0) {
            if (n) {
                P.meta[META_HEAD] = head2;
                P.meta[META_SIZE] = size2;
                P.meta[META_TOTAL] = total + 1ull;
            }
            *P.counter = ctr + 1;
            if (P.spec_rec_in) {
                P.spec_rec_in[SP_REC_VALID] = 0;               // (a record serves one timestep)
                P.spec_rec_in[take ? SP_REC_TAKEN : SP_REC_DRAWN] += 1;
            }
        }
    };
    if (APPEND && (take || MODE == 2)) {
        // the minibatch, its moments and its indices are in place: the append and the counters are all that is left
        if (MODE == 2 && !take && tid == 0 && P.pipe_errors)
            __hip_atomic_fetch_add((unsigned long long*)P.pipe_errors, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __syncthreads();                                // every thread has read {head, size, counter, record} before thread 0 rewrites them
        store_row_and_counters();
        if (P.idx_out && P.idx_spec_in) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (tid + SP_THREADS * k < B) P.idx_out[tid + SP_THREADS * k] = ispec[k];
        }
        if (MODE == 0) {
            NAF_TL(g_tl_sp, NAF_TL_STEP_PREP, 6);
            return;
        }
        // ... and on to the next timestep's minibatch, on the ring as it now is (the stores above have landed before the first of the
        // draw's barriers lets anyone read the row back: the gather may well draw it)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        head = head2;
        size = size2;
        total += (uint64_t)n;
        ctr += 1;
    }
    int newpos1 = -1;                                   // (depth 2: the second row that does not exist yet)
    uint64_t rec_ctr = ctr, rec_head = head, rec_size = size;      // what the launch that CONSUMES the prefetch must find before its append
    if (SPEC) {
        n = 1;                                          // the ring as the next append will leave it
        head2 = head + 1 == P.cap ? 0 : head + 1;
        size2 = size + 1 > P.cap ? P.cap : size + 1;
        newpos = (int)head;
        if (P.depth == 2) {                             // ... and the one behind it: another timestep's append and draw lie in between
            rec_ctr = ctr + 1;
            rec_head = head2;
            rec_size = size2;
            newpos1 = (int)head2;
            head2 = head2 + 1 == P.cap ? 0 : head2 + 1;
            size2 = size2 + 1 > P.cap ? P.cap : size2 + 1;
        }
        if (tid == 0) *sHit = 0;                        // (visible behind the draw's barriers)
    }
    if (!SPEC && !cache) {
        __syncthreads();                                // every thread has read {head, size, counter} before thread 0 rewrites them
        store_row_and_counters();
    }
    NAF_TL(g_tl_sp, NAF_TL_STEP_PREP, 1);

    // ---- random.sample: the chunked sampler's body on the ring as the append leaves it ---------------------------------------
    replay_sample_body(vals, tid, SP_THREADS, size2, SPEC ? rec_ctr : ctr, P.seed, B, P.without_replacement, P.hash_bits);
    NAF_TL(g_tl_sp, NAF_TL_STEP_PREP, 2);
    const uint64_t base = head2 + P.cap - size2;        // physical position of deque element 0 (oldest)
    int mypos[4], myidx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int t = tid + SP_THREADS * k;
        int64_t i = t < B ? (int64_t)vals[t] : 0;
        myidx[k] = (int)i;
        bool bad = i < 0 || (uint64_t)i >= size2;
        if (!SPEC && bad && t < B) atomicAdd((unsigned long long*)&P.meta[META_BAD_IDX], 1ull);
        if (SPEC && bad && t < B) *sHit = 1;            // (left to the launch that counts it)
        if (bad) i = 0;
        uint64_t pos = base + (uint64_t)i;
        pos = pos >= P.cap ? pos - P.cap : pos;
        pos = pos >= P.cap ? pos - P.cap : pos;
        mypos[k] = (int)pos;
        if (SPEC && t < B && (mypos[k] == newpos || mypos[k] == newpos1)) *sHit = 1;       // a row that does not exist yet
    }
    // (sPos is LDS of its own; the draw's table is dead and becomes the moments' staging area behind the barriers below)
    int32_t* const idx_dst = SPEC ? P.idx_spec : P.idx_out;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int t = tid + SP_THREADS * k;
        if (t < B) {
            sPos[t] = mypos[k];
            if (!cache && idx_dst) idx_dst[t] = myidx[k];
        }
    }
    __syncthreads();                                    // (!cache: the appended row's store has completed too — a workgroup-scope release)
    NAF_TL(g_tl_sp, NAF_TL_STEP_PREP, 3);
    // the verdict for the host (pinned words, system scope: [1] = does the prefetch hold, [0] = this workgroup's ordinal). Thread 0,
    // behind a barrier every wave reaches with its stores acknowledged: the host launches work on ANOTHER stream on the strength of
    // this word (the graph whose chain reads the minibatch; the next prefetch, which reads the ring and the counters), so
    // everything the workgroup wrote is released to the device first.
    auto tell_host = [&](int valid) {
        if (P.pf_seq) *P.pf_seq = pf_epoch;
        if (P.host_spec) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            // ONE 8-byte store {ordinal, valid}: whole in host memory or not at all (two words could pass one another on the way)
            __hip_atomic_store((unsigned long long*)P.host_spec, ((unsigned long long)(unsigned)valid << 32) | (unsigned long long)pf_epoch,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    };
    if (SPEC && *sHit) {                                // (uniform) nothing usable: say so and go
        if (tid == 0 && P.spec_rec) P.spec_rec[SP_REC_VALID] = 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) tell_host(0);
        return;
    }

    // ---- the minibatch rows: one lane per output float4, SP_RPT in flight ---------------------------------------------------
    const unsigned w4 = (unsigned)P.w4;
    const int total4 = B * P.w4;
    for (int j0 = tid; j0 < total4; j0 += SP_THREADS * SP_RPT) {
        float4 v[SP_RPT];
        int col[SP_RPT], pos[SP_RPT];
#pragma unroll
        for (int k = 0; k < SP_RPT; ++k) {
            const int j = j0 + SP_THREADS * k;
            const unsigned jj = j < total4 ? (unsigned)j : 0u;
            const unsigned r = jj / w4;
            col[k] = (int)(jj - r * w4);
            pos[k] = sPos[r];
            v[k] = P.ring[((int64_t)pos[k] << P.rf4_shift) + col[k]];
        }
#pragma unroll
        for (int k = 0; k < SP_RPT; ++k) {
            const int j = j0 + SP_THREADS * k;
            if (j < total4) {
                if (!SPEC && cache && pos[k] == newpos) v[k] = sNew[col[k]];        // (the row this launch appends: not in the ring yet)
                const float4 t = sp_trunc(v[k], 4 * col[k], P.trunc_lo, P.trunc_hi);
                if (cache) sRows[j] = t;
                else P.out_rows[j] = t;
            }
        }
    }

    NAF_TL(g_tl_sp, NAF_TL_STEP_PREP, 4);
    // ---- the moments of layer 1's inputs: threads 0 .. 511 the states (net 0), 512 .. 1023 the next states (net 1) ------------
    const int net = tid >> 9;
    const int c0 = net ? P.off_s2_4 : 0;
    // (a record is K4 float4 wide from the net's first column: where that runs past the minibatch row — S = 26 with A >= 2 — the
    //  chunked launch reads on into the NEXT minibatch row, zeros behind the last one; columns beyond the state size meet zero
    //  weights, but the record is the same bits here)
    bb_moments_body<K4>(
        [&](int row, int q) {
            int c = c0 + q;
            if (c >= P.w4) {
                c -= P.w4;
                row += 1;
            }
            const int rr = row < B ? row : 0;
            float4 v = cache ? sRows[rr * P.w4 + c]
                             : sp_trunc(P.ring[((int64_t)sPos[rr] << P.rf4_shift) + c], 4 * c, P.trunc_lo, P.trunc_hi);
            if (row >= B) v = make_float4(0.f, 0.f, 0.f, 0.f);
            return v;
        },
        S[net], tid & (BM_THREADS - 1), P.mom + net * REC, B);
    NAF_TL(g_tl_sp, NAF_TL_STEP_PREP, 5);
    if (cache) {
        // everything this launch leaves in memory, in one go (the moments' records left just above)
        if (!SPEC) store_row_and_counters();
        for (int j = tid; j < total4; j += SP_THREADS) P.out_rows[j] = sRows[j];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int t = tid + SP_THREADS * k;
            if (t < B && idx_dst) idx_dst[t] = myidx[k];
        }
    }
    if (SPEC) {
        if (tid == 0 && P.spec_rec) {
            // what was assumed: the state the launch that consumes this minibatch must find before its append — at depth 1 the state
            // this workgroup found (MODE 2: left), at depth 2 that state one append and one draw further on
            int4 ra, rb;
            ra.x = 1;
            ra.y = B;
            ra.z = (int)(uint32_t)rec_ctr;
            ra.w = (int)(uint32_t)(rec_ctr >> 32);
            rb.x = (int)(uint32_t)rec_head;
            rb.y = (int)(uint32_t)(rec_head >> 32);
            rb.z = (int)(uint32_t)rec_size;
            rb.w = (int)(uint32_t)(rec_size >> 32);
            ((int4*)P.spec_rec)[1] = rb;
            ((int4*)P.spec_rec)[0] = ra;
        }
        if (P.host_spec || P.pf_seq) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) tell_host(1);
        }
    }
    NAF_TL(g_tl_sp, NAF_TL_STEP_PREP, 6);
}

template <int K4, bool CACHE>
__global__ __launch_bounds__(SP_THREADS) void step_prep_kernel(const StepPrepArgs P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_smem[];
    __shared__ float4 sNew[SP_MAX_RF4];                                // the appended row (rf4 <= 32 float4)
    __shared__ int sHit;
    step_prep_body<K4, CACHE, 0>(P, sp_smem, sNew, &sHit);
}

// The prefetch as a launch of its own (round 6): MODE 1 = the prefetch alone, MODE 2 = the timestep's append, the hand-over of the
// indices of the minibatch it consumes, and the prefetch on the ring as that leaves it. One workgroup, launched on a stream of its
// own beside the timestep's graph (naf_step_launch) or inside the graph that starts a timestep over.
template <int K4, bool CACHE, int MODE>
__global__ __launch_bounds__(SP_THREADS) void step_prefetch_kernel(const StepPrepArgs P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_smem[];
    __shared__ float4 sNew[SP_MAX_RF4];
    __shared__ int sHit;
    step_prep_body<K4, CACHE, MODE>(P, sp_smem, sNew, &sHit);
}

// arguments of a launch of step_prep_body, checked; lds = its dynamic LDS
static int sp_build(naf_replay_t* h, const float* src_row, const int32_t* n_word, float* row_out, uint64_t seed, uint64_t* counter_dev,
                    int32_t* idx_out, float* out_rows, int out_ld, int action_mode, float* mom, int B, int without_replacement,
                    int32_t* spec_rec, int32_t* idx_spec, const naf_step_copies_t* copies, uint32_t* host_spec, uint64_t* pipe_errors,
                    int32_t* spec_rec_in, int32_t* idx_spec_in, int depth, uint32_t* pf_seq, StepPrepArgs& P, size_t& lds, int& k4) {
    if (!h || h->magic != NAF_REPLAY_MAGIC) return NAF_ERR_STATE;
    if (!counter_dev || !out_rows || !mom || B <= 0 || B > 4096 || (((uintptr_t)out_rows | (uintptr_t)mom) & 15) != 0)
        return NAF_ERR_ARG;
    if ((src_row == nullptr) != (n_word == nullptr) || ((uintptr_t)src_row & 15) != 0 || ((uintptr_t)n_word & 3) != 0) return NAF_ERR_ARG;
    if (((uintptr_t)row_out & 15) != 0 || (row_out && !src_row)) return NAF_ERR_ARG;
    if (((uintptr_t)spec_rec & 15) != 0 || ((uintptr_t)idx_spec & 3) != 0 || ((uintptr_t)host_spec & 7) != 0) return NAF_ERR_ARG;
    if (((uintptr_t)spec_rec_in & 15) != 0 || ((uintptr_t)idx_spec_in & 3) != 0 || ((uintptr_t)pf_seq & 3) != 0) return NAF_ERR_ARG;
    if ((depth != 1 && depth != 2) || (host_spec && !pf_seq)) return NAF_ERR_ARG;      // (a verdict carries the ordinal pf_seq counts)
    if (action_mode != NAF_ACTION_TRUNC_INT && action_mode != NAF_ACTION_FLOAT) return NAF_ERR_ARG;
    if ((out_ld & 3) != 0 || out_ld < naf_round_up(naf_row_off_done(h->S, h->A) + 1, 4) || out_ld > h->row_floats) return NAF_ERR_ARG;
    k4 = (h->S + 3) / 4;
    if (h->row_floats > 4 * SP_MAX_RF4) return NAF_ERR_ARG;   // (the appended row is held in 32 float4 of LDS)
    if (k4 > 8 || out_ld != naf_replay_batch_row_floats(h->S, h->A)) return NAF_ERR_ARG;   // (the row the learner's kernels and the moments expect)
    P.ring = (float4*)h->rows;
    P.meta = h->meta;
    P.cap = h->capacity;
    int sh = 0;
    while ((1 << sh) < h->row_floats / 4) ++sh;
    P.rf4_shift = sh;
    P.src_row = (const float4*)src_row;
    P.n_word = n_word;
    P.row_out = (float4*)row_out;
    P.seed = seed;
    P.counter = counter_dev;
    P.idx_out = idx_out;
    P.out_rows = (float4*)out_rows;
    P.w4 = out_ld / 4;
    P.trunc_lo = h->S;
    P.trunc_hi = h->S + h->A;
    if (action_mode == NAF_ACTION_FLOAT) P.trunc_lo = P.trunc_hi = 0x7fffffff;
    P.off_s2_4 = naf_row_off_s2(h->S, h->A) / 4;
    P.mom = mom;
    P.B = B;
    P.without_replacement = without_replacement;
    P.hash_bits = sample_hash_bits(B);
    P.spec_rec = spec_rec;
    P.idx_spec = idx_spec;
    P.spec_rec_in = spec_rec_in;
    P.idx_spec_in = idx_spec_in;
    P.depth = depth;
    P.pf_seq = pf_seq;
    P.host_spec = host_spec;
    P.pipe_errors = pipe_errors;
    for (int c = 0; c < 3; ++c) {
        P.cp_src[c] = nullptr;
        P.cp_dst[c] = nullptr;
        P.cp_n[c] = 0;
        if (copies && copies->n_words[c] > 0) {
            if (!copies->src[c] || !copies->dst[c] || copies->n_words[c] > SP_COPY_WPT * SP_THREADS ||
                (((uintptr_t)copies->src[c] | (uintptr_t)copies->dst[c]) & 3) != 0)
                return NAF_ERR_ARG;
            P.cp_src[c] = (const unsigned*)copies->src[c];
            P.cp_dst[c] = (unsigned*)copies->dst[c];
            P.cp_n[c] = copies->n_words[c];
        }
    }
    size_t draw = sample_lds_ints(B, P.hash_bits) * sizeof(int);
    lds = 2 * sizeof(BmShared);
    if (draw > lds) return NAF_ERR_ARG;                  // (cannot happen for B <= 4096: 81,920 <= 82,176)
    lds += (size_t)naf_round_up(B, 4) * sizeof(int);
    if (sp_cached(B, out_ld)) lds += (size_t)B * out_ld * sizeof(float);     // the minibatch itself (<= 64 KB)
    return NAF_OK;
}
#define SP_MAX_DYN_LDS (146 * 1024)                      // 82,176 + 1,024 + 65,536 = 148,736 at most; adam_act_kernel's own arrays (up to
                                                         // 11.9 KB with the 16-lane noise body's tile) beside it inside the CU's 160 KB

extern "C" int naf_step_prep(naf_replay_t* h, const float* src_row, const int32_t* n_word, float* row_out, uint64_t seed,
                             uint64_t* counter_dev, int32_t* idx_out, float* out_rows, int out_ld, int action_mode, float* mom, int B,
                             int without_replacement, int32_t* spec_rec, const int32_t* idx_spec, const naf_step_copies_t* copies,
                             void* stream) {
    StepPrepArgs P;
    size_t lds = 0;
    int k4 = 0;
    if ((spec_rec == nullptr) != (idx_spec == nullptr) && idx_out) return NAF_ERR_ARG;     // (a record without its indices: only if nobody asks for them)
    const int rc = sp_build(h, src_row, n_word, row_out, seed, counter_dev, idx_out, out_rows, out_ld, action_mode, mom, B,
                            without_replacement, spec_rec, (int32_t*)idx_spec, copies, nullptr, nullptr, spec_rec, (int32_t*)idx_spec, 1,
                            nullptr, P, lds, k4);
    if (rc != NAF_OK) return rc;
    static int raised_dev[64];                           // per device: the kernels' dynamic-LDS limit raised once
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!raised_dev[dev]) {
        const void* ks[4] = {(const void*)step_prep_kernel<6, true>, (const void*)step_prep_kernel<6, false>,
                             (const void*)step_prep_kernel<8, true>, (const void*)step_prep_kernel<8, false>};
        for (const void* k : ks) {
            hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, SP_MAX_DYN_LDS);
            if (e != hipSuccess) return (int)e;
        }
        raised_dev[dev] = 1;
    }
    const bool cache = sp_cached(B, out_ld);
    if (k4 <= 6 && cache) step_prep_kernel<6, true><<<1, SP_THREADS, lds, (hipStream_t)stream>>>(P);
    else if (k4 <= 6) step_prep_kernel<6, false><<<1, SP_THREADS, lds, (hipStream_t)stream>>>(P);
    else if (cache) step_prep_kernel<8, true><<<1, SP_THREADS, lds, (hipStream_t)stream>>>(P);
    else step_prep_kernel<8, false><<<1, SP_THREADS, lds, (hipStream_t)stream>>>(P);
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}

// =====================================================================================================================
// adam_act_kernel
// =====================================================================================================================
#define AA_THREADS 512
#define AA_H 256
#define AA_L1_WGS 8                       // layer-1 workgroups: 32 rows of W1 each (one float4 of the slice per thread)
#define AA_L1_ROWS (AA_H / AA_L1_WGS)
#define AA_L2_WGS 32                      // layer-2 workgroups: 8 rows of W2 each (one wave per row)
#define AA_SYNC_RIDE (16 + 4 * 2 * AA_H)  // ints: the riders' words (naf_adam_polyak_act_layer1) lie behind the widest launch's records
#define AA_POLL_TICKS 200000LL            // 2 ms at 100 MHz: a hang guard, not a schedule (records arrive within ~2 us)

typedef float aa_f4 __attribute__((ext_vector_type(4)));

struct AdamActArgs {
    AdamArgs ad;
    int64_t off_W1, off_b1, off_g1, off_be1, off_W2, off_b2, off_g2, off_be2, off_Wh;   // floats, inside the flat buffers
    int S, NH, NHP, HP, A;
    const float* obs;                     // [S] (pinned host or device)
    int obs_sys;                          // the observation lies in device memory the HOST stored into: system-scope loads
    const float *rm1, *rv1, *rm2, *rv2;   // BatchNorm running statistics of the main net
    float eps;
    float* heads_out;                     // nullable [NH]
    float* action_out;                    // [A] (pinned host or device)
    uint64_t seed;
    uint64_t* counter_dev;                // noise stream position: read, advanced by one
    float noise_scale;
    // device scratch, zero-initialised once by the caller and owned by this kernel: ints {epoch, wh_arrivals, timeouts, 0 ...} in the
    // first 64 bytes, then 256 (value, epoch) records of layer 1's activations and 256 of layer 2's
    int* sync;
    int wh_wgs;                           // workgroups that step the heads' weights
    uint64_t* host_errors;                // nullable pinned host word: timed-out polls
    uint32_t* host_seq;                   // nullable pinned host word: = the launch's epoch once the action has been written
    uint32_t* act_rec;                    // nullable pinned host record: 3 x {a[3j], a[3j + 1], a[3j + 2], epoch}, a store each
};

// Layer 1 of the NEXT update's chain riding on this launch (round 6; naf_adam_polyak_act_layer1): the arguments of bb_layer1_kernel
// (csrc/big_batch.hip, csrc/layer1_body.h). The chain's first launch needs the parameters this launch's optimizer step leaves and
// nothing else of it — as a launch of its own behind adam_act it also waited for the act() tail (heads, noise, the action: 4 of the
// launch's 7.6 us) and for a launch boundary. Here n_main extra workgroups run its body IN ITS ADAM FORM: they evaluate the layer-1
// parameters they read as this launch's step will leave them (the form the chunked chain runs when the previous update's step rides
// on layer 1: the same bits the layer-1 workgroups of this launch write), so they start with the launch and wait for nothing it
// computes. Two orderings, both off the critical path: the launch's layer-1 workgroups store their stepped slices only once every
// rider has consumed the old values (`l1_loaded`), and the riders' statistics lanes overwrite the running statistics only once this
// launch's readers of them — the layer-1 workgroups, the commit — are through (`l1_arrived`, `l1_commit`); the last party out
// (`l1_done`) clears the words and moves the launch's ordinal. Four words of `sync` behind the records, a cache line each.
// (Riders BEHIND the step — a flag, then the stand-alone form — were built first and gained a third of this: a flag between XCDs is
// 1.5 - 2 us under this launch's traffic; NOTEBOOK section 11.16.)
struct L1RideArgs {
    const float* x;
    int64_t x_net_stride;
    int ldx, K;
    const float *W, *bias, *gamma, *beta;
    int64_t param_net_stride;
    const float* mom;
    float *running_mean, *running_var;
    int64_t stat_net_stride;
    float* out;
    int64_t out_net_stride;
    int ldo;
    float *save_mean, *save_invstd, *wc_out, *xhat_out;
    int B, H;
    float momentum, eps;
    int n_main, xcd_rows;
};

// one float4 / one float of the flat buffers through the pending update (adam_one: the code every other launch of the step runs);
// loads and arithmetic are separate so that a workgroup has every operand in flight before it derives the step's scalars
struct AaOld4 {
    float4 th, gr, mm, vv, tg;
};
__device__ __forceinline__ static AaOld4 aa_load4(const AdamArgs& A, int64_t f4) {
    AaOld4 o;
    o.th = ((float4*)A.theta)[f4];
    o.gr = ((const float4*)A.g)[f4];
    o.mm = ((float4*)A.m)[f4];
    o.vv = ((float4*)A.v)[f4];
    o.tg = ((float4*)(A.target ? A.target : A.theta))[f4];
    return o;
}
// (the same in two halves — the new values now, their stores later: the layer-1 workgroups of a launch whose riders read the OLD
//  values of the same addresses, adam_act_kernel<.., L1K4 != 0>)
__device__ __forceinline__ static void aa_step4(const AdamArgs& A, const AdamScalars& sc, AaOld4& o) {
    if (!sc.skip) {
        const bool ht = A.target != nullptr;
        adam_one(o.th.x, o.gr.x, o.mm.x, o.vv.x, o.tg.x, ht, sc, A.beta1, A.beta2, A.eps, A.tau, A.one_minus_tau);
        adam_one(o.th.y, o.gr.y, o.mm.y, o.vv.y, o.tg.y, ht, sc, A.beta1, A.beta2, A.eps, A.tau, A.one_minus_tau);
        adam_one(o.th.z, o.gr.z, o.mm.z, o.vv.z, o.tg.z, ht, sc, A.beta1, A.beta2, A.eps, A.tau, A.one_minus_tau);
        adam_one(o.th.w, o.gr.w, o.mm.w, o.vv.w, o.tg.w, ht, sc, A.beta1, A.beta2, A.eps, A.tau, A.one_minus_tau);
    }
}
__device__ __forceinline__ static void aa_store4(const AdamArgs& A, const AdamScalars& sc, const AaOld4& o, int64_t f4) {
    if (!sc.skip) {
        ((float4*)A.theta)[f4] = o.th;
        ((float4*)A.m)[f4] = o.mm;
        ((float4*)A.v)[f4] = o.vv;
        if (A.target != nullptr) ((float4*)A.target)[f4] = o.tg;
    }
}
__device__ __forceinline__ static aa_f4 aa_apply4(const AdamArgs& A, const AdamScalars& sc, AaOld4 o, int64_t f4, bool through,
                                                  bool through_target = false) {
    if (!sc.skip) {
        const bool ht = A.target != nullptr;
        adam_one(o.th.x, o.gr.x, o.mm.x, o.vv.x, o.tg.x, ht, sc, A.beta1, A.beta2, A.eps, A.tau, A.one_minus_tau);
        adam_one(o.th.y, o.gr.y, o.mm.y, o.vv.y, o.tg.y, ht, sc, A.beta1, A.beta2, A.eps, A.tau, A.one_minus_tau);
        adam_one(o.th.z, o.gr.z, o.mm.z, o.vv.z, o.tg.z, ht, sc, A.beta1, A.beta2, A.eps, A.tau, A.one_minus_tau);
        adam_one(o.th.w, o.gr.w, o.mm.w, o.vv.w, o.tg.w, ht, sc, A.beta1, A.beta2, A.eps, A.tau, A.one_minus_tau);
        const unsigned off = (unsigned)(f4 * 16);
        if (through) naf_buf_st_f4_sc1(naf_buf(A.theta), off, 0, (naf_f32x4){o.th.x, o.th.y, o.th.z, o.th.w});
        else ((float4*)A.theta)[f4] = o.th;
        ((float4*)A.m)[f4] = o.mm;
        ((float4*)A.v)[f4] = o.vv;
        if (ht && through_target) naf_buf_st_f4_sc1(naf_buf(A.target), off, 0, (naf_f32x4){o.tg.x, o.tg.y, o.tg.z, o.tg.w});
        else if (ht) ((float4*)A.target)[f4] = o.tg;
    }
    return (aa_f4){o.th.x, o.th.y, o.th.z, o.th.w};
}
struct AaOld1 {
    float th, gr, mm, vv, tg;
};
__device__ __forceinline__ static AaOld1 aa_load1(const AdamArgs& A, int64_t e) {
    AaOld1 o;
    o.th = A.theta[e];
    o.gr = A.g[e];
    o.mm = A.m[e];
    o.vv = A.v[e];
    o.tg = (A.target ? A.target : A.theta)[e];
    return o;
}
__device__ __forceinline__ static float aa_apply1(const AdamArgs& A, const AdamScalars& sc, AaOld1 o, int64_t e) {
    if (!sc.skip) {
        const bool ht = A.target != nullptr;
        adam_one(o.th, o.gr, o.mm, o.vv, o.tg, ht, sc, A.beta1, A.beta2, A.eps, A.tau, A.one_minus_tau);
        A.theta[e] = o.th;
        A.m[e] = o.mm;
        A.v[e] = o.vv;
        if (ht) A.target[e] = o.tg;
    }
    return o.th;
}

// ---- (value, epoch) records: 8 bytes, one sc1 store by the producer; the reader polls two of them per 16-byte sc1 load -------
__device__ __forceinline__ static void aa_publish(int* recs, int i, float v, int epoch) {
    const naf_u32x2 r = {__builtin_bit_cast(unsigned, v), (unsigned)epoch};
    __builtin_amdgcn_raw_buffer_store_b64(r, naf_buf(recs), 8u * (unsigned)i, 0, 16);
}
// the four activations 4 l .. 4 l + 3 (records 4 l .. 4 l + 3 = two 16-byte granules); NaN in every slot if the bound ran out
__device__ __forceinline__ static aa_f4 aa_poll4(int* recs, int l, int epoch, bool* timed_out) {
    const __amdgpu_buffer_rsrc_t rb = naf_buf(recs);
    const long long t0 = wall_clock64();
    while (true) {
        const naf_u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rb, 32u * (unsigned)l, 0, 16);
        const naf_u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rb, 32u * (unsigned)l + 16u, 0, 16);
        const unsigned a1 = a[1], a3 = a[3], b1 = b[1], b3 = b[3];
        if ((int)a1 == epoch && (int)a3 == epoch && (int)b1 == epoch && (int)b3 == epoch) {
            const unsigned x0 = a[0], x1 = a[2], x2 = b[0], x3 = b[2];
            return (aa_f4){__builtin_bit_cast(float, x0), __builtin_bit_cast(float, x1), __builtin_bit_cast(float, x2),
                           __builtin_bit_cast(float, x3)};
        }
        if (wall_clock64() - t0 > AA_POLL_TICKS) break;
        __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");                  // a poll: the loads above are issued again every trip
    }
    *timed_out = true;
    const float nan = __builtin_nanf("");
    return (aa_f4){nan, nan, nan, nan};
}
__device__ static inline void aa_count_timeout(const AdamActArgs& P) {
    atomicAdd(&P.sync[2], 1);
    if (P.host_errors) __hip_atomic_fetch_add((unsigned long long*)P.host_errors, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// SPEC: one more workgroup — the launch's last, of SP_THREADS threads like all of them then (the others' upper half leaves at once) —
// prefetches the next timestep's minibatch (step_prep_body above, MODE 1): 0 = none, 1 / 2 = K <= 24 with / without the rows cached
// in LDS, 3 / 4 = K <= 32. It depends on nothing in this launch and nothing in this launch depends on it. (Round 5's pipelined
// timestep also ran its append + prefetch here — SPEC 5 .. 8 — and the chain behind the launch waited 4 - 9 us for this one
// workgroup; since round 6 that is step_prefetch_kernel on a stream of its own, one minibatch further ahead.)
// SPEC == 0 with SP.cp_n set: the COMMIT of the pipelined timestep (working -> public state of the learner, `copies`) by the launch's
// last workgroup while it waits for the heads' weights.
// G (round 6): lanes of the state's group in the noise body — 8, or 16 for 9 .. 11 joints: up to 78 heads rows over the last workgroup's
// eight waves (ten per wave instead of six), a fourth 16-byte chunk of the action's record. Kernels of their own: the 8-lane ones keep
// their registers.
#define AA_MAX_NH_WIDE 80
#define AA_MAX_A_WIDE 11
// HV (round 6): the layer size — 256, or 512 (widths in (256, 512] are stored as 512): 16 layer-1 and 64 layer-2 workgroups, a row of
// W2 / Wh as TWO float4 per lane (inputs 4 l .. 4 l + 3 and 256 + 4 l .. 256 + 4 l + 3, one fmaf chain through both: act_dot4 ->
// act_dot4_acc, as policy_act_512_kernel), 512 + 512 records.
// L1K4 (round 6): 6 | 8 = the next update's layer 1 rides on this launch (L1RideArgs above; K4 = 6 | 8 float4 of state, L1FULL: whole
// 64-row blocks): L1.n_main more workgroups behind the last one, 512 threads each (bb_layer1_impl<.., ADAM = true, ..>). 0: no riders,
// the kernel as it was.
#ifndef AA_EARLY_RIDE_OUT
#define AA_EARLY_RIDE_OUT 1               // (0: the last workgroup counts itself out at its end, as the riders' first form did — A/B)
#endif
template <int PMODE, int SPEC, int G = 8, int HV = AA_H, int L1K4 = 0, bool L1FULL = true>
__global__ __launch_bounds__(SPEC ? SP_THREADS : AA_THREADS) void adam_act_kernel(const AdamActArgs P, const StepPrepArgs SP, const L1RideArgs L1) {
    static_assert(L1K4 == 0 || SPEC == 0, "layer 1 rides on the launch without the prefetching workgroup");
    constexpr int HL = G == 8 ? HEAD_MAX_LDH : AA_MAX_NH_WIDE;
    constexpr int NQ = HV / 256;                        // float4 per lane of a 256- | 512-wide row
    constexpr int L1_WGS = HV / AA_L1_ROWS, L2_WGS = HV / 8;
    static_assert(HV == 256 || HV == 512, "layer size");
    if (SPEC) {
        extern __shared__ __attribute__((aligned(16))) unsigned char aa_smem[];
        __shared__ float4 sNewS[SP_MAX_RF4];
        __shared__ int sHitS;
        if (blockIdx.x == gridDim.x - 1) {
            NAF_TL_FL(g_tl_sp, NAF_TL_ADAM_ACT, 13, true, false);
            step_prep_body<(((SPEC - 1) & 3) < 2 ? 6 : 8), ((SPEC - 1) & 1) == 0, 1>(SP, aa_smem, sNewS, &sHitS);
            NAF_TL_FL(g_tl_sp, NAF_TL_ADAM_ACT, 14, true, false);
            return;
        }
        if (threadIdx.x >= AA_THREADS) return;          // (whole waves: the barriers below count the waves that are left)
    }
#define AA_TL(slot) NAF_TL_FL(g_tl_sp, NAF_TL_ADAM_ACT, slot, blockIdx.x == 0, (int)blockIdx.x == (int)gridDim.x - (SPEC ? 2 : 1) - (L1K4 ? L1.n_main : 0))
    __shared__ AdamScalars sSc;
    __shared__ __attribute__((aligned(16))) float sAct[HV];                   // a1 (layer-2 workgroups) / a2 (the last workgroup)
    __shared__ __attribute__((aligned(16))) float sW[64 * ACT_MAX_S + 3 * 64]; // layer-1 workgroups: their 64 rows of W1, b1, g1, be1
    __shared__ float sObs[ACT_MAX_S];
    __shared__ float sHeads[HL];
    __shared__ float sActOut[G == 8 ? NAF_MAX_A + 1 : 12];
    __shared__ float sL[PMODE == NAF_P_MATMUL ? G * (G + 1) : 1];
    __shared__ int sTimed;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wg = blockIdx.x;
    const AdamArgs& A = P.ad;
    // (only the last workgroup writes the epoch, behind everything it waits for; unsigned arithmetic: it may wrap)
    const int epoch = (int)((unsigned)P.sync[0] + 1u);
    int* rec1 = P.sync + 16;
    int* rec2 = P.sync + 16 + 2 * HV;
    // the riders' three words, each on a 128-byte line of its own BEHIND the records (polled with atomics by every rider: on the line
    // of the launch's counters they delayed the first records of layer 1 by 0.85 us — the action's critical path)
    int* const l1_arrived = P.sync + AA_SYNC_RIDE;          // layer-1 workgroups that have stepped their slice
    int* const l1_done = P.sync + AA_SYNC_RIDE + 32;        // riders that are through
    int* const l1_commit = P.sync + AA_SYNC_RIDE + 64;      // = the launch's ordinal once the commit has been copied
    int* const l1_loaded = P.sync + AA_SYNC_RIDE + 96;      // riders that have consumed the old layer-1 parameters they read
    // Nobody waits for the launch's end: every party that uses these words — the riders, the layer-1 workgroups, the last workgroup —
    // counts itself out in `l1_done`, and the LAST one out clears the words and moves the launch's ordinal (one that starts late must
    // still read the old one; a wait for the riders in the last workgroup cost the launch 2 us: a flag takes that long to cross the chip)
    // (in two halves: the count comes back across the chip, 1.2 us — a party with work left issues it BEHIND ITS LAST USE of the words and
    //  looks at the answer when it leaves)
    auto ride_out_begin = [&]() { return __hip_atomic_fetch_add(l1_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto ride_out_end = [&](int before) {
        if (before == L1.n_main + L1_WGS) {
            __hip_atomic_store(l1_arrived, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(l1_loaded, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(l1_done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            P.sync[0] = epoch;
        }
    };
    auto ride_out = [&]() { ride_out_end(ride_out_begin()); };
    if (tid == 0) sTimed = 0;
    const int wh_wgs = P.wh_wgs;                        // workgroups that step Wh: ceil(NHP * HP / 4 / AA_THREADS)
    const bool tl_l2 = wg == L1_WGS + wh_wgs;           // (timeline: the first layer-2 workgroup leaves slots 8 ...)
    AA_TL(0);

    if (wg < L1_WGS) {
        // ---- layer 1: rows 32 wg .. 32 wg + 31 of W1 and of b1 / g1 / be1 --------------------------------------------------
        constexpr int R = AA_L1_ROWS;
        const int S = P.S, row0 = R * wg;
        const int nW4 = R * S / 4, items = nW4 + 3 * R / 4;   // float4 of this workgroup's slice (R S floats start 16-byte aligned)
        if (tid < ACT_MAX_S) {
            float o = 0.f;
            if (tid < S) {
                if (P.obs_sys) o = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(naf_buf(P.obs), 4u * (unsigned)tid, 0, 17));
                else o = P.obs[tid];
            }
            sObs[tid] = o;
        }
        // this thread's float4 of the flat buffers, and where its new value goes in LDS
        int lds;
        int64_t f4;
        {
            const int it = tid < items ? tid : 0;
            if (it < nW4) {
                lds = 4 * it;
                f4 = (P.off_W1 + (int64_t)row0 * S) / 4 + it;
            } else {
                const int j = it - nW4, seg = j / (R / 4), c = j % (R / 4);
                lds = R * ACT_MAX_S + R * seg + 4 * c;
                const int64_t off = seg == 0 ? P.off_b1 : (seg == 1 ? P.off_g1 : P.off_be1);
                f4 = (off + row0) / 4 + c;
            }
        }
        const AaOld4 o = aa_load4(A, f4);
        const float rm = P.rm1[row0 + (tid & (R - 1))], rv = P.rv1[row0 + (tid & (R - 1))];
        const AdamPrefetch pf = adam_prefetch(A, tid);
        adam_derive(A, pf, &sSc, tid);
        if constexpr (L1K4 != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (rm / rv above have been READ)
        __syncthreads();
        if constexpr (L1K4 != 0) {
            // the riders overwrite the running statistics this workgroup has just read — once every one of the launch's layer-1
            // workgroups has said so (raised here, early: a flag takes 1.5 - 2 us to cross the chip, and the riders get to their
            // statistics 4 us into the launch)
            if (tid == 0) __hip_atomic_fetch_add(l1_arrived, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        AA_TL(1);
        const AdamScalars sc = sSc;
        AaOld4 on = o;                                      // (L1K4: the stepped values, stored at the end)
        if (tid < items) {
            aa_f4 nv;
            if constexpr (L1K4 != 0) {
                aa_step4(A, sc, on);
                nv = (aa_f4){on.th.x, on.th.y, on.th.z, on.th.w};
            } else {
                nv = aa_apply4(A, sc, o, f4, false);
            }
            *(aa_f4*)(sW + lds) = nv;
        }
        __syncthreads();
        AA_TL(2);
        if (tid < R) {
            const int row = row0 + tid;
            float w1[ACT_MAX_S];
#pragma unroll
            for (int k = 0; k < ACT_MAX_S; ++k) w1[k] = k < S ? sW[tid * S + k] : 0.f;
            const float a1 = act_layer1_row(w1, sObs, sW[R * ACT_MAX_S + tid], sW[R * ACT_MAX_S + R + tid],
                                            sW[R * ACT_MAX_S + 2 * R + tid], rm, rv, P.eps);
            aa_publish(rec1, row, a1, epoch);
        }
        AA_TL(3);
        if constexpr (L1K4 != 0) {
            // The riders evaluate this slice themselves, from its OLD values, gradient and optimizer state — so the stepped values go
            // to memory only once every rider has consumed what it reads (they have, microseconds ago: their loads are the first
            // thing they do; act() above ran on the copies in LDS). Bounded like every wait here.
            if (tid == 0) {
                const long long t0 = wall_clock64();
                while (__hip_atomic_load(l1_loaded, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < L1.n_main) {
                    if (wall_clock64() - t0 > AA_POLL_TICKS) { sTimed = 1; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            __syncthreads();
            if (tid < items) aa_store4(A, sc, on, f4);
            if (tid == 0) {
                if (sTimed) aa_count_timeout(P);
                ride_out();
            }
        }
        return;
    }

    if (wg < L1_WGS + wh_wgs) {
        // ---- the heads' weights: one float4 per thread, written THROUGH (the last workgroup reads them in this launch), then one
        // arrival per workgroup once the stores have landed. Dispatched ahead of the layer-2 workgroups and depending on nothing:
        // long done when the last workgroup asks.
        const int64_t i = (int64_t)(wg - L1_WGS) * AA_THREADS + tid, n4 = (int64_t)P.NHP * (P.HP / 4);
        const bool on = i < n4;
        const int64_t f = P.off_Wh / 4 + (on ? i : 0);
        const AaOld4 o = aa_load4(A, f);
        const AdamPrefetch pf = adam_prefetch(A, tid);
        adam_derive(A, pf, &sSc, tid);
        __syncthreads();
        const AdamScalars sc = sSc;
        if (on) aa_apply4(A, sc, o, f, true);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(&P.sync[1], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }

    if (wg < L1_WGS + wh_wgs + L2_WGS) {
        // ---- layer 2: wave = one row of W2 (64 float4 per 256 inputs, lane l holds inputs 4 l .. 4 l + 3 of each) -------------------
        NAF_TL_FL(g_tl_sp, NAF_TL_ADAM_ACT, 8, tl_l2, false);
        const int w = wg - L1_WGS - wh_wgs, row = 8 * w + wave;
        const int64_t fW2 = P.off_W2 / 4 + (HV / 4) * (int64_t)row + lane;
        AaOld4 o2[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) o2[q] = aa_load4(A, fW2 + 64 * q);
        AaOld1 ob = {}, og = {}, obe = {};
        if (lane == 0) {
            ob = aa_load1(A, P.off_b2 + row);
            og = aa_load1(A, P.off_g2 + row);
            obe = aa_load1(A, P.off_be2 + row);
        }
        const float rm = P.rm2[row], rv = P.rv2[row];
        const AdamPrefetch pf = adam_prefetch(A, tid);
        adam_derive(A, pf, &sSc, tid);
        __syncthreads();
        NAF_TL_FL(g_tl_sp, NAF_TL_ADAM_ACT, 9, tl_l2, false);
        const AdamScalars sc = sSc;
        aa_f4 w2[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) w2[q] = aa_apply4(A, sc, o2[q], fW2 + 64 * q, false);
        float b2 = 0.f, g2 = 0.f, be2 = 0.f;
        if (lane == 0) {
            b2 = aa_apply1(A, sc, ob, P.off_b2 + row);
            g2 = aa_apply1(A, sc, og, P.off_g2 + row);
            be2 = aa_apply1(A, sc, obe, P.off_be2 + row);
        }
        NAF_TL_FL(g_tl_sp, NAF_TL_ADAM_ACT, 10, tl_l2, false);
        // layer 1's activations: polled by the first wave, shared through LDS
        if (wave == 0) {
            bool timed = false;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const aa_f4 x = aa_poll4(rec1, lane + 64 * q, epoch, &timed);
                *(aa_f4*)(sAct + 256 * q + 4 * lane) = x;
            }
            if (timed) sTimed = 1;
        }
        __syncthreads();
        NAF_TL_FL(g_tl_sp, NAF_TL_ADAM_ACT, 11, tl_l2, false);
        if (sTimed && tid == 0) aa_count_timeout(P);
        float p = act_dot4(w2[0], *(const aa_f4*)(sAct + 4 * lane));
        if (NQ == 2) p = act_dot4_acc(w2[NQ - 1], *(const aa_f4*)(sAct + 256 + 4 * lane), p);
        p = act_sum64(p);
        if (lane == 0) aa_publish(rec2, row, act_bn_relu(p + b2, rm, rv, g2, be2, P.eps), epoch);
        NAF_TL_FL(g_tl_sp, NAF_TL_ADAM_ACT, 12, tl_l2, false);
        return;
    }

    if constexpr (L1K4 != 0) {
        const int last_wg = L1_WGS + wh_wgs + L2_WGS;
        if (wg > last_wg) {
            // ---- layer 1 of the next update's chain (csrc/layer1_body.h), in its ADAM form: the workgroup evaluates the layer-1
            // parameters it reads AS THIS LAUNCH'S STEP WILL LEAVE THEM (the form the chunked chain runs when the previous update's
            // step rides on it: same gradient, same norm partials, same step count — the same bits the layer-1 workgroups above
            // write), so it depends on nothing in this launch and starts with it. Its one hazard is the running statistics of
            // layer 1, which the workgroups above READ (act() is eval-mode BatchNorm) and the commit copies: the lanes that write
            // them wait for both first (the hook; both have long happened by then).
            __shared__ BbL1Shared<L1K4, true> sL1;
            NAF_TL_FL(g_tl_sp, NAF_TL_BB_LAYER1, 13, wg == last_wg + 1, wg == last_wg + L1.n_main);
            struct Guard {
                const int *arrived, *commit;
                int want_arrived, want_commit;
                int* timed;
                int* loaded;
                __device__ __forceinline__ void parameters_loaded() const {
                    __hip_atomic_fetch_add(loaded, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __device__ __forceinline__ void before_running_stats() const {
                    // ONE lane of each wave that gets here polls, the others take its word: with all 64 column lanes of 8 riders
                    // polling, the two lines were so busy that the layer-1 workgroups' arrivals took 3 us to land
                    const int lane_ = (int)(threadIdx.x & 63);
                    const bool leader = __builtin_amdgcn_readfirstlane(lane_) == lane_;
                    const long long t0 = wall_clock64();
                    for (;;) {
                        int a = 0, c = 0;
                        if (leader) {
                            a = __hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            c = __hip_atomic_load(commit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        a = __builtin_amdgcn_readfirstlane(a);
                        c = __builtin_amdgcn_readfirstlane(c);
                        if (a >= want_arrived && c == want_commit) break;
                        if (wall_clock64() - t0 > AA_POLL_TICKS) { *timed = 1; break; }
                        __builtin_amdgcn_s_sleep(2);
                    }
                }
            };
            const Guard guard = {l1_arrived, l1_commit, L1_WGS, epoch, &sTimed, l1_loaded};
            bb_layer1_impl<L1K4, true, L1FULL, Guard>(sL1, wg - last_wg - 1, L1.x, L1.x_net_stride, L1.ldx, L1.K, L1.W, L1.bias, L1.gamma,
                                                      L1.beta, L1.param_net_stride, L1.mom, L1.running_mean, L1.running_var,
                                                      L1.stat_net_stride, L1.out, L1.out_net_stride, L1.ldo, L1.save_mean, L1.save_invstd,
                                                      L1.wc_out, L1.B, L1.H, L1.momentum, L1.eps, L1.n_main, A, 0, 0, 0, L1.xcd_rows,
                                                      L1.xhat_out, guard);
            NAF_TL_FL(g_tl_sp, NAF_TL_BB_LAYER1, 14, wg == last_wg + 1, wg == last_wg + L1.n_main);
            __syncthreads();
            if (tid == 0) {
                if (sTimed) aa_count_timeout(P);            // (counted where the host raises: the chain's result is not valid)
                ride_out();
            }
            return;
        }
    }

    // ---- the last workgroup: heads, exploration noise, clamp ----------------------------------------------------------------------
    {
        const uint64_t ctr = *P.counter_dev;
        const int NH = P.NH;
        if (!SPEC) {
            // the pipelined timestep's commit: working -> public copies of what the chain advances besides the gradient (ranges no
            // workgroup of this launch writes; the chain that overwrites the working copies is the NEXT launch)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const unsigned* src = SP.cp_src[c];
                unsigned* dst = SP.cp_dst[c];
                const int nw = SP.cp_n[c];
                if (src && dst)
                    for (int w = tid; w < nw; w += AA_THREADS) dst[w] = src[w];
            }
        }
        if constexpr (L1K4 != 0) {
            // the riders advance the working BatchNorm statistics the commit above has just copied: they wait for this word (a barrier
            // of its own: behind the poll below it kept them waiting until the heads' weights had arrived, 4 us into the launch).
            // What they must not overtake is the copy's READS of the working statistics: done once its stores have been issued — no
            // release (nobody in this launch reads the public copy).
            __syncthreads();
            if (tid == 0) __hip_atomic_store(l1_commit, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // every float4 of Wh has been stepped and written through once its workgroups have arrived
        if (tid == 0) {
            const unsigned want = (unsigned)epoch * (unsigned)wh_wgs;
            const long long t0 = wall_clock64();
            while ((int)((unsigned)__hip_atomic_load(&P.sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
                if (wall_clock64() - t0 > AA_POLL_TICKS) { sTimed = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        AA_TL(1);
        const __amdgpu_buffer_rsrc_t whb = naf_buf(A.theta + P.off_Wh);
        aa_f4 wh[HL / 8][NQ];
        float bias[HL / 8];
#pragma unroll
        for (int k = 0; k < HL / 8; ++k) {
            const int h = wave + 8 * k;
            const unsigned roff = (unsigned)(h < NH ? h : 0) * (unsigned)P.HP * 4u;
#pragma unroll
            for (int q = 0; q < NQ; ++q) wh[k][q] = naf_buf_f4_sc1(whb, 16u * (unsigned)lane + 1024u * (unsigned)q, roff);
            const naf_f32x4 bq = naf_buf_f4_sc1(whb, 16u * (unsigned)(HV / 4), roff);
            bias[k] = bq[0];
        }
        // the standard normal draw of the noise depends on nothing the launch computes: taken while the activations are under way
        const float zn = naf_act_noise_z(P.seed, ctr, 0, tid & (G - 1), tid < G && (tid & (G - 1)) < P.A);
        if (wave == 0) {
            bool timed = false;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const aa_f4 x = aa_poll4(rec2, lane + 64 * q, epoch, &timed);
                *(aa_f4*)(sAct + 256 * q + 4 * lane) = x;
            }
            if (timed) sTimed = 1;
        }
        __syncthreads();
        AA_TL(2);
        // The last workgroup counts itself out of the riders' words HERE: every workgroup of the launch has started (the records
        // above are the layer-2 workgroups', the arrivals the heads' weights'), so the launch's ordinal may move, and this workgroup
        // does not touch the words again. The answer is read at the very end — waited for there, the count's way across the chip was
        // the launch's last 1.2 us (r06_step_timeline_b256: action stored at 8.6 us, end at 9.85).
        int rode = 0;
        if constexpr (L1K4 != 0 && AA_EARLY_RIDE_OUT != 0) {
            if (tid == 0) rode = ride_out_begin();
        }
        const bool timed = sTimed != 0;
        const aa_f4 x = *(const aa_f4*)(sAct + 4 * lane);
        float ph[HL / 8];
#pragma unroll
        for (int k = 0; k < HL / 8; ++k) {
            ph[k] = act_dot4(wh[k][0], x);
            if (NQ == 2) ph[k] = act_dot4_acc(wh[k][NQ - 1], *(const aa_f4*)(sAct + 256 + 4 * lane), ph[k]);
        }
#pragma unroll
        for (int k = 0; k < HL / 8; ++k) {
            const int h = wave + 8 * k;
            if (h < NH) {                                   // (wave-uniform)
                float p = act_sum64(ph[k]);
                if (lane == 0) {
                    p += bias[k];
                    sHeads[h] = timed ? __builtin_nanf("") : p;
                    if (P.heads_out) P.heads_out[h] = p;
                }
            }
        }
        __syncthreads();
        AA_TL(3);
        // (the action lands in LDS first: its lanes hand it to memory below, once as plain words and once as a self-validating record)
        naf_act_noise_body_z<PMODE, G>(sHeads, sL, sActOut, zn, P.noise_scale, 0, tid < G, P.A, tid);
        __syncthreads();
        AA_TL(4);
        if (tid < 64) {                                     // (the first wave)
            if (tid < P.A) P.action_out[tid] = sActOut[tid];
            if (P.act_rec && tid < (G == 8 ? 3 : 4)) {
                // How the HOST learns the action without synchronising the stream: 16-byte chunks {a[3j], a[3j+1], a[3j+2], ordinal},
                // one store each — a chunk is in host memory whole or not at all, and the host takes the action from the chunks
                // once every chunk it needs carries this launch's ordinal. (Round 5's first form announced the action by a word of
                // its own, stored after the action's stores had been acknowledged: writes to host memory may pass one another on
                // the way — relaxed ordering — and one timestep in 1e4 .. 1e5 read the ordinal before the action had landed.)
                const float a0 = 3 * tid + 0 < P.A ? sActOut[3 * tid + 0] : 0.f, a1 = 3 * tid + 1 < P.A ? sActOut[3 * tid + 1] : 0.f;
                const float a2 = 3 * tid + 2 < P.A ? sActOut[3 * tid + 2] : 0.f;
                const naf_u32x4 rec = {__builtin_bit_cast(unsigned, a0), __builtin_bit_cast(unsigned, a1), __builtin_bit_cast(unsigned, a2),
                                       (unsigned)epoch};
                __builtin_amdgcn_raw_buffer_store_b128(rec, naf_buf(P.act_rec), 16u * (unsigned)tid, 0, 17);
            }
            if (tid == 0) {
                if (timed) aa_count_timeout(P);
                *P.counter_dev = ctr + 1;
                if constexpr (L1K4 != 0) ride_out_end(AA_EARLY_RIDE_OUT != 0 ? rode : ride_out_begin());    // (the last party out moves the ordinal)
                else P.sync[0] = epoch;
            }
            if (P.host_seq) {
                // (the launch's ordinal as a word of its own, behind the action's plain words: for readers that synchronise the stream)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (tid == 0) __hip_atomic_store(P.host_seq, (uint32_t)epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        AA_TL(5);
    }
#undef AA_TL
}

static int aa_launch(const naf_adam_args_t* adam, const naf_act_net_t* net, const float* obs, float* heads_out,
                     float* action_out, uint64_t seed, uint64_t* counter_dev, float noise_scale, int p_mode,
                     int32_t* sync, uint64_t* host_errors, uint32_t* host_seq, uint32_t* action_rec,
                     const naf_step_prefetch_t* prefetch, int obs_system_scope, const naf_bb_layer1_t* l1, void* stream) {
    if (!adam || !net || !obs || !action_out || !counter_dev || !sync || ((uintptr_t)sync & 15) != 0) return NAF_ERR_ARG;
    L1RideArgs L1;
    memset(&L1, 0, sizeof(L1));
    int l1k4 = 0;
    bool l1full = true;
    if (l1) {
        // (the checks of naf_bb_layer1_adam, csrc/big_batch.hip: the same body runs)
        if (prefetch && prefetch->mode != 0) return NAF_ERR_ARG;      // (the launch without the prefetching workgroup only)
        if (!l1->x || !l1->W || !l1->bias || !l1->mom || !l1->gamma || !l1->beta || !l1->running_mean || !l1->running_var || !l1->out ||
            !l1->save_mean || !l1->save_invstd || l1->B < 16 || l1->B > 4096 || l1->H < BB_COLS || (l1->H % BB_COLS) != 0 || l1->nets != 2 ||
            l1->K <= 0 || l1->K > 4 * BB_MAX_K4 || l1->ldo < l1->H || (l1->ldo & 3))
            return NAF_ERR_ARG;
        const int k4 = (l1->K + 3) / 4, k4d = k4 <= 6 ? 6 : 8;
        if (((uintptr_t)l1->x & 15) != 0 || (l1->ldx & 3) != 0 || l1->ldx < 4 * k4d || (l1->x_net_stride & 3) != 0) return NAF_ERR_ARG;
        if ((((uintptr_t)l1->bias | (uintptr_t)l1->out | (uintptr_t)l1->mom | (uintptr_t)l1->W) & 15) != 0 || (l1->param_net_stride & 3) != 0 ||
            (l1->out_net_stride & 3) != 0 || (l1->xhat_out && ((uintptr_t)l1->xhat_out & 15)))
            return NAF_ERR_ARG;
        {
            // (the riders evaluate these parameters through the pending step: they must be the main network's, inside the flat buffer
            //  this launch steps, the target's param_net_stride floats behind — what naf_bb_layer1_adam asks of a riding step)
            const float* lo = adam->theta;
            const float* hi = adam->theta + adam->n;
            if ((((uintptr_t)l1->gamma | (uintptr_t)l1->beta) & 15) != 0) return NAF_ERR_ARG;
            if (l1->W < lo || l1->W + (int64_t)l1->H * l1->K > hi || l1->bias < lo || l1->bias + l1->H > hi || l1->gamma < lo ||
                l1->gamma + l1->H > hi || l1->beta < lo || l1->beta + l1->H > hi || adam->theta_target != adam->theta + l1->param_net_stride)
                return NAF_ERR_ARG;
        }
        L1.x = l1->x; L1.x_net_stride = l1->x_net_stride; L1.ldx = l1->ldx; L1.K = l1->K; L1.W = l1->W; L1.bias = l1->bias;
        L1.gamma = l1->gamma; L1.beta = l1->beta; L1.param_net_stride = l1->param_net_stride; L1.mom = l1->mom;
        L1.running_mean = l1->running_mean; L1.running_var = l1->running_var; L1.stat_net_stride = l1->stat_net_stride; L1.out = l1->out;
        L1.out_net_stride = l1->out_net_stride; L1.ldo = l1->ldo; L1.save_mean = l1->save_mean; L1.save_invstd = l1->save_invstd;
        L1.wc_out = l1->wc_out; L1.xhat_out = l1->xhat_out; L1.B = l1->B; L1.H = l1->H; L1.momentum = l1->momentum; L1.eps = l1->eps;
        L1.n_main = ((l1->B + BB_ROWS - 1) / BB_ROWS) * (l1->H / BB_COLS) * l1->nets;
        L1.xcd_rows = 1;
        l1k4 = k4d;
        l1full = l1->B % BB_ROWS == 0;
    }
    AdamActArgs P;
    memset(&P, 0, sizeof(P));
    if (!adam_args_from(*adam, P.ad)) return NAF_ERR_ARG;
    P.ad.bc = nullptr;
    const int S = net->S, A = net->A, H = net->H, NHP = net->NHP, HP = net->HP;
    const int NH = A + A * (A + 1) / 2 + 1;
    if ((H != AA_H && H != 2 * AA_H) || S <= 0 || S > ACT_MAX_S || A <= 0 || A > AA_MAX_A_WIDE || NH > (A > NAF_MAX_A ? AA_MAX_NH_WIDE : HEAD_MAX_LDH) ||
        NHP < NH || HP <= H || (HP & 3) != 0 || HP / 4 > AA_THREADS)
        return NAF_ERR_ARG;
    const bool wide = A > NAF_MAX_A;                     // (one sample per 16-lane group, four chunks in the action's record)
    if (p_mode != NAF_P_HADAMARD && p_mode != NAF_P_MATMUL) return NAF_ERR_ARG;
    // the flat layout this kernel walks: [W1 | b1 | g1 | be1 | W2 | b2 | g2 | be2 | Wh] back to back (offsets in floats, multiples of
    // 4; W1 rows of 64 start on 16-byte boundaries because 64 S floats do), covering the whole buffer — every parameter is stepped
    const int64_t o[9] = {net->off_W1, net->off_b1, net->off_g1, net->off_be1, net->off_W2, net->off_b2, net->off_g2, net->off_be2, net->off_Wh};
    const int64_t len[9] = {(int64_t)H * S, H, H, H, (int64_t)H * H, H, H, H, (int64_t)NHP * HP};
    int64_t at = 0;
    for (int i = 0; i < 9; ++i) {
        if (o[i] != at || (o[i] & 3) != 0) return NAF_ERR_ARG;
        at += len[i];
    }
    if (at != adam->n) return NAF_ERR_ARG;
    if (!net->running_mean1 || !net->running_var1 || !net->running_mean2 || !net->running_var2) return NAF_ERR_ARG;
    P.off_W1 = o[0]; P.off_b1 = o[1]; P.off_g1 = o[2]; P.off_be1 = o[3]; P.off_W2 = o[4]; P.off_b2 = o[5]; P.off_g2 = o[6];
    P.off_be2 = o[7]; P.off_Wh = o[8];
    P.S = S; P.NH = NH; P.NHP = NHP; P.HP = HP; P.A = A;
    P.obs = obs;
    P.obs_sys = obs_system_scope ? 1 : 0;
    P.rm1 = net->running_mean1; P.rv1 = net->running_var1; P.rm2 = net->running_mean2; P.rv2 = net->running_var2;
    P.eps = net->eps;
    P.heads_out = heads_out;
    P.action_out = action_out;
    P.seed = seed;
    P.counter_dev = counter_dev;
    P.noise_scale = noise_scale;
    P.sync = sync;
    P.host_errors = host_errors;
    P.host_seq = host_seq;
    P.act_rec = action_rec;
    if (((uintptr_t)action_rec & 15) != 0) return NAF_ERR_ARG;
    P.wh_wgs = (int)(((int64_t)NHP * (HP / 4) + AA_THREADS - 1) / AA_THREADS);
    const bool h512 = H == 2 * AA_H;
    const int grid = H / AA_L1_ROWS + P.wh_wgs + H / 8 + 1;
    StepPrepArgs SP;
    memset(&SP, 0, sizeof(SP));
    if (prefetch && prefetch->mode == 0) {
        // no prefetching workgroup: the launch's last workgroup commits `copies` (working -> public state of the pipelined timestep)
        for (int c = 0; c < 3; ++c) {
            const int nw = prefetch->copies.n_words[c];
            if (nw <= 0) continue;
            if (!prefetch->copies.src[c] || !prefetch->copies.dst[c] ||
                (((uintptr_t)prefetch->copies.src[c] | (uintptr_t)prefetch->copies.dst[c]) & 3) != 0)
                return NAF_ERR_ARG;
            SP.cp_src[c] = (const unsigned*)prefetch->copies.src[c];
            SP.cp_dst[c] = (unsigned*)prefetch->copies.dst[c];
            SP.cp_n[c] = nw;
        }
    }
    if (!prefetch || prefetch->mode == 0) {
#define AA_GO0(PM, GV, HVV, LK, LF) adam_act_kernel<PM, 0, GV, HVV, LK, LF><<<grid + L1.n_main, AA_THREADS, 0, (hipStream_t)stream>>>(P, SP, L1)
#define AA_GO0_L(PM, GV, HVV)                                 \
    do {                                                      \
        if (l1k4 == 0) AA_GO0(PM, GV, HVV, 0, true);          \
        else if (l1k4 == 6 && l1full) AA_GO0(PM, GV, HVV, 6, true);  \
        else if (l1k4 == 6) AA_GO0(PM, GV, HVV, 6, false);    \
        else if (l1full) AA_GO0(PM, GV, HVV, 8, true);        \
        else AA_GO0(PM, GV, HVV, 8, false);                   \
    } while (0)
#define AA_GO0_H(PM, GV)                    \
    do {                                    \
        if (h512) AA_GO0_L(PM, GV, 512);    \
        else AA_GO0_L(PM, GV, 256);         \
    } while (0)
        if (p_mode == NAF_P_HADAMARD) {
            if (wide) AA_GO0_H(NAF_P_HADAMARD, 16);
            else AA_GO0_H(NAF_P_HADAMARD, 8);
        } else {
            if (wide) AA_GO0_H(NAF_P_MATMUL, 16);
            else AA_GO0_H(NAF_P_MATMUL, 8);
        }
#undef AA_GO0_H
#undef AA_GO0_L
#undef AA_GO0
        NAF_CHECK_LAUNCH();
        return NAF_OK;
    }
    // with the prefetch of the next timestep's minibatch (step_prep_body<.., 1>): the record and the indices are its own, nothing
    // of the ring or the sampler's stream is committed
    if (!prefetch->spec_rec || !prefetch->idx_spec || prefetch->mode != 1) return NAF_ERR_ARG;
    size_t lds = 0;
    int k4 = 0;
    const int rc = sp_build(prefetch->replay, nullptr, nullptr, nullptr, prefetch->seed, prefetch->counter_dev, nullptr,
                            prefetch->out_rows, prefetch->out_ld, prefetch->action_mode, prefetch->mom, prefetch->B,
                            prefetch->without_replacement, prefetch->spec_rec, prefetch->idx_spec, &prefetch->copies,
                            prefetch->host_spec, prefetch->pipe_errors, nullptr, nullptr, prefetch->depth ? prefetch->depth : 1,
                            prefetch->pf_seq, SP, lds, k4);
    if (rc != NAF_OK) return rc;
    static int raised_dev[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!raised_dev[dev]) {
        const void* ks[32] = {
            (const void*)adam_act_kernel<NAF_P_HADAMARD, 1>, (const void*)adam_act_kernel<NAF_P_HADAMARD, 2>,
            (const void*)adam_act_kernel<NAF_P_HADAMARD, 3>, (const void*)adam_act_kernel<NAF_P_HADAMARD, 4>,
            (const void*)adam_act_kernel<NAF_P_MATMUL, 1>, (const void*)adam_act_kernel<NAF_P_MATMUL, 2>,
            (const void*)adam_act_kernel<NAF_P_MATMUL, 3>, (const void*)adam_act_kernel<NAF_P_MATMUL, 4>,
            (const void*)adam_act_kernel<NAF_P_HADAMARD, 1, 16>, (const void*)adam_act_kernel<NAF_P_HADAMARD, 2, 16>,
            (const void*)adam_act_kernel<NAF_P_HADAMARD, 3, 16>, (const void*)adam_act_kernel<NAF_P_HADAMARD, 4, 16>,
            (const void*)adam_act_kernel<NAF_P_MATMUL, 1, 16>, (const void*)adam_act_kernel<NAF_P_MATMUL, 2, 16>,
            (const void*)adam_act_kernel<NAF_P_MATMUL, 3, 16>, (const void*)adam_act_kernel<NAF_P_MATMUL, 4, 16>,
            (const void*)adam_act_kernel<NAF_P_HADAMARD, 1, 8, 512>, (const void*)adam_act_kernel<NAF_P_HADAMARD, 2, 8, 512>,
            (const void*)adam_act_kernel<NAF_P_HADAMARD, 3, 8, 512>, (const void*)adam_act_kernel<NAF_P_HADAMARD, 4, 8, 512>,
            (const void*)adam_act_kernel<NAF_P_MATMUL, 1, 8, 512>, (const void*)adam_act_kernel<NAF_P_MATMUL, 2, 8, 512>,
            (const void*)adam_act_kernel<NAF_P_MATMUL, 3, 8, 512>, (const void*)adam_act_kernel<NAF_P_MATMUL, 4, 8, 512>,
            (const void*)adam_act_kernel<NAF_P_HADAMARD, 1, 16, 512>, (const void*)adam_act_kernel<NAF_P_HADAMARD, 2, 16, 512>,
            (const void*)adam_act_kernel<NAF_P_HADAMARD, 3, 16, 512>, (const void*)adam_act_kernel<NAF_P_HADAMARD, 4, 16, 512>,
            (const void*)adam_act_kernel<NAF_P_MATMUL, 1, 16, 512>, (const void*)adam_act_kernel<NAF_P_MATMUL, 2, 16, 512>,
            (const void*)adam_act_kernel<NAF_P_MATMUL, 3, 16, 512>, (const void*)adam_act_kernel<NAF_P_MATMUL, 4, 16, 512>};
        for (const void* k : ks) {
            hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, SP_MAX_DYN_LDS);
            if (e != hipSuccess) return (int)e;
        }
        raised_dev[dev] = 1;
    }
    const int spec = (k4 <= 6 ? 1 : 3) + (sp_cached(prefetch->B, prefetch->out_ld) ? 0 : 1);
    const hipStream_t st = (hipStream_t)stream;
#define AA_LAUNCH(PM, SV, GV, HVV) adam_act_kernel<PM, SV, GV, HVV><<<grid + 1, SP_THREADS, lds, st>>>(P, SP, L1)
#define AA_LAUNCH_SV(PM, SV)                              \
    do {                                                  \
        if (wide && h512) AA_LAUNCH(PM, SV, 16, 512);     \
        else if (wide) AA_LAUNCH(PM, SV, 16, 256);        \
        else if (h512) AA_LAUNCH(PM, SV, 8, 512);         \
        else AA_LAUNCH(PM, SV, 8, 256);                   \
    } while (0)
#define AA_LAUNCH_PM(PM)                    \
    switch (spec) {                         \
        case 1: AA_LAUNCH_SV(PM, 1); break; \
        case 2: AA_LAUNCH_SV(PM, 2); break; \
        case 3: AA_LAUNCH_SV(PM, 3); break; \
        default: AA_LAUNCH_SV(PM, 4); break; \
    }
    if (p_mode == NAF_P_HADAMARD) { AA_LAUNCH_PM(NAF_P_HADAMARD) } else { AA_LAUNCH_PM(NAF_P_MATMUL) }
#undef AA_LAUNCH_PM
#undef AA_LAUNCH_SV
#undef AA_LAUNCH
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}

extern "C" int naf_adam_polyak_act(const naf_adam_args_t* adam, const naf_act_net_t* net, const float* obs, float* heads_out,
                                   float* action_out, uint64_t seed, uint64_t* counter_dev, float noise_scale, int p_mode,
                                   int32_t* sync, uint64_t* host_errors, uint32_t* host_seq, uint32_t* action_rec,
                                   const naf_step_prefetch_t* prefetch, int obs_system_scope, void* stream) {
    return aa_launch(adam, net, obs, heads_out, action_out, seed, counter_dev, noise_scale, p_mode, sync, host_errors, host_seq, action_rec,
                     prefetch, obs_system_scope, nullptr, stream);
}
extern "C" int naf_adam_polyak_act_layer1(const naf_adam_args_t* adam, const naf_act_net_t* net, const float* obs, float* heads_out,
                                          float* action_out, uint64_t seed, uint64_t* counter_dev, float noise_scale, int p_mode,
                                          int32_t* sync, uint64_t* host_errors, uint32_t* host_seq, uint32_t* action_rec,
                                          const naf_step_prefetch_t* prefetch, int obs_system_scope, const naf_bb_layer1_t* layer1,
                                          void* stream) {
    if (!layer1) return NAF_ERR_ARG;
    return aa_launch(adam, net, obs, heads_out, action_out, seed, counter_dev, noise_scale, p_mode, sync, host_errors, host_seq, action_rec,
                     prefetch, obs_system_scope, layer1, stream);
}

// ---- the prefetch as a launch of its own ------------------------------------------------------------------------------------------
static int spf_launch(const naf_step_prefetch_t* pf, hipStream_t st) {
    if (!pf || !pf->spec_rec || !pf->idx_spec || (pf->mode != 1 && pf->mode != 2)) return NAF_ERR_ARG;
    if (pf->mode == 2 && (!pf->src_row || !pf->n_word)) return NAF_ERR_ARG;
    const bool app = pf->mode == 2;
    StepPrepArgs SP;
    memset(&SP, 0, sizeof(SP));
    size_t lds = 0;
    int k4 = 0;
    // (mode 2 consumes the record / indices `spec_rec_in` / `idx_spec_in` name — NULL: the ones it leaves, as at depth 1)
    int32_t* rec_in = app ? (pf->spec_rec_in ? pf->spec_rec_in : pf->spec_rec) : nullptr;
    int32_t* idx_in = app ? (pf->idx_spec_in ? pf->idx_spec_in : pf->idx_spec) : nullptr;
    const int rc = sp_build(pf->replay, app ? pf->src_row : nullptr, app ? pf->n_word : nullptr, app ? pf->row_out : nullptr, pf->seed,
                            pf->counter_dev, app ? pf->idx_out : nullptr, pf->out_rows, pf->out_ld, pf->action_mode, pf->mom, pf->B,
                            pf->without_replacement, pf->spec_rec, pf->idx_spec, &pf->copies, pf->host_spec, pf->pipe_errors, rec_in,
                            idx_in, pf->depth ? pf->depth : 1, pf->pf_seq, SP, lds, k4);
    if (rc != NAF_OK) return rc;
    static int raised_dev[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!raised_dev[dev]) {
        const void* ks[8] = {(const void*)step_prefetch_kernel<6, true, 1>, (const void*)step_prefetch_kernel<6, false, 1>,
                             (const void*)step_prefetch_kernel<8, true, 1>, (const void*)step_prefetch_kernel<8, false, 1>,
                             (const void*)step_prefetch_kernel<6, true, 2>, (const void*)step_prefetch_kernel<6, false, 2>,
                             (const void*)step_prefetch_kernel<8, true, 2>, (const void*)step_prefetch_kernel<8, false, 2>};
        for (const void* k : ks) {
            hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, SP_MAX_DYN_LDS);
            if (e != hipSuccess) return (int)e;
        }
        raised_dev[dev] = 1;
    }
    const bool cache = sp_cached(pf->B, pf->out_ld);
#define SPF(K, C, M) step_prefetch_kernel<K, C, M><<<1, SP_THREADS, lds, st>>>(SP)
    if (app) {
        if (k4 <= 6 && cache) SPF(6, true, 2); else if (k4 <= 6) SPF(6, false, 2); else if (cache) SPF(8, true, 2); else SPF(8, false, 2);
    } else {
        if (k4 <= 6 && cache) SPF(6, true, 1); else if (k4 <= 6) SPF(6, false, 1); else if (cache) SPF(8, true, 1); else SPF(8, false, 1);
    }
#undef SPF
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}
extern "C" int naf_step_prefetch(const naf_step_prefetch_t* prefetch, void* stream) { return spf_launch(prefetch, (hipStream_t)stream); }

// One timestep of the pipelined path in ONE foreign call: the transition row into device memory (naf_host_publish), the timestep's
// graph on `stream`, and — prefetch != NULL — the append + depth-2 prefetch on `side_stream`, beside the graph.
extern "C" int naf_step_launch(void* dst_device, const void* src_host, size_t bytes, void* graph_exec, void* stream,
                               const naf_step_prefetch_t* prefetch, void* side_stream, int prefetch_first) {
    if (!graph_exec) return NAF_ERR_ARG;
    if (bytes) {
        if (!dst_device || !src_host) return NAF_ERR_ARG;
        memcpy(dst_device, src_host, bytes);
        __builtin_ia32_sfence();
    }
    // Which of the two launches goes first is free (neither reads what the other writes: engine._Pipeline's docstring) and decides what
    // the host gets sooner: hipGraphLaunch keeps this thread 8 us, so the launch behind it starts 8 us late. The graph first: the
    // action. The prefetch first: its verdict, which the host must have read before the NEXT call — the caller asks for that when it
    // had to wait for the last one (see _Pipeline.collect: the loop has a second stable state in which every tick waits for a
    // verdict that left 8 us late because the tick before did)
    if (prefetch && prefetch_first) {
        const int rc = spf_launch(prefetch, (hipStream_t)side_stream);
        if (rc != NAF_OK) return rc;
    }
    hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    return prefetch && !prefetch_first ? spf_launch(prefetch, (hipStream_t)side_stream) : NAF_OK;
}

extern "C" int naf_adam_polyak_act_sync_ints(void) { return AA_SYNC_RIDE + 128; }     // (the 512-wide launch's records, the riders' words)
