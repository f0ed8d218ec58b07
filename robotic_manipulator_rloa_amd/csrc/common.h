// Shared host/device helpers for the NAF hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NAF_OK 0
#define NAF_ERR_ARG (-1)       // bad argument (null pointer, out-of-range size, unsupported A)
#define NAF_ERR_STATE (-2)     // bad handle / state
// positive return values are hipError_t codes

#define NAF_MAX_A 8            // one sample per 8-lane group: A <= 8 (every fused kernel)
#define NAF_MAX_A_WIDE 64      // ... per 16- / 32- / 64-lane group: the stand-alone head kernels and the replay ring take up to 64 joints
#define NAF_WAVE 64

#define NAF_CHECK_LAUNCH()                                    \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) return (int)e__;               \
    } while (0)

// ---------------------------------------------------------------------------------------------
// Kernel timeline (development aid, compiled in with NAF_BUILD_DEFINES=-DNAF_TIMELINE only): thread 0 of the FIRST and of
// the LAST workgroup of a launch leaves the 100 MHz wall clock at up to 16 marks per kernel; naf_timeline_read() copies
// them out (benchmarks/kernel_timeline.py lays one update's launches side by side: phases inside a kernel, gaps between
// kernels). Without the define the marks compile to nothing and naf_timeline_read() returns NAF_ERR_STATE.
// ---------------------------------------------------------------------------------------------
#define NAF_TL_SLOTS 16
enum { NAF_TL_BB_LAYER1 = 0, NAF_TL_BB_LINEAR_STATS, NAF_TL_BB_LAYER2_HEAD, NAF_TL_BB_STAGE2, NAF_TL_GEMM_BUNDLE, NAF_TL_BB_FINISH,
       NAF_TL_ADAM, NAF_TL_STEP_PREP, NAF_TL_ADAM_ACT, NAF_TL_KERNELS };
#ifdef NAF_TIMELINE
#define NAF_TL_DECL(arr) __device__ long long arr[NAF_TL_KERNELS][2][NAF_TL_SLOTS]
#define NAF_TL(arr, kid, slot)                                                                                       \
    do {                                                                                                             \
        __builtin_amdgcn_sched_barrier(0);       /* the clock read stays where the mark is written */                \
        if (threadIdx.x == 0) {                                                                                      \
            if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) arr[kid][0][slot] = wall_clock64();             \
            if (blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1 && blockIdx.z == gridDim.z - 1)             \
                arr[kid][1][slot] = wall_clock64();                                                                  \
        }                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    } while (0)
// the same with the caller saying which workgroup is the first / the last (one-dimensional grids that carry the workgroups
// of another job behind the kernel's own)
#define NAF_TL_FL(arr, kid, slot, is_first, is_last)                                                                 \
    do {                                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        if (threadIdx.x == 0) {                                                                                      \
            if (is_first) arr[kid][0][slot] = wall_clock64();                                                        \
            if (is_last) arr[kid][1][slot] = wall_clock64();                                                         \
        }                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    } while (0)
#define NAF_TL_FL_T(arr, kid, slot, is_first, is_last, thread) /* the mark written by thread `thread` instead of thread 0 */ \
    do {                                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        if ((int)threadIdx.x == (thread)) {                                                                          \
            if (is_first) arr[kid][0][slot] = wall_clock64();                                                        \
            if (is_last) arr[kid][1][slot] = wall_clock64();                                                         \
        }                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    } while (0)
#define NAF_TL_READER(fn, arr)                                                                                        \
    int fn(int kid, long long* out) {                                                                                 \
        return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(arr), 2 * NAF_TL_SLOTS * sizeof(long long),                    \
                                        (size_t)kid * 2 * NAF_TL_SLOTS * sizeof(long long), hipMemcpyDeviceToHost);   \
    }
#else
#define NAF_TL_DECL(arr)
#define NAF_TL(arr, kid, slot) do { } while (0)
#define NAF_TL_FL(arr, kid, slot, is_first, is_last) do { } while (0)
#define NAF_TL_FL_T(arr, kid, slot, is_first, is_last, thread) do { } while (0)
#define NAF_TL_READER(fn, arr) int fn(int, long long*) { return NAF_ERR_STATE; }
#endif

// ---------------------------------------------------------------------------------------------
// Buffer loads with SCALAR address arithmetic. A kernel prologue that requests ~50 operands per thread through flat/global
// loads spends ~5 vector instructions per load on 64-bit addresses; with 8 waves per workgroup (two per SIMD) that was 2 us
// of VALU issue in front of bb_layer2_head_kernel's first byte (benchmarks/kernel_timeline.py + the ISA). buffer_load takes
// a wave-uniform resource (SGPRs), a wave-uniform byte offset (SGPR) and ONE 32-bit lane offset (VGPR) shared by every load
// of the batch: the per-load arithmetic moves to the scalar unit. Bases and scalar offsets MUST be wave-uniform (derive them
// from blockIdx, kernel arguments and __builtin_amdgcn_readfirstlane(wave)); all offsets are bytes, < 2 GiB from the base.
// Reads past `bytes` return 0.
// ---------------------------------------------------------------------------------------------
#ifdef __HIPCC__
typedef float naf_f32x2 __attribute__((ext_vector_type(2)));
typedef float naf_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned naf_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned naf_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ static __amdgpu_buffer_rsrc_t naf_buf(const void* base, unsigned bytes = 0x7fffffffu) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ static float naf_buf_f1(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned wave_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane_off, wave_off, 0));
}
__device__ __forceinline__ static naf_f32x2 naf_buf_f2(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned wave_off) {
    return __builtin_bit_cast(naf_f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, lane_off, wave_off, 0));
}
__device__ __forceinline__ static naf_f32x4 naf_buf_f4(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned wave_off) {
    return __builtin_bit_cast(naf_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, wave_off, 0));
}
// Stores of a kernel's bulk outputs (what the NEXT launch reads, on other XCDs), optionally sc0 sc1 = write-through to the
// level every XCD sees: the lines are then clean when the kernel ends and the release at the kernel boundary has less to
// write back. A/B over the whole chain (updates/s, write-through | plain): B = 256 30.0k | 30.2k, 1024 22.7k | 22.4k, 2048
// 16.6k | 16.1k — it pays once a launch leaves megabytes dirty, so `wt` (wave-uniform) is B >= NAF_WT_MIN_B. (nt, the
// streaming hint, changed nothing.)
#ifndef NAF_WT_MIN_B
#define NAF_WT_MIN_B 1024
#endif
__device__ __forceinline__ static void naf_buf_st_f1(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned wave_off, float v, bool wt) {
    if (wt) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, lane_off, wave_off, 17);
    else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, lane_off, wave_off, 0);
}
// sc1 accesses (aux bit 4): the two sides of a hand-off between workgroups of ONE launch (csrc/gemm_bundle.hip, the
// BatchNorm-backward constants): stores written through to where every XCD reads them, loads that do not take a stale line
__device__ __forceinline__ static naf_f32x4 naf_buf_f4_sc1(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned wave_off) {
    return __builtin_bit_cast(naf_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, wave_off, 16));
}
__device__ __forceinline__ static void naf_buf_st_f4_sc1(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned wave_off, naf_f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(naf_u32x4, v), r, lane_off, wave_off, 16);
}
__device__ __forceinline__ static void naf_buf_st_f4(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned wave_off, naf_f32x4 v, bool wt) {
    if (wt) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(naf_u32x4, v), r, lane_off, wave_off, 17);
    else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(naf_u32x4, v), r, lane_off, wave_off, 0);
}
#endif

// ---------------------------------------------------------------------------------------------
// Cross-lane sums without the LDS crossbar. hipcc lowers __shfl_xor / __shfl to ds_bpermute_b32 (an LDS-pipe round trip,
// ~100 cycles each, and these kernels are chains of short dependent phases); DPP modifiers ride on the add itself, and
// gfx950's v_permlane16_swap / v_permlane32_swap exchange 16- / 32-lane halves in the VALU.
//   naf_xor1/2/8_add : v + v[lane ^ m], exact partners (quad_perm, row_ror:8)
//   naf_xor16/32_add : v + v[lane ^ 16 / 32] (permlane swaps)
//   naf_sum8 / 16 / 64: all-reduce over aligned groups of 8 / 16 / 64 lanes, every lane gets the sum. Levels 4 and 8 use
//     row_half_mirror / row_mirror: after the quad levels every lane of a quad holds the quad's sum, so the mirrored
//     partner holds exactly what the xor partner would — bitwise the xor-tree result in the order 1, 2, 4, 8, 16, 32.
// ---------------------------------------------------------------------------------------------
#ifdef __HIPCC__
template <int CTRL>
__device__ __forceinline__ static float naf_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ static float naf_xor1_add(float v) { return v + naf_dpp<0xB1>(v); }     // quad_perm [1,0,3,2]
__device__ __forceinline__ static float naf_xor2_add(float v) { return v + naf_dpp<0x4E>(v); }     // quad_perm [2,3,0,1]
__device__ __forceinline__ static float naf_xor8_add(float v) { return v + naf_dpp<0x128>(v); }    // row_ror:8
// (the swap instructions exchange rows BETWEEN their two operand registers and return both. Given ONE value for both
//  operands, hipcc 7.2 treats the two results as equal and adds the first to itself — the sum came out as 2 v
//  (benchmarks/probe/lane_ops_probe.hip caught it; the ISA showed v_add_f32 v, v6, v6). The empty asm statements make the
//  second operand and the two results values of their own.)
__device__ __forceinline__ static float naf_xor16_add(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    unsigned u2 = u;
    asm volatile("" : "+v"(u2));
    const auto p = __builtin_amdgcn_permlane16_swap(u, u2, false, false);  // (rows 0,0,2,2) and (rows 1,1,3,3)
    unsigned a = p[0], b = p[1];
    asm volatile("" : "+v"(a), "+v"(b));
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__device__ __forceinline__ static float naf_xor32_add(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    unsigned u2 = u;
    asm volatile("" : "+v"(u2));
    const auto p = __builtin_amdgcn_permlane32_swap(u, u2, false, false);  // (low half twice) and (high half twice)
    unsigned a = p[0], b = p[1];
    asm volatile("" : "+v"(a), "+v"(b));
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__device__ __forceinline__ static float naf_sum8(float v) {
    v = naf_xor2_add(naf_xor1_add(v));
    return v + naf_dpp<0x141>(v);                           // row_half_mirror
}
__device__ __forceinline__ static float naf_sum16(float v) {
    v = naf_sum8(v);
    return v + naf_dpp<0x140>(v);                           // row_mirror
}
__device__ __forceinline__ static float naf_sum64(float v) { return naf_xor32_add(naf_xor16_add(naf_sum16(v))); }
#endif

__host__ __device__ static inline int naf_round_up(int x, int m) { return (x + m - 1) / m * m; }

// experiment switches of the host-side launchers: read once per call site, not per launch
#include <stdlib.h>
#define NAF_ENV_INT(name, dflt) ([]() -> int { static const int v = getenv(name) ? atoi(getenv(name)) : (dflt); return v; }())

// transition row layout: [state(S) | action(A) | reward | 0-pad to a multiple of 4 floats | next_state(S) | done | 0-pad]
// next_state starts on a 16-byte boundary so both observations of a row can be read as float4
__host__ __device__ static inline int naf_row_off_s2(int S, int A) { return naf_round_up(S + A + 1, 4); }
__host__ __device__ static inline int naf_row_off_done(int S, int A) { return naf_row_off_s2(S, A) + S; }

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011), counter-based: used by the replay sampler and by the
// exploration noise. The numpy restatement in oracle/naf_oracle.py must produce the same words.
// ---------------------------------------------------------------------------------------------
struct Philox4 {
    uint32_t v[4];
};

__host__ __device__ static inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                        uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c0;
        uint64_t p1 = (uint64_t)M1 * c2;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += W0; k1 += W1;
    }
    Philox4 out;
    out.v[0] = c0; out.v[1] = c1; out.v[2] = c2; out.v[3] = c3;
    return out;
}

// uniform in (0,1) from the top 24 bits of a 32-bit word (exactly representable in f32)
__host__ __device__ static inline float naf_u01(uint32_t x) {
    return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f);
}
