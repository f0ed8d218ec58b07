// Probe of the LDS-DMA instruction the ring GEMM (csrc/gemm_ring.hip) is built on: `buffer_load_dwordx4 ... offen lds` on gfx950.
// Checks, against a host copy: (1) the destination of one wave-instruction is M0 + 16 * lane (1 KB, lane-linear) while the SOURCE
// address is per lane (a swizzle goes on the source side); (2) M0 may point anywhere in a 120-KB LDS allocation (beyond 64 KB);
// (3) a counted s_waitcnt vmcnt(N) + s_barrier makes pieces issued by OTHER waves readable.
// Build + run (GPU box):  hipcc -O3 --offload-arch=gfx950 benchmarks/probe/ldsdma_probe.hip -o /tmp/ldsdma_probe && /tmp/ldsdma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define LDS_FLOATS (30 * 1024)        // 120 KB

__device__ __forceinline__ static void dma16(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, unsigned lds_byte) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rs), "s"(soff), "s"(lds_byte)
                 : "memory");
}

// 4 waves; wave w issues pieces w, w + 4, ... of `n_pieces`; piece p lands at LDS byte base[p]; lane L of piece p reads
// source float4 index p * 64 + (L ^ (p & 7))  (a per-lane source permutation)
__global__ __launch_bounds__(256) void probe(const float* src, float* out, const unsigned* base, int n_pieces) {
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < LDS_FLOATS; i += 256) lds[i] = -1.f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
    const unsigned lds0 = (unsigned)(uintptr_t)(&lds[0]);      // LDS byte address of the array (0 for the only array)
    for (int p = wave; p < n_pieces; p += 4) {
        const unsigned b = __builtin_amdgcn_readfirstlane(base[p]);
        dma16(rs, (unsigned)((lane ^ (p & 7)) * 16), (unsigned)(p * 1024), lds0 + b);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = tid; i < LDS_FLOATS; i += 256) out[i] = lds[i];
}

int main() {
    const int n_pieces = 24;
    std::vector<unsigned> base(n_pieces);
    // pieces scattered over the allocation, including beyond 64 KB and up to the last KB
    for (int p = 0; p < n_pieces; ++p) base[p] = (unsigned)((p * 5 * 1024) % (119 * 1024));
    base[n_pieces - 1] = 119 * 1024;
    base[n_pieces - 2] = 65 * 1024;
    std::vector<float> src(n_pieces * 256), out(LDS_FLOATS), want(LDS_FLOATS, -1.f);
    for (size_t i = 0; i < src.size(); ++i) src[i] = (float)i;
    for (int p = 0; p < n_pieces; ++p)
        for (int L = 0; L < 64; ++L)
            for (int e = 0; e < 4; ++e) want[base[p] / 4 + L * 4 + e] = src[(size_t)p * 256 + (L ^ (p & 7)) * 4 + e];
    float *dsrc, *dout;
    unsigned* dbase;
    hipMalloc(&dsrc, src.size() * 4);
    hipMalloc(&dout, out.size() * 4);
    hipMalloc(&dbase, base.size() * 4);
    hipMemcpy(dsrc, src.data(), src.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dbase, base.data(), base.size() * 4, hipMemcpyHostToDevice);
    probe<<<1, 256>>>(dsrc, dout, dbase, n_pieces);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return 1; }
    hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < LDS_FLOATS; ++i)
        if (out[i] != want[i]) {
            if (bad < 10) printf("  lds float %d (byte %d): got %g want %g\n", i, 4 * i, out[i], want[i]);
            ++bad;
        }
    printf("LDSDMA_PROBE %s: %d mismatches over %d floats, %d pieces (bases up to byte %u)\n", bad ? "FAIL" : "OK", bad, LDS_FLOATS,
           n_pieces, 119 * 1024);
    return bad != 0;
}
