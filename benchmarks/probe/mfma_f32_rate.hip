// What the f32 matrix cores sustain on THIS chip under load: v_mfma_f32_16x16x4_f32 back to back from registers, 16 independent
// accumulators per wave, W waves per SIMD on every CU, random operands — TFLOP/s by HIP events, and the shader clock the chip holds
// meanwhile (s_memtime ticks per 100-MHz wall-clock tick). The data-sheet peak (157.3 TFLOP/s) is 64 FLOP/clk/SIMD at 2.4 GHz; the
// backward GEMM launches of the learner are priced against what this probe reads, not against the data sheet.
// Build + run (GPU box):  hipcc -O3 --offload-arch=gfx950 benchmarks/probe/mfma_f32_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mfma_loop(const float* in, float* out, long long* clocks, int iters) {
    const int tid = threadIdx.x, gid = blockIdx.x * blockDim.x + tid;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = in[(gid * 8 + i) & 0xffff];
        b[i] = in[(gid * 8 + 4 + i) & 0xffff];
    }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    f32x4 s = acc[0];
    for (int i = 1; i < 16; ++i) s += acc[i];
    out[gid] = s[0] + s[1] + s[2] + s[3];
    if (tid == 0 && blockIdx.x == 0) {
        clocks[0] = c1 - c0;
        clocks[1] = w1 - w0;
    }
}

int main() {
    const int n_in = 1 << 16;
    std::vector<float> h(n_in);
    srand(7);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *din, *dout;
    long long* dclk;
    hipMalloc(&din, n_in * 4);
    hipMalloc(&dout, 4096 * 256 * 4);
    hipMalloc(&dclk, 16);
    hipMemcpy(din, h.data(), n_in * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4000;
    printf("waves/SIMD  workgroups  ms        TFLOP/s   shader clock (GHz, while the loop runs)\n");
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int wgs = 256 * wps;                       // 4 waves per workgroup = one per SIMD of a CU
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            mfma_loop<<<wgs, 256>>>(din, dout, dclk, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            long long clk[2];
            hipMemcpy(clk, dclk, 16, hipMemcpyDeviceToHost);
            const double flop = (double)wgs * 4 * iters * 16 * 2048.0;
            if (rep) printf("%9d  %10d  %8.3f  %8.1f  %6.2f\n", wps, wgs, ms, flop / (ms * 1e-3) / 1e12, (double)clk[0] / clk[1] * 0.1);
        }
    }
    return 0;
}
