// Cross-lane helpers of csrc/common.h against their definition, lane by lane (exact small integers in f32).
//   hipcc --offload-arch=gfx950 -O2 -I robotic_manipulator_rloa_amd/csrc benchmarks/probe/lane_ops_probe.hip -o /tmp/lane_ops && /tmp/lane_ops
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include "common.h"

__global__ void k(const float* in, float* out) {
    const int l = threadIdx.x;
    const float v = in[l];
    out[0 * 64 + l] = naf_xor1_add(v);
    out[1 * 64 + l] = naf_xor2_add(v);
    out[2 * 64 + l] = naf_xor8_add(v);
    out[3 * 64 + l] = naf_xor16_add(v);
    out[4 * 64 + l] = naf_xor32_add(v);
    out[5 * 64 + l] = naf_sum8(v);
    out[6 * 64 + l] = naf_sum16(v);
    out[7 * 64 + l] = naf_sum64(v);
}

int main() {
    float h[64], *din, *dout, r[8 * 64];
    for (int i = 0; i < 64; ++i) h[i] = (float)(1 + i * i % 37 + 3 * i);
    hipMalloc(&din, sizeof(h));
    hipMalloc(&dout, sizeof(r));
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 64>>>(din, dout);
    hipMemcpy(r, dout, sizeof(r), hipMemcpyDeviceToHost);
    const char* names[8] = {"xor1_add", "xor2_add", "xor8_add", "xor16_add", "xor32_add", "sum8", "sum16", "sum64"};
    int bad = 0;
    for (int t = 0; t < 8; ++t) {
        int nb = 0;
        for (int l = 0; l < 64; ++l) {
            float want = 0.f;
            if (t < 5) {
                const int m = t == 0 ? 1 : t == 1 ? 2 : t == 2 ? 8 : t == 3 ? 16 : 32;
                want = h[l] + h[l ^ m];
            } else {
                const int gsz = t == 5 ? 8 : t == 6 ? 16 : 64;
                for (int j = 0; j < gsz; ++j) want += h[(l / gsz) * gsz + j];
            }
            if (r[t * 64 + l] != want) {
                if (nb < 4) printf("  %s lane %d: got %g want %g\n", names[t], l, r[t * 64 + l], want);
                ++nb;
            }
        }
        printf("%-10s %s (%d lanes differ)\n", names[t], nb ? "WRONG" : "ok", nb);
        bad += nb;
    }
    return bad != 0;
}
