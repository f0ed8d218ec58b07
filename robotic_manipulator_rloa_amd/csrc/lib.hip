// Library identity for libnaf_hip.so (gfx950 only: no other code object is built).
#include "common.h"
#include "../../include/naf_hip.h"

extern "C" int naf_hip_abi_version(void) { return NAF_HIP_ABI_VERSION; }
extern "C" const char* naf_hip_arch(void) { return "gfx950"; }

// kernel timeline (common.h): one array per translation unit (no relocatable device code in this build)
int naf_tl_read_bb(int kid, long long* out);
int naf_tl_read_gb(int kid, long long* out);
int naf_tl_read_opt(int kid, long long* out);
int naf_tl_read_gb_wg(int first, long long* out);
int naf_tl_read_sp(int kid, long long* out);
extern "C" int naf_timeline_read(int kernel_id, long long* out) {
    if (!out || kernel_id < 0 || (kernel_id >= NAF_TL_KERNELS && kernel_id < 1024)) return NAF_ERR_ARG;
    if (hipDeviceSynchronize() != hipSuccess) return NAF_ERR_STATE;
    if (kernel_id >= 2048) return naf_tl_read_sp(kernel_id - 2048, out);               // (layer 1 riding on adam_act_kernel: its marks, 2048 + NAF_TL_BB_LAYER1)
    if (kernel_id >= 1024) return naf_tl_read_gb_wg(16 * (kernel_id - 1024), out);   // gemm_bundle, entry / exit per workgroup
    if (kernel_id == NAF_TL_GEMM_BUNDLE) return naf_tl_read_gb(kernel_id, out);
    if (kernel_id == NAF_TL_ADAM) return naf_tl_read_opt(kernel_id, out);
    if (kernel_id == NAF_TL_STEP_PREP || kernel_id == NAF_TL_ADAM_ACT) return naf_tl_read_sp(kernel_id, out);
    return naf_tl_read_bb(kernel_id, out);
}

// Host-side half of a hand-over THROUGH DEVICE MEMORY: the host stores `bytes` bytes straight into device memory (every device
// allocation is mapped for the CPU on this platform: large BAR) and fences, so that the stores have left the CPU's write-combining
// buffers before whatever the caller does next — ring a doorbell. The reading kernel must load with system scope (sc0 sc1): the
// L2 of its XCD may still hold the line from the previous read. A kernel that reads its input this way pays a local memory
// latency on its first dependent load instead of a PCIe round trip to pinned host memory (~0.8 against ~2.8 us on MI355X).
#include <string.h>
// 1: the CPU can store into this device's memory (the runtime reports a large BAR); 0: it cannot — callers keep pinned host
// memory and let the kernel read across PCIe; < 0: error. (A store through an unmapped device pointer is a segmentation fault:
// nobody calls naf_host_publish without asking here first.)
extern "C" int naf_host_store_supported(int device) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeIsLargeBar, device) != hipSuccess) return NAF_ERR_STATE;
    return v ? 1 : 0;
}
extern "C" int naf_host_publish(void* dst_device, const void* src_host, size_t bytes) {
    if (!dst_device || !src_host) return NAF_ERR_ARG;
    memcpy(dst_device, src_host, bytes);
    __builtin_ia32_sfence();
    return NAF_OK;
}

// ... and the launch that reads them, in the same call: naf_host_publish (bytes == 0: nothing to publish) followed by
// hipGraphLaunch of an instantiated graph on `stream` — what NAFAgent.step() does per timestep, in one trip through the
// foreign-function interface instead of three (the store, the fence, torch's CUDAGraph.replay).
extern "C" int naf_host_publish_launch(void* dst_device, const void* src_host, size_t bytes, void* graph_exec, void* stream) {
    if (!graph_exec) return NAF_ERR_ARG;
    if (bytes) {
        if (!dst_device || !src_host) return NAF_ERR_ARG;
        memcpy(dst_device, src_host, bytes);
        __builtin_ia32_sfence();
    }
    hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
    return e == hipSuccess ? NAF_OK : (int)e;
}

// ---- a device allocation of the library's own for such hand-overs, and the proof that it works ------------------------------------
// The framework allocator's segments need not be CPU-mapped (torch's expandable segments come from hipMemCreate / hipMemMap with
// access for the device only; a memory pool may have none for the host): a store through such a pointer is a segmentation fault,
// not an error code. A plain hipMalloc on a large-BAR device is mapped — and naf_host_store_selftest shows it: `n` distinct patterns
// stored through naf_host_publish and read back by a kernel with the loads the per-timestep kernels use (system scope: sc0 sc1).
// Returns the number of mismatching patterns (0 = the hand-over works), < 0 on an error.
#define HS_WORDS 68                       // a ring row of 64 floats + its count, rounded up
extern "C" int naf_host_store_alloc(size_t bytes, void** out) {
    if (!out || bytes == 0) return NAF_ERR_ARG;
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) return NAF_ERR_STATE;
    if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipFree(p);
        return NAF_ERR_STATE;
    }
    *out = p;
    return NAF_OK;
}
extern "C" int naf_host_store_free(void* p) { return p && hipFree(p) != hipSuccess ? NAF_ERR_STATE : NAF_OK; }

__host__ __device__ static inline unsigned hs_pattern(unsigned i, unsigned w) {
    unsigned x = (i + 1u) * 2654435761u ^ (w + 1u) * 40503u;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    return x;
}
__global__ void host_store_check_kernel(const unsigned* p, unsigned i, int words, int* bad) {
    const int w = threadIdx.x;
    if (w >= words) return;
    const unsigned got = __builtin_amdgcn_raw_buffer_load_b32(naf_buf(p, 4u * (unsigned)words), 4u * (unsigned)w, 0, 17);
    if (got != hs_pattern(i, (unsigned)w)) atomicAdd(bad, 1);
}
extern "C" int naf_host_store_selftest(void* dst_device, int n, void* stream) {
    if (!dst_device || n <= 0) return NAF_ERR_ARG;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, dst_device) != hipSuccess || at.type != hipMemoryTypeDevice) return NAF_ERR_ARG;
    const hipStream_t st = (hipStream_t)stream;
    int* bad = nullptr;
    if (hipMalloc(&bad, sizeof(int)) != hipSuccess) return NAF_ERR_STATE;
    int rc = NAF_OK, failed = 0;
    unsigned host[HS_WORDS];
    const int words = 65;                // the [row | count] the path publishes
    if (hipMemsetAsync(bad, 0, sizeof(int), st) != hipSuccess) rc = NAF_ERR_STATE;
    for (int i = 0; i < n && rc == NAF_OK; ++i) {
        for (int w = 0; w < words; ++w) host[w] = hs_pattern((unsigned)i, (unsigned)w);
        memcpy(dst_device, host, sizeof(unsigned) * words);
        __builtin_ia32_sfence();
        host_store_check_kernel<<<1, 128, 0, st>>>((const unsigned*)dst_device, (unsigned)i, words, bad);
        int b = 0;
        // (the pattern is rewritten only after its reader has passed: the path's own protocol)
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&b, bad, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)
            rc = NAF_ERR_STATE;
        else if (b != failed) { failed = b; }
    }
    memset(host, 0, sizeof(host));
    memcpy(dst_device, host, sizeof(unsigned) * words);
    __builtin_ia32_sfence();
    (void)hipFree(bad);
    if (rc != NAF_OK) return rc;
    // (mismatching WORDS were counted; report patterns: at least one word wrong)
    return failed > 0 ? (failed + words - 1) / words : 0;
}
