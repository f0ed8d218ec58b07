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
