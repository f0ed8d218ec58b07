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
extern "C" int naf_timeline_read(int kernel_id, long long* out) {
    if (!out || kernel_id < 0 || (kernel_id >= NAF_TL_KERNELS && kernel_id < 1024)) return NAF_ERR_ARG;
    if (hipDeviceSynchronize() != hipSuccess) return NAF_ERR_STATE;
    if (kernel_id >= 1024) return naf_tl_read_gb_wg(16 * (kernel_id - 1024), out);   // gemm_bundle, entry / exit per workgroup
    if (kernel_id == NAF_TL_GEMM_BUNDLE) return naf_tl_read_gb(kernel_id, out);
    if (kernel_id == NAF_TL_ADAM) return naf_tl_read_opt(kernel_id, out);
    return naf_tl_read_bb(kernel_id, out);
}
