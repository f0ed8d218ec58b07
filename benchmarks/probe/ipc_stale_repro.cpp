// Reproducer for the "stale slab" observation parked in naf_xgmi_destroy (csrc/xgmi_reduce.hip) — VERDICT r01 item 8.
// Two processes on ONE GPU (fork before any HIP call, a socketpair as the control plane), exactly the memory life cycle of
// the one-shot all-reduce's receive slab:
//   owner A: hipExtMallocWithFlags(uncached) -> hipIpcGetMemHandle -> ... -> hipFree -> hipMalloc (the pages come back)
//   peer  B: hipIpcOpenMemHandle -> a kernel WRITES the slab through the mapping (all XCDs) -> hipIpcCloseMemHandle
// After the slab is gone A allocates ordinary memory of the same size, writes a new pattern with one kernel and reads it with
// another whose workgroups read what OTHER workgroups (other XCDs) wrote. Any word that still shows the peer's pattern, or
// anything but the new one, is counted. Modes = the order of {B closes its mapping, A frees the slab}:
//   0  B closes, then A frees          (the orderly teardown)
//   1  A frees while B still holds the mapping, B closes afterwards
//   2  as 0, and B also READ the slab through its mapping before closing (clean lines left in B-side L2s)
//   3  as 0, but the slab is ordinary hipMalloc memory instead of uncached
// Build: hipcc -O2 --offload-arch=gfx950 ipc_stale_repro.cpp -o ipc_stale_repro ; run: ./ipc_stale_repro [iters] [MiB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(3); } } while (0)

__global__ void fill(uint32_t* p, size_t n, uint32_t tag) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = tag ^ (uint32_t)i;
    __threadfence_system();
}
__global__ void touch(const uint32_t* p, size_t n, unsigned long long* sink) {
    unsigned long long s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s == 0x1234567ull) *sink = s;
}
// workgroup b checks the chunk that workgroup (b + 3) of `fill` wrote: a different XCD under round-robin placement
__global__ void check(const uint32_t* p, size_t n, uint32_t tag, uint32_t old_tag, unsigned long long* bad) {
    const size_t per = (size_t)gridDim.x * blockDim.x;
    const size_t shift = (size_t)3 * blockDim.x;
    unsigned long long wrong = 0, old = 0;
    for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += per) {
        const size_t i = (i0 + shift) % n;
        const uint32_t v = p[i];
        if (v != (tag ^ (uint32_t)i)) { ++wrong; if (v == (old_tag ^ (uint32_t)i)) ++old; }
    }
    if (wrong) { atomicAdd(&bad[0], wrong); atomicAdd(&bad[1], old); }
}

static void xsend(int fd, const void* p, size_t n) { if (write(fd, p, n) != (ssize_t)n) exit(4); }
static void xrecv(int fd, void* p, size_t n) { size_t g = 0; while (g < n) { ssize_t r = read(fd, (char*)p + g, n - g); if (r <= 0) exit(5); g += r; } }

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 40;
    const size_t bytes = (size_t)(argc > 2 ? atoi(argv[2]) : 6) << 20, n = bytes / 4;
    int sp[2];
    if (socketpair(AF_UNIX, SOCK_STREAM, 0, sp)) return 1;
    const pid_t pid = fork();                      // before ANY HIP call in either process
    if (pid == 0) {                                // ---- peer B ----
        close(sp[0]);
        const int fd = sp[1];
        unsigned long long* sink; CK(hipMalloc(&sink, 8));
        for (;;) {
            int cmd; xrecv(fd, &cmd, 4);
            if (cmd < 0) break;
            hipIpcMemHandle_t h; xrecv(fd, &h, sizeof(h));
            uint32_t tag; xrecv(fd, &tag, 4);
            void* m = nullptr;
            CK(hipIpcOpenMemHandle(&m, h, hipIpcMemLazyEnablePeerAccess));
            fill<<<1024, 256>>>((uint32_t*)m, n, tag);
            if (cmd == 2) touch<<<1024, 256>>>((const uint32_t*)m, n, sink);
            CK(hipDeviceSynchronize());
            int ack = 1; xsend(fd, &ack, 4);       // written
            int go; xrecv(fd, &go, 4);             // "close now"
            CK(hipIpcCloseMemHandle(m));
            CK(hipDeviceSynchronize());
            xsend(fd, &ack, 4);                    // closed
        }
        return 0;
    }
    close(sp[1]);
    const int fd = sp[0];
    unsigned long long* bad; CK(hipMalloc(&bad, 16));
    printf("%d iterations per mode, %zu MiB slab\n", iters, bytes >> 20);
    for (int mode = 0; mode < 4; ++mode) {
        unsigned long long tot_wrong = 0, tot_old = 0, same_addr = 0, runs_bad = 0;
        for (int it = 0; it < iters; ++it) {
            void* slab = nullptr;
            if (mode == 3) CK(hipMalloc(&slab, bytes)); else CK(hipExtMallocWithFlags(&slab, bytes, hipDeviceMallocUncached));
            CK(hipMemset(slab, 0, bytes)); CK(hipDeviceSynchronize());
            hipIpcMemHandle_t h; CK(hipIpcGetMemHandle(&h, slab));
            const uint32_t old_tag = 0xA5000000u + (uint32_t)(mode * 1000 + it), new_tag = 0x3C000000u + (uint32_t)(mode * 1000 + it);
            int cmd = mode == 2 ? 2 : 0; xsend(fd, &cmd, 4); xsend(fd, &h, sizeof(h)); xsend(fd, &old_tag, 4);
            int ack; xrecv(fd, &ack, 4);           // B wrote the slab
            int go = 1;
            if (mode == 1) { CK(hipFree(slab)); xsend(fd, &go, 4); xrecv(fd, &ack, 4); }
            else { xsend(fd, &go, 4); xrecv(fd, &ack, 4); CK(hipDeviceSynchronize()); CK(hipFree(slab)); }
            void* x = nullptr; CK(hipMalloc(&x, bytes));
            same_addr += (x == slab);
            CK(hipMemset(bad, 0, 16));
            fill<<<1024, 256>>>((uint32_t*)x, n, new_tag);
            check<<<1024, 256>>>((const uint32_t*)x, n, new_tag, old_tag, bad);
            CK(hipDeviceSynchronize());
            unsigned long long hb[2]; CK(hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost));
            tot_wrong += hb[0]; tot_old += hb[1]; runs_bad += hb[0] != 0;
            CK(hipFree(x));
        }
        printf("mode %d: %llu wrong words (%llu of them the peer's old pattern) in %llu of %d runs; new allocation reused the slab's address %llu times\n",
               mode, tot_wrong, tot_old, runs_bad, iters, same_addr);
    }
    int stop = -1; xsend(fd, &stop, 4);
    waitpid(pid, nullptr, 0);
    return 0;
}
