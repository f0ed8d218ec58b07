"""Times the learn() GEMM shapes under torch's BLAS back-ends (hipBLASLt default, rocBLAS, TunableOp)."""
import os, sys, time
import torch
B, S, H, HP, NHP = 256, 21, 256, 264, 32
dev = "cuda"
def shapes():
    f = dict(device=dev, dtype=torch.float32)
    X2 = torch.randn(2, B, 64, **f)[:, :, :S]
    W1 = torch.randn(2, H, S, **f); W2 = torch.randn(2, H, H, **f); Wh = torch.randn(2, NHP, HP, **f)
    A1 = torch.randn(2, B, H, **f); A2 = torch.randn(2, B, HP, **f)
    dH = torch.randn(B, NHP, **f); dZ = torch.randn(B, H, **f)
    return {
        "bmm1 2x[256x21]@[21x256]": lambda o=torch.empty(2, B, H, **f): torch.bmm(X2, W1.transpose(1, 2), out=o),
        "bmm2 2x[256x256]@[256x256]": lambda o=torch.empty(2, B, H, **f): torch.bmm(A1, W2.transpose(1, 2), out=o),
        "bmmh 2x[256x264]@[264x32]": lambda o=torch.empty(2, B, NHP, **f): torch.bmm(A2, Wh.transpose(1, 2), out=o),
        "gWh [32x256]@[256x264]": lambda o=torch.empty(NHP, HP, **f): torch.mm(dH.t(), A2[0], out=o),
        "dA2 [256x32]@[32x264]": lambda o=torch.empty(B, HP, **f): torch.mm(dH, Wh[0], out=o),
        "gW2 [256x256]^T@[256x256]": lambda o=torch.empty(H, H, **f): torch.mm(dZ.t(), A1[0], out=o),
        "dA1 [256x256]@[256x256]": lambda o=torch.empty(B, H, **f): torch.mm(dZ, W2[0], out=o),
        "gW1 [256x256]^T@[256x21]": lambda o=torch.empty(H, S, **f): torch.mm(dZ.t(), X2[0], out=o),
    }
def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
if mode == "rocblas":
    torch.backends.cuda.preferred_blas_library("cublas")
elif mode == "tunable":
    torch.cuda.tunable.enable(True); torch.cuda.tunable.tuning_enable(True)
    torch.cuda.tunable.set_max_tuning_duration(200); torch.cuda.tunable.set_max_tuning_iterations(50)
print("mode", mode, "blas:", torch.backends.cuda.preferred_blas_library())
tot = 0
for name, fn in shapes().items():
    t = timeit(fn); tot += t
    print(f"{name:34s} {t:8.2f} us")
print("sum", round(tot, 1))
if mode == "tunable":
    torch.cuda.tunable.write_file("/root/repo/gpurun_out/tunable.csv")
