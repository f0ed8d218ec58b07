"""Bulk launches of replay_gather_rows_kernel for rocprofv3 (kernel-trace and --pmc passes): the roofline evidence
for the gather. N-row ring (default 4e6 rows = 1.02 GB, beyond the 256 MiB Infinity Cache), M uniformly random rows
per launch. Prints algorithmic and physical GB/s measured with events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robotic_manipulator_rloa_amd.utils.replay_buffer import ReplayBuffer
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000
M = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1 << 22
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
S, A = 21, 6
buf = ReplayBuffer(N, 256, "cuda", 0, state_size=S, action_size=A)
for lo in range(0, N, 1 << 20):
    n = min(1 << 20, N - lo)
    buf.add_rows_device(torch.randn(n, 64, device="cuda"), n)
idx = torch.randint(0, N, (M,), device="cuda", dtype=torch.int32)
out = torch.empty(M, buf.batch_row_floats, device="cuda")   # packed minibatch rows
def measure():
    for _ in range(2):
        buf.gather_rows(idx, out, M)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        buf.gather_rows(idx, out, M)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
ms = measure()
alg = M * (200 + 200 + 4); phys = M * (256 + buf.batch_row_floats * 4 + 4)   # 4*(2S+A+2) B read + the same written per row
print(f"ring {N} rows ({N*256/2**20:.0f} MiB), {M} rows/launch: {ms:.4f} ms  algorithmic {alg/ms/1e6:.1f} GB/s ({alg/ms/1e6/8000:.3f} of 8 TB/s)  "
      f"physical {phys/ms/1e6:.1f} GB/s; alg bytes/launch {alg}, physical {phys}")
