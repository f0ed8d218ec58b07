"""Bulk-launch roofline sweep of the hand-written streaming kernels: at the bench's real sizes every
launch is latency-bound, so achieved GB/s is quoted on launches big enough to stream —
  replay gather of M rows out of a 4e6-row ring (M = 256 ... 16 Mi), 404 algorithmic B/row
  Polyak (12 B/param) and clip+Adam+Polyak (36 B/param) over R stacked replicas of the flat parameter buffer
  NAF head fwd+bwd over B samples (136 B in + 8 B (r, V') + 4 B q + 128 B d_heads per sample at A=6)
Writes a markdown table to stdout (profiles/r01_roofline_sweep.md)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robotic_manipulator_rloa_amd import _lib
from robotic_manipulator_rloa_amd.learner import NetLayout
from robotic_manipulator_rloa_amd.utils.replay_buffer import ReplayBuffer
from robotic_manipulator_rloa_amd._lib import ptr, stream_ptr

lib = _lib.load()
dev = "cuda"

def timed(fn, reps):
    for _ in range(2):
        rc = fn()
        assert rc in (0, None), f"launch failed with status {rc}"      # a refused launch must not be timed as a fast one
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3

print("| kernel | units per launch | algorithmic bytes | time | achieved | of 8 TB/s |")
print("|---|---|---|---|---|---|")
N = 4_000_000
buf = ReplayBuffer(N, 256, dev, 0, state_size=21, action_size=6)
for lo in range(0, N, 1 << 20):
    n = min(1 << 20, N - lo)
    buf.add_rows_device(torch.randn(n, 64, device=dev), n)
for M in (256, 16384, 65536, 1 << 20, 1 << 22, 1 << 24):
    idx = torch.randint(0, N, (M,), device=dev, dtype=torch.int32)
    out = torch.empty(M, buf.batch_row_floats, device=dev)
    t = timed(lambda: buf.gather_rows(idx, out, M), 200 if M <= 65536 else 10)
    alg = M * 404
    print(f"| replay_gather_rows (ring 4e6 rows = 977 MiB) | {M} rows | {alg/1e6:.2f} MB | {t*1e6:.1f} us | {alg/t/1e9:.0f} GB/s | {alg/t/8e12:.3f} |")
    del idx, out
del buf
torch.cuda.empty_cache()
# the same bulk launch on a ring 16 x the 256-MiB Infinity Cache (16e6 rows = 3.8 GiB): at most 1 / 16 of the row fetches can be
# LLC hits — the HBM-only figure beside the workload's (configs[4]'s 4e6-row ring, where up to a quarter can)
N = 16_000_000
buf = ReplayBuffer(N, 256, dev, 0, state_size=21, action_size=6)
for lo in range(0, N, 1 << 20):
    n = min(1 << 20, N - lo)
    buf.add_rows_device(torch.randn(n, 64, device=dev), n)
for M in (1 << 22, 1 << 24):
    idx = torch.randint(0, N, (M,), device=dev, dtype=torch.int32)
    out = torch.empty(M, buf.batch_row_floats, device=dev)
    t = timed(lambda: buf.gather_rows(idx, out, M), 10)
    alg = M * 404
    print(f"| replay_gather_rows (ring 16e6 rows = 3.8 GiB) | {M} rows | {alg/1e6:.2f} MB | {t*1e6:.1f} us | {alg/t/1e9:.0f} GB/s | {alg/t/8e12:.3f} |")
    del idx, out
del buf
torch.cuda.empty_cache()
P = NetLayout(21, 6, 256).P
for R in (1, 64, 1024):
    n = P * R
    main, tgt = torch.randn(n, device=dev), torch.randn(n, device=dev)
    t = timed(lambda: lib.naf_polyak_update(ptr(tgt), ptr(main), 1e-3, 1 - 1e-3, n, stream_ptr()), 200 if R == 1 else 20)
    alg = 12 * n
    print(f"| polyak_update | {R} x {P} params | {alg/1e6:.2f} MB | {t*1e6:.1f} us | {alg/t/1e9:.0f} GB/s | {alg/t/8e12:.3f} |")
    g, m, v = torch.randn(n, device=dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    npart = (n + 4095) // 4096
    parts = torch.zeros(npart, device=dev)
    step = torch.ones(1, dtype=torch.int32, device=dev)
    lib.naf_grad_norm_partials(ptr(g), n, ptr(parts), None, stream_ptr())
    parts2 = parts[:min(npart, 4096)].contiguous()          # the clip scale only needs SOME partials for timing purposes
    t = timed(lambda: lib.naf_adam_polyak_fused(ptr(main), ptr(g), ptr(m), ptr(v), ptr(tgt), ptr(parts2), parts2.numel(), 1.0, 1e-3, .9, .999,
                                                1e-8, 1e-3, 1 - 1e-3, ptr(step), 1.0, n, stream_ptr()), 200 if R == 1 else 20)
    alg = 36 * n
    print(f"| adam_polyak_fused | {R} x {P} params | {alg/1e6:.2f} MB | {t*1e6:.1f} us | {alg/t/1e9:.0f} GB/s | {alg/t/8e12:.3f} |")
    t = timed(lambda: lib.naf_grad_norm_partials(ptr(g), n, ptr(parts), None, stream_ptr()), 200 if R == 1 else 20)
    alg = 4 * n
    print(f"| grad_norm_partials | {R} x {P} params | {alg/1e6:.2f} MB | {t*1e6:.1f} us | {alg/t/1e9:.0f} GB/s | {alg/t/8e12:.3f} |")
    del main, tgt, g, m, v
for B in (256, 2048, 1 << 20):
    heads = torch.randn(B, 32, device=dev)
    u = torch.trunc(torch.rand(B, 6, device=dev) * 2 - 1)
    r, vn = torch.randn(B, device=dev), torch.randn(B, device=dev)
    q, dh = torch.empty(B, device=dev), torch.empty(B, 32, device=dev)
    lp = torch.zeros((B + 7) // 8, device=dev)
    t = timed(lambda: lib.naf_head_fwd_bwd_mse(ptr(heads), 32, ptr(u), 6, ptr(r), 1, ptr(vn), 1, 0.99, ptr(q), ptr(dh), ptr(lp), B, 6, 0,
                                               stream_ptr()), 200 if B <= 2048 else 20)
    alg = B * (4 * 28 + 24 + 8 + 4 + 4 * 28)
    print(f"| naf_head_fwd_bwd_mse (A=6, Hadamard) | {B} samples | {alg/1e6:.2f} MB | {t*1e6:.1f} us | {alg/t/1e9:.0f} GB/s | {alg/t/8e12:.3f} |")
