"""Per-launch cost (graph replay of N back-to-back launches) of each libnaf_hip.so kernel at the bench shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robotic_manipulator_rloa_amd import _lib
from robotic_manipulator_rloa_amd.learner import Learner
from robotic_manipulator_rloa_amd.utils.replay_buffer import ReplayBuffer
from robotic_manipulator_rloa_amd._lib import ptr, stream_ptr, check

def timeit(fn, n=200):
    for _ in range(5):
        rc = fn()
        assert not isinstance(rc, int) or rc == 0, f"launch refused with status {rc}"
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = Learner(21, 6, 256, B, 1e-3, 1e-3, 0.99, torch.device("cuda"))
lay = L.lay; lib = L.lib; seg = lay.seg; H, HP, NHP, P = lay.H, lay.HP, lay.NHP, lay.P
L.theta2.normal_(0, 0.05); L.G1.normal_(); L.G2.normal_(); L.Gh.normal_(); L.dA2.normal_(); L.dA1.normal_(); L.grad.normal_()
rows = torch.randn(B, 64, device="cuda")
t2p, gp, bnp = L.theta2.data_ptr(), L.grad.data_ptr(), L.bn_stats.data_ptr()
tests = {
 "bn_relu_fwd_train x2nets": lambda: lib.naf_bn_relu_fwd_train(ptr(L.G1), B*H, H, t2p+4*seg["b1"].offset, t2p+4*seg["g1"].offset, t2p+4*seg["be1"].offset, P, bnp, bnp+4*H, 4*H, ptr(L.A1), B*H, H, ptr(L.save_mean[0]), ptr(L.save_invstd[0]), B, H, 2, 0.1, 1e-5, stream_ptr()),
 "bn_relu_bwd": lambda: lib.naf_bn_relu_bwd(ptr(L.dA1), H, ptr(L.G1[0]), H, t2p+4*seg["b1"].offset, ptr(L.A1[0]), H, t2p+4*seg["g1"].offset, ptr(L.save_mean[0,0]), ptr(L.save_invstd[0,0]), ptr(L.dZ1), H, gp+4*seg["g1"].offset, gp+4*seg["be1"].offset, gp+4*seg["b1"].offset, B, H, stream_ptr()),
 "head_fwd_bwd_mse": lambda: lib.naf_head_fwd_bwd_mse(ptr(L.Gh[0]), NHP, rows.data_ptr()+4*lay.off_u, 64, rows.data_ptr()+4*lay.off_r, 64, L.Gh[1].data_ptr()+4*(lay.A+lay.T), NHP, 0.99, ptr(L.q_out), ptr(L.dH), None, B, lay.A, 0, stream_ptr()),
 "grad_norm_partials": lambda: lib.naf_grad_norm_partials(ptr(L.grad), P, ptr(L.partials), ptr(L.step_dev), stream_ptr()),
 "adam_polyak_fused": lambda: lib.naf_adam_polyak_fused(ptr(L.theta2[0]), ptr(L.grad), ptr(L.adam_m), ptr(L.adam_v), ptr(L.theta2[1]), ptr(L.partials), L.n_partials, 1.0, 1e-3, .9, .999, 1e-8, 1e-3, 1-1e-3, ptr(L.step_dev), 1.0, P, stream_ptr()),
 "polyak": lambda: lib.naf_polyak_update(ptr(L.theta2[1]), ptr(L.theta2[0]), 1e-3, 1-1e-3, P, stream_ptr()),
 "counter_add (1 thread)": lambda: lib.naf_counter_add(ptr(L.step_dev), 0, stream_ptr()),
}
HPv = HP
tests["F1 linear_bn_relu_fwd x2"] = lambda: lib.naf_linear_bn_relu_fwd_train(rows.data_ptr(), lay.off_s2, 64, lay.S, t2p+4*seg["W1"].offset, t2p+4*seg["b1"].offset, t2p+4*seg["g1"].offset, t2p+4*seg["be1"].offset, P, bnp, bnp+4*H, 4*H, ptr(L.A1), B*H, H, ptr(L.save_mean[0]), ptr(L.save_invstd[0]), B, H, 2, 0.1, 1e-5, stream_ptr())
tests["B1 bn_relu_bwd_wgrad"] = lambda: lib.naf_bn_relu_bwd_wgrad(ptr(L.dA1), H, rows.data_ptr(), 64, lay.S, t2p+4*seg["W1"].offset, t2p+4*seg["b1"].offset, ptr(L.A1[0]), H, t2p+4*seg["g1"].offset, ptr(L.save_mean[0,0]), ptr(L.save_invstd[0,0]), gp+4*seg["g1"].offset, gp+4*seg["be1"].offset, gp+4*seg["b1"].offset, gp+4*seg["W1"].offset, None, None, B, H, stream_ptr())
tests["B2 heads_bwd_bn_relu_bwd"] = lambda: lib.naf_heads_bwd_bn_relu_bwd(ptr(L.dH), NHP, t2p+4*seg["Wh"].offset, HP, ptr(L.G2[0]), H, t2p+4*seg["b2"].offset, ptr(L.A2[0]), HP, t2p+4*seg["g2"].offset, ptr(L.save_mean[1,0]), ptr(L.save_invstd[1,0]), ptr(L.dZ2), H, gp+4*seg["g2"].offset, gp+4*seg["be2"].offset, gp+4*seg["b2"].offset, None, B, H, stream_ptr())
tests["F3 heads_gemm_head"] = lambda: lib.naf_heads_gemm_head_fwd_bwd_mse(ptr(L.A2), B*HP, HP, HP, t2p+4*seg["Wh"].offset, P, HP, NHP, rows.data_ptr()+4*lay.off_u, 64, rows.data_ptr()+4*lay.off_r, 64, 0.99, None, ptr(L.q_out), ptr(L.dH), None, B, lay.A, 0, stream_ptr())
D = _lib.GemmDesc
bun = L._bundle
one = lambda i: (D * 1)(bun[i])
tests["GB bundle dWh+dW2+dA1"] = lambda: lib.naf_gemm_bundle(bun, 3, stream_ptr())
tests["GB dWh only (kmaj,kmaj)"] = lambda a=one(0): lib.naf_gemm_bundle(a, 1, stream_ptr())
tests["GB dW2 only (kmaj,kmaj)"] = lambda a=one(1): lib.naf_gemm_bundle(a, 1, stream_ptr())
tests["GB dA1 only (kcont,kmaj)"] = lambda a=one(2): lib.naf_gemm_bundle(a, 1, stream_ptr())
tests["torch mm dW2"] = lambda: torch.mm(L.dZ2.t(), L.A1[0], out=L.gW2)
tests["torch mm dA1"] = lambda: torch.mm(L.dZ2, L.W2_main, out=L.dA1)
tests["torch mm dWh"] = lambda: torch.mm(L.dH.t(), L.A2[0], out=L.gWh)
buf = ReplayBuffer(1_000_000, B, "cuda", 0, state_size=21, action_size=6)
buf.add_rows_device(torch.randn(1_000_000, 64, device="cuda"), 1_000_000)
idx = torch.randint(0, 1_000_000, (64, B), device="cuda", dtype=torch.int32); out = torch.empty(64*B, buf.batch_row_floats, device="cuda")
tests["sample 64xB"] = lambda: buf.sample_indices(idx, 64)
tests["gather 64xB rows"] = lambda: buf.gather_rows(idx, out, 64*B)
L.step_dev.fill_(1)
for k, f in tests.items():
    print(f"{k:28s} {timeit(f):8.2f} us")
names = {0: "32x32", 1: "32x16", 2: "32x8", 3: "16x16", 4: "16x32", 5: "64x8", 6: "64x16", 7: "16x64", 8: "8x32", 9: "8x64", 10: "8x128", 11: "4x64"}
for cfg, nm in names.items():
    lib.naf_debug_set(0, cfg)
    print(f"bn tile {nm:6s} fwd {timeit(tests['bn_relu_fwd_train x2nets']):6.2f} us   bwd {timeit(tests['bn_relu_bwd']):6.2f} us")
lib.naf_debug_set(0, -1)
