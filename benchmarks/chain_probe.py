"""Where do the ~12 us go that one learn() update costs in situ beyond the sum of its kernels replayed alone?

The same 8 launches in the same order, captured N times into one graph, in two wirings:
  real    : every kernel reads what its predecessor just wrote (the true update)
  scratch : every kernel reads buffers nobody writes inside the loop (a frozen copy of a real update's state) and
            writes into a second set of buffers — same code, same order, same kernel switching, but no input that
            was produced a moment ago on other XCDs
If `scratch` is as slow as `real`, the in-situ penalty is the price of SWITCHING kernels (instruction/scalar caches,
dispatch); if it drops to the sum of the stand-alone times, it is the price of FRESH data (cross-XCD L2 misses).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from robotic_manipulator_rloa_amd import _lib
from robotic_manipulator_rloa_amd._lib import ptr, stream_ptr
from robotic_manipulator_rloa_amd.learner import Learner, BN_EPS, BN_MOMENTUM
from kernel_probe_util import timeit

B = 256
dev = torch.device("cuda")


def make():
    L = Learner(21, 6, 256, B, 1e-3, 1e-3, 0.99, dev)
    L.theta2.normal_(0, 0.05)
    return L


def chain(I, O, rows_i, bundle):
    """One update's launches: inputs from learner I, outputs into learner O."""
    lay, lib, seg = I.lay, I.lib, I.lay.seg
    H, HP, NHP, P = lay.H, lay.HP, lay.NHP, lay.P
    st = stream_ptr()
    ti, to = I.theta2.data_ptr(), O.theta2.data_ptr()
    go, bno = O.grad.data_ptr(), O.bn_stats.data_ptr()
    rp = rows_i.data_ptr()
    po = O.partials.data_ptr()
    lib.naf_linear_bn_relu_fwd_train(rp, lay.off_s2, 64, lay.S, ti + 4 * seg["W1"].offset, ti + 4 * seg["b1"].offset,
                                     ti + 4 * seg["g1"].offset, ti + 4 * seg["be1"].offset, P, bno, bno + 4 * H, 4 * H,
                                     ptr(O.A1), B * H, H, ptr(O.save_mean[0]), ptr(O.save_invstd[0]), B, H, 2, BN_MOMENTUM,
                                     BN_EPS, st)
    torch.bmm(I.A1, I.W2T2, out=O.G2)
    lib.naf_bn_relu_fwd_heads_partial(ptr(I.G2), B * H, H, ti + 4 * seg["b2"].offset, ti + 4 * seg["g2"].offset,
                                      ti + 4 * seg["be2"].offset, P, bno + 8 * H, bno + 12 * H, 4 * H, ptr(O.A2), B * HP, HP,
                                      ptr(O.save_mean[1]), ptr(O.save_invstd[1]), ti + 4 * seg["Wh"].offset, P, HP, NHP,
                                      lay.A + lay.T, ptr(O.heads_partial), O.slab_stride, ptr(O.vnext_partial), B, H,
                                      BN_MOMENTUM, BN_EPS, st)
    lib.naf_head_fwd_bwd_mse_splitk(ptr(I.heads_partial), I.slab_stride, ptr(I.vnext_partial), I.n_slabs, NHP,
                                    rp + 4 * lay.off_u, 64, rp + 4 * lay.off_r, 64, 0.99, ptr(O.q_out), ptr(O.dH), None, B,
                                    lay.A, 0, st)
    lib.naf_heads_bwd_bn_relu_bwd(ptr(I.dH), NHP, ti + 4 * seg["Wh"].offset, HP, ptr(I.G2[0]), H, ti + 4 * seg["b2"].offset,
                                  ptr(I.A2[0]), HP, ti + 4 * seg["g2"].offset, ptr(I.save_mean[1, 0]),
                                  ptr(I.save_invstd[1, 0]), ptr(O.dZ2), H, go + 4 * seg["g2"].offset,
                                  go + 4 * seg["be2"].offset, go + 4 * seg["b2"].offset,
                                  po + 4 * (O._gb_blocks + O._ft_blocks), B, H, st)
    lib.naf_gemm_bundle(bundle, 3, st)
    lib.naf_bn_relu_bwd_wgrad(ptr(I.dA1), H, rp, 64, lay.S, ti + 4 * seg["W1"].offset, ti + 4 * seg["b1"].offset,
                              ptr(I.A1[0]), H, ti + 4 * seg["g1"].offset, ptr(I.save_mean[0, 0]), ptr(I.save_invstd[0, 0]),
                              go + 4 * seg["g1"].offset, go + 4 * seg["be1"].offset, go + 4 * seg["b1"].offset,
                              go + 4 * seg["W1"].offset, po + 4 * O._gb_blocks, ptr(O.step_dev), B, H, st)
    lib.naf_adam_polyak_fused(ptr(O.theta2[0]), ptr(I.grad), ptr(O.adam_m), ptr(O.adam_v), ptr(O.theta2[1]),
                              ptr(I.partials), I.n_partials, 1.0, 1e-3, .9, .999, 1e-8, 1e-3, 1 - 1e-3, ptr(I.step_dev), 1.0, P,
                              st)


def bundle_desc(I, O):
    D = _lib.GemmDesc
    lay = I.lay
    H, HP, NHP = lay.H, lay.HP, lay.NHP
    pp = O.partials.data_ptr()
    return (D * 3)(
        D(ptr(I.dH), ptr(I.A2[0]), ptr(O.gWh), pp, NHP, HP, B, NHP, HP, HP, 1, 1),
        D(ptr(I.dZ2), ptr(I.A1[0]), ptr(O.gW2), pp + 4 * O._gb_wh_blocks, H, H, B, H, H, H, 1, 1),
        D(ptr(I.dZ2), ptr(I.W2_main), ptr(O.dA1), None, B, H, H, H, H, H, 0, 1))


A, S = make(), make()
rows = torch.randn(B, 64, device=dev)
rows[:, 21:27] = torch.trunc(rows[:, 21:27])
for _ in range(3):                       # a few real updates so every buffer of A holds realistic values
    A.learn_rows(rows)
torch.cuda.synchronize()
for name in ("theta2", "grad", "adam_m", "adam_v", "bn_stats", "partials", "A1", "G2", "A2", "heads_partial",
             "vnext_partial", "dH", "dZ2", "dA1", "save_mean", "save_invstd", "step_dev"):
    getattr(S, name).copy_(getattr(A, name))
real_b, scr_b = bundle_desc(A, A), bundle_desc(A, S)
t_real = timeit(lambda: chain(A, A, rows, real_b), n=100)
# restore A (the real chain advanced it) so the scratch wiring reads a sane frozen state
for name in ("theta2", "grad", "adam_m", "adam_v", "bn_stats", "partials", "step_dev"):
    getattr(A, name).copy_(getattr(S, name))
t_scr = timeit(lambda: chain(A, S, rows, scr_b), n=100)
print(f"chain, true dependencies (reads what was just written) {t_real:7.2f} us per update")
print(f"chain, frozen inputs / scratch outputs                 {t_scr:7.2f} us per update")
