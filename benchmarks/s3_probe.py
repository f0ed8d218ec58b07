"""Per-launch cost of the split-K heads pair (S3) against the three launches it replaces (graph replay, B=256)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NAF_FUSE"] = "l1,b2,gb,s3"
import torch
from robotic_manipulator_rloa_amd.learner import Learner
from robotic_manipulator_rloa_amd._lib import ptr, stream_ptr
from kernel_probe_util import timeit

B = 256
L = Learner(21, 6, 256, B, 1e-3, 1e-3, 0.99, torch.device("cuda"))
lay, lib, seg = L.lay, L.lib, L.lay.seg
H, HP, NHP, P = lay.H, lay.HP, lay.NHP, lay.P
L.theta2.normal_(0, 0.05); L.G2.normal_(); L.Gh.normal_()
rows = torch.randn(B, 64, device="cuda")
t2p, bnp = L.theta2.data_ptr(), L.bn_stats.data_ptr()
bn2 = lambda: lib.naf_bn_relu_fwd_train(ptr(L.G2), B*H, H, t2p+4*seg["b2"].offset, t2p+4*seg["g2"].offset, t2p+4*seg["be2"].offset, P, bnp+8*H, bnp+12*H, 4*H, ptr(L.A2), B*HP, HP, ptr(L.save_mean[1]), ptr(L.save_invstd[1]), B, H, 2, 0.1, 1e-5, stream_ptr())
s3 = lambda: lib.naf_bn_relu_fwd_heads_partial(ptr(L.G2), B*H, H, t2p+4*seg["b2"].offset, t2p+4*seg["g2"].offset, t2p+4*seg["be2"].offset, P, bnp+8*H, bnp+12*H, 4*H, ptr(L.A2), B*HP, HP, ptr(L.save_mean[1]), ptr(L.save_invstd[1]), t2p+4*seg["Wh"].offset, P, HP, NHP, lay.A+lay.T, ptr(L.heads_partial), L.slab_stride, ptr(L.vnext_partial), B, H, 0.1, 1e-5, stream_ptr())
bmm = lambda: torch.bmm(L.A2, L.WhT2, out=L.Gh)
head = lambda: lib.naf_head_fwd_bwd_mse(ptr(L.Gh[0]), NHP, rows.data_ptr()+4*lay.off_u, 64, rows.data_ptr()+4*lay.off_r, 64, L.Gh[1].data_ptr()+4*(lay.A+lay.T), NHP, 0.99, ptr(L.q_out), ptr(L.dH), None, B, lay.A, 0, stream_ptr())
heads = lambda: lib.naf_head_fwd_bwd_mse_splitk(ptr(L.heads_partial), L.slab_stride, ptr(L.vnext_partial), L.n_slabs, NHP, rows.data_ptr()+4*lay.off_u, 64, rows.data_ptr()+4*lay.off_r, 64, 0.99, ptr(L.q_out), ptr(L.dH), None, B, lay.A, 0, stream_ptr())
def chain_old(): bn2(); bmm(); head()
def chain_new(): s3(); heads()
for name, f in (("bn2 fwd", bn2), ("heads bmm", bmm), ("head", head), ("S3 bn2+partials", s3), ("head split-K", heads),
                ("chain bn2+bmm+head", chain_old), ("chain S3+head split-K", chain_new)):
    print(f"{name:26s} {timeit(f):7.2f} us")
