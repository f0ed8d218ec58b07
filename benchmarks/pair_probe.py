import sys, os
sys.path.insert(0, "/root/repo")
import torch
from robotic_manipulator_rloa_amd import _lib
from robotic_manipulator_rloa_amd.learner import Learner
from robotic_manipulator_rloa_amd._lib import ptr, stream_ptr
def timeit(fn, n=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n
B=256
L = Learner(21, 6, 256, B, 1e-3, 1e-3, 0.99, torch.device("cuda"))
lib=L.lib; P=L.lay.P
L.theta2.normal_(0,0.05); L.grad.normal_(); L.step_dev.fill_(1)
src = torch.randn(P, device="cuda")
adam = lambda: lib.naf_adam_polyak_fused(ptr(L.theta2[0]), ptr(L.grad), ptr(L.adam_m), ptr(L.adam_v), ptr(L.theta2[1]), ptr(L.partials), L.n_partials, 1.0, 1e-3, .9, .999, 1e-8, 1e-3, 1-1e-3, ptr(L.step_dev), 1.0, P, stream_ptr())
wr_grad = lambda: lib.naf_polyak_update(ptr(L.grad), ptr(src), 0.5, 0.5, P, stream_ptr())   # rewrites grad (fresh data for adam)
wr_other = lambda: lib.naf_polyak_update(ptr(L.G1.view(-1)[:P]), ptr(src), 0.5, 0.5, P, stream_ptr())  # same kernel, unrelated buffer
print("adam alone        %.2f" % timeit(adam))
print("writer alone      %.2f" % timeit(wr_grad))
print("writer(other)+adam %.2f" % timeit(lambda: (wr_other(), adam())))
print("writer(grad)+adam  %.2f" % timeit(lambda: (wr_grad(), adam())))
