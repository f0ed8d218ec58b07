"""Where the learner's work buffers land (device addresses modulo a few powers of two): run from a checkout's root on the GPU box."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from robotic_manipulator_rloa_amd.learner import Learner

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = Learner(21, 6, 256, B, 1e-3, 1e-3, 0.99, torch.device("cuda"))
names = [n for n in dir(L) if isinstance(getattr(L, n, None), torch.Tensor) and getattr(L, n).is_cuda]
rows = sorted((getattr(L, n).data_ptr(), n, getattr(L, n).numel() * getattr(L, n).element_size()) for n in names)
base = rows[0][0]
for p, n, sz in rows:
    print(f"{n:16s} +{p - base:10d}  size {sz:9d}  mod 4K {p % 4096:5d}  mod 64K {p % 65536:6d}  mod 2M {p % (1 << 21):8d}")
