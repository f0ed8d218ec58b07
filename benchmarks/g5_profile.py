import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tests/golden")
import numpy as np, torch
from test_learner_gpu import _kuka_learner_and_replay
from synth_data import batch_indices
from robotic_manipulator_rloa_amd.engine import TrainChunk
g = np.load("/root/repo/tests/golden/g5_curve.npz")
S, A, B, NROWS, n_upd = [int(x) for x in g["dims"]]
L, buf = _kuka_learner_and_replay(NROWS, B, rare_events=False, structured_reward=True)
idx = torch.from_numpy(batch_indices(NROWS, B, n_upd, seed=99)).cuda()
U = 100
chunk = TrainChunk(L, buf, U, teacher_forced=True); chunk.capture()
losses = torch.zeros(n_upd, device="cuda")
for c in range(n_upd // U):
    chunk.idx.copy_(idx[c * U:(c + 1) * U]); chunk.run(); losses[c * U:(c + 1) * U] = chunk.losses()
torch.cuda.synchronize()
got, ref = losses.cpu().numpy().astype(np.float64), g["losses"].astype(np.float64)
for w in (500, 2000):
    sm = lambda x: np.convolve(x, np.ones(w) / w, mode="valid")
    rel = np.abs(sm(got) - sm(ref)) / sm(ref)
    seg = [rel[i:i + 10000].max() for i in range(0, len(rel), 10000)]
    print(os.environ.get("TAG", ""), "w=%d max %.4f mean %.4f per-10k max:" % (w, rel.max(), rel.mean()), " ".join("%.3f" % s for s in seg))
