"""Generates robotic_manipulator_rloa_amd/tuning/tunableop_gfx950.csv: PyTorch TunableOp results (best rocBLAS /
hipBLASLt solution per GEMM shape) for the learn()/act() shapes of the BASELINE configs. Run on an MI355X:
    python benchmarks/tune_gemms.py gpurun_out/tunableop_gfx950.csv
then copy the file into robotic_manipulator_rloa_amd/tuning/."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NAF_BLAS_TUNING_FILE"] = "none"     # do not load existing results
import torch
out = os.path.abspath(sys.argv[1])
torch.cuda.tunable.enable(True)
torch.cuda.tunable.tuning_enable(True)
torch.cuda.tunable.set_max_tuning_duration(100)
torch.cuda.tunable.set_max_tuning_iterations(30)
torch.cuda.tunable.set_filename(out)
from robotic_manipulator_rloa_amd.learner import Learner, ActPath
configs = [(21, 6, 256), (21, 6, 64), (21, 6, 128), (21, 6, 1024), (23, 7, 2048), (23, 7, 64)]
if len(sys.argv) > 2:
    configs = configs[:int(sys.argv[2])]
for (S, A, B) in configs:
    L = Learner(S, A, 256, B, 1e-3, 1e-3, 0.99, torch.device("cuda"))
    L.theta2.normal_(0, 0.05)
    rows = torch.randn(B, L.lay.row_floats, device="cuda")
    # TunableOp sizes its scratch copies from (batch stride x batch): the learner's overlapping [state|next_state]
    # view (batch stride 28 floats) would make it read out of bounds WHILE TUNING. Same GEMM key (sizes + leading
    # dimensions), non-overlapping operands:
    safe = torch.randn(2, B, L.lay.row_floats, device="cuda")[:, :, :S]
    L._x2 = lambda r, _s=safe: _s
    for _ in range(2):
        L.learn_rows(rows)
    for E in (1, 64):
        ap = ActPath(L, E, 0)
        ap.heads()
    torch.cuda.synchronize()
    print("tuned", S, A, B, len(torch.cuda.tunable.get_results()), flush=True)
print("validators", torch.cuda.tunable.get_validators())
