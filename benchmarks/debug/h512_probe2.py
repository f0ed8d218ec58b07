"""layer 1 and GEMM 2 of the row-split chain at layer_size 512 against torch, buffer by buffer (round 6 bring-up)"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
from oracle import naf_oracle as O
from synth_data import make_transitions
from test_learner_gpu import _random_init_sd, make_learner, rows_device
torch.backends.cuda.matmul.allow_tf32 = False
for (S, A, H, B) in [(21, 6, 512, 512), (21, 6, 512, 576), (21, 6, 512, 1024)]:
    st, ac, rw, ns, dn = make_transitions(B, S, A, seed=21, rare_events=False, structured_reward=True)
    sd = _random_init_sd(S, A, H)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        L = make_learner(S, A, B, sd, sd, H=H)
    rows = rows_device(L, st, ac, rw, ns, dn)
    L.forward_train(rows[:B])
    torch.cuda.synchronize()
    v = L.lay.param_views(L.theta2[0])
    x = torch.from_numpy(st).cuda()
    z1 = x @ v["input_layer.weight"].T + v["input_layer.bias"]
    a1 = torch.relu((z1 - z1.mean(0)) / torch.sqrt(z1.var(0, unbiased=False) + 1e-5) * v["bn1.weight"] + v["bn1.bias"])
    z2 = a1 @ v["hidden_layer.weight"].T + v["hidden_layer.bias"]
    A1 = L.A1[0, :B]; G2 = L.G2[0, :B]
    d1 = (A1 - a1).abs().max().item(); d2 = (G2 - z2).abs()
    bad_cols = (d2.max(0).values > 1e-3).nonzero().flatten().cpu().numpy()
    bad_rows = (d2.max(1).values > 1e-3).nonzero().flatten().cpu().numpy()
    st2 = L.bb_st2[0]                                  # [NB][H][2]
    mean_p = st2[:, :, 0].sum(0) / B
    dm = (mean_p - z2.mean(0)).abs().max().item()
    # the fused layer-2 launch alone: A2 against torch
    lp = torch.zeros(L.n_loss_wg, device="cuda")
    L.learn_rows(rows[:B], lp)
    torch.cuda.synchronize()
    a2 = torch.relu((z2 - z2.mean(0)) / torch.sqrt(z2.var(0, unbiased=False) + 1e-5) * v["bn2.weight"] + v["bn2.bias"])
    A2 = L.A2[0, :B, :H]
    d3 = (A2 - a2).abs()
    print("   stats mean diff", dm, "A2 max diff", d3.max().item(), "bad cols", (d3.max(0).values > 1e-3).sum().item(), "bad rows", (d3.max(1).values > 1e-3).sum().item(),
          "save_mean diff", (L.save_mean[1, 0] - z2.mean(0)).abs().max().item(), flush=True)
    print((S, A, H, B), "A1 max diff", d1, "Z2 max diff", d2.max().item(), "bad cols", len(bad_cols), bad_cols[:8], "bad rows", len(bad_rows), bad_rows[:8], flush=True)
