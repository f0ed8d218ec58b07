"""which rows of Q / which loss parts are wrong at layer_size 512 beyond B = 512 (round 6 bring-up)"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
from oracle import naf_oracle as O
from synth_data import make_transitions
from test_learner_gpu import _random_init_sd, make_learner, rows_device
for (S, A, H, B) in [(21, 6, 512, 512), (21, 6, 512, 576)]:
    st, ac, rw, ns, dn = make_transitions(B, S, A, seed=21, rare_events=False, structured_reward=True)
    sd = _random_init_sd(S, A, H)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        L = make_learner(S, A, B, sd, sd, H=H)
        L256 = make_learner(S, A, B, _random_init_sd(S, A, 256), _random_init_sd(S, A, 256), H=256)
    rows = rows_device(L, st, ac, rw, ns, dn)
    lp = torch.zeros(L.n_loss_wg, device="cuda")
    L.learn_rows(rows[:B], lp)
    torch.cuda.synchronize()
    Or = O.LearnerOracle(sd, p_mode=0, dtype=np.float32)
    loss = Or.learn(st, ac, rw, ns, dn)
    q = L.q_out.cpu().numpy()
    lpn = lp.cpu().numpy()
    print((S, A, H, B), "loss", lpn.sum(), loss, "zero loss parts", int((lpn == 0).sum()), "of", len(lpn), "first parts", lpn[:6], "q[:4]", q[:4],
          "q nonfinite", int((~np.isfinite(q)).sum()), "fold fallbacks", L.fold_fallbacks, "err", [int(e) for e in L.err_host[:4]], flush=True)
    nz = np.nonzero(lpn == 0)[0]
    print("   zero parts at", nz[:20], "..." if len(nz) > 20 else "")
