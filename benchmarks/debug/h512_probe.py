"""first-update loss of the row-split chain at layer_size 512 against the f32 oracle, shape by shape (round 6 bring-up)"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
from oracle import naf_oracle as O
from synth_data import make_transitions
from test_learner_gpu import _random_init_sd, make_learner, rows_device
for (S, A, H, B) in [(21, 6, 512, 256), (21, 6, 512, 512), (21, 6, 512, 576), (21, 6, 512, 1024), (23, 7, 512, 256), (23, 7, 512, 1024),
                     (21, 6, 512, 2048), (21, 6, 512, 4096), (21, 6, 512, 1000)]:
    st, ac, rw, ns, dn = make_transitions(2 * B, S, A, seed=21, rare_events=False, structured_reward=True)
    sd = _random_init_sd(S, A, H)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        L = make_learner(S, A, B, sd, sd, H=H)
    Or = O.LearnerOracle(sd, p_mode=0, dtype=np.float32)
    rows = rows_device(L, st, ac, rw, ns, dn)
    lp = torch.zeros(2, L.n_loss_wg, device="cuda")
    out = []
    for k in range(2):
        L.learn_rows(rows[k * B:(k + 1) * B], lp[k])
        out.append(Or.learn(st[k * B:(k + 1) * B], ac[k * B:(k + 1) * B], rw[k * B:(k + 1) * B], ns[k * B:(k + 1) * B], dn[k * B:(k + 1) * B]))
    torch.cuda.synchronize()
    got = lp.sum(1).cpu().numpy()
    print((S, A, H, B), L.chain, got, np.array(out), "OK" if np.allclose(got, out, rtol=2e-4) else "MISMATCH", "fallbacks", L.fold_fallbacks, flush=True)
