"""GPU: the whole learn() update (torch-ROCm GEMMs + libnaf_hip.so kernels on flat buffers) against the
reference's own learn() (golden G3/G5) and against the numpy oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_group
from oracle import naf_oracle as O
from test_oracle_golden import assert_adam_stepped_close

pytestmark = pytest.mark.gpu


def make_learner(S, A, B, sd_main, sd_target, **kw):
    from robotic_manipulator_rloa_amd.learner import Learner
    L = Learner(S, A, 256, B, 1e-3, 1e-3, 0.99, torch.device("cuda"), **kw)
    L.load_params(0, sd_main)
    L.load_params(1, sd_target)
    return L


def rows_device(L, st, ac, rw, ns, dn, trunc=True):
    a = np.trunc(ac) if trunc else ac
    return torch.from_numpy(O.pack_rows(st, a, rw, ns, dn, L.lay.row_floats)).cuda()


def current_sd(L, net):
    sd = {k: v.detach().cpu().numpy().copy() for k, v in L.lay.param_views(L.theta2[net]).items()}
    sd.update({k: v.cpu().numpy().copy() for k, v in L.bn_views(net).items()})
    return sd


@pytest.mark.parametrize("tag", ["kuka", "panda"])
def test_learn_vs_reference_golden_g3(tag):
    from synth_data import make_transitions
    g = np.load(os.path.join(GOLDEN, "g3_learn.npz"))
    S, A, B = [int(x) for x in g[f"{tag}/dims"]]
    st, ac, rw, ns, dn = make_transitions(5 * B, S, A, seed=7)
    L = make_learner(S, A, B, load_group(g, f"{tag}/main0"), load_group(g, f"{tag}/target0"))
    rows = rows_device(L, st, ac, rw, ns, dn)
    lp = torch.zeros(5, L.n_loss_wg, device="cuda")
    for k in range(5):
        L.learn_rows(rows[k * B:(k + 1) * B], lp[k])
        if k == 0:
            torch.cuda.synchronize()
            np.testing.assert_allclose(L.q_out.cpu().numpy(), g[f"{tag}/q1"].ravel(), rtol=1e-3, atol=1e-3)
            gr = load_group(g, f"{tag}/grads1")
            gv = {k_: v.cpu().numpy() for k_, v in L.lay.param_views(L.grad).items()}
            norm = float(g[f"{tag}/grad_norm1"])
            np.testing.assert_allclose(np.sqrt(L.partials.sum().item()), norm, rtol=2e-4)
            for name in O.PARAM_ORDER:
                if name in ("input_layer.bias", "hidden_layer.bias"):
                    continue
                np.testing.assert_allclose(gv[name].reshape(gr[name].shape), gr[name], rtol=5e-3, atol=1e-5 * norm,
                                           err_msg=name)
            for grp, net in (("main1", 0), ("target1", 1)):
                ref = load_group(g, f"{tag}/{grp}")
                cur = current_sd(L, net)
                for name, val in ref.items():
                    if "num_batches" in name:
                        continue
                    if name in ("input_layer.bias", "hidden_layer.bias"):
                        np.testing.assert_allclose(cur[name], val, atol=1.01e-3)
                    elif "running" in name:
                        np.testing.assert_allclose(cur[name], val, rtol=1e-4, atol=5e-5, err_msg=f"{grp}/{name}")
                    else:
                        assert_adam_stepped_close(cur[name].reshape(val.shape), val, lr=1e-3, msg=f"{grp}/{name}")
    torch.cuda.synchronize()
    losses = lp.sum(1).cpu().numpy()
    np.testing.assert_allclose(losses, g[f"{tag}/losses5"], rtol=5e-3)
    assert int(L.step_dev.item()) == 5
    # pad regions of the flat buffers stay exactly zero (they are never given a gradient)
    mask = torch.ones(L.lay.P, dtype=torch.bool, device="cuda")
    for v in L.lay.param_views(torch.arange(L.lay.P, device="cuda", dtype=torch.float32)).values():
        mask[v.reshape(-1).long()] = False
    Whp = L.lay.view(torch.arange(L.lay.P, device="cuda", dtype=torch.float32), "Wh")
    assert (L.grad[mask] == 0).all() and (L.theta2[0][mask] == 0).all() and (L.adam_v[mask] == 0).all()


@pytest.mark.parametrize("p_mode", [0, 1])
@pytest.mark.parametrize("S,A,B", [(21, 6, 256), (23, 7, 2048), (19, 5, 64)])
def test_learn_vs_oracle_both_modes(p_mode, S, A, B):
    """20 updates against the f32 numpy oracle (Hadamard = reference semantics; matmul = textbook NAF)."""
    from synth_data import make_transitions
    g = np.load(os.path.join(GOLDEN, "g3_learn.npz"))
    n_upd = 20
    st, ac, rw, ns, dn = make_transitions(n_upd * B, S, A, seed=21, rare_events=False, structured_reward=True)
    torch.manual_seed(3)
    import torch.nn as nn
    # random but reproducible init in the reference's key layout
    T = A * (A + 1) // 2
    lin = {"input_layer": nn.Linear(S, 256), "hidden_layer": nn.Linear(256, 256), "action_values": nn.Linear(256, A),
           "value": nn.Linear(256, 1), "matrix_entries": nn.Linear(256, T)}
    sd = {}
    for k, l in lin.items():
        sd[f"{k}.weight"], sd[f"{k}.bias"] = l.weight.detach().numpy(), l.bias.detach().numpy()
    for b in ("bn1", "bn2"):
        sd[f"{b}.weight"], sd[f"{b}.bias"] = np.ones(256, np.float32), np.zeros(256, np.float32)
        sd[f"{b}.running_mean"], sd[f"{b}.running_var"] = np.zeros(256, np.float32), np.ones(256, np.float32)
        sd[f"{b}.num_batches_tracked"] = np.array(0)
    L = make_learner(S, A, B, sd, sd, p_mode=p_mode)
    Or = O.LearnerOracle(sd, p_mode=p_mode, dtype=np.float32)
    rows = rows_device(L, st, ac, rw, ns, dn)
    lp = torch.zeros(n_upd, L.n_loss_wg, device="cuda")
    ol = []
    for k in range(n_upd):
        sl = slice(k * B, (k + 1) * B)
        L.learn_rows(rows[sl], lp[k])
        ol.append(Or.learn(st[sl], ac[sl], rw[sl], ns[sl], dn[sl]))
    torch.cuda.synchronize()
    np.testing.assert_allclose(lp.sum(1).cpu().numpy(), ol, rtol=2e-2)    # f32 vs f32, 20 chaotic Adam steps
    np.testing.assert_allclose(lp.sum(1).cpu().numpy()[:3], ol[:3], rtol=2e-4)
    cur = current_sd(L, 0)
    for name in ("bn1.running_mean", "bn2.running_var"):
        np.testing.assert_allclose(cur[name], Or.main[name], rtol=2e-2, atol=2e-3)


def test_learn_bitwise_reproducible_run_to_run():
    from synth_data import make_transitions
    g = np.load(os.path.join(GOLDEN, "g3_learn.npz"))
    S, A, B = 21, 6, 256
    st, ac, rw, ns, dn = make_transitions(5 * B, S, A, seed=7)
    outs = []
    for rep in range(2):
        L = make_learner(S, A, B, load_group(g, "kuka/main0"), load_group(g, "kuka/target0"))
        rows = rows_device(L, st, ac, rw, ns, dn)
        for k in range(5):
            L.learn_rows(rows[k * B:(k + 1) * B])
        torch.cuda.synchronize()
        outs.append((L.theta2.clone(), L.bn_stats.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
