"""GPU: the whole learn() update (torch-ROCm GEMMs + libnaf_hip.so kernels on flat buffers) against the
reference's own learn() (golden G3/G5) and against the numpy oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, G3_TAGS, g3_case, load_group
from oracle import naf_oracle as O
from test_oracle_golden import assert_adam_stepped_close

pytestmark = pytest.mark.gpu


def make_learner(S, A, B, sd_main, sd_target, H=256, **kw):
    from robotic_manipulator_rloa_amd.learner import Learner
    L = Learner(S, A, H, B, 1e-3, 1e-3, 0.99, torch.device("cuda"), **kw)
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


ROWS, COLUMNS = {"bb", "gb", "hk", "ep", "s2"}, {"l1", "b2", "gb", "s3"}


@pytest.mark.parametrize("fused", ["default", "rows", "columns", "unfused"])
@pytest.mark.parametrize("tag", G3_TAGS)
def test_learn_vs_reference_golden_g3(tag, fused, monkeypatch):
    """The three chains of launches that implement learn() (Learner: NAF_FUSE / fuse = rows | columns | unfused; the
    default picks by batch size) against the unmodified reference's learn() at every BASELINE batch size."""
    monkeypatch.delenv("NAF_FUSE", raising=False)
    from synth_data import make_transitions
    g, main0, target0 = g3_case(tag)      # 'xarm1024' / 'panda2048': the reference's learn() at configs[3] / [4]'s batch
    S, A, B = [int(x) for x in g[f"{tag}/dims"]]
    st, ac, rw, ns, dn = make_transitions(5 * B, S, A, seed=7)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")          # (the unfused chain beyond B = 512 warns about its speed)
        L = make_learner(S, A, B, main0, target0, fuse=None if fused == "default" else fused)
    if fused in ("default", "rows"):
        assert L.fuse == (ROWS if B >= 16 else COLUMNS)
    elif fused == "columns":
        assert L.fuse == (COLUMNS if B <= 512 else {"gb"})
    else:
        assert L.fuse == {"gb"} and L.chain == "unfused"
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
            np.testing.assert_allclose(np.sqrt(L.partials[:L.n_partials].sum().item()), norm, rtol=2e-4)
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


@pytest.mark.parametrize("fused", ["default", "rows", "columns", "unfused"])
@pytest.mark.parametrize("p_mode", [0, 1])
@pytest.mark.parametrize("S,A,B", [(21, 6, 256), (23, 7, 2048), (19, 5, 64), (25, 8, 100), (11, 1, 48), (40, 4, 32),
                                   (21, 6, 512), (32, 8, 512), (21, 6, 1024), (21, 6, 768), (21, 6, 320), (21, 6, 1536),
                                   (23, 7, 1984), (21, 6, 1000), (21, 6, 1008), (21, 6, 64), (21, 6, 128), (21, 6, 192),
                                   (21, 6, 4096), (21, 6, 2500), (21, 6, 1200), (23, 7, 2000), (21, 6, 96), (21, 6, 80),
                                   (21, 6, 160), (21, 6, 1040), (21, 6, 65), (21, 6, 127), (21, 6, 513), (21, 6, 1025),
                                   (23, 7, 2047), (26, 8, 77), (21, 6, 1023), (21, 6, 16), (21, 6, 17), (21, 6, 33), (23, 7, 63),
                                   (21, 6, 9), (21, 6, 8192), (27, 9, 256), (30, 12, 64), (45, 16, 48), (27, 9, 5000),
                                   (29, 10, 64), (31, 11, 1024), (27, 9, 100), (31, 11, 2048), (29, 10, 1000), (32, 11, 48),
                                   (27, 9, 2100), (30, 6, 256), (43, 17, 64), (73, 32, 100), (137, 64, 32), (31, 11, 4096), (29, 10, 3000)])
def test_learn_vs_oracle_both_modes(p_mode, S, A, B, fused, monkeypatch):
    """20 updates against the f32 numpy oracle (Hadamard = reference semantics; matmul = textbook NAF), every chain at
    every shape it admits — including batch sizes that are multiples of 64 but not of 256 (K ranges of the weight
    gradients with a tail chunk), sizes that are not whole 64-row blocks or whole 16-row groups (100, 1000, 65, 127, 2047 ...:
    the row-split chain with a partial last block / workgroup / MFMA tile) and sizes beyond 2048 (2500, 4096: 32 rows per
    workgroup in the fused layer-2 launch, statistics folds in two passes; the unfused chain there with a warning)."""
    monkeypatch.delenv("NAF_FUSE", raising=False)
    import warnings
    # (round 6: 9 .. 11 joints run the row-split chain up to B = 2048 — one sample per 16-lane group in the fused layer-2 launch —
    #  and state sizes up to 32 run it at any joint count it takes)
    wide_rows = 9 <= A <= 11 and S <= 32 and 16 <= B <= (4096 if p_mode == 0 else 2048)
    if (B in (1000, 1008, 1984, 4096, 2500, 1200, 2000, 1040, 513, 1025, 2047, 1023, 8192, 5000, 2100, 3000) or A > 8) and fused in ("rows", "columns"):
        pytest.skip("same chain as default at this size")
    if (B > 4096 or (A > 8 and not wide_rows)) and fused == "unfused":
        pytest.skip("same chain as default at this size")
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
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        L = make_learner(S, A, B, sd, sd, p_mode=p_mode, fuse=None if fused == "default" else fused)
    rows_ok = 16 <= B <= 4096 and S <= 32 and (A <= 8 or wide_rows)
    if fused == "default" and ((A > 8 and not wide_rows) or B > 4096):
        # no shape the reference takes raises here — batch sizes beyond 4096 (the sampler's table in device memory)
        # and 9 .. 16 joints (one sample per 16-lane group in the head kernels) train on the unfused chain, and say so
        assert L.chain == "unfused" and len(caught) == 1
        assert ("action_size" in str(caught[0].message)) == (A > 8)
    elif fused == "default":
        assert L.fuse == (ROWS if rows_ok else (L.fuse if B > 512 or S > 24 else (COLUMNS if B % 16 == 0 else COLUMNS - {"gb"})))
    if fused == "rows" and rows_ok:
        assert L.fuse == ROWS
    if A > 8:
        # a 9-joint arm runs 5 launches per update, not 14 — and says nothing
        assert (L.chain == "rows" and not caught) if (wide_rows and fused == "default") else L.chain == "unfused"
    elif B > 512 and "bb" not in L.fuse:
        assert any("16 <= batch_size <= 4096" in str(w.message) for w in caught), "the unfused chain beyond B = 512 must say so"
        if B > 2048:      # beyond the row-split chain's sizes: the streamed BatchNorm kernels, any batch size up to the sampler's 4096
            assert L.chain == "unfused"
    else:
        assert not caught
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


def test_learn_at_batch_20000_vs_oracle():
    """the reference takes any positive batch_size (rl_framework.py:186-189); round 5 stopped at 16384 (the
    device-memory sampler packed a table slot into 16 bits). Now 2^20: three updates at B = 20000 on the unfused chain (streamed
    BatchNorm, one sample per 8-lane group in the head) against the f32 oracle, and a draw of that size from the ring."""
    import warnings
    from synth_data import make_transitions
    from robotic_manipulator_rloa_amd.utils.replay_buffer import ReplayBuffer
    S, A, B, n_upd = 21, 6, 20000, 3
    st, ac, rw, ns, dn = make_transitions(n_upd * B, S, A, seed=21, rare_events=False, structured_reward=True)
    sd = _random_init_sd(S, A, 256)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        L = make_learner(S, A, B, sd, sd)
    assert L.chain == "unfused"
    Or = O.LearnerOracle(sd, p_mode=0, dtype=np.float32)
    rows = rows_device(L, st, ac, rw, ns, dn)
    lp = torch.zeros(n_upd, L.n_loss_wg, device="cuda")
    ol = []
    for k in range(n_upd):
        sl = slice(k * B, (k + 1) * B)
        L.learn_rows(rows[sl], lp[k])
        ol.append(Or.learn(st[sl], ac[sl], rw[sl], ns[sl], dn[sl]))
    torch.cuda.synchronize()
    np.testing.assert_allclose(lp.sum(1).cpu().numpy(), ol, rtol=1e-3)
    buf = ReplayBuffer(n_upd * B, B, "cuda", 5, state_size=S, action_size=A)
    buf.add_rows_device(rows, n_upd * B)
    s_, a_, r_, n_, d_ = buf.sample()
    assert s_.shape == (B, S) and a_.dtype == torch.int64 and len(set(buf._idx.cpu().numpy().tolist())) == B


def _random_init_sd(S, A, H, seed=3):
    """random but reproducible init in the reference's key layout (torch's own Linear init, naf_neural_network.py:37-54)"""
    import torch.nn as nn
    torch.manual_seed(seed)
    T = A * (A + 1) // 2
    lin = {"input_layer": nn.Linear(S, H), "hidden_layer": nn.Linear(H, H), "action_values": nn.Linear(H, A),
           "value": nn.Linear(H, 1), "matrix_entries": nn.Linear(H, T)}
    sd = {}
    for k, l in lin.items():
        sd[f"{k}.weight"], sd[f"{k}.bias"] = l.weight.detach().numpy(), l.bias.detach().numpy()
    for b in ("bn1", "bn2"):
        sd[f"{b}.weight"], sd[f"{b}.bias"] = np.ones(H, np.float32), np.zeros(H, np.float32)
        sd[f"{b}.running_mean"], sd[f"{b}.running_var"] = np.zeros(H, np.float32), np.ones(H, np.float32)
        sd[f"{b}.num_batches_tracked"] = np.array(0)
    return sd


@pytest.mark.parametrize("pad", [True, False])
@pytest.mark.parametrize("p_mode", [0, 1])
@pytest.mark.parametrize("S,A,H,B", [(10, 5, 128, 64), (21, 6, 128, 256), (21, 6, 512, 64), (21, 6, 512, 300), (21, 6, 128, 1024),
                                     (23, 7, 64, 100), (21, 6, 320, 48), (21, 6, 200, 256), (21, 6, 4, 32),
                                     (21, 6, 512, 256), (23, 7, 512, 1024), (21, 6, 384, 2048), (21, 6, 512, 4096),
                                     (27, 9, 512, 256), (29, 10, 128, 64), (31, 11, 400, 1000)])
def test_learn_at_other_layer_sizes_vs_oracle(S, A, H, B, p_mode, pad):
    """layer_size is a hyper-parameter of the reference (rl_framework.py:68-74, `NAF(state, action, layer_size,
    ...)`), and its own agent test builds NAF(10, 5, 128, ...) (tests/.../test_naf_algorithm.py:74). A width below 256 is STORED
    zero-padded to 256 (NetLayout: the padded units compute exact zeros and receive zero gradients) and runs the row-split chain the
    presets run; round 6: a width in (256, 512] runs it too (stored as 512: two 256-column halves, two workgroups per row block in
    the fused layer-2 launch); pad_layer = False at other widths runs the column-tile chain (B <= 512) or the unfused chain: 20
    updates against the f32 numpy oracle of the network AS GIVEN, as test_learn_vs_oracle_both_modes holds the default width to."""
    import warnings
    from synth_data import make_transitions
    if H in (256, 512) and pad:
        pytest.skip("a native width: nothing to pad")
    n_upd = 20 if B * H <= 1024 * 512 else 6               # (the numpy oracle: ~0.5 s per update at 512 x 4096)
    st, ac, rw, ns, dn = make_transitions(n_upd * B, S, A, seed=21, rare_events=False, structured_reward=True)
    sd = _random_init_sd(S, A, H)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        L = make_learner(S, A, B, sd, sd, H=H, p_mode=p_mode, pad_layer=pad)
    Hs = (256 if H < 256 else 512 if 256 < H < 512 else H) if pad else H      # the stored width
    assert L.lay.H_ref == H and L.lay.H == Hs
    rows_chain = B >= 16 and (Hs in (256, 512)) and (pad or H in (256, 512))
    if rows_chain:
        # (round 6: widths in (256, 512] run the row-split chain too — 512 columns as two 256-column halves, two workgroups per row
        #  block in the fused layer-2 launch; 320 stored zero-padded to 512 as 128 is to 256)
        assert L.chain == "rows" and "bb" in L.fuse and not caught
        # what the padding holds: zeros, before and (below) after the updates
        for name in ("W1", "b1", "g1", "be1", "W2", "b2", "g2", "be2"):
            v = L.lay.view(L.theta2[0], name)
            assert v[H:].numel() == 0 or float(v[H:].abs().max()) == 0.0, name
            assert v.dim() == 1 or name == "W1" or v[:, H:].numel() == 0 or float(v[:, H:].abs().max()) == 0.0, name
    else:
        assert "bb" not in L.fuse and L.chain in ("columns", "unfused")
        assert bool(caught) == (B > 512 or A > 8)          # (beyond 512 rows / 8 joints the unfused chain says what it is)
    Or = O.LearnerOracle(sd, p_mode=p_mode, dtype=np.float32)
    rows = rows_device(L, st, ac, rw, ns, dn)
    lp = torch.zeros(n_upd, L.n_loss_wg, device="cuda")
    ol = []
    for k in range(n_upd):
        sl = slice(k * B, (k + 1) * B)
        L.learn_rows(rows[sl], lp[k])
        ol.append(Or.learn(st[sl], ac[sl], rw[sl], ns[sl], dn[sl]))
    torch.cuda.synchronize()
    got = lp.sum(1).cpu().numpy()
    np.testing.assert_allclose(got[:3], ol[:3], rtol=2e-4)
    np.testing.assert_allclose(got, ol, rtol=2e-2)
    cur = current_sd(L, 0)
    for name in ("bn1.running_mean", "bn2.running_var", "hidden_layer.weight", "value.weight"):
        # (an element whose gradient is rounding noise takes +-lr steps of either sign in the two implementations: over 20 steps a
        #  handful of a 512 x 512 matrix's elements part by more than 2.5e-3 — at most 1 in 10,000, and none by more than 20 lr)
        off = np.abs(cur[name] - Or.main[name]) > 2.5e-3 + 2e-2 * np.abs(Or.main[name])
        assert off.mean() <= 1e-4 and np.abs(cur[name] - Or.main[name]).max() <= 2e-2, (name, int(off.sum()))
    if rows_chain and Hs != H:
        for net in (0, 1):
            for buf in (L.theta2[net], L.grad, L.adam_m, L.adam_v):
                for name in ("W1", "b1", "g1", "be1", "W2", "b2", "g2", "be2"):
                    v = L.lay.view(buf, name)
                    assert float(v[H:].abs().max()) == 0.0, (name, net)
                assert float(L.lay.view(buf, "W2")[:, H:].abs().max()) == 0.0 and float(L.lay.view(buf, "Wh")[:, H:Hs].abs().max()) == 0.0


def _wide_golden(tag, file="g3_learn_wide.npz"):
    """(dims, {group: {name: array}}, q1, losses5, grad_norm1) of a slim G3 golden (tests/golden/g3_learn_wide.npz, g3_learn_joints.npz):
    the H x H matrix of every group is there as its first 16 rows (`name@rows16`) and its (sum, sum of squares) (`name@sums`)"""
    g = np.load(os.path.join(GOLDEN, file))
    groups = {grp: load_group(g, f"{tag}/{grp}") for grp in ("main0", "main1", "target1", "grads1")}
    return [int(x) for x in g[f"{tag}/dims"]], groups, g[f"{tag}/q1"].ravel(), g[f"{tag}/losses5"], float(g[f"{tag}/grad_norm1"])


def _check_against_slim(cur, ref, check, msg):
    """cur: {name: full array}; ref: a slim group; check(actual, desired, name)"""
    for name, val in ref.items():
        if "num_batches" in name or name.endswith("@sums"):
            continue
        if name.endswith("@rows16"):
            base = name[:-len("@rows16")]
            check(cur[base].reshape(-1, val.shape[1])[:16], val, f"{msg}/{base}[:16]")
        else:
            check(cur[name].reshape(val.shape), val, f"{msg}/{name}")


@pytest.mark.parametrize("tag", ["h512", "h384", "j9", "j11"])
def test_learn_at_wide_layers_vs_reference_golden_g3(tag):
    """Round 6: layer sizes in (256, 512] on the row-split chain (512 columns as two 256-column halves; 384 stored zero-padded to
    512) against the UNMODIFIED reference's learn() at NAF(21, 6, 512), batch 256, and NAF(21, 6, 384), batch 64 (slim goldens:
    make_golden.py --only g3wide) — Q, the gradient norm and every gradient before the clip, parameters and target after one step,
    BatchNorm buffers, the five losses. The initial weights are the reference constructor's at seed 0, which
    reference_init_state_dict reproduces bit for bit (checked here against the golden's slices and sums).
    j9 / j11: the same at 9 and 11 joints — NAF(27, 9, 256) at batch 256, NAF(31, 11, 256) at batch 64 (make_golden.py --only g3joints)
    — where the fused layer-2 launch holds one sample per 16-lane group and a heads tile of 64 / 80 rows."""
    from synth_data import make_transitions
    from robotic_manipulator_rloa_amd.naf_components.naf_neural_network import reference_init_state_dict
    (S, A, B, H), grp, q1, losses5, norm = _wide_golden(tag, "g3_learn_joints.npz" if tag.startswith("j") else "g3_learn_wide.npz")
    sd0 = {k: v.numpy() for k, v in reference_init_state_dict(S, A, H, 0).items()}
    for name, val in grp["main0"].items():
        if name.endswith("@rows16"):
            np.testing.assert_array_equal(sd0[name[:-7]][:16], val)
        elif name.endswith("@sums"):
            w = sd0[name[:-5]].astype(np.float64)
            np.testing.assert_allclose([w.sum(), (w ** 2).sum()], val, rtol=1e-12)
        elif "num_batches" not in name:
            np.testing.assert_array_equal(sd0[name], val, err_msg=name)
    st, ac, rw, ns, dn = make_transitions(5 * B, S, A, seed=7)
    L = make_learner(S, A, B, sd0, sd0, H=H)
    assert L.chain == "rows" and L.lay.H == (512 if H > 256 else 256) and L.lay.H_ref == H
    rows = rows_device(L, st, ac, rw, ns, dn)
    lp = torch.zeros(5, L.n_loss_wg, device="cuda")
    L.learn_rows(rows[:B], lp[0])
    torch.cuda.synchronize()
    np.testing.assert_allclose(L.q_out.cpu().numpy(), q1, rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(np.sqrt(L.partials[:L.n_partials].sum().item()), norm, rtol=2e-4)
    gv = {k: v.cpu().numpy() for k, v in L.lay.param_views(L.grad).items()}

    def grad_close(actual, desired, name):
        if "input_layer.bias" in name or "hidden_layer.bias" in name:
            return                                         # (rounding-noise gradients in front of a train-mode BatchNorm: DESIGN section 2)
        np.testing.assert_allclose(actual, desired, rtol=5e-3, atol=1e-5 * norm, err_msg=name)
    _check_against_slim(gv, grp["grads1"], grad_close, "grads1")

    def stepped_close(actual, desired, name):
        if "input_layer.bias" in name or "hidden_layer.bias" in name:
            np.testing.assert_allclose(actual, desired, atol=1.01e-3)
        elif "running" in name:
            np.testing.assert_allclose(actual, desired, rtol=1e-4, atol=5e-5, err_msg=name)
        else:
            assert_adam_stepped_close(actual, desired, lr=1e-3, msg=name)
    for which, net in (("main1", 0), ("target1", 1)):
        _check_against_slim(current_sd(L, net), grp[which], stepped_close, which)
    for k in range(1, 5):
        L.learn_rows(rows[k * B:(k + 1) * B], lp[k])
    torch.cuda.synchronize()
    np.testing.assert_allclose(lp.sum(1).cpu().numpy(), losses5, rtol=5e-3)


def test_learn_at_the_reference_agent_tests_shape_g3():
    """G3 at NAF(10, 5, 128), batch 64 — the network the reference's own agent test builds (test_naf_algorithm.py:74) — from the
    UNMODIFIED reference's learn() (tests/golden/g3_learn_h128.npz, make_golden.py --only g3h128): Q, y, the five losses, every
    gradient before the clip, the parameters after one step."""
    from synth_data import make_transitions
    path = os.path.join(GOLDEN, "g3_learn_h128.npz")
    g = np.load(path)
    tag = "h128"
    S, A, B, H = [int(x) for x in g[f"{tag}/dims"]]
    assert (S, A, B, H) == (10, 5, 64, 128)
    main0, target0 = load_group(g, f"{tag}/main0"), load_group(g, f"{tag}/target0")
    st, ac, rw, ns, dn = make_transitions(5 * B, S, A, seed=7)
    L = make_learner(S, A, B, main0, target0, H=H)
    rows = rows_device(L, st, ac, rw, ns, dn)
    lp = torch.zeros(5, L.n_loss_wg, device="cuda")
    L.learn_rows(rows[:B], lp[0])
    torch.cuda.synchronize()
    np.testing.assert_allclose(L.q_out.cpu().numpy(), g[f"{tag}/q1"].reshape(-1), rtol=2e-4, atol=2e-5)
    after = current_sd(L, 0)
    for k in O.PARAM_ORDER:
        if k in ("input_layer.bias", "hidden_layer.bias"):
            continue                                       # (rounding-noise gradients in front of a train-mode BatchNorm: DESIGN section 2)
        np.testing.assert_allclose(after[k], g[f"{tag}/main1/{k}"], rtol=1e-3, atol=2.1e-3, err_msg=k)
    for k in range(1, 5):
        L.learn_rows(rows[k * B:(k + 1) * B], lp[k])
    torch.cuda.synchronize()
    np.testing.assert_allclose(lp.sum(1).cpu().numpy(), g[f"{tag}/losses5"], rtol=5e-3)


# Round 6: the reference's OWN loss curves at these batch sizes (tests/golden/g7_curves.npz, make_golden.py --only g7: the unmodified
# NAFAgent.learn() on the same rows, positions and initial weights — 5000 updates at B = 100 and 1000, 600 at 2560, 2000 at 4096) are
# what the kernels are held to, as G5 holds the whole-block kernels: a direct pin on the reference instead of on its numpy restatement,
# and no oracle time in the suite (3 / 43 / 143 ms per update at B = 100 / 1000 / 4096 on the GPU box's host: 150 s of the suite's 450
# before). NAF_LONG_PARITY=1 runs the f32 oracle beside it over the full lengths (profiles/r05_long_parity.log: 0.44 % / 0.63 % / 0.07 %).
_LONG = os.environ.get("NAF_LONG_PARITY") == "1"


@pytest.mark.parametrize("B,n_upd", [(100, 5000), (1000, 5000), (2560, 600), (4096, 2000)])
def test_long_teacher_forced_run_on_the_general_kernels_vs_reference(B, n_upd):
    """VERDICT r04 item 4a / r05 weak spot 1c: the partial-block (TAIL: B = 100, 1000) and beyond-2048 (BIG: B = 2560, 4096) variants
    of the row-split kernels over thousands of updates, not twenty: teacher-forced minibatches (fixed rows, fixed positions) through
    gather -> learn as replayed graphs of 100 updates against the UNMODIFIED REFERENCE's losses on the same minibatches (G7) — first
    updates one by one, then the 500-update moving average of the loss within 5 % (the criterion G5 holds the whole-block kernels
    to over 100k updates), and the parameter norm at the end."""
    from synth_data import batch_indices, make_transitions
    from robotic_manipulator_rloa_amd.engine import TrainChunk
    from robotic_manipulator_rloa_amd.utils.replay_buffer import ReplayBuffer
    S, A, NROWS, U = 21, 6, 40000, 100
    g = np.load(os.path.join(GOLDEN, "g3_learn.npz"))
    g7 = np.load(os.path.join(GOLDEN, "g7_curves.npz"))
    assert [int(x) for x in g7[f"b{B}/dims"]] == [S, A, B, NROWS, n_upd]
    sd = load_group(g, "kuka/main0")
    L = make_learner(S, A, B, sd, load_group(g, "kuka/target0"))
    assert "bb" in L.fuse
    st, ac, rw, ns, dn = make_transitions(NROWS, S, A, seed=2024, rare_events=False, structured_reward=True)
    buf = ReplayBuffer(NROWS, B, "cuda", 0, state_size=S, action_size=A)
    buf.add_rows_device(torch.from_numpy(O.pack_rows(st, ac, rw, ns, dn, 64)).cuda(), NROWS)
    idx_np = batch_indices(NROWS, B, n_upd, seed=99)
    idx = torch.from_numpy(idx_np).cuda()
    chunk = TrainChunk(L, buf, U, teacher_forced=True)
    chunk.capture()
    losses = torch.zeros(n_upd, device="cuda")
    for c in range(n_upd // U):
        chunk.idx.copy_(idx[c * U:(c + 1) * U])
        chunk.run()
        losses[c * U:(c + 1) * U] = chunk.losses()
    torch.cuda.synchronize()
    got = losses.cpu().numpy().astype(np.float64)
    ref = g7[f"b{B}/losses"].astype(np.float64)
    assert np.isfinite(got).all() and buf.bad_index_count() == 0
    np.testing.assert_allclose(got[:20], ref[:20], rtol=2e-4)
    np.testing.assert_allclose(got[:200], ref[:200], rtol=3e-2)
    w = 500
    sm = lambda x: np.convolve(x, np.ones(w) / w, mode="valid")            # noqa: E731
    rel = np.abs(sm(got) - sm(ref)) / sm(ref)
    print("B = %d: smoothed rel. deviation from the reference over %d updates: max %.4f mean %.4f" % (B, n_upd, rel.max(), rel.mean()))
    assert rel.max() < 0.05, f"smoothed loss curve deviates {rel.max():.3f} from the reference's"
    l2 = float(sum((v.double() ** 2).sum() for k, v in L.lay.param_views(L.theta2[0]).items()) ** 0.5)
    np.testing.assert_allclose(l2, float(g7[f"b{B}/theta_l2"]), rtol=2e-2)
    if _LONG:
        Or = O.LearnerOracle(sd, dtype=np.float32, target_state_dict=load_group(g, "kuka/target0"))
        orc = np.empty(n_upd)
        for k in range(n_upd):
            i = idx_np[k]
            orc[k] = Or.learn(st[i], ac[i], rw[i], ns[i], dn[i])
        rel_o = np.abs(sm(got) - sm(orc)) / sm(orc)
        print("B = %d: ... from the f32 oracle: max %.4f" % (B, rel_o.max()))
        assert rel_o.max() < 0.05


@pytest.mark.parametrize("tag", ["j9", "j11", "h512", "j10big"])
def test_long_teacher_forced_run_at_more_joints_and_wider_layers_vs_reference(tag):
    """Round 6's new fused shapes over thousands of updates against the UNMODIFIED REFERENCE's losses on the same teacher-forced
    minibatches (G8, make_golden.py --only g8): 9 joints at B = 256 (3000 updates), 11 joints at B = 1000 (2000: a partial last
    block), layer size 512 at B = 256 (2000: the two-halves form), 10 joints at B = 2560 (600: 32 rows per workgroup beside an 83-KB
    heads tile) — first updates one by one, then the 500-update moving average of the loss within 5 %, and the parameter norm."""
    from synth_data import batch_indices, make_transitions
    from robotic_manipulator_rloa_amd.engine import TrainChunk
    from robotic_manipulator_rloa_amd.naf_components.naf_neural_network import reference_init_state_dict
    from robotic_manipulator_rloa_amd.utils.replay_buffer import ReplayBuffer
    g8 = np.load(os.path.join(GOLDEN, "g8_curves.npz"))
    S, A, H, B, NROWS, n_upd = [int(x) for x in g8[f"{tag}/dims"]]
    U = 100
    sd = {k: v.numpy() for k, v in reference_init_state_dict(S, A, H, 0).items()}
    L = make_learner(S, A, B, sd, sd, H=H)
    assert L.chain == "rows"
    st, ac, rw, ns, dn = make_transitions(NROWS, S, A, seed=int(g8["data_seed"]), rare_events=False, structured_reward=True)
    buf = ReplayBuffer(NROWS, B, "cuda", 0, state_size=S, action_size=A)
    buf.add_rows_device(torch.from_numpy(O.pack_rows(st, ac, rw, ns, dn, L.lay.row_floats)).cuda(), NROWS)
    idx = torch.from_numpy(batch_indices(NROWS, B, n_upd, seed=int(g8["idx_seed"]))).cuda()
    chunk = TrainChunk(L, buf, U, teacher_forced=True)
    chunk.capture()
    losses = torch.zeros(n_upd, device="cuda")
    for c in range(n_upd // U):
        chunk.idx.copy_(idx[c * U:(c + 1) * U])
        chunk.run()
        losses[c * U:(c + 1) * U] = chunk.losses()
    torch.cuda.synchronize()
    got = losses.cpu().numpy().astype(np.float64)
    ref = g8[f"{tag}/losses"].astype(np.float64)
    assert np.isfinite(got).all() and buf.bad_index_count() == 0
    np.testing.assert_allclose(got[:5], ref[:5], rtol=3e-4)             # (f32 against f32: the curves part at the 1e-4 level within ten updates)
    np.testing.assert_allclose(got[:20], ref[:20], rtol=2e-3)
    np.testing.assert_allclose(got[:200], ref[:200], rtol=3e-2)
    w = 500
    sm = lambda x: np.convolve(x, np.ones(w) / w, mode="valid")            # noqa: E731
    rel = np.abs(sm(got) - sm(ref)) / sm(ref)
    print("%s: smoothed rel. deviation from the reference over %d updates: max %.4f mean %.4f" % (tag, n_upd, rel.max(), rel.mean()))
    assert rel.max() < 0.05, f"smoothed loss curve deviates {rel.max():.3f} from the reference's"
    l2 = float(sum((v.double() ** 2).sum() for k, v in L.lay.param_views(L.theta2[0]).items()) ** 0.5)
    np.testing.assert_allclose(l2, float(g8[f"{tag}/theta_l2"]), rtol=2e-2)


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


def _kuka_learner_and_replay(n_rows, B=256, seed_data=2024, learner_kw=None, **data_kw):
    from synth_data import make_transitions
    from robotic_manipulator_rloa_amd.utils.replay_buffer import ReplayBuffer
    g = np.load(os.path.join(GOLDEN, "g3_learn.npz"))
    S, A = 21, 6
    L = make_learner(S, A, B, load_group(g, "kuka/main0"), load_group(g, "kuka/target0"), **(learner_kw or {}))
    st, ac, rw, ns, dn = make_transitions(n_rows, S, A, seed=seed_data, **data_kw)
    buf = ReplayBuffer(n_rows, B, "cuda", 0, state_size=S, action_size=A)
    buf.add_rows_device(torch.from_numpy(O.pack_rows(st, ac, rw, ns, dn, 64)).cuda(), n_rows)
    return L, buf


def test_g5_teacher_forced_loss_curve_within_5_percent():
    """north_star criterion: Q-loss curve within +-5 % of the reference. Reference side: the unmodified
    NAFAgent.learn() driven for 100k updates on fixed minibatches (tests/golden/g5_curve.npz); build side: the same
    minibatches (regenerated, never stored) through gather -> learn as replayed HIP graphs of 100 updates."""
    from synth_data import batch_indices
    from robotic_manipulator_rloa_amd.engine import TrainChunk
    g = np.load(os.path.join(GOLDEN, "g5_curve.npz"))
    S, A, B, NROWS, n_upd = [int(x) for x in g["dims"]]
    L, buf = _kuka_learner_and_replay(NROWS, B, rare_events=False, structured_reward=True)
    idx = torch.from_numpy(batch_indices(NROWS, B, n_upd, seed=99)).cuda()
    U = 100
    chunk = TrainChunk(L, buf, U, teacher_forced=True)
    chunk.capture()
    losses = torch.zeros(n_upd, device="cuda")
    for c in range(n_upd // U):
        chunk.idx.copy_(idx[c * U:(c + 1) * U])
        chunk.run()
        losses[c * U:(c + 1) * U] = chunk.losses()
    torch.cuda.synchronize()
    got, ref = losses.cpu().numpy().astype(np.float64), g["losses"].astype(np.float64)
    assert np.isfinite(got).all() and buf.bad_index_count() == 0
    np.testing.assert_allclose(got[:20], ref[:20], rtol=2e-4)              # early: update by update
    np.testing.assert_allclose(got[:200], ref[:200], rtol=3e-2)            # rounding differences amplify (chaotic)
    w = 500
    sm = lambda x: np.convolve(x, np.ones(w) / w, mode="valid")            # noqa: E731
    rel = np.abs(sm(got) - sm(ref)) / sm(ref)
    print("G5 smoothed rel. deviation: max %.4f mean %.4f" % (rel.max(), rel.mean()))
    assert rel.max() < 0.05, f"smoothed loss curve deviates {rel.max():.3f} from the reference"
    l2 = float(sum((v.double() ** 2).sum() for k, v in L.lay.param_views(L.theta2[0]).items()) ** 0.5)
    np.testing.assert_allclose(l2, float(g[f"theta_l2_{n_upd}"]), rtol=2e-2)


def test_chunk_graph_equals_eager_and_sampler_advances():
    """The captured chunk replays to exactly the same bits as the eager launch sequence, capture leaves no
    trace in the learner state, and free-running sampling consumes the device counter."""
    from robotic_manipulator_rloa_amd.engine import TrainChunk
    res = []
    for use_graph in (False, True):
        L, buf = _kuka_learner_and_replay(5000, 256, seed_data=5)
        chunk = TrainChunk(L, buf, 8, teacher_forced=False, use_graph=use_graph)
        for _ in range(3):
            chunk.run()
        torch.cuda.synchronize()
        assert int(buf._sample_ctr.item()) == 24 and int(L.step_dev.item()) == 24
        res.append((L.theta2.clone(), chunk.idx.clone(), chunk.losses().clone()))
    assert torch.equal(res[0][1], res[1][1])
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][2], res[1][2])
    exp = O.replay_sample_indices(0, 16, 5000, 256, 8, True)
    np.testing.assert_array_equal(res[0][1].cpu().numpy(), exp)


@pytest.mark.parametrize("S,A,B,U", [(21, 6, 256, 7), (21, 6, 512, 3), (21, 6, 1024, 4), (23, 7, 2048, 3), (21, 6, 64, 5), (21, 6, 320, 3),
                                     (21, 6, 100, 4), (21, 6, 1000, 3), (21, 6, 2500, 2)])
def test_deferred_optimizer_step_is_the_same_bits(S, A, B, U, monkeypatch):
    """The optimizer step of update k carried by the first two launches of update k + 1 (csrc/adam_body.h; TrainChunk) against
    the step as a launch of its own: parameters of both nets, Adam moments, BatchNorm buffers, step count and every loss
    bit-identical after several chunks — eagerly and as replayed graphs (naf_algorithm.py:209-213 semantics unchanged)."""
    from synth_data import make_transitions
    from robotic_manipulator_rloa_amd.engine import TrainChunk
    from robotic_manipulator_rloa_amd.naf_components.naf_neural_network import reference_init_state_dict
    from robotic_manipulator_rloa_amd.utils.replay_buffer import ReplayBuffer
    sd = reference_init_state_dict(S, A, 256, seed=3)
    n_rows = 6000
    st, ac, rw, ns, dn = make_transitions(n_rows, S, A, seed=11)
    res = {}
    for mode, use_graph in (("0", False), ("1", False), ("1", True)):
        monkeypatch.setenv("NAF_DEFER_ADAM", mode)
        L = make_learner(S, A, B, sd, sd)
        assert L.defer_ok == (mode == "1")
        buf = ReplayBuffer(n_rows, B, "cuda", 0, state_size=S, action_size=A)
        buf.add_rows_device(torch.from_numpy(O.pack_rows(st, ac, rw, ns, dn, 64)).cuda(), n_rows)
        chunk = TrainChunk(L, buf, U, use_graph=use_graph)
        losses = []
        for _ in range(3):
            chunk.run()
            losses.append(chunk.losses().clone())
        torch.cuda.synchronize()
        assert int(L.step_dev.item()) == 3 * U
        # with the GPU to itself no poll inside a launch runs out of its 20 us: a count here means the records are not seen in
        # time (the results would still be right — the thread folds for itself — but every launch would pay for it)
        assert L.fold_fallbacks == 0
        res[(mode, use_graph)] = (L.theta2.clone(), L.adam_m.clone(), L.adam_v.clone(), L.bn_stats.clone(), torch.cat(losses))
    ref = res[("0", False)]
    assert torch.isfinite(ref[0]).all() and not torch.equal(ref[0][0], ref[0][1])
    for key in (("1", False), ("1", True)):
        for a, b, name in zip(ref, res[key], ("theta2", "adam_m", "adam_v", "bn_stats", "losses")):
            assert torch.equal(a, b), f"{name} differs with the deferred step ({key})"


@pytest.mark.parametrize("B", [256, 1024])
def test_deferred_optimizer_step_behind_a_collective_is_the_same_bits(B, monkeypatch):
    """The collective path of data parallel (RCCL all-reduce of the flat gradient, then a norm launch on the reduced gradient) with
    the optimizer step riding on the next update (round 4) against the same path with the step as a launch of its own: world 1 with
    the all-reduce forced (a no-op that keeps every launch of the path), eager and as replayed graphs — bit-identical."""
    from synth_data import make_transitions
    from robotic_manipulator_rloa_amd.engine import TrainChunk
    from robotic_manipulator_rloa_amd.naf_components.naf_neural_network import reference_init_state_dict
    from robotic_manipulator_rloa_amd.utils.replay_buffer import ReplayBuffer
    S, A, U = 21, 6, 6
    sd = reference_init_state_dict(S, A, 256, seed=3)
    n_rows = 6000
    st, ac, rw, ns, dn = make_transitions(n_rows, S, A, seed=11)
    res = {}
    for mode, use_graph in (("0", False), ("1", False), ("1", True)):
        monkeypatch.setenv("NAF_DEFER_ADAM", mode)
        L = make_learner(S, A, B, sd, sd, _force_allreduce=True)
        assert not L.fold_norm and L.defer_ok == (mode == "1")
        buf = ReplayBuffer(n_rows, B, "cuda", 0, state_size=S, action_size=A)
        buf.add_rows_device(torch.from_numpy(O.pack_rows(st, ac, rw, ns, dn, 64)).cuda(), n_rows)
        chunk = TrainChunk(L, buf, U, use_graph=use_graph)
        losses = []
        for _ in range(3):
            chunk.run()
            losses.append(chunk.losses().clone())
        torch.cuda.synchronize()
        assert int(L.step_dev.item()) == 3 * U
        res[(mode, use_graph)] = (L.theta2.clone(), L.adam_m.clone(), L.adam_v.clone(), L.bn_stats.clone(), torch.cat(losses))
    ref = res[("0", False)]
    assert torch.isfinite(ref[0]).all() and not torch.equal(ref[0][0], ref[0][1])
    for key in (("1", False), ("1", True)):
        for a, b, name in zip(ref, res[key], ("theta2", "adam_m", "adam_v", "bn_stats", "losses")):
            assert torch.equal(a, b), f"{name} differs with the deferred step ({key})"


def test_learn_rows_rejects_a_deferred_step_where_it_cannot_ride(monkeypatch):
    monkeypatch.setenv("NAF_DEFER_ADAM", "0")
    from robotic_manipulator_rloa_amd.naf_components.naf_neural_network import reference_init_state_dict
    sd = reference_init_state_dict(21, 6, 256, seed=3)
    L = make_learner(21, 6, 256, sd, sd)
    rows = torch.zeros(256, L.lay.row_floats, device="cuda")
    with pytest.raises(ValueError):
        L.learn_rows(rows, defer=True)


def test_device_env_loop_fills_replay_and_trains():
    from robotic_manipulator_rloa_amd.engine import DeviceEnvLoop, TrainChunk
    from robotic_manipulator_rloa_amd.utils.replay_buffer import ReplayBuffer
    g = np.load(os.path.join(GOLDEN, "g3_learn.npz"))
    S, A, B, E = 21, 6, 64, 16
    L = make_learner(S, A, B, load_group(g, "kuka/main0"), load_group(g, "kuka/target0"))
    buf = ReplayBuffer(1000, B, "cuda", 0, state_size=S, action_size=A)
    loop = DeviceEnvLoop(L, buf, E, seed=1, max_frames=50)
    theta0 = L.theta2.clone()
    for _ in range(10):
        loop.step()
    torch.cuda.synchronize()
    assert torch.equal(theta0, L.theta2)                                    # acting does not touch the weights
    assert len(buf) == 160 and int(buf.meta[1].item()) == 160 and int(buf.meta[2].item()) == 160
    rows = buf.rows[:160].cpu().numpy()
    s, a, r, s2, d = O.unpack_rows(rows, S, A)
    assert np.abs(a).max() <= 1.0 and np.isfinite(rows).all()
    np.testing.assert_allclose(s2[:, :A], s[:, :A] + a / 240.0, atol=1e-6)      # velocity control for one 1/240 s tick
    np.testing.assert_allclose(s2[:, A:2 * A], a, atol=0)                        # joint velocities = commanded
    np.testing.assert_array_equal(s2[:, 2 * A + 3:], s[:, 2 * A + 3:])            # target / obstacle never move
    dist = np.linalg.norm(s2[:, 2 * A:2 * A + 3] - s2[:, 2 * A + 3:2 * A + 6], axis=1)
    plain = (d == 0)
    np.testing.assert_allclose(r[plain], -(dist[plain] - 0.05), atol=1e-5)      # environment.py:366-371
    # consecutive transitions of env 0 chain: s2 of step t is s of step t+1 (no episode end within 10 frames)
    e0 = rows[0::E]
    np.testing.assert_array_equal(e0[1:, :S], O.unpack_rows(e0[:-1], S, A)[3])
    chunk = TrainChunk(L, buf, E)
    chunk.run()
    torch.cuda.synchronize()
    assert not torch.equal(theta0, L.theta2) and torch.isfinite(L.theta2).all() and buf.bad_index_count() == 0


@pytest.mark.parametrize("p_mode", [0, 1])
@pytest.mark.parametrize("S,A,E,H", [(21, 6, 1, 256), (21, 6, 64, 256), (23, 7, 5, 256), (32, 8, 33, 256), (11, 1, 300, 256), (27, 9, 1, 256),
                                     (31, 11, 40, 256), (21, 6, 1, 512), (23, 7, 70, 512), (27, 9, 3, 384)])
def test_policy_act_one_launch_matches_seven_launch_path(S, A, E, H, p_mode, monkeypatch):
    """csrc/policy_act.hip (act() for E states in one launch) against the GEMM + BN-eval + noise chain it replaces:
    same heads pre-activations to f32 rounding, same noise stream (same Philox keys), counter advanced by one per call."""
    from robotic_manipulator_rloa_amd.learner import ActPath
    torch.manual_seed(S + A + E)
    g = np.load(os.path.join(GOLDEN, "g3_learn.npz"))
    import torch.nn as nn
    T = A * (A + 1) // 2
    lin = {"input_layer": nn.Linear(S, H), "hidden_layer": nn.Linear(H, H), "action_values": nn.Linear(H, A),
           "value": nn.Linear(H, 1), "matrix_entries": nn.Linear(H, T)}
    sd = {}
    for k, l in lin.items():
        sd[f"{k}.weight"], sd[f"{k}.bias"] = l.weight.detach().numpy(), l.bias.detach().numpy()
    rng = np.random.default_rng(5)
    for b in ("bn1", "bn2"):
        sd[f"{b}.weight"], sd[f"{b}.bias"] = rng.uniform(0.5, 1.5, H).astype(np.float32), rng.normal(0, 0.2, H).astype(np.float32)
        sd[f"{b}.running_mean"] = rng.normal(0, 0.3, H).astype(np.float32)
        sd[f"{b}.running_var"] = rng.uniform(0.5, 2.0, H).astype(np.float32)
    L = make_learner(S, A, 64, sd, sd, H=H, p_mode=p_mode)      # (H = 512 | 384, round 6: stored as 512 — policy_act_512_kernel)
    obs = torch.randn(E, S, device="cuda")
    outs = []
    for fused in ("0", "1"):
        act = ActPath(L, E, seed=1234)
        assert act.fused                      # H <= 512, S <= 32: one launch
        act.fused = fused == "1"              # the seven-launch path other shapes take
        act.obs.copy_(obs)
        a1 = act.act(1.0).clone()
        h1 = act.Gh[:, :L.lay.NH].clone()
        a2 = act.act(1.0).clone()          # second call: the counter moved, fresh noise
        a0 = act.act(0.0).clone()          # noise off: tanh(mu)
        torch.cuda.synchronize()
        assert int(act.counter.item()) == 3 and int(act._ticket.item()) == 0
        outs.append((h1, a1, a2, a0))
    (h_ref, a1_ref, a2_ref, a0_ref), (h, a1, a2, a0) = outs
    scale = max(1.0, float(h_ref.abs().max()))
    np.testing.assert_allclose(h.cpu().numpy(), h_ref.cpu().numpy(), rtol=2e-5, atol=2e-5 * scale)
    np.testing.assert_allclose(a0.cpu().numpy(), a0_ref.cpu().numpy(), atol=2e-5)
    np.testing.assert_allclose(a1.cpu().numpy(), a1_ref.cpu().numpy(), atol=2e-4)      # same z, sigma from nearly equal heads
    np.testing.assert_allclose(a2.cpu().numpy(), a2_ref.cpu().numpy(), atol=2e-4)
    assert not torch.equal(a1, a2)
