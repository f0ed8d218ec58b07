"""The per-timestep path (csrc/step_path.hip): NAFAgent.step -> act of the reference's own loop (one env, one transition, one
minibatch, one update per timestep; naf_algorithm.py:129-178, :249-261) in seven launches instead of twelve. Everything here is
BIT-EXACT: the two fused launches against the launches they replace (same draw, same rows, same moments record; same parameters,
Adam state, target and action), and the whole path against the chunked path on the same minibatches."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import naf_oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.fixture(scope="module")
def lib():
    from robotic_manipulator_rloa_amd import _lib
    _lib.require_gpu()
    return _lib.load(allow_build=False)


@pytest.fixture()
def scratch_cwd(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    return tmp_path


def st():
    return torch.cuda.current_stream().cuda_stream


def _filled_buffer(cap, B, S, A, n_rows, seed):
    from robotic_manipulator_rloa_amd.utils.replay_buffer import ReplayBuffer
    from synth_data import make_transitions
    buf = ReplayBuffer(cap, B, "cuda", seed, state_size=S, action_size=A)
    if n_rows:
        s_, ac, rw, ns, dn = make_transitions(n_rows, S, A, seed=seed + 1)
        rows = torch.from_numpy(O.pack_rows(s_, ac, rw, ns, dn, buf.row_floats)).cuda()
        for lo in range(0, n_rows, cap):
            buf.add_rows_device(rows[lo:lo + cap], min(cap, n_rows - lo))
    torch.cuda.synchronize()
    return buf


# (S, A, B, capacity, rows in the ring before the launch, append a row?)
PREP_CASES = [
    (21, 6, 64, 1000, 500, 1),          # configs[0]'s shape
    (21, 6, 256, 5000, 3000, 1),        # configs[1]'s batch
    (21, 6, 256, 5000, 3000, 0),        # an idle tick: the count word says 0
    (21, 6, 64, 1000, 100, 1),          # dense regime: B <= population < 4 B (partial Fisher-Yates)
    (21, 6, 64, 1000, 0, 1),            # the very first row: population 1 < B (with replacement, as the chunked sampler)
    (21, 6, 64, 300, 899, 1),           # full ring, head about to wrap (899 = 3 * 300 - 1 rows went in)
    (21, 6, 100, 1000, 700, 1),         # a batch that is not whole 64-row blocks
    (23, 7, 2048, 20000, 15000, 1),     # configs[4]'s batch: 8 chunks of the moments passes, 7 x 7 tiles' rows
    (26, 6, 128, 1000, 600, 1),         # 32-column moments records (K4 = 7 -> 8)
    (21, 6, 4096, 40000, 30000, 1),     # the sampler's largest batch: an 80-KB hash table under the moments' staging area
    (21, 6, 256, 5000, 3000, None),     # no append node at all (schedules other than update_freq = 1)
    (27, 9, 64, 1000, 500, 1),          # round 6: 9 joints — ring rows of 128 floats, 72 of them gathered, the appended row in 32 float4
    (31, 11, 256, 5000, 3000, 1),       # 11 joints at B = 256: 80 floats per gathered row, more than the 64 KB the launch keeps in LDS
    (32, 8, 100, 1000, 700, 1),         # the widest state the layer-1 kernels take, a batch that is not whole blocks
    (29, 10, 64, 300, 899, 1),          # 10 joints, full ring, head about to wrap
]


@pytest.mark.parametrize("S,A,B,cap,n0,append", PREP_CASES)
def test_step_prep_equals_the_launches_it_replaces(lib, S, A, B, cap, n0, append):
    """naf_step_prep == naf_replay_add_counted + naf_replay_sample_indices + naf_counter_add + naf_replay_gather_rows +
    naf_bb_moments, bit for bit: ring, {head, size, total}, sampler counter, indices, minibatch rows, both moments records."""
    from robotic_manipulator_rloa_amd import _lib
    from synth_data import make_transitions
    bufs = [_filled_buffer(cap, B, S, A, n0, seed=5) for _ in range(2)]
    brf = bufs[0].batch_row_floats
    mf = lib.naf_bb_moments_floats(S)
    off_s2 = bufs[0].off_s2
    s_, ac, rw, ns, dn = make_transitions(3, S, A, seed=77)
    new_rows = O.pack_rows(s_, ac, rw, ns, dn, bufs[0].row_floats)
    outs = []
    for which, buf in enumerate(bufs):
        row_pin = torch.zeros(1, buf.row_floats).pin_memory()
        cnt = torch.zeros(1, dtype=torch.int32).pin_memory()
        idx = torch.full((B,), -1, dtype=torch.int32, device="cuda")
        batch = torch.zeros(B * brf + 64, device="cuda")
        mom = torch.zeros(2, mf, device="cuda")
        row_dev = torch.zeros(buf.row_floats, device="cuda")
        buf._sample_ctr.fill_(41)
        for rep in range(3):                                 # three timesteps in a row: the counters carry over
            row_pin.copy_(torch.from_numpy(new_rows[rep:rep + 1]))
            cnt[0] = 1 if append else 0
            if which == 0:
                if append is not None:
                    _lib.check(lib.naf_replay_add_counted(buf.handle, row_pin.data_ptr(), cnt.data_ptr(), 1, st()), "add")
                buf.sample_indices(idx, 1)
                buf.gather_rows(idx, batch[:B * brf].view(B, brf), B)
                _lib.check(lib.naf_bb_moments(batch.data_ptr(), B * brf, off_s2, brf, S, mom.data_ptr(), B, 1, 2, st()), "moments")
            else:
                rp, cp = (row_pin.data_ptr(), cnt.data_ptr()) if append is not None else (None, None)
                _lib.check(lib.naf_step_prep(buf.handle, rp, cp, row_dev.data_ptr() if rp else None, buf.seed,
                                             buf._sample_ctr.data_ptr(), idx.data_ptr(), batch.data_ptr(), brf, buf.action_mode,
                                             mom.data_ptr(), B, 1, None, None, None, st()), "step_prep")
            torch.cuda.synchronize()
            if which == 1 and append is not None:
                np.testing.assert_array_equal(row_dev.cpu().numpy(), new_rows[rep])     # the row's device copy, count or no count
        outs.append((buf.rows.clone(), buf.meta.clone(), buf._sample_ctr.clone(), idx.clone(), batch.clone(), mom.clone()))
    names = ("ring", "meta", "sampler counter", "indices", "minibatch rows", "moments")
    for name, a, b in zip(names, outs[0], outs[1]):
        assert torch.equal(a, b), name
    meta = outs[1][1].cpu().numpy()
    n_app = 3 if append else 0
    assert meta[1] == min(cap, n0 + n_app) and meta[2] == n0 + n_app and int(outs[1][2].item()) == 44
    if n0 + n_app >= B:
        assert len(set(outs[1][3].cpu().numpy().tolist())) == B          # without replacement
    assert meta[7] == 0 or n0 + n_app == 0


def test_step_prep_refuses_bad_arguments(lib):
    buf = _filled_buffer(100, 16, 21, 6, 50, seed=1)
    brf, mf = buf.batch_row_floats, lib.naf_bb_moments_floats(21)
    idx = torch.zeros(16, dtype=torch.int32, device="cuda")
    batch, mom = torch.zeros(16 * brf, device="cuda"), torch.zeros(2, mf, device="cuda")
    row = torch.zeros(1, buf.row_floats).pin_memory()
    ok = lambda **kw: lib.naf_step_prep(buf.handle, kw.get("row"), kw.get("cnt"), None, 0, buf._sample_ctr.data_ptr(), idx.data_ptr(),   # noqa: E731
                                       batch.data_ptr(), kw.get("ld", brf), 0, mom.data_ptr(), kw.get("B", 16), 1, kw.get("rec"), kw.get("ispec"),
                                       None, st())
    assert ok() == 0
    assert ok(row=row.data_ptr()) == -1                       # a row without its count word
    assert ok(ld=brf - 4) == -1 and ok(ld=brf + 4) == -1 and ok(B=0) == -1 and ok(B=5000) == -1
    rec = torch.zeros(12, dtype=torch.int32, device="cuda")
    assert ok(rec=rec.data_ptr()) == -1                       # a prefetch record without the prefetch's indices
    assert ok(rec=rec.data_ptr() + 4, ispec=idx.data_ptr()) == -1 and ok(rec=rec.data_ptr(), ispec=idx.data_ptr()) == 0
    torch.cuda.synchronize()


def _two_learners(S, A, B, seed, H=256):
    from robotic_manipulator_rloa_amd.learner import Learner
    Ls = []
    for _ in range(2):
        L = Learner(S, A, H, B, 1e-3, 1e-3, 0.99, DEV)
        g = torch.Generator(device="cuda").manual_seed(seed)
        L.theta2.copy_(0.1 * torch.randn(L.theta2.shape, generator=g, device="cuda"))
        L.grad.copy_(0.05 * torch.randn(L.grad.shape, generator=g, device="cuda"))
        L.adam_m.copy_(0.01 * torch.randn(L.grad.shape, generator=g, device="cuda"))
        L.adam_v.copy_(1e-4 * torch.rand(L.grad.shape, generator=g, device="cuda"))
        L.bn_stats[:, 0::2].copy_(0.3 * torch.randn(2, 2, L.lay.H, generator=g, device="cuda"))
        L.bn_stats[:, 1::2].copy_(0.5 + torch.rand(2, 2, L.lay.H, generator=g, device="cuda"))
        L.step_dev.fill_(7)
        Ls.append(L)
    return Ls


@pytest.mark.parametrize("p_mode", [0, 1])
@pytest.mark.parametrize("S,A,H", [(21, 6, 256), (23, 7, 256), (10, 5, 256), (21, 8, 256), (27, 9, 256), (29, 10, 256), (31, 11, 256),
                                   (32, 8, 256), (20, 9, 256), (21, 6, 512), (23, 7, 512), (27, 9, 512), (31, 11, 512)])
def test_adam_polyak_act_equals_the_two_launches_it_replaces(lib, S, A, H, p_mode):
    """naf_adam_polyak_act == naf_adam_polyak_fused followed by naf_policy_act: theta, theta', m, v, the heads' pre-activations
    and the (noisy, clamped) action bit for bit, over three consecutive launches (epochs of the records, the noise counter and the
    pinned ordinal move on); A = 8 takes two rows of Wh per layer-2 workgroup. H = 512 (round 6): 16 + 64 layer workgroups, rows as
    two float4 per lane — against policy_act_512_kernel."""
    from robotic_manipulator_rloa_amd import _lib
    from robotic_manipulator_rloa_amd.learner import ActPath
    La, Lb = _two_learners(S, A, 64, seed=3, H=H)
    La.p_mode = Lb.p_mode = p_mode
    acts = [ActPath(La, 1, seed=99, host_io=True), ActPath(Lb, 1, seed=99, host_io=True)]
    assert acts[1].can_ride
    rng = np.random.default_rng(0)
    for rep in range(3):
        obs = rng.standard_normal(S).astype(np.float32)
        for L in (La, Lb):
            _lib.check(lib.naf_grad_norm_partials(L.grad.data_ptr(), L.lay.P, L.partials.data_ptr(), L.step_dev.data_ptr(), st()), "norm")
            L._adam_args.n_partials = L.n_partials_norm
        acts[0].obs_np[0] = acts[1].obs_np[0] = obs
        _lib.check(lib.naf_adam_polyak_fused(
            La.theta2[0].data_ptr(), La.grad.data_ptr(), La.adam_m.data_ptr(), La.adam_v.data_ptr(), La.theta2[1].data_ptr(),
            La.partials.data_ptr(), La.n_partials_norm, 1.0, La.lr, 0.9, 0.999, 1e-8, La.tau, float(1.0 - La.tau),
            La.step_dev.data_ptr(), 1.0, La.lay.P, st()), "adam")
        acts[0].act(1.0)
        acts[1].act_with_optimizer_step(1.0)
        torch.cuda.synchronize()
        assert int(acts[1].seq_np[0]) == rep + 1 and acts[1].act_timeouts == 0
        for name in ("theta2", "adam_m", "adam_v"):
            assert torch.equal(getattr(La, name), getattr(Lb, name)), (name, rep)
        NH = La.lay.NH
        assert torch.equal(acts[0].Gh[0, :NH], acts[1].Gh[0, :NH]), rep
        np.testing.assert_array_equal(acts[0].actions_np, acts[1].actions_np)
        assert np.isfinite(acts[1].actions_np).all() and (np.abs(acts[1].actions_np) <= 1).all()
        assert int(acts[0].counter.item()) == int(acts[1].counter.item()) == rep + 1
        # move the gradient on so that the three launches differ
        for L in (La, Lb):
            L.grad.mul_(-0.7)
    # the new parameters differ from the old ones (the step did run)
    assert not torch.equal(La.theta2[0], _two_learners(S, A, 64, seed=3, H=H)[0].theta2[0])


@pytest.mark.parametrize("p_mode", [0, 1])
@pytest.mark.parametrize("S,A,H,B", [(21, 6, 256, 64), (21, 6, 256, 256), (23, 7, 256, 100), (27, 9, 256, 64), (21, 6, 512, 256),
                                     (31, 11, 384, 1000), (32, 8, 256, 2048)])
def test_adam_polyak_act_layer1_equals_the_three_launches_it_replaces(lib, S, A, H, B, p_mode):
    """naf_adam_polyak_act_layer1 (end of round 6: layer 1 of the next update's chain riding on the per-timestep path's first launch, in
    workgroups that evaluate the layer-1 parameters as the launch's own step will leave them) == naf_adam_polyak_fused +
    naf_policy_act + naf_bb_layer1_adam as launches of their own: parameters, optimizer state, target, the action, and everything
    layer 1 leaves — A1 of both networks, x-hat, the saved statistics, w_c C, the running statistics — bit for bit, three launches
    in a row; whole and partial 64-row blocks, both state widths (K4 = 6 | 8), 9 / 11 joints, layer sizes 512 / 384."""
    from robotic_manipulator_rloa_amd import _lib
    from robotic_manipulator_rloa_amd.learner import ActPath, BN_EPS, BN_MOMENTUM
    La, Lb = _two_learners(S, A, B, seed=5, H=H)
    La.p_mode = Lb.p_mode = p_mode
    acts = [ActPath(La, 1, seed=7, host_io=True), ActPath(Lb, 1, seed=7, host_io=True)]
    assert acts[1].can_ride
    g = torch.Generator(device="cuda").manual_seed(11)
    brf = La.lay.batch_row_floats
    rng = np.random.default_rng(1)
    for rep in range(3):
        rows = torch.randn(B + 1, brf, generator=g, device="cuda")[:B]
        obs = rng.standard_normal(S).astype(np.float32)
        outs = []
        for which, (L, act) in enumerate(zip((La, Lb), acts)):
            _lib.check(lib.naf_grad_norm_partials(L.grad.data_ptr(), L.lay.P, L.partials.data_ptr(), L.step_dev.data_ptr(), st()), "norm")
            L._adam_args.n_partials = L.n_partials_norm
            mom = torch.zeros(2, L.mom_floats, device="cuda")
            L.moments(rows, mom)
            act.obs_np[0] = obs
            l1 = L.layer1_args(rows, mom)
            if which == 0:
                _lib.check(lib.naf_adam_polyak_fused(
                    L.theta2[0].data_ptr(), L.grad.data_ptr(), L.adam_m.data_ptr(), L.adam_v.data_ptr(), L.theta2[1].data_ptr(),
                    L.partials.data_ptr(), L.n_partials_norm, 1.0, L.lr, 0.9, 0.999, 1e-8, L.tau, float(1.0 - L.tau),
                    L.step_dev.data_ptr(), 1.0, L.lay.P, st()), "adam")
                act.act(1.0)
                _lib.check(lib.naf_bb_layer1_adam(l1.x, l1.x_net_stride, l1.ldx, l1.K, l1.W, l1.bias, l1.gamma, l1.beta, l1.param_net_stride,
                                                  l1.mom, l1.running_mean, l1.running_var, l1.stat_net_stride, l1.out, l1.out_net_stride,
                                                  l1.ldo, l1.save_mean, l1.save_invstd, l1.wc_out, l1.xhat_out, l1.B, l1.H, 2, BN_MOMENTUM,
                                                  BN_EPS, None, st()), "layer1")
            else:
                act.act_with_optimizer_step(1.0, layer1=l1)
            torch.cuda.synchronize()
            assert act.act_timeouts == 0
            outs.append(dict(theta=L.theta2.clone(), m=L.adam_m.clone(), v=L.adam_v.clone(), A1=L.A1[:, :B].clone(),
                             xh=None if L.XH1 is None else L.XH1[:B].clone(), sm=L.save_mean[0].clone(), si=L.save_invstd[0].clone(),
                             wc=L.bb_wc.clone(), bn=L.bn_stats.clone(), act=act.actions_np.copy(), gh=act.Gh[0, :L.lay.NH].clone()))
        a, b = outs
        for k in a:
            if a[k] is None:
                continue
            if isinstance(a[k], np.ndarray):
                np.testing.assert_array_equal(a[k], b[k], err_msg=f"{k} (launch {rep})")
            else:
                assert torch.equal(a[k], b[k]), (k, rep)
        for L in (La, Lb):
            L.grad.mul_(-0.7)


def test_adam_polyak_act_skips_a_poisoned_update_and_still_acts(lib):
    """A -inf norm partial (a timed-out gradient exchange under data parallel, csrc/xgmi_reduce.hip) makes the optimizer leave
    every buffer alone; the action is then the policy's on the OLD parameters — as naf_adam_polyak_fused + naf_policy_act."""
    from robotic_manipulator_rloa_amd.learner import ActPath
    La, Lb = _two_learners(21, 6, 64, seed=4)
    before = La.theta2.clone()
    acts = [ActPath(La, 1, seed=5, host_io=True), ActPath(Lb, 1, seed=5, host_io=True)]
    for L in (La, Lb):
        L.partials.fill_(1.0)
        L.partials[3] = float("-inf")
    obs = np.linspace(-1, 1, 21).astype(np.float32)
    acts[0].obs_np[0] = acts[1].obs_np[0] = obs
    La.optimizer_step(norm_ready=True)
    acts[0].act(1.0)
    acts[1].act_with_optimizer_step(1.0)
    torch.cuda.synchronize()
    assert torch.equal(La.theta2, before) and torch.equal(Lb.theta2, before)
    np.testing.assert_array_equal(acts[0].actions_np, acts[1].actions_np)


def test_adam_polyak_act_layer1_skips_a_poisoned_update_like_the_launches_it_replaces(lib):
    """The same with layer 1 riding (naf_adam_polyak_act_layer1): a -inf norm partial leaves every parameter alone in the launch's
    own workgroups AND in the riders' evaluation of the layer-1 parameters — A1, the statistics and the action are those of the
    separate launches on the OLD parameters."""
    from robotic_manipulator_rloa_amd import _lib
    from robotic_manipulator_rloa_amd.learner import ActPath, BN_EPS, BN_MOMENTUM
    B = 128
    La, Lb = _two_learners(21, 6, B, seed=4)
    before = La.theta2.clone()
    acts = [ActPath(La, 1, seed=5, host_io=True), ActPath(Lb, 1, seed=5, host_io=True)]
    g = torch.Generator(device="cuda").manual_seed(3)
    rows = torch.randn(B + 1, La.lay.batch_row_floats, generator=g, device="cuda")[:B]
    obs = np.linspace(-1, 1, 21).astype(np.float32)
    outs = []
    for which, (L, act) in enumerate(zip((La, Lb), acts)):
        L.partials.fill_(1.0)
        L.partials[3] = float("-inf")
        mom = torch.zeros(2, L.mom_floats, device="cuda")
        L.moments(rows, mom)
        act.obs_np[0] = obs
        l1 = L.layer1_args(rows, mom)
        if which == 0:
            L.optimizer_step(norm_ready=True)
            act.act(1.0)
            _lib.check(lib.naf_bb_layer1_adam(l1.x, l1.x_net_stride, l1.ldx, l1.K, l1.W, l1.bias, l1.gamma, l1.beta, l1.param_net_stride,
                                              l1.mom, l1.running_mean, l1.running_var, l1.stat_net_stride, l1.out, l1.out_net_stride,
                                              l1.ldo, l1.save_mean, l1.save_invstd, l1.wc_out, l1.xhat_out, l1.B, l1.H, 2, BN_MOMENTUM,
                                              BN_EPS, None, st()), "layer1")
        else:
            act.act_with_optimizer_step(1.0, layer1=l1)
        torch.cuda.synchronize()
        outs.append((L.theta2.clone(), L.adam_m.clone(), L.A1[:, :B].clone(), L.bn_stats.clone(), L.save_invstd[0].clone(), act.actions_np.copy()))
    assert torch.equal(outs[0][0], before) and torch.equal(outs[1][0], before)
    for x, y in zip(outs[0][:-1], outs[1][:-1]):
        assert torch.equal(x, y)
    np.testing.assert_array_equal(outs[0][-1], outs[1][-1])


PREFETCH_CASES = [
    # S, A, B, capacity, rows in the ring, counts of the timesteps (1 = brings a row, 0 = an idle tick)
    (21, 6, 64, 100000, 50000, (1, 1, 1, 1, 1, 1)),         # the steady state: the new row is hardly ever drawn
    (21, 6, 64, 5000, 70, (1, 1, 1, 1, 1, 1, 1, 1)),         # a young ring: the new row is drawn most of the time
    (21, 6, 256, 100000, 3000, (1, 1, 0, 1, 1, 0, 0, 1)),    # idle ticks in between: the record assumed an append
    (23, 7, 64, 512, 512, (1, 1, 1, 1, 1, 1)),               # a full ring: every append evicts the oldest row
    (21, 6, 512, 100000, 20000, (1, 1, 1, 1)),               # B > 256: the form that gathers through memory
    (26, 6, 100, 4000, 3990, (1,) * 14),                     # K = 28 (the wide moments record), the ring fills up and wraps
    (27, 9, 64, 5000, 70, (1, 1, 1, 1, 1, 1, 1, 1)),         # round 6: 9 joints (ring rows of 128 floats), a young ring
    (31, 11, 256, 100000, 3000, (1, 1, 0, 1, 1, 0, 0, 1)),   # 11 joints, B = 256 (gathered through memory: 80 floats per row), idle ticks
]


@pytest.mark.parametrize("S,A,B,cap,n0,counts", PREFETCH_CASES)
def test_prefetched_minibatch_is_the_one_the_timestep_would_draw(lib, S, A, B, cap, n0, counts):
    """naf_adam_polyak_act's prefetching workgroup + the naf_step_prep that checks its record == naf_step_prep alone, timestep by
    timestep: ring, {head, size, total}, sampler counter, indices, minibatch rows and both moments records bit for bit — whether the
    record held (the launch only appended) or not (an idle tick, the new row among the positions drawn, the first timestep)."""
    from robotic_manipulator_rloa_amd import _lib
    from robotic_manipulator_rloa_amd.learner import ActPath
    from synth_data import make_transitions
    bufs = [_filled_buffer(cap, B, S, A, n0, seed=11) for _ in range(2)]
    brf, mf, rf = bufs[0].batch_row_floats, lib.naf_bb_moments_floats(S), bufs[0].row_floats
    s_, ac, rw, ns, dn = make_transitions(len(counts), S, A, seed=78)
    new_rows = O.pack_rows(s_, ac, rw, ns, dn, rf)
    L = _two_learners(S, A, 64, seed=3)[0]
    act = ActPath(L, 1, seed=9, host_io=True)
    _lib.check(lib.naf_grad_norm_partials(L.grad.data_ptr(), L.lay.P, L.partials.data_ptr(), L.step_dev.data_ptr(), st()), "norm")
    L._adam_args.n_partials = L.n_partials_norm
    state = []
    for buf in bufs:
        buf._sample_ctr.fill_(17)
        state.append(dict(idx=torch.full((B,), -1, dtype=torch.int32, device="cuda"), batch=torch.zeros(B * brf + 64, device="cuda"),
                          mom=torch.zeros(2, mf, device="cuda"), row_dev=torch.zeros(rf, device="cuda")))
    rec = torch.zeros(12, dtype=torch.int32, device="cuda")
    idx_spec = torch.zeros(B, dtype=torch.int32, device="cuda")
    b1, s1 = bufs[1], state[1]
    pf = _lib.StepPrefetch(b1.handle, b1.seed, b1._sample_ctr.data_ptr(), idx_spec.data_ptr(), s1["batch"].data_ptr(), brf,
                           b1.action_mode, s1["mom"].data_ptr(), B, 1, rec.data_ptr(), 1)
    row_pin = torch.zeros(1, rf).pin_memory()
    cnt = torch.zeros(1, dtype=torch.int32).pin_memory()
    for t, c in enumerate(counts):
        row_pin.copy_(torch.from_numpy(new_rows[t:t + 1]))
        cnt[0] = c
        for which, (buf, s) in enumerate(zip(bufs, state)):
            _lib.check(lib.naf_step_prep(buf.handle, row_pin.data_ptr(), cnt.data_ptr(), s["row_dev"].data_ptr(), buf.seed,
                                         buf._sample_ctr.data_ptr(), s["idx"].data_ptr(), s["batch"].data_ptr(), brf, buf.action_mode,
                                         s["mom"].data_ptr(), B, 1, rec.data_ptr() if which else None,
                                         idx_spec.data_ptr() if which else None, None, st()), "step_prep")
        torch.cuda.synchronize()
        for name in ("idx", "batch", "mom", "row_dev"):
            assert torch.equal(state[0][name], state[1][name]), (name, t)
        assert torch.equal(bufs[0].rows, bufs[1].rows) and torch.equal(bufs[0].meta, bufs[1].meta), t
        assert int(bufs[0]._sample_ctr.item()) == int(bufs[1]._sample_ctr.item()) == 18 + t
        assert int(rec[0].item()) == 0                         # a record serves one timestep
        # the launch that ends the timestep: the optimizer step, act() and the prefetch for timestep t + 1
        act.obs_np[0] = new_rows[t][b1.off_s2:b1.off_s2 + S]
        act.act_with_optimizer_step(1.0, prefetch=pf)
        torch.cuda.synchronize()
        assert act.act_timeouts == 0 and int(bufs[1]._sample_ctr.item()) == 18 + t      # nothing committed
        assert torch.equal(bufs[0].meta, bufs[1].meta)
    taken, drawn = int(rec[8].item()), int(rec[9].item())
    assert taken + drawn == len(counts) and drawn >= 1 + sum(1 for c in counts[1:] if c == 0)
    if n0 >= 20000:
        assert taken == sum(1 for c in counts[1:] if c == 1)   # (B / fill < 2 %: no draw met the new row with these seeds)
    if (cap, n0) == (5000, 70):
        assert drawn >= 3                                      # (B / fill = 0.9: the new row is among the positions most of the time)



DEPTH2_CASES = [
    # (S, A, B, capacity, rows in the ring before the first timestep, timesteps)
    (21, 6, 64, 100000, 20000, 12),              # configs[0]'s shape: B / fill = 0.3 %, every prefetch holds
    (21, 6, 64, 300, 299, 40),                   # a ring that fills up and wraps at once: four prefetches in ten are void
    (21, 6, 512, 100000, 20000, 8),              # B > 256: the form that gathers through memory
    (26, 6, 100, 4000, 3990, 16),                # K = 28 (the wide moments record), the ring fills up and wraps
    (23, 7, 64, 512, 512, 24),                   # a full ring: every append evicts the oldest row
    (27, 9, 64, 300, 299, 40),                   # round 6: 9 joints (ring rows of 128 floats), a ring that fills up and wraps
    (31, 11, 256, 100000, 20000, 8),             # 11 joints at B = 256: gathered through memory
]


@pytest.mark.parametrize("S,A,B,cap,n0,T", DEPTH2_CASES)
def test_depth2_prefetch_launch_is_what_the_timestep_would_draw(lib, S, A, B, cap, n0, T):
    """naf_step_prefetch (round 6: the append + the prefetch of the minibatch TWO timesteps ahead, a launch of its own) against
    naf_step_prep alone, timestep by timestep: the ring, {head, size, total}, the sampler's counter always; and whenever the verdict
    the launch gave for a minibatch says `valid`, that minibatch — rows, both moments records, indices — is bit for bit the one the
    timestep draws for itself two timesteps later, and the launch that consumes it finds its record to hold (no pipe error). A
    verdict says `void` exactly when one of the two rows to come is among the positions drawn."""
    from robotic_manipulator_rloa_amd import _lib
    from synth_data import make_transitions
    bufs = [_filled_buffer(cap, B, S, A, n0, seed=11) for _ in range(2)]
    base, pipe = bufs
    brf, mf, rf = base.batch_row_floats, lib.naf_bb_moments_floats(S), base.row_floats
    s_, ac, rw, ns, dn = make_transitions(T, S, A, seed=78)
    new_rows = O.pack_rows(s_, ac, rw, ns, dn, rf)
    for buf in bufs:
        buf._sample_ctr.fill_(17)
    z = lambda *shape, dt=torch.float32: torch.zeros(*shape, dtype=dt, device="cuda")            # noqa: E731
    ref = dict(idx=torch.full((B,), -1, dtype=torch.int32, device="cuda"), batch=z(B * brf + 64), mom=z(2, mf), row_dev=z(rf))
    sets = [dict(batch=z(B * brf + 64), mom=z(2, mf), idx=z(B, dt=torch.int32), rec=z(12, dt=torch.int32)) for _ in range(3)]
    idx_out = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    pf_seq = z(4, dt=torch.int32)
    host_spec = torch.zeros(3, 2, dtype=torch.int32).pin_memory()
    errs = torch.zeros(2, dtype=torch.int64).pin_memory()
    row_pin = torch.zeros(1, rf + 4).pin_memory()
    row_pin.view(-1)[rf:rf + 1].view(torch.int32)[0] = 1
    cnt_ptr = row_pin.data_ptr() + 4 * rf

    def pf(k, mode, depth, consume=None):
        x = sets[k]
        s = _lib.StepPrefetch(pipe.handle, pipe.seed, pipe._sample_ctr.data_ptr(), x["idx"].data_ptr(), x["batch"].data_ptr(), brf,
                              pipe.action_mode, x["mom"].data_ptr(), B, 1, x["rec"].data_ptr(), mode)
        s.host_spec, s.pipe_errors, s.depth, s.pf_seq = host_spec[k].data_ptr(), errs.data_ptr(), depth, pf_seq.data_ptr()
        if consume is not None:
            s.src_row, s.n_word, s.idx_out = row_pin.data_ptr(), cnt_ptr, idx_out.data_ptr()
            s.spec_rec_in, s.idx_spec_in = sets[consume]["rec"].data_ptr(), sets[consume]["idx"].data_ptr()
        return s
    # the pipeline's start, as the graph that starts a timestep over leaves it: minibatch 0 one append ahead, minibatch 1 two
    _lib.check(lib.naf_step_prefetch(C.byref(pf(0, 1, 1)), st()), "prefetch d1")
    _lib.check(lib.naf_step_prefetch(C.byref(pf(1, 1, 2)), st()), "prefetch d2")
    torch.cuda.synchronize()
    hs = host_spec.numpy()
    assert hs[0, 0] == 1 and hs[1, 0] == 2                      # (the ordinals pf_seq counts)
    valid = {0: bool(hs[0, 1]), 1: bool(hs[1, 1])}
    n_valid = n_void = 0
    for t in range(T):
        row_pin[0, :rf].copy_(torch.from_numpy(new_rows[t]))
        p = t % 3
        _lib.check(lib.naf_step_prep(base.handle, row_pin.data_ptr(), cnt_ptr, ref["row_dev"].data_ptr(), base.seed,
                                     base._sample_ctr.data_ptr(), ref["idx"].data_ptr(), ref["batch"].data_ptr(), brf, base.action_mode,
                                     ref["mom"].data_ptr(), B, 1, None, None, None, st()), "step_prep")
        before = int(errs[0])
        torch.cuda.synchronize()
        snap = {k: sets[p][k].clone() for k in ("batch", "mom", "idx")}          # (what was prefetched for THIS timestep)
        _lib.check(lib.naf_step_prefetch(C.byref(pf((p + 2) % 3, 2, 2, consume=p)), st()), "prefetch launch")
        torch.cuda.synchronize()
        assert torch.equal(base.rows, pipe.rows) and torch.equal(base.meta, pipe.meta), t
        assert int(base._sample_ctr.item()) == int(pipe._sample_ctr.item()) == 18 + t
        assert hs[(p + 2) % 3, 0] == 3 + t
        # does the void verdict say the truth? positions the baseline will draw at t + 2 are checked when t + 2 comes; here: this one
        head, size = int(base.meta[0].item()), int(base.meta[1].item())
        if valid[t]:
            n_valid += 1
            assert int(errs[0]) == before, t                   # the record held where it was consumed
            for k in ("batch", "mom", "idx"):
                assert torch.equal(snap[k], ref[k]), (k, t)
            assert torch.equal(idx_out, ref["idx"]), t
        else:
            n_void += 1
            # void = one of the rows that did not exist yet when it was drawn is among the positions the timestep draws: the row
            # appended now (physical head - 1) or, for a depth-2 draw, the one before it
            base_pos = (head + cap - size) % cap
            phys = (base_pos + ref["idx"].cpu().numpy().astype(np.int64)) % cap
            newest = {(head - 1) % cap} | ({(head - 2) % cap} if t >= 1 else set())
            assert newest & set(phys.tolist()), t
        valid[t + 2] = bool(hs[(p + 2) % 3, 1])
    assert n_valid >= 3, (n_valid, n_void)
    if cap == 300:
        assert n_void >= 5, (n_valid, n_void)
    if n0 >= 20000:
        assert n_void == 0


def _drive(agent, env_seed, warm, steps, record):
    """the reference's loop body (naf_algorithm.py:249-262) on a scripted stream of transitions; returns the actions taken"""
    from synth_data import make_transitions
    S, A = agent.state_size, agent.action_size
    st_, ac, rw, ns, dn = make_transitions(warm + steps + 1, S, A, seed=env_seed)
    actions = []
    state = st_[0].astype(np.float64)
    for t in range(warm + steps):
        a = agent.act(state)
        actions.append(np.array(a, copy=True))
        nxt = ns[t].astype(np.float64)
        agent.step(state, a, float(rw[t]), nxt, 0)
        if record is not None and t >= warm:
            torch.cuda.synchronize()
            record.append(agent._chunk.idx.cpu().numpy().copy())
        state = nxt
    torch.cuda.synchronize()
    return np.array(actions)


@pytest.mark.parametrize("B", [64, 256, 100])
def test_per_timestep_path_is_the_separate_launches_and_the_chunked_path_bit_for_bit(scratch_cwd, monkeypatch, B):
    """(a) NAFAgent.act / step through the two fused launches == the same loop through the twelve separate
    launches (NAF_STEP_FORM=separate): every action the policy took, parameters, target, Adam state, BatchNorm buffers, ring and counters
    bit-equal after 150 updates; (b) == the CHUNKED path: a TrainChunk of 150 teacher-forced updates (deferred optimizer steps,
    one moments launch for all) on the minibatches the per-timestep path drew, from the same initial state."""
    from robotic_manipulator_rloa_amd.engine import TrainChunk
    from robotic_manipulator_rloa_amd.learner import Learner
    from robotic_manipulator_rloa_amd.naf_components.naf_algorithm import NAFAgent
    S, A, N, T = 21, 6, 5000, 150
    warm = B                                                  # the gate: len(memory) > batch_size (naf_algorithm.py:150) opens at step B
    runs = []
    for form in ("pipelined", "separate", "fused", "prefetch"):
        # (separate: twelve launches; fused: the two launches of csrc/step_path.hip; prefetch: the next timestep's minibatch drawn
        #  by the last launch; pipelined: ... and its learn() chain run before its transition exists)
        monkeypatch.setenv("NAF_STEP_FORM", form)
        fused, prefetch, pipeline = ("0" if form == "separate" else "1", "0" if form in ("separate", "fused") else "1",
                                     "1" if form == "pipelined" else "0")
        agent = NAFAgent(object(), S, A, 256, B, N, 1e-3, 1e-3, 0.99, 1, 1, 500, DEV, 0)
        theta0 = agent.learner.theta2.clone()
        idx = [] if fused == "1" else None
        acts = _drive(agent, 21, warm, T, idx)
        ch = agent._chunk
        assert ch.fused_prep == ch.fused_tail == (fused == "1") and ch.head_row is not None and (agent._fast is not None)
        assert (ch.spec_rec is not None) == (fused == "1" and prefetch == "1")
        assert ch.pipelined == (form == "pipelined") and ch.form == form
        if ch.pipelined:
            # both graphs ran: the one that starts with the waiting gradient, and the one that starts over
            # (a timestep takes the six-launch graph only if NEITHER of the two rows to come was among the positions its minibatch's
            #  prefetch drew, two timesteps running: with B = 256 of 257 ... 406 rows that is rare)
            assert ch.fast_runs >= (5 if B < 256 else 1) and ch.slow_runs >= 5 and ch.fast_runs + ch.slow_runs in (T - 1, T), \
                (ch.fast_runs, ch.slow_runs)
        if ch.spec_rec is not None:
            taken, drawn = ch.prefetch_stats()
            # the ring holds B + 1 ... B + T rows: the new row is among the B positions drawn about as often as not — both ways
            # of a timestep are in this run
            assert taken + drawn in (T - 1, T) and taken >= (5 if B < 256 or not ch.pipelined else 1) and drawn >= 5, (taken, drawn)
        L = agent.learner
        runs.append(dict(acts=acts, theta=L.theta2.clone(), m=L.adam_m.clone(), v=L.adam_v.clone(), bn=L.bn_stats.clone(),
                         ring=agent.memory.rows.clone(), meta=agent.memory.meta.clone(), step=int(L.step_dev.item()),
                         loss=agent.last_loss(), idx=idx, theta0=theta0))
    a = runs[0]
    for b in runs[1:]:
        assert a["step"] == b["step"] == T
        np.testing.assert_array_equal(a["acts"], b["acts"])
        for k in ("theta", "m", "v", "bn", "ring", "meta"):
            assert torch.equal(a[k], b[k]), k
        assert a["loss"] == b["loss"] and torch.isfinite(a["theta"]).all()
    for other in (runs[2], runs[3]):
        for i, j in zip(runs[0]["idx"], other["idx"]):
            np.testing.assert_array_equal(i, j)                # (the indices a reader finds in chunk.idx: the timestep's own)
    assert not torch.equal(a["theta"], a["theta0"])
    # (b) the chunked path on the same minibatches: positions are stable while the ring only grows (no eviction here)
    from robotic_manipulator_rloa_amd.utils.replay_buffer import ReplayBuffer
    monkeypatch.delenv("NAF_STEP_FORM")
    L2 = Learner(S, A, 256, B, 1e-3, 1e-3, 0.99, DEV)
    L2.theta2.copy_(a["theta0"])
    buf = ReplayBuffer(N, B, DEV, 0, state_size=S, action_size=A)
    n_rows = int(a["meta"][1].item())
    buf.add_rows_device(a["ring"][:n_rows].contiguous(), n_rows)
    chunk = TrainChunk(L2, buf, T, teacher_forced=True)
    chunk.idx.copy_(torch.from_numpy(np.concatenate(a["idx"], 0)).view(T, B))
    chunk.run()
    torch.cuda.synchronize()
    assert int(L2.step_dev.item()) == T
    assert torch.equal(L2.theta2, a["theta"]) and torch.equal(L2.adam_m, a["m"]) and torch.equal(L2.adam_v, a["v"])
    assert torch.equal(L2.bn_stats, a["bn"])
    assert float(chunk.losses()[-1].item()) == a["loss"]



@pytest.mark.parametrize("B", [64, 256])
def test_pipelined_path_is_the_same_bits_whichever_of_its_two_launches_goes_first(scratch_cwd, monkeypatch, B):
    """naf_step_launch(..., prefetch_first): a tick whose verdict the host had to wait for launches the prefetch BEFORE the graph
    (engine._Pipeline.waited; DESIGN 4d, "two stable states"). Which ticks do depends on timing — here both orders are forced for a
    whole run each and held against the twelve-launch loop: every action, the parameters, Adam's state, the BatchNorm buffers, the ring."""
    from robotic_manipulator_rloa_amd import engine
    from robotic_manipulator_rloa_amd.naf_components.naf_algorithm import NAFAgent
    S, A, N, T = 21, 6, 20000, 400
    inner = engine._Pipeline.collect
    runs = []
    for form, first in (("pipelined", True), ("pipelined", False), ("separate", None)):
        monkeypatch.setenv("NAF_STEP_FORM", form)
        if first is not None:
            def collect(self, _first=first):
                inner(self)
                self.waited = _first
            monkeypatch.setattr(engine._Pipeline, "collect", collect)
        agent = NAFAgent(object(), S, A, 256, B, N, 1e-3, 1e-3, 0.99, 1, 1, 500, DEV, 0)
        # (a ring of 6000 rows before the first timestep: the prefetches hold in all but a few percent of the ticks)
        rng = np.random.default_rng(5)
        m = agent.memory
        r = np.zeros((6000, m.row_floats), np.float32)
        r[:, :S] = rng.standard_normal((6000, S))
        r[:, S:S + A] = rng.uniform(-1, 1, (6000, A))
        r[:, S + A] = -rng.random(6000)
        r[:, m.off_s2:m.off_s2 + S] = r[:, :S]
        m.add_rows_device(torch.from_numpy(r).to(DEV), 6000)
        acts = _drive(agent, 33, 0, T, None)
        ch = agent._chunk
        if first is not None:
            assert ch.pipelined and ch.fast_runs >= T * 3 // 4, (ch.fast_runs, ch.slow_runs)
            assert ch.pipe.side_first_runs == (ch.fast_runs if first else 0), (ch.pipe.side_first_runs, ch.fast_runs)
            assert not any(ch.error_words().values()), ch.error_words()
        L = agent.learner
        runs.append(dict(acts=acts, theta=L.theta2.clone(), m=L.adam_m.clone(), v=L.adam_v.clone(), bn=L.bn_stats.clone(),
                         ring=agent.memory.rows.clone(), meta=agent.memory.meta.clone(), step=int(L.step_dev.item()), loss=agent.last_loss()))
    a = runs[2]
    for b in runs[:2]:
        assert a["step"] == b["step"] == T
        np.testing.assert_array_equal(a["acts"], b["acts"])
        for k in ("theta", "m", "v", "bn", "ring", "meta"):
            assert torch.equal(a[k], b[k]), k
        assert a["loss"] == b["loss"] and torch.isfinite(b["theta"]).all()


@pytest.mark.parametrize("tag", ["kuka64", "kuka"])
def test_agent_step_on_the_reference_goldens_minibatches(scratch_cwd, monkeypatch, tag):
    """A direct pin of the per-timestep path on the unmodified reference's learn() (G3: Q, 14 gradients' norm, parameters and
    target after one step, BatchNorm buffers, the 5-step loss trace at KUKA 21 / 6, B = 64 and 256) — not through a chain of
    equalities with other launches: NAFAgent.act / NAFAgent.step themselves run five timesteps whose minibatches ARE the golden's.
    The sampler is not stubbed, it is predicted: its Philox stream is restated bit for bit by the oracle, so the ring is laid out
    such that the positions timestep k draws hold the golden's k-th minibatch, in the golden's order. Timestep 1 builds the graph
    (its update runs eagerly), timestep 2 starts the pipeline (the graph that starts over), 3 - 5 run the six-launch graph beside
    the prefetch launch."""
    from conftest import g3_case, load_group
    from robotic_manipulator_rloa_amd.naf_components.naf_algorithm import NAFAgent
    from synth_data import make_transitions
    from test_learner_gpu import current_sd
    from test_oracle_golden import assert_adam_stepped_close
    monkeypatch.setenv("NAF_STEP_FORM", "pipelined")
    g, main0, target0 = g3_case(tag)
    S, A, B = [int(x) for x in g[f"{tag}/dims"]]
    K = 5
    n0 = 50000 if B <= 64 else 1000000                                     # (five disjoint draws of B need a ring of >> 10 B^2 rows)
    N = n0 + 1000
    st_, ac, rw, ns, dn = make_transitions(K * B, S, A, seed=7)             # the golden's transitions (make_golden.py)
    # a sampler seed under which the five draws are disjoint and stay among the rows laid out beforehand
    for seed in range(200):
        draws = [O.replay_sample_indices(seed, k, n0 + k + 1, B, 1, True)[0] for k in range(K)]
        flat = np.concatenate(draws)
        if len(set(flat.tolist())) == K * B and flat.max() < n0:
            break
    else:
        raise AssertionError("no seed with five disjoint draws")
    agent = NAFAgent(object(), S, A, 256, B, N, 1e-3, 1e-3, 0.99, 1, 1, 500, DEV, seed)
    agent.qnetwork_main.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in main0.items()})
    agent.qnetwork_target.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in target0.items()})
    m = agent.memory
    f_s, f_a, f_r, f_n, f_d = make_transitions(K + 1, S, A, seed=123)         # the transitions the loop appends (never drawn)
    rows = torch.zeros(n0, m.row_floats, device="cuda")                       # (rows nobody draws: zeros)
    gold = torch.from_numpy(O.pack_rows(st_, ac, rw, ns, dn, m.row_floats)).cuda()
    for k in range(K):
        rows[torch.from_numpy(draws[k].astype(np.int64)).cuda()] = gold[k * B:(k + 1) * B]
    m.add_rows_device(rows, n0)
    del rows
    L = agent.learner
    losses, idx_seen = [], []
    state = f_s[0].astype(np.float64)
    for k in range(K):
        a = agent.act(state)
        nxt = f_n[k].astype(np.float64)
        agent.step(state, a, float(f_r[k]), nxt, 0)
        state = nxt
        torch.cuda.synchronize()
        losses.append(agent.last_loss())
        idx_seen.append(agent._chunk.idx.cpu().numpy().ravel().copy())
        if k == 0:
            np.testing.assert_allclose(L.q_out.cpu().numpy(), g[f"{tag}/q1"].ravel(), rtol=1e-3, atol=1e-3)
            norm = float(g[f"{tag}/grad_norm1"])
            np.testing.assert_allclose(np.sqrt(L.partials[:L.n_partials].sum().item()), norm, rtol=2e-4)
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
    for k in range(K):
        np.testing.assert_array_equal(idx_seen[k], draws[k])                 # the minibatch each timestep learned from: the golden's
    np.testing.assert_allclose(losses, g[f"{tag}/losses5"], rtol=5e-3)
    ch = agent._chunk
    assert ch.pipelined and int(L.step_dev.item()) == K and ch.slow_runs == 1 and ch.fast_runs == K - 2, (ch.fast_runs, ch.slow_runs)
    assert ch.error_words() == {"act_poll_timeouts": 0, "pipe_errors": 0, "verdict_waits_synchronised": 0}


def test_pipelined_path_survives_api_calls_between_timesteps(scratch_cwd, monkeypatch):
    """The pipelined per-timestep path keeps a gradient waiting between timesteps (taken on the prefetched minibatch while the
    host stepped the environment). Anything a user does to the agent in between — memory.sample(), learn() on an explicit
    minibatch, soft_update(), loading weights, adding a transition by hand, a second step() without an act() — voids it (call
    counters of the replay buffer and the learner, engine.TrainChunk._holds): the next timestep starts over, and the run equals
    the twelve-launch loop's bit for bit."""
    from robotic_manipulator_rloa_amd.naf_components.naf_algorithm import NAFAgent
    from synth_data import make_transitions
    S, A, B, N, T = 21, 6, 64, 4000, 120
    st_, ac, rw, ns, dn = make_transitions(B + T + 40, S, A, seed=5)
    runs = []
    for fused in ("1", "0"):
        monkeypatch.setenv("NAF_STEP_FORM", "pipelined" if fused == "1" else "separate")
        agent = NAFAgent(object(), S, A, 256, B, N, 1e-3, 1e-3, 0.99, 1, 1, 500, DEV, 0)
        state = st_[0].astype(np.float64)
        acts, extra = [], B + T
        for t in range(B + T):
            a = agent.act(state)
            acts.append(np.array(a, copy=True))
            nxt = ns[t].astype(np.float64)
            agent.step(state, a, float(rw[t]), nxt, 0)
            state = nxt
            k = t - B
            if k > 5 and k % 9 == 0:
                agent.memory.sample()                                        # moves the sampler's stream
            if k > 5 and k % 13 == 0:
                agent.learn(agent.memory.sample())                           # an update of the user's own
            if k > 5 and k % 17 == 0:
                agent.soft_update(agent.qnetwork_main, agent.qnetwork_target)
            if k > 5 and k % 23 == 0:
                agent.memory.add(st_[extra], ac[extra], float(rw[extra]), ns[extra], 0)      # a transition added by hand
                extra += 1
            if k > 5 and k % 29 == 0:
                agent.step(state, a, 0.25, state, 0)                         # a second step() without an act() in between
            if k == 60:
                sd = {kk: v.clone() for kk, v in agent.qnetwork_main.state_dict().items()}
                agent.qnetwork_target.load_state_dict(sd)                    # (weights loaded mid-run)
        torch.cuda.synchronize()
        ch, L = agent._chunk, agent.learner
        assert ch.pipelined == (fused == "1")
        if ch.pipelined:
            assert ch.fast_runs >= 10 and ch.slow_runs >= 20, (ch.fast_runs, ch.slow_runs)
            assert int(L.err_host[2]) == 0                                   # never launched on a prefetch that did not hold
        runs.append(dict(acts=np.array(acts), theta=L.theta2.clone(), m=L.adam_m.clone(), v=L.adam_v.clone(), bn=L.bn_stats.clone(),
                         ring=agent.memory.rows.clone(), meta=agent.memory.meta.clone(), step=int(L.step_dev.item()),
                         loss=agent.last_loss()))
    a, b = runs
    assert a["step"] == b["step"] and a["step"] > T
    np.testing.assert_array_equal(a["acts"], b["acts"])
    for k in ("theta", "m", "v", "bn", "ring", "meta"):
        assert torch.equal(a[k], b[k]), k
    assert a["loss"] == b["loss"]


@pytest.mark.parametrize("S,A,H,p_mode,action_mode", [(21, 6, 256, "hadamard", "trunc_int"), (23, 7, 256, "matmul", "float"),
                                                      (10, 5, 128, "matmul", "trunc_int"), (21, 6, 100, "hadamard", "trunc_int"),
                                                      (27, 9, 256, "hadamard", "trunc_int"), (31, 11, 256, "matmul", "float"),
                                                      (32, 8, 256, "hadamard", "trunc_int"), (21, 6, 512, "hadamard", "trunc_int"),
                                                      (23, 7, 384, "matmul", "float"), (27, 9, 512, "hadamard", "trunc_int")])
def test_pipelined_path_on_a_ring_that_wraps(scratch_cwd, monkeypatch, S, A, H, p_mode, action_mode):
    """600 timesteps on a ring of 300 rows at B = 64: every append evicts the oldest row after the first 300, the prefetch does not
    hold four times in ten (one of the two rows to come among the positions drawn) — the pipelined loop equals the twelve-launch loop all the way;
    also with the textbook P = L L^T head, float actions, the Panda's shapes, the reference agent test's network NAF(10, 5, 128) and a
    width of 100 (layers narrower than 256 are stored zero-padded: the same launches; state_dict() in the reference's shapes)."""
    from robotic_manipulator_rloa_amd.naf_components.naf_algorithm import NAFAgent
    B, N, T = 64, 300, 600
    runs = []
    for fused in ("1", "0"):
        monkeypatch.setenv("NAF_STEP_FORM", "pipelined" if fused == "1" else "separate")
        agent = NAFAgent(object(), S, A, H, B, N, 1e-3, 1e-3, 0.99, 1, 1, 500, DEV, 0, p_mode=p_mode, action_mode=action_mode)
        acts = _drive(agent, 33, B, T, None)
        ch, L = agent._chunk, agent.learner
        assert ch.pipelined == (fused == "1")
        sd = agent.qnetwork_main.state_dict()
        assert tuple(sd["hidden_layer.weight"].shape) == (H, H) and tuple(sd["input_layer.weight"].shape) == (H, S) and \
            tuple(sd["bn2.running_var"].shape) == (H,) and tuple(sd["value.weight"].shape) == (1, H)
        assert int(sd["bn1.num_batches_tracked"]) == T
        if ch.pipelined:
            # (64 of 300 rows drawn, two rows to come: a prefetch holds six times in ten, and a timestep needs two that do)
            assert ch.fast_runs > 150 and ch.slow_runs > 60 and int(L.err_host[2]) == 0, (ch.fast_runs, ch.slow_runs)
        runs.append(dict(acts=acts, theta=L.theta2.clone(), m=L.adam_m.clone(), v=L.adam_v.clone(), bn=L.bn_stats.clone(),
                         ring=agent.memory.rows.clone(), meta=agent.memory.meta.clone(), step=int(L.step_dev.item()),
                         loss=agent.last_loss()))
    a, b = runs
    assert a["step"] == b["step"] == T and int(a["meta"][1].item()) == N
    np.testing.assert_array_equal(a["acts"], b["acts"])
    for k in ("theta", "m", "v", "bn", "ring", "meta"):
        assert torch.equal(a[k], b[k]), k
    assert a["loss"] == b["loss"]


def _soak(form, steps, B, N, chunk=20000):
    """`steps` free-running timesteps of the reference's loop body on scripted transitions; the SHA-256 of every action taken and
    the final state of the agent"""
    import hashlib
    from robotic_manipulator_rloa_amd.naf_components.naf_algorithm import NAFAgent
    from synth_data import make_transitions
    os.environ["NAF_STEP_FORM"] = form
    S, A = 21, 6
    agent = NAFAgent(object(), S, A, 256, B, N, 1e-3, 1e-3, 0.99, 1, 1, 500, DEV, 0)
    h = hashlib.sha256()
    state, done = None, 0
    while done < steps:
        n = min(chunk, steps - done)
        st_, ac, rw, ns, dn = make_transitions(n + 1, S, A, seed=7000 + done)
        if state is None:
            state = st_[0].astype(np.float64)
        acts = np.empty((n, A), np.float32)
        for t in range(n):
            a = agent.act(state)
            acts[t] = a
            nxt = ns[t].astype(np.float64)
            agent.step(state, a, float(rw[t]), nxt, 0)
            state = nxt
        h.update(acts.tobytes())
        done += n
    torch.cuda.synchronize()
    L, ch = agent.learner, agent._chunk
    return dict(digest=h.hexdigest(), theta=L.theta2.clone(), m=L.adam_m.clone(), v=L.adam_v.clone(), bn=L.bn_stats.clone(),
                ring=agent.memory.rows.clone(), meta=agent.memory.meta.clone(), step=int(L.step_dev.item()), errors=ch.error_words(),
                runs=(ch.fast_runs, ch.slow_runs), form=ch.form)


def test_free_running_soak_200k_timesteps_equals_the_twelve_launch_loop(scratch_cwd, monkeypatch):
    """The host learns the graph's action, and the prefetch launch's verdict, WITHOUT synchronising a stream: hand-overs by stores
    and polls. Round 5's first form of the action's hand-over took a stale action once in 1e4 - 1e5 timesteps on some boxes (stores
    to host memory may pass one another; benchmarks/debug/soak.py found it): a bug of that rate needs a run of this length to show.
    200,000 free-running timesteps of the shipped (pipelined) loop at B = 64 on a ring of 20,000 rows that wraps nine times, against
    the twelve separate launches on the same transitions: the SHA-256 of every action taken, parameters, target, Adam state,
    BatchNorm buffers, ring and counters equal; no error word raised, no wait that had to synchronise."""
    monkeypatch.setenv("NAF_STEP_FORM", "pipelined")           # (restored by monkeypatch; _soak sets the form per run)
    T, B, N = 200000, 64, 20000
    a = _soak("pipelined", T, B, N)
    b = _soak("separate", T, B, N)
    assert a["form"] == "pipelined" and b["form"] == "separate"
    assert a["step"] == b["step"] == T - B and a["digest"] == b["digest"]
    for k in ("theta", "m", "v", "bn", "ring", "meta"):
        assert torch.equal(a[k], b[k]), k
    assert a["errors"] == {"act_poll_timeouts": 0, "pipe_errors": 0, "verdict_waits_synchronised": 0}, a["errors"]
    fast, slow = a["runs"]
    assert fast + slow in (T - B - 1, T - B) and fast > 0.98 * (T - B), (fast, slow)    # (2 B / N = 0.6 % of the prefetches void)


def test_per_timestep_path_is_five_to_seven_launches(scratch_cwd, monkeypatch):
    """The update graph of NAFAgent.step() at num_updates = 1 (profiles/r06_api_path_kernel_stats.csv has the same counts from
    rocprofv3). Pipelined (default): naf_adam_polyak_act_layer1 — the waiting gradient's optimizer step, act(), the commit, AND layer 1
    of the chain in extra workgroups behind the step — + the other four launches of the row-split chain on a minibatch prefetched two
    timesteps ago: five launches in the graph (six with NAF_STEP_L1_RIDE=0: layer 1 as a launch of its own), and naf_step_prefetch
    (the append + the prefetch of the minibatch two timesteps ahead) beside it on a stream of its own. The graph that starts a
    timestep over is naf_step_prep + chain + naf_adam_polyak_act (with the depth-1 prefetch) + naf_step_prefetch (depth 2) + chain. NAF_STEP_FORM=prefetch: naf_step_prep + the chain + naf_adam_polyak_act."""
    from robotic_manipulator_rloa_amd.naf_components.naf_algorithm import NAFAgent
    chain = ["naf_bb_layer1_adam", "naf_bb_linear_stats_adam", "naf_bb_layer2_head", "naf_gemm_bundle", "naf_bb_layer1_bwd_finish"]
    names = ["naf_step_prep", "naf_adam_polyak_act", "naf_adam_polyak_act_layer1", "naf_step_prefetch"] + chain + [
        "naf_replay_add_counted", "naf_replay_sample_indices", "naf_counter_add", "naf_replay_gather_rows", "naf_bb_moments",
        "naf_adam_polyak_fused", "naf_policy_act", "naf_grad_norm_partials"]
    for pipeline in ("1", "1-no-ride", "0"):
        monkeypatch.setenv("NAF_STEP_FORM", "pipelined" if pipeline != "0" else "prefetch")
        monkeypatch.setenv("NAF_STEP_L1_RIDE", "0" if pipeline == "1-no-ride" else "1")
        agent = NAFAgent(object(), 21, 6, 256, 64, 1000, 1e-3, 1e-3, 0.99, 1, 1, 500, DEV, 0)
        _drive(agent, 3, 65, 5, None)
        ch = agent._chunk
        assert ch.fused_prep and ch.fused_tail and ch.graph is not None and ch.pipelined == (pipeline != "0")
        calls = []
        lib = agent.learner.lib

        class Spy:
            def __init__(self, inner):
                self._inner = inner

            def __getattr__(self, name):
                fn = getattr(self._inner, name)
                if name in names:
                    def wrapped(*a, **k):
                        calls.append(name)
                        return fn(*a, **k)
                    return wrapped
                return fn
        spy = Spy(lib)
        agent.learner.lib = agent.learner._f = spy
        ch.L.lib = spy
        torch.cuda.synchronize()
        ch._body()                                             # one eager pass through exactly what the (slower) graph holds
        torch.cuda.synchronize()
        if pipeline != "0":
            assert calls == ["naf_step_prep"] + chain + ["naf_adam_polyak_act", "naf_step_prefetch"] + chain, calls
            del calls[:]
            ch.pipe.body_fast(1)                               # (the graph of a timestep in phase 1: what _body() leaves)
            torch.cuda.synchronize()
            assert calls == (["naf_adam_polyak_act_layer1"] + chain[1:] if pipeline == "1" else ["naf_adam_polyak_act"] + chain), calls
            agent.learner.err_host[2] = 0                      # (an eager pass outside the host's bookkeeping may have counted a mismatch)
        else:
            assert calls == ["naf_step_prep"] + chain + ["naf_adam_polyak_act"], calls
        agent.learner.lib = agent.learner._f = lib
