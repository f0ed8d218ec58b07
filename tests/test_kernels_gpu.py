"""GPU parity tests proper: every entry point of libnaf_hip.so (through the C ABI, via ctypes) against the
numpy oracle and the golden vectors generated from the unmodified reference. Bit-exact for byte/index work
(replay add/gather/sample, Polyak), tolerance stated per test for floating point."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_group
from oracle import naf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from robotic_manipulator_rloa_amd import _lib
    _lib.require_gpu()
    return _lib.load(allow_build=False)


def dev(x, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(x)).to("cuda", dtype)


def st():
    return torch.cuda.current_stream().cuda_stream


def heads_rows(mu_pre, l_pre, V, ldh):
    B, A = mu_pre.shape
    T = l_pre.shape[1]
    h = np.zeros((B, ldh), np.float32)
    h[:, :A], h[:, A:A + T], h[:, A + T] = mu_pre, l_pre, np.asarray(V).reshape(-1)
    return h


# ------------------------------------------------------------------------------------------------------------
# replay
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("S,A", [(21, 6), (23, 7), (19, 5), (10, 5)])
def test_replay_add_gather_fifo_and_trunc(lib, S, A):
    from robotic_manipulator_rloa_amd.utils.replay_buffer import ReplayBuffer
    from robotic_manipulator_rloa_amd import _lib
    from synth_data import make_transitions
    cap = 1000
    buf = ReplayBuffer(cap, 64, "cuda", 0, state_size=S, action_size=A)
    rf = buf.row_floats
    assert rf == (64 if S > 10 else 32)
    st_, ac, rw, ns, dn = make_transitions(2600, S, A, seed=3)
    rows = O.pack_rows(st_, ac, rw, ns, dn, rf)
    batches = [rows[0:1], rows[1:65], rows[65:700], rows[700:1300], rows[1300:2300], rows[2300:2600]]  # ragged, wraps
    for b in batches:
        buf.add_rows_device(dev(b), b.shape[0])
        torch.cuda.synchronize()
    assert len(buf) == cap
    meta = buf.meta.cpu().numpy()
    assert meta[1] == cap and meta[2] == 2600 and meta[0] == 2600 % cap
    expect = O.ring_after_adds(cap, batches)              # deque order: oldest first
    idx = torch.arange(cap, dtype=torch.int32, device="cuda")
    out = torch.empty(cap, rf, device="cuda")
    buf.action_mode = _lib.ACTION_FLOAT
    buf.gather_rows(idx, out, cap)
    np.testing.assert_array_equal(out.cpu().numpy(), expect)          # byte-exact row copy, FIFO eviction
    buf.action_mode = _lib.ACTION_TRUNC_INT
    perm = torch.from_numpy(np.random.default_rng(0).permutation(cap).astype(np.int32)).cuda()
    buf.gather_rows(perm, out, cap)
    exp_t = expect[perm.cpu().numpy()].copy()
    exp_t[:, S:S + A] = np.trunc(exp_t[:, S:S + A])                    # the reference's .long() (replay_buffer.py:60)
    np.testing.assert_array_equal(out.cpu().numpy(), exp_t)
    # bulk path (4 rows in flight per lane) on a big index list with repeats
    big = torch.from_numpy(np.random.default_rng(1).integers(0, cap, 70000).astype(np.int32)).cuda()
    outb = torch.empty(70000, rf, device="cuda")
    buf.gather_rows(big, outb, 70000)
    np.testing.assert_array_equal(outb.cpu().numpy(), exp_t_full(expect, big.cpu().numpy(), S, A))
    assert buf.bad_index_count() == 0
    # packed minibatch rows (what TrainChunk gathers): the leading batch_row_floats of every row, nothing else written
    brf = buf.batch_row_floats
    assert brf == lib.naf_replay_batch_row_floats(S, A) and brf % 4 == 0 and O.row_offsets(S, A)[3] + 1 <= brf <= rf
    assert (S, A, brf) in ((21, 6, 52), (23, 7, 56), (19, 5, 52), (10, 5, 32))      # (10, 5): the run-time-width kernel instance
    for n_rows, src in ((cap, perm), (70000, big)):
        outp = torch.full((n_rows * brf + 8,), -7.0, device="cuda")
        buf.gather_rows(src, outp[:n_rows * brf].view(n_rows, brf), n_rows)
        np.testing.assert_array_equal(outp[:n_rows * brf].view(n_rows, brf).cpu().numpy(),
                                      exp_t_full(expect, src.cpu().numpy(), S, A)[:, :brf])
        assert (outp[n_rows * brf:] == -7.0).all()
    assert lib.naf_replay_gather_rows(buf.handle, perm.data_ptr(), out.data_ptr(), 4, brf - 8 if S > 10 else 24, 0, st()) == -1
    assert lib.naf_replay_gather_rows(buf.handle, perm.data_ptr(), out.data_ptr(), 4, rf + 4, 0, st()) == -1
    assert lib.naf_replay_gather_rows(buf.handle, perm.data_ptr(), out.data_ptr(), 4, brf + 2, 0, st()) == -1
    # out-of-range positions are counted, not silently used
    bad = torch.tensor([0, cap, -1, 5], dtype=torch.int32, device="cuda")
    buf.gather_rows(bad, out, 4)
    assert buf.bad_index_count() == 2


@pytest.mark.parametrize("cap,S,A,B", [(1_000_000, 21, 6, 256), (4_000_000, 23, 7, 2048)])
def test_replay_ring_at_baseline_sizes(lib, cap, S, A, B):
    """The rings of BASELINE configs[1] (1e6 rows = 256 MB) and configs[4] (4e6 rows = 1.02 GB, beyond the Infinity
    Cache): filled past wrap-around with id-tagged rows in ragged appends, then the ReplayBuffer.sample contract
    (utils/replay_buffer.py:47-67) at the config's batch size — positions in range and distinct inside a minibatch,
    sampler bit-exact against the oracle's restatement, gathered rows byte-exact (every float of a row is a function of
    its id, so each of the U*B gathered rows is checked in full), FIFO eviction = the newest `cap` ids survive."""
    from robotic_manipulator_rloa_amd.utils.replay_buffer import ReplayBuffer
    buf = ReplayBuffer(cap, B, "cuda", 0xBEEF, state_size=S, action_size=A)
    rf, brf, off2 = buf.row_floats, buf.batch_row_floats, buf.off_s2
    total = cap + cap // 3 + 12345                                  # wraps; < 2^24 so ids are exact in f32
    assert total < (1 << 24)

    def rows_of(ids):                                               # [n] int64 ids -> [n, rf] f32 rows, exact arithmetic
        idf = ids.to(torch.float32)
        r = torch.zeros(ids.numel(), rf, device="cuda")
        cols = torch.arange(S, device="cuda", dtype=torch.float32)
        r[:, :S] = torch.where(cols[None, :] == 0, idf[:, None], (idf[:, None] % 4096.0) + cols[None, :] * 0.5)
        r[:, S:S + A] = ((idf[:, None] % 7.0) - 3.0) * 0.75 + torch.arange(A, device="cuda")[None, :] * 0.125
        r[:, S + A] = -(idf % 1000.0)
        r[:, off2:off2 + S] = (idf[:, None] % 8192.0) * 0.25 - cols[None, :]
        r[:, off2 + S] = (ids % 5 == 0).to(torch.float32)
        return r

    lo = 0
    for n in [1, 63, 64, 65, 1000, 100_003] + [1 << 20] * 8:        # ragged appends: single rows up to 1 Mi rows at once
        n = min(n, total - lo, cap)
        if n > 0:
            buf.add_rows_device(rows_of(torch.arange(lo, lo + n, device="cuda")), n)
            lo += n
    assert lo == total
    torch.cuda.synchronize()
    assert len(buf) == cap and buf.device_len() == cap
    meta = buf.meta.cpu().numpy()
    assert meta[0] == total % cap and meta[2] == total
    U = 8
    idx = torch.zeros(U, B, dtype=torch.int32, device="cuda")
    buf.sample_indices(idx, U)
    got = idx.cpu().numpy()
    np.testing.assert_array_equal(got, O.replay_sample_indices(0xBEEF, 0, cap, B, U, True))
    assert got.min() >= 0 and got.max() < cap and all(len(set(g.tolist())) == B for g in got)
    out = torch.empty(U, B, brf, device="cuda")
    buf.gather_rows(idx, out, U * B)                                # the minibatch-sized launch, packed rows
    ids = idx.long().reshape(-1) + (total - cap)                    # deque position p holds id (total - cap) + p
    exp = rows_of(ids)
    exp[:, S:S + A] = torch.trunc(exp[:, S:S + A])                  # `.long()` (replay_buffer.py:60)
    assert torch.equal(out.view(-1, brf), exp[:, :brf])
    # the bulk launch (4 float4 in flight per lane, nontemporal stores) over 1 Mi random positions incl. both ends
    M = 1 << 20
    pos = torch.randint(0, cap, (M,), device="cuda", dtype=torch.int32)
    pos[0], pos[1] = 0, cap - 1
    outb = torch.empty(M, brf, device="cuda")
    buf.gather_rows(pos, outb, M)
    expb = rows_of(pos.long() + (total - cap))
    expb[:, S:S + A] = torch.trunc(expb[:, S:S + A])
    assert torch.equal(outb, expb[:, :brf])
    # the reference-contract path: five tensors
    s, a, r, s2, d = buf.sample(idx=idx[0])
    assert torch.equal(s, exp[:B, :S]) and torch.equal(a, exp[:B, S:S + A].long()) and torch.equal(r[:, 0], exp[:B, S + A])
    assert torch.equal(s2, exp[:B, off2:off2 + S]) and torch.equal(d[:, 0], exp[:B, off2 + S])
    assert buf.bad_index_count() == 0


def exp_t_full(expect, idx, S, A):
    e = expect[idx].copy()
    e[:, S:S + A] = np.trunc(e[:, S:S + A])
    return e


def test_replay_sample_api_contract_matches_reference_golden(lib):
    """ReplayBuffer.add/sample through the host API against G4 (the reference's own sample() output for the
    same deque positions)."""
    from robotic_manipulator_rloa_amd.utils.replay_buffer import ReplayBuffer
    from synth_data import make_transitions
    g = np.load(os.path.join(GOLDEN, "g4_replay.npz"))
    S, A, cap, B = [int(x) for x in g["dims"]]
    st_, ac, rw, ns, dn = make_transitions(500, S, A, seed=11)
    buf = ReplayBuffer(cap, B, torch.device("cuda"), 0)
    for i in range(500):
        s = st_[i].astype(np.float64).copy()
        s[0] = float(i)
        buf.add(s, ac[i], float(rw[i]), ns[i].astype(np.float64), int(dn[i]))
    assert len(buf) == cap
    buf.flush()
    assert buf.device_len() == cap                  # naf_replay_size: the fill level as the device holds it
    pos = torch.from_numpy(g["positions_from_range"][0].astype(np.int32))
    s, a, r, s2, d = buf.sample(idx=pos)
    assert [str(t.dtype) for t in (s, a, r, s2, d)] == list(g["dtypes"])
    assert [tuple(t.shape) for t in (s, a, r, s2, d)] == [(B, S), (B, A), (B, 1), (B, S), (B, 1)]
    np.testing.assert_array_equal(s.cpu().numpy(), g["s"])
    np.testing.assert_array_equal(a.cpu().numpy(), g["a"])
    np.testing.assert_array_equal(r.cpu().numpy(), g["r"])
    np.testing.assert_array_equal(s2.cpu().numpy(), g["s2"])
    np.testing.assert_array_equal(d.cpu().numpy(), g["d"])
    # free-running sample(): right shapes, in range, no duplicates
    s, a, r, s2, d = buf.sample()
    ids = s[:, 0].cpu().numpy()
    assert len(set(ids.tolist())) == B and ids.min() >= 200 and ids.max() <= 499


# (duplicates are resolved through an LDS hash table — up to 80 KB of LDS at B = 4096, csrc/replay.hip: (300, 64), (5000, 1024),
#  (9000, 2048), (20000, 2048), (20000, 4096) and (9000, 3000) redraw dozens to hundreds of elements over several rounds; 100, 48
#  and 3000 are not powers of two)
# (beyond 4096 the table is in device memory, csrc/replay.hip replay_sample_big_kernel — the same rule, so the same restatement:
#  (100000, 8192) a few redraws, (30000, 8192) a thousand over several rounds, (20000, 8192) the dense regime's partial
#  Fisher-Yates, (40000, 5000) not a power of two, (200000, 16384) round 5's largest minibatch; round 6: the table slot of an element has
#  a word of its own and the limit is 2^20 — (300000, 20000), (150000, 65536) a few thousand redraws, (100000, 40000) the dense regime)
@pytest.mark.parametrize("size,B,nb", [(1000, 256, 8), (257, 256, 3), (300, 64, 5), (100000, 2048, 2), (5000, 1024, 2),
                                       (9000, 2048, 3), (20000, 2048, 4), (50000, 4096, 2), (20000, 4096, 3), (9000, 3000, 2),
                                       (700, 100, 6), (200, 48, 9),
                                       (1_000_000, 256, 64),
                                       (100000, 8192, 2), (30000, 8192, 3), (20000, 8192, 2), (40000, 5000, 2), (200000, 16384, 1),
                                       (300000, 20000, 1), (150000, 65536, 1), (100000, 40000, 1)])
def test_replay_sampler_bit_exact_vs_oracle(lib, size, B, nb):
    from robotic_manipulator_rloa_amd.utils.replay_buffer import ReplayBuffer
    buf = ReplayBuffer(size, B, "cuda", 0x1234ABCD5678, state_size=21, action_size=6)
    buf.add_rows_device(torch.zeros(size, 64, device="cuda"), size)
    idx = torch.zeros(nb, B, dtype=torch.int32, device="cuda")
    buf._sample_ctr.fill_(7)
    buf.sample_indices(idx, nb)
    got = idx.cpu().numpy()
    exp = O.replay_sample_indices(0x1234ABCD5678, 7, size, B, nb, True)
    np.testing.assert_array_equal(got, exp)
    for b in range(nb):
        assert len(set(got[b].tolist())) == B
    assert int(buf._sample_ctr.item()) == 7 + nb
    # with replacement: plain draws
    buf.without_replacement = False
    buf.sample_indices(idx, nb)
    np.testing.assert_array_equal(idx.cpu().numpy(), O.replay_sample_indices(0x1234ABCD5678, 7 + nb, size, B, nb, False))


def test_replay_sampler_uniformity(lib):
    from robotic_manipulator_rloa_amd.utils.replay_buffer import ReplayBuffer
    size, B, nb = 4096, 256, 4000
    buf = ReplayBuffer(size, B, "cuda", 99, state_size=21, action_size=6)
    buf.add_rows_device(torch.zeros(size, 64, device="cuda"), size)
    idx = torch.zeros(nb, B, dtype=torch.int32, device="cuda")
    buf.sample_indices(idx, nb)
    counts = np.bincount(idx.cpu().numpy().ravel(), minlength=size).astype(np.float64)
    expected = nb * B / size
    chi2 = ((counts - expected) ** 2 / expected).sum()
    assert abs(chi2 - size) < 6 * np.sqrt(2 * size)      # chi-square with ~size dof


# ------------------------------------------------------------------------------------------------------------
# NAF head
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("A", [5, 6, 7])
@pytest.mark.parametrize("B", [2, 256])
@pytest.mark.parametrize("tag", ["rand", "wide"])
def test_head_vs_reference_golden(lib, A, B, tag):
    """Hadamard mode against the reference's own forward/autograd outputs (G2). rtol 2e-5 forward (f32 tanh/exp
    differ by an ulp or two between libm and the device), 2e-4 backward."""
    g = load_group(np.load(os.path.join(GOLDEN, "g2_head.npz")), f"A{A}_B{B}_{tag}")
    T = A * (A + 1) // 2
    ldh = (A + T + 1 + 7) // 8 * 8
    h = dev(heads_rows(g["mu_pre"], g["l_pre"], g["V"], ldh))
    u = dev(g["u_trunc"])
    q = torch.empty(B, device="cuda")
    mu = torch.empty(B, A, device="cuda")
    assert lib.naf_head_fwd(h.data_ptr(), ldh, u.data_ptr(), A, q.data_ptr(), mu.data_ptr(), B, A, 0, st()) == 0
    np.testing.assert_allclose(q.cpu().numpy(), g["q"].ravel(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(mu.cpu().numpy(), np.tanh(g["mu_pre"]), rtol=1e-5, atol=1e-6)
    dq = dev(g["dq"].ravel())
    dh = torch.full((B, ldh), 7.0, device="cuda")
    assert lib.naf_head_bwd(h.data_ptr(), ldh, u.data_ptr(), A, dq.data_ptr(), dh.data_ptr(), B, A, 0, st()) == 0
    dh = dh.cpu().numpy()
    np.testing.assert_allclose(dh[:, :A], g["d_mu_pre"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(dh[:, A:A + T], g["d_l_pre"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(dh[:, A + T], g["d_V"].ravel(), rtol=1e-6)
    assert (dh[:, A + T + 1:] == 0).all()                              # pad columns are written as zeros
    # off-diagonal l entries get EXACTLY zero gradient under the reference's Hadamard P
    diag = [k * (k + 3) // 2 for k in range(A)]
    off = [k for k in range(T) if k not in diag]
    assert (dh[:, A:A + T][:, off] == 0).all()


@pytest.mark.parametrize("mode", [0, 1])
# (A > 8: one sample per 16-lane group, csrc/naf_head_wide.hip — the reference builds matrix_entries for any action size,
#  naf_neural_network.py:53-54)
#  beyond 16 joints (round 6): per 32- / 64-lane group through the shared body, naf_head_any_kernel)
@pytest.mark.parametrize("A,B", [(6, 256), (7, 2048), (3, 33), (8, 100), (1, 5), (9, 256), (12, 100), (16, 37), (10, 1),
                                 (17, 64), (24, 100), (32, 9), (33, 50), (48, 256), (64, 21)])
def test_head_both_modes_vs_oracle_f64(lib, mode, A, B):
    rng = np.random.default_rng(10 * A + B + mode)
    T = A * (A + 1) // 2
    ldh = (A + T + 1 + 7) // 8 * 8
    mu_pre, l_pre, V = rng.standard_normal((B, A)), rng.standard_normal((B, T)), rng.standard_normal(B)
    u = np.trunc(rng.uniform(-1.5, 1.5, (B, A)))
    r, vn = rng.standard_normal(B), rng.standard_normal(B)
    gamma = 0.99
    f = O.head_forward(mu_pre, l_pre, V, u, mode)
    y = r + gamma * vn
    dq = 2 * (f["Q"] - y) / B
    d_mu, d_l, d_V = O.head_backward(mu_pre, l_pre, u, dq, mode)
    h = dev(heads_rows(mu_pre, l_pre, V, ldh))
    ud, rd, vnd = dev(u), dev(r), dev(vn)
    q = torch.empty(B, device="cuda")
    dh = torch.empty(B, ldh, device="cuda")
    nwg = (B + 7) // 8 if A <= 32 else (B + 3) // 4      # (one loss part per workgroup: 8 samples each, 4 beyond 32 joints)
    lp = torch.zeros(nwg, device="cuda")
    assert lib.naf_head_fwd_bwd_mse(h.data_ptr(), ldh, ud.data_ptr(), A, rd.data_ptr(), 1, vnd.data_ptr(), 1, gamma,
                                    q.data_ptr(), dh.data_ptr(), lp.data_ptr(), B, A, mode, st()) == 0
    if A > 8:
        # the forward-only and the backward-given-dq entry points on the same inputs
        q0, mu0 = torch.empty(B, device="cuda"), torch.empty(B, A, device="cuda")
        assert lib.naf_head_fwd(h.data_ptr(), ldh, ud.data_ptr(), A, q0.data_ptr(), mu0.data_ptr(), B, A, mode, st()) == 0
        dh0 = torch.full((B, ldh), 7.0, device="cuda")
        assert lib.naf_head_bwd(h.data_ptr(), ldh, ud.data_ptr(), A, dev(dq).data_ptr(), dh0.data_ptr(), B, A, mode, st()) == 0
        torch.cuda.synchronize()
        assert torch.equal(q0, q)
        np.testing.assert_allclose(mu0.cpu().numpy(), np.tanh(mu_pre), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(dh0.cpu().numpy(), dh.cpu().numpy(), rtol=2e-4, atol=1e-7)
        assert (dh0[:, A + T + 1:] == 0).all()
    np.testing.assert_allclose(q.cpu().numpy(), f["Q"], rtol=3e-5, atol=3e-5)
    dhn = dh.cpu().numpy()
    scale = np.abs(dq).max() * 10
    np.testing.assert_allclose(dhn[:, :A], d_mu, rtol=3e-4, atol=1e-6 * scale + 1e-7)
    np.testing.assert_allclose(dhn[:, A:A + T], d_l, rtol=3e-4, atol=1e-6 * scale + 1e-7)
    np.testing.assert_allclose(dhn[:, A + T], d_V, rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(lp.sum().item(), ((f["Q"] - y) ** 2).mean(), rtol=1e-4)
    # run-to-run bitwise reproducible
    dh2 = torch.empty_like(dh)
    lp2 = torch.zeros_like(lp)
    lib.naf_head_fwd_bwd_mse(h.data_ptr(), ldh, ud.data_ptr(), A, rd.data_ptr(), 1, vnd.data_ptr(), 1, gamma,
                             q.data_ptr(), dh2.data_ptr(), lp2.data_ptr(), B, A, mode, st())
    assert torch.equal(dh, dh2) and torch.equal(lp, lp2)


def test_head_argument_errors(lib):
    h = torch.zeros(4, 32, device="cuda")
    u = torch.zeros(4, 6, device="cuda")
    q = torch.zeros(4, device="cuda")
    assert lib.naf_head_fwd(h.data_ptr(), 32, u.data_ptr(), 6, q.data_ptr(), None, 4, 65, 0, st()) == -1  # A > 64
    assert lib.naf_head_fwd(h.data_ptr(), 32, u.data_ptr(), 17, q.data_ptr(), None, 4, 17, 0, st()) == -1  # A = 17 needs ldh >= 171
    assert lib.naf_head_fwd(h.data_ptr(), 32, u.data_ptr(), 9, q.data_ptr(), None, 4, 9, 0, st()) == -1   # A = 9 needs ldh >= 55
    assert lib.naf_head_fwd(h.data_ptr(), 24, u.data_ptr(), 6, q.data_ptr(), None, 4, 6, 0, st()) == -1   # ldh too small
    assert lib.naf_head_fwd(h.data_ptr(), 32, u.data_ptr(), 6, q.data_ptr(), None, 4, 6, 2, st()) == -1   # bad mode
    assert lib.naf_head_fwd(None, 32, u.data_ptr(), 6, q.data_ptr(), None, 4, 6, 0, st()) == -1


@pytest.mark.parametrize("mode", [0, 1])
def test_act_noise_distribution(lib, mode):
    """clamp(mu + P^-1/2 z): first two moments against the analytic covariance inverse(P) (what the reference's
    MultivariateNormal(mu, inverse(P)) samples from, naf_neural_network.py:119), and z bit-stream against the
    oracle's Philox restatement."""
    A, E = 6, 64
    T = A * (A + 1) // 2
    rng = np.random.default_rng(5)
    mu_pre, l_pre = 0.1 * rng.standard_normal((E, A)), rng.standard_normal((E, T))
    h = dev(heads_rows(mu_pre, l_pre, np.zeros(E), 32))
    n_draws = 4000
    acts = torch.empty(n_draws, E, A, device="cuda")
    ctr = torch.zeros(1, dtype=torch.int64, device="cuda")
    for k in range(n_draws):
        assert lib.naf_act_noise(h.data_ptr(), 32, acts[k].data_ptr(), 42, ctr.data_ptr(), k, 0.05, E, A, mode, st()) == 0
    a = acts.cpu().numpy().astype(np.float64)
    f = O.head_forward(mu_pre, l_pre, np.zeros(E), np.zeros((E, A)), mode)
    cov = np.linalg.inv(f["P"]) * 0.05 ** 2                      # noise_scale^2 * inverse(P); small so the clamp is idle
    sd = np.sqrt(np.einsum("eii->ei", cov))
    free = (np.abs(f["mu"]) + 4.5 * sd < 1.0).all(axis=1)          # states where the +-1 clamp never bites
    assert free.sum() >= E // 2
    np.testing.assert_allclose(a.mean(0)[free], f["mu"][free], atol=5 * sd[free].max() / np.sqrt(n_draws))
    emp = np.einsum("kei,kej->eij", a - a.mean(0), a - a.mean(0)) / (n_draws - 1)
    np.testing.assert_allclose(emp[free], cov[free], rtol=0.25, atol=0.12 * np.abs(cov[free]).max())
    # Hadamard: exact stream check for draw 0
    if mode == 0:
        z = O.normal_from_philox(42, 0, np.arange(E)[:, None], np.arange(A)[None, :])
        sigma = O.noise_std_hadamard(l_pre, A)
        np.testing.assert_allclose(a[0], np.clip(f["mu"] + 0.05 * sigma * z, -1, 1), rtol=1e-4, atol=2e-6)
    # noise_scale = 1 and big sigma: the clamp holds
    lib.naf_act_noise(h.data_ptr(), 32, acts[0].data_ptr(), 1, None, 0, 50.0, E, A, mode, st())
    assert acts[0].abs().max().item() <= 1.0


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("A,E", [(9, 40), (16, 5), (20, 33), (32, 8), (40, 17), (64, 6)])
def test_act_noise_beyond_8_joints_is_the_oracles_draw(lib, mode, A, E):
    """naf_act_noise for 9 .. 64 joints (one sample per 16- / 32- / 64-lane group): the action is clamp(mu + s x) with x the solution
    of P^(1/2) x = z for the z of the oracle's Philox restatement — Hadamard: x_i = z_i / L_ii; matmul: L^T x = z (checked as that
    identity: back substitution over up to 64 lanes)."""
    T = A * (A + 1) // 2
    ldh = (A + T + 1 + 7) // 8 * 8
    rng = np.random.default_rng(A + E)
    # (small off-diagonals beyond 16 joints: a random triangular matrix of order 64 with entries of 0.5 is hopelessly ill-conditioned)
    mu_pre, l_pre = 0.05 * rng.standard_normal((E, A)), (0.5 if A <= 16 else 0.05) * rng.standard_normal((E, T))
    h = dev(heads_rows(mu_pre, l_pre, np.zeros(E), ldh))
    act = torch.empty(E, A, device="cuda")
    ctr = torch.full((1,), 3, dtype=torch.int64, device="cuda")
    scale = 1e-3                                                  # small: the clamp never bites, x is read back exactly
    assert lib.naf_act_noise(h.data_ptr(), ldh, act.data_ptr(), 77, ctr.data_ptr(), 2, scale, E, A, mode, st()) == 0
    torch.cuda.synchronize()
    f = O.head_forward(mu_pre, l_pre, np.zeros(E), np.zeros((E, A)), mode)
    z = O.normal_from_philox(77, 5, np.arange(E)[:, None], np.arange(A)[None, :])
    x = (act.cpu().numpy().astype(np.float64) - np.tanh(mu_pre)) / scale
    if mode == 0:
        np.testing.assert_allclose(x, z * O.noise_std_hadamard(l_pre, A), rtol=2e-3, atol=2e-3)
    else:
        np.testing.assert_allclose(np.einsum("eji,ej->ei", f["L"], x), z, rtol=5e-3, atol=5e-3 * np.abs(z).max())


# ------------------------------------------------------------------------------------------------------------
# BatchNorm + ReLU
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,H", [(256, 256), (2, 256), (64, 256), (2048, 256), (33, 40), (1024, 64)])
def test_bn_relu_train_fwd_bwd_vs_oracle(lib, B, H):
    rng = np.random.default_rng(B + H)
    nets = 2
    g = rng.standard_normal((nets, B, H)) * 2 + 0.5
    bias, gamma, beta = rng.standard_normal((nets, H)), rng.uniform(0.5, 1.5, (nets, H)), rng.standard_normal((nets, H))
    rm, rv = rng.standard_normal((nets, H)), rng.uniform(0.5, 2, (nets, H))
    P = 3 * H + 16   # fake flat layout: [bias | gamma | beta | pad] per net
    flat = np.zeros((nets, P), np.float32)
    flat[:, :H], flat[:, H:2 * H], flat[:, 2 * H:3 * H] = bias, gamma, beta
    flat_d, g_d = dev(flat), dev(g)
    stats = dev(np.stack([rm, rv], 1))                                 # [nets][2][H]
    ldo = H + 8
    out = torch.zeros(nets, B, ldo, device="cuda")
    sm, si = torch.empty(nets, H, device="cuda"), torch.empty(nets, H, device="cuda")
    fp = flat_d.data_ptr()
    assert lib.naf_bn_relu_fwd_train(g_d.data_ptr(), B * H, H, fp, fp + 4 * H, fp + 8 * H, P, stats.data_ptr(),
                                     stats.data_ptr() + 4 * H, 2 * H, out.data_ptr(), B * ldo, ldo, sm.data_ptr(),
                                     si.data_ptr(), B, H, nets, 0.1, 1e-5, st()) == 0
    outn, statn = out.cpu().numpy(), stats.cpu().numpy()
    tol = 5e-5 if B > 2 else 2e-3     # a batch of 2 amplifies rounding (x - mean)/sqrt(var+eps) with var ~ eps
    caches = []
    for n in range(nets):
        z = (g[n] + bias[n])
        y, cache, nrm, nrv = O.bn_train_forward(z, gamma[n], beta[n], rm[n], rv[n])
        caches.append((z, cache))
        np.testing.assert_allclose(outn[n, :, :H], np.maximum(y, 0), rtol=tol, atol=tol)
        assert (outn[n, :, H:] == 0).all()                             # columns beyond H are never touched
        np.testing.assert_allclose(statn[n, 0], nrm, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(statn[n, 1], nrv, rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(sm[n].cpu().numpy(), cache["mean"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(si[n].cpu().numpy(), cache["invstd"], rtol=1e-4)
    # backward for net 0
    d_out = rng.standard_normal((B, ldo))
    z, cache = caches[0]
    dy = d_out[:, :H] * (outn[0, :, :H] > 0)
    dz, dgam, dbet = O.bn_train_backward(dy, {"xhat": (z - sm[0].cpu().numpy().astype(np.float64)) * si[0].cpu().numpy(),
                                              "invstd": si[0].cpu().numpy().astype(np.float64)}, gamma[0])
    dzd = torch.empty(B, H, device="cuda")
    dg, db, dbias = (torch.empty(H, device="cuda") for _ in range(3))
    d_out_d = dev(d_out)
    assert lib.naf_bn_relu_bwd(d_out_d.data_ptr(), ldo, g_d.data_ptr(), H, fp, out.data_ptr(), ldo, fp + 4 * H,
                               sm.data_ptr(), si.data_ptr(), dzd.data_ptr(), H, dg.data_ptr(), db.data_ptr(),
                               dbias.data_ptr(), B, H, st()) == 0
    s = max(1.0, np.abs(dz).max())
    np.testing.assert_allclose(dzd.cpu().numpy(), dz, rtol=1e-3, atol=2e-5 * s)
    np.testing.assert_allclose(dg.cpu().numpy(), dgam, rtol=1e-3, atol=1e-4 * np.abs(dgam).max())
    np.testing.assert_allclose(db.cpu().numpy(), dbet, rtol=1e-3, atol=1e-4 * np.abs(dbet).max())
    np.testing.assert_allclose(dbias.cpu().numpy(), 0, atol=1e-3 * s)   # bias under train-mode BN: gradient ~ 0


def test_bn_relu_eval_vs_oracle(lib):
    rng = np.random.default_rng(0)
    B, H = 64, 256
    g, bias = rng.standard_normal((B, H)), rng.standard_normal(H)
    gamma, beta, rm, rv = rng.uniform(.5, 1.5, H), rng.standard_normal(H), rng.standard_normal(H), rng.uniform(.5, 2, H)
    out = torch.zeros(B, H + 8, device="cuda")
    t = [dev(x) for x in (g, bias, gamma, beta, rm, rv)]
    assert lib.naf_bn_relu_fwd_eval(t[0].data_ptr(), H, t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), t[4].data_ptr(),
                                    t[5].data_ptr(), out.data_ptr(), H + 8, B, H, 1e-5, st()) == 0
    exp = np.maximum(O.bn_eval_forward(g + bias, gamma, beta, rm, rv), 0)
    np.testing.assert_allclose(out.cpu().numpy()[:, :H], exp, rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------------------------------------------------
# optimizer
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [83264, 4096, 5000, 1023])
def test_polyak_bit_exact(lib, n):
    rng = np.random.default_rng(n)
    main, tgt = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    tau = 1e-3
    md, td = dev(main), dev(tgt)
    assert lib.naf_polyak_update(td.data_ptr(), md.data_ptr(), tau, float(1.0 - tau), n, st()) == 0
    exp = O.polyak(tgt, main, tau)      # f32: fl(fl(tau*main) + fl((1-tau)*target)), exactly the reference's expression
    np.testing.assert_array_equal(td.cpu().numpy(), exp)


@pytest.mark.parametrize("n,gscale,world", [(83264, 1.0, 1), (83264, 1e-3, 1), (5000, 30.0, 1), (83264, 1.0, 8)])
def test_clip_adam_polyak_vs_oracle(lib, n, gscale, world):
    """5 consecutive optimizer steps. Tolerance: 2e-6 absolute on parameters (step size is lr = 1e-3, f32)."""
    rng = np.random.default_rng(1)
    th = rng.standard_normal(n).astype(np.float32) * 0.05
    tg = th.copy()
    m, v = np.zeros(n, np.float32), np.zeros(n, np.float32)
    thd, tgd, md, vd = dev(th), dev(tg), dev(m), dev(v)
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    nparts = (n + 4095) // 4096
    parts = torch.zeros(nparts, device="cuda")
    lr, tau = 1e-3, 1e-3
    for t in range(1, 6):
        g = (rng.standard_normal(n) * gscale).astype(np.float32)
        g[rng.random(n) < 0.3] = 0.0          # exact zeros: padding and Hadamard-dead weights
        gd = dev(g)
        assert lib.naf_grad_norm_partials(gd.data_ptr(), n, parts.data_ptr(), step.data_ptr(), st()) == 0
        assert lib.naf_adam_polyak_fused(thd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(), tgd.data_ptr(),
                                         parts.data_ptr(), nparts, 1.0, lr, 0.9, 0.999, 1e-8, tau, float(1 - tau),
                                         step.data_ptr(), 1.0 / world, n, st()) == 0
        gavg = (g.astype(np.float64) / world)
        grads, total = O.clip_grad_norm({"g": gavg.astype(np.float32)}, 1.0)
        np.testing.assert_allclose(np.sqrt(parts.sum().item()) / world, total, rtol=1e-5)
        th, m, v = O.adam_step(th, grads["g"], m, v, t, lr)
        tg = O.polyak(tg, th, tau)
        assert int(step.item()) == t
        np.testing.assert_allclose(thd.cpu().numpy(), th, rtol=0, atol=2e-6)
        np.testing.assert_allclose(md.cpu().numpy(), m, rtol=5e-5, atol=1e-9)      # clip factor from an f32 norm
        np.testing.assert_allclose(vd.cpu().numpy(), v, rtol=1e-4, atol=1e-12)
        np.testing.assert_allclose(tgd.cpu().numpy(), tg, rtol=0, atol=1e-6)
        zero = g == 0
        if t == 1:
            assert (thd.cpu().numpy()[zero] == th[zero]).all()        # zero gradient + zero moments: no movement


@pytest.mark.parametrize("S,A,B,U", [(21, 6, 1024, 3), (23, 7, 2048, 2), (26, 8, 64, 1), (11, 1, 192, 2)])
def test_bb_moments_vs_numpy_f64(lib, S, A, B, U):
    """naf_bb_moments (csrc/big_batch.hip): column sums and centred second moments of the state / next-state columns of U
    minibatches in one launch, against numpy in float64; and the identities the large-batch chain builds on them —
    mean_c = b_c + w_c . m, var_c = w_c^T C w_c / B — against the statistics of z = X W^T + b computed directly."""
    from synth_data import make_transitions
    rf = lib.naf_replay_row_floats(S, A)
    brf = lib.naf_replay_batch_row_floats(S, A)
    off2 = lib.naf_replay_row_off_next_state(S, A)
    st_, ac, rw, ns, dn = make_transitions(U * B, S, A, seed=S + B)
    st_[:, -3:] = 0.4 + 1e-3 * st_[:, -3:]                        # nearly constant columns, like target / obstacle positions
    rows = O.pack_rows(st_, ac, rw, ns, dn, rf)[:, :brf].copy()
    d = dev(np.concatenate([rows.reshape(-1), np.zeros(64, np.float32)]))
    n = lib.naf_bb_moments_floats(S)
    kp = 24 if S <= 24 else 32
    assert n == kp + kp * kp
    mom = torch.zeros(U, 2, n, device="cuda")
    assert lib.naf_bb_moments(d.data_ptr(), B * brf, off2, brf, S, mom.data_ptr(), B, U, 2, st()) == 0
    torch.cuda.synchronize()
    got = mom.cpu().numpy().astype(np.float64)
    rng = np.random.default_rng(0)
    W = rng.normal(0, 0.3, (256, S))
    bias = rng.normal(0, 0.1, 256)
    for u in range(U):
        blk = rows[u * B:(u + 1) * B].astype(np.float64)
        for net, off in ((0, 0), (1, off2)):
            X = blk[:, off:off + S]                              # (the record's columns beyond S meet zero weights: not compared)
            Sx, m = X.sum(0), X.mean(0)
            Cm = (X - m).T @ (X - m)
            np.testing.assert_allclose(got[u, net, :S], Sx, rtol=2e-6, atol=2e-4)
            np.testing.assert_allclose(got[u, net, kp:].reshape(kp, kp)[:S, :S], Cm, rtol=2e-5,
                                       atol=2e-4 * max(1.0, np.abs(Cm).max() * 1e-3))
            z = X @ W.T + bias
            Cg = got[u, net, kp:].reshape(kp, kp)[:S, :S]
            np.testing.assert_allclose(bias + W @ (got[u, net, :S] / B), z.mean(0), rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(np.einsum("cj,jk,ck->c", W, Cg, W) / B, z.var(0), rtol=2e-5, atol=1e-9)


def test_adam_skips_the_update_when_the_norm_partials_are_poisoned(lib):
    """A timed-out one-shot gradient exchange leaves -inf in its sum-of-squares partial (csrc/xgmi_reduce.hip): the
    optimizer kernel must then leave theta, m, v and the target untouched — and behave normally otherwise."""
    n = 83264
    torch.manual_seed(0)
    theta, g, m, v, tgt = (torch.randn(n, device="cuda") for _ in range(5))
    v = v.abs()
    part = torch.rand(21, device="cuda")
    step = torch.ones(1, dtype=torch.int32, device="cuda")
    saved = [t.clone() for t in (theta, m, v, tgt)]

    def run():
        assert lib.naf_adam_polyak_fused(theta.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), tgt.data_ptr(),
                                         part.data_ptr(), 21, 1.0, 1e-3, 0.9, 0.999, 1e-8, 1e-3, 0.999, step.data_ptr(), 1.0,
                                         n, st()) == 0
        torch.cuda.synchronize()
    part[13] = float("-inf")
    run()
    assert all(torch.equal(a, b) for a, b in zip((theta, m, v, tgt), saved))
    part[13] = 0.5
    run()
    assert not torch.equal(theta, saved[0]) and not torch.equal(tgt, saved[3]) and torch.isfinite(theta).all()


def test_symbols_exported(lib):
    from robotic_manipulator_rloa_amd import _lib
    for name in _lib.EXPORTED_SYMBOLS:
        assert hasattr(lib, name)
    assert lib.naf_hip_arch() == b"gfx950"


# ------------------------------------------------------------------------------------------------------------
# fused small-GEMM layers (csrc/fused_layers.hip)
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,K,H", [(256, 21, 256), (64, 23, 256), (2, 10, 256), (512, 23, 256), (100, 32, 40), (256, 32, 256), (33, 5, 70)])
def test_linear_bn_relu_fused_fwd_and_wgrad_vs_oracle(lib, B, K, H):
    rng = np.random.default_rng(B + K)
    nets, ldx, xoff = 2, 64, 28 if K <= 24 else 32
    rows = rng.standard_normal((B, ldx)) * 3.0                  # columns beyond K hold other (finite) row fields
    W = rng.standard_normal((nets, H, K)) / np.sqrt(K)
    bias, gamma, beta = rng.standard_normal((nets, H)), rng.uniform(0.5, 1.5, (nets, H)), rng.standard_normal((nets, H))
    rm, rv = rng.standard_normal((nets, H)), rng.uniform(0.5, 2, (nets, H))
    P = H * K + 3 * H + 8
    flat = np.zeros((nets, P), np.float32)
    flat[:, :H * K], flat[:, H * K:H * K + H] = W.reshape(nets, -1), bias
    flat[:, H * K + H:H * K + 2 * H], flat[:, H * K + 2 * H:H * K + 3 * H] = gamma, beta
    fd, xd = dev(flat), dev(rows)
    stats = dev(np.stack([rm, rv], 1))
    out = torch.zeros(nets, B, H, device="cuda")
    sm, si = torch.empty(nets, H, device="cuda"), torch.empty(nets, H, device="cuda")
    fp = fd.data_ptr()
    assert lib.naf_linear_bn_relu_fwd_train(xd.data_ptr(), xoff, ldx, K, fp, fp + 4 * H * K, fp + 4 * (H * K + H),
                                            fp + 4 * (H * K + 2 * H), P, stats.data_ptr(), stats.data_ptr() + 4 * H, 2 * H,
                                            out.data_ptr(), B * H, H, sm.data_ptr(), si.data_ptr(), B, H, nets, 0.1, 1e-5,
                                            st()) == 0
    tol = 5e-5 if B > 2 else 2e-3
    caches = []
    for n in range(nets):
        x = rows[:, n * xoff:n * xoff + K]
        z = x @ W[n].T + bias[n]
        y, cache, nrm, nrv = O.bn_train_forward(z, gamma[n], beta[n], rm[n], rv[n])
        caches.append((x, z, cache))
        np.testing.assert_allclose(out[n].cpu().numpy(), np.maximum(y, 0), rtol=tol, atol=tol)
        np.testing.assert_allclose(stats[n, 0].cpu().numpy(), nrm, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(stats[n, 1].cpu().numpy(), nrv, rtol=1e-4, atol=1e-6)
    # backward + weight gradient of net 0
    x, z, cache = caches[0]
    d_out = rng.standard_normal((B, H))
    mean, invstd = sm[0].cpu().numpy().astype(np.float64), si[0].cpu().numpy().astype(np.float64)
    dy = d_out * (out[0].cpu().numpy() > 0)
    dz, dgam, dbet = O.bn_train_backward(dy, {"xhat": (z - mean) * invstd, "invstd": invstd}, gamma[0])
    dg, db, dbias = (torch.empty(H, device="cuda") for _ in range(3))
    dW = torch.full((H, K), 9.0, device="cuda")
    dod = dev(d_out)
    nblk = (H + 7) // 8
    sq = torch.zeros(nblk, device="cuda")
    stepc = torch.zeros(1, dtype=torch.int32, device="cuda")
    assert lib.naf_bn_relu_bwd_wgrad(dod.data_ptr(), H, xd.data_ptr(), ldx, K, fp, fp + 4 * H * K, out.data_ptr(), H,
                                     fp + 4 * (H * K + H), sm.data_ptr(), si.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                     dbias.data_ptr(), dW.data_ptr(), sq.data_ptr(), stepc.data_ptr(), B, H, st()) == 0
    total = (dW.double() ** 2).sum() + (dg.double() ** 2).sum() + (db.double() ** 2).sum() + (dbias.double() ** 2).sum()
    np.testing.assert_allclose(sq.double().sum().item(), total.item(), rtol=1e-5)      # folded grad-norm partials
    assert int(stepc.item()) == 1
    gW = dz.T @ x
    s = max(1.0, np.abs(gW).max())
    np.testing.assert_allclose(dW.cpu().numpy(), gW, rtol=2e-3, atol=3e-5 * s)
    np.testing.assert_allclose(dg.cpu().numpy(), dgam, rtol=1e-3, atol=1e-4 * np.abs(dgam).max())
    np.testing.assert_allclose(db.cpu().numpy(), dbet, rtol=1e-3, atol=1e-4 * np.abs(dbet).max())
    np.testing.assert_allclose(dbias.cpu().numpy(), 0, atol=1e-3 * max(1.0, np.abs(dz).max()))
    # alignment contract
    assert lib.naf_linear_bn_relu_fwd_train(xd.data_ptr() + 4, xoff, ldx, K, fp, fp, fp, fp, P, stats.data_ptr(), stats.data_ptr(),
                                            0, out.data_ptr(), 0, H, sm.data_ptr(), si.data_ptr(), B, H, 1, 0.1, 1e-5, st()) == -1


@pytest.mark.parametrize("B,NHP,H", [(256, 32, 256), (64, 48, 256), (512, 48, 256), (33, 16, 70)])
def test_heads_bwd_bn_relu_bwd_fused_vs_oracle(lib, B, NHP, H):
    rng = np.random.default_rng(B + NHP)
    ldw = H + 16
    dH = rng.standard_normal((B, NHP))
    Wh = rng.standard_normal((NHP, ldw)) * 0.1
    g, bias = rng.standard_normal((B, H)) * 2, rng.standard_normal(H)
    gamma = rng.uniform(0.5, 1.5, H)
    z = g + bias
    y, cache, _, _ = O.bn_train_forward(z, gamma, rng.standard_normal(H), np.zeros(H), np.ones(H))
    outp = np.maximum(y, 0)
    d_out = dH @ Wh[:, :H]
    dz, dgam, dbet = O.bn_train_backward(d_out * (outp > 0), cache, gamma)
    t = {k: dev(v) for k, v in dict(dH=dH, Wh=Wh, g=g, bias=bias, out=outp, gamma=gamma, mean=cache["mean"], inv=cache["invstd"]).items()}
    dzd = torch.empty(B, H, device="cuda")
    dg, db, dbias = (torch.empty(H, device="cuda") for _ in range(3))
    sq = torch.zeros((H + 7) // 8, device="cuda")
    assert lib.naf_heads_bwd_bn_relu_bwd(t["dH"].data_ptr(), NHP, t["Wh"].data_ptr(), ldw, t["g"].data_ptr(), H,
                                         t["bias"].data_ptr(), t["out"].data_ptr(), H, t["gamma"].data_ptr(), t["mean"].data_ptr(),
                                         t["inv"].data_ptr(), dzd.data_ptr(), H, dg.data_ptr(), db.data_ptr(), dbias.data_ptr(),
                                         sq.data_ptr(), B, H, st()) == 0
    total = (dg.double() ** 2).sum() + (db.double() ** 2).sum() + (dbias.double() ** 2).sum()
    np.testing.assert_allclose(sq.double().sum().item(), total.item(), rtol=1e-5)
    s = max(1.0, np.abs(dz).max())
    np.testing.assert_allclose(dzd.cpu().numpy(), dz, rtol=1e-3, atol=3e-5 * s)
    np.testing.assert_allclose(dg.cpu().numpy(), dgam, rtol=1e-3, atol=1e-4 * np.abs(dgam).max())
    np.testing.assert_allclose(db.cpu().numpy(), dbet, rtol=1e-3, atol=1e-4 * np.abs(dbet).max())
    assert lib.naf_heads_bwd_bn_relu_bwd(t["dH"].data_ptr(), 40, t["Wh"].data_ptr(), ldw, t["g"].data_ptr(), H, None,
                                         t["out"].data_ptr(), H, t["gamma"].data_ptr(), t["mean"].data_ptr(), t["inv"].data_ptr(),
                                         dzd.data_ptr(), H, dg.data_ptr(), db.data_ptr(), None, None, B, H, st()) == -1


@pytest.mark.parametrize("ak,bk", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_bundle_mfma_vs_numpy(lib, ak, bk):
    """Three GEMMs of different shapes in one launch of f32-MFMA tiles, every operand storage combination; asymmetric
    random operands (a symmetric B would hide a row/column swap)."""
    from robotic_manipulator_rloa_amd import _lib
    rng = np.random.default_rng(10 * ak + bk)
    shapes = [(32, 272, 256), (256, 256, 64), (48, 16, 272)]
    descs, keep, expect, sums = [], [], [], []
    for (M, N, K) in shapes:
        A = rng.standard_normal((M, K))
        Bm = rng.standard_normal((N, K))
        lda, ldb, ldc = (M + 16 if ak else K + 8), (N + 4 if bk else K + 12), N + 16
        a_store = np.zeros((K, lda)) if ak else np.zeros((M, lda))
        b_store = np.zeros((K, ldb)) if bk else np.zeros((N, ldb))
        if ak: a_store[:, :M] = A.T
        else: a_store[:, :K] = A
        if bk: b_store[:, :N] = Bm.T
        else: b_store[:, :K] = Bm
        ad, bd = dev(a_store), dev(b_store)
        cd = torch.full((M, ldc), -7.0, device="cuda")
        sqd = torch.zeros(((M + 31) // 32) * ((N + 31) // 32), device="cuda")
        keep += [ad, bd, cd]
        sums.append(sqd)
        descs.append(_lib.GemmDesc(ad.data_ptr(), bd.data_ptr(), cd.data_ptr(), sqd.data_ptr(), M, N, K, lda, ldb, ldc, ak, bk))
        expect.append(A @ Bm.T)
    arr = (_lib.GemmDesc * 3)(*descs)
    assert lib.naf_gemm_bundle(arr, 3, st()) == 0
    for i, (M, N, K) in enumerate(shapes):
        got = keep[3 * i + 2].cpu().numpy()
        np.testing.assert_allclose(got[:, :N], expect[i], rtol=1e-4, atol=1e-4 * np.sqrt(K))
        assert (got[:, N:] == -7.0).all()
        np.testing.assert_allclose(sums[i].double().sum().item(), (expect[i] ** 2).sum(), rtol=1e-4)   # per-block sum of C^2
    bad = (_lib.GemmDesc * 1)(_lib.GemmDesc(keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(), None, 30, 272, 256, 300, 300, 300, ak, bk))
    assert lib.naf_gemm_bundle(bad, 1, st()) == -1            # M not a multiple of 16


@pytest.mark.parametrize("B", [128, 2048])
def test_gemm_bundle_bn2bwd_prologue_fold_once_and_fallback(lib, B):
    """naf_gemm_bn2bwd_t through the C ABI: the second stage of layer 2's BatchNorm backward (autograd of
    naf_neural_network.py:79-80) applied to the A panels while they are staged, its block sums folded ONCE per launch by the
    launch's first workgroups and handed on as tagged records — against numpy in double (both operand orders, i.e. the dA1- and
    the dW2-shaped product; B = 2048: the four-wave form of the kernel) — and the way out of a wait that lasts: the threads of a
    product whose records nobody folds poll for 20 us, then fold their columns themselves — the same bits as the folded product — and
    count the event in the pinned host word (no expired wait may turn into a quiet wrong number; a GPU shared by several
    processes can keep a folding workgroup queued behind another process's waiting blocks)."""
    from robotic_manipulator_rloa_amd import _lib
    rng = np.random.default_rng(B)
    H, N, rows = 256, 64, 16
    N1 = 256 if B == 2048 else N                         # (B = 2048: 648 blocks > 512 -> gemm_bundle_kernel<256, 128>)
    npb = B // rows
    dy = rng.standard_normal((B, H)) * (rng.random((B, H)) > 0.4)
    z = rng.standard_normal((B, H)) * 1.5 + 0.3
    W = rng.standard_normal((H, N1))                     # k-major B operand of the dA1-shaped product: [K = H][N]
    A1 = rng.standard_normal((B, N))                     # k-major B operand of the dW2-shaped product: [K = B][N]
    gamma = rng.standard_normal(H) + 1.5
    mean, var = z.mean(0), z.var(0)
    invstd = 1.0 / np.sqrt(var + 1e-5)
    xhat = (z - mean) * invstd
    sdy, sdx = dy.sum(0), (dy * xhat).sum(0)
    k1 = gamma * invstd
    dz = k1 * dy - k1 * (sdy / B) - (z - mean) * (invstd * k1 * (sdx / B))
    parts = np.stack([np.stack([dy[i * rows:(i + 1) * rows].sum(0), (dy * xhat)[i * rows:(i + 1) * rows].sum(0)], -1)
                      for i in range(npb)])             # [npb][H][2]
    t = dict(dy=dev(dy), z=dev(z), W=dev(W), A1=dev(A1), gamma=dev(gamma), mean=dev(mean), inv=dev(invstd), parts=dev(parts))
    dg, db = torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda")
    cst = torch.zeros(H, 4, device="cuda")
    epoch = torch.full((1,), 5, dtype=torch.int32, device="cuda")
    err = torch.zeros(8, dtype=torch.int64).pin_memory()
    pro = _lib.GemmBn2Bwd(t["z"].data_ptr(), t["parts"].data_ptr(), t["gamma"].data_ptr(), t["mean"].data_ptr(), t["inv"].data_ptr(),
                          dg.data_ptr(), db.data_ptr(), npb, B, H, cst.data_ptr(), epoch.data_ptr(), err.data_ptr())
    ks = max(1, B // 256)
    c1 = torch.full((B, N1), -7.0, device="cuda")
    slabs = torch.full((ks, H, N), -7.0, device="cuda")
    D = _lib.GemmDesc
    arr = (D * 2)(D(t["dy"].data_ptr(), t["W"].data_ptr(), c1.data_ptr(), None, B, N1, H, H, N1, N1, 0, 1, 1, 0, None, C.addressof(pro)),
                  D(t["dy"].data_ptr(), t["A1"].data_ptr(), slabs.data_ptr(), None, H, N, B, H, N, N, 1, 1, ks, H * N, None,
                    C.addressof(pro)))
    assert lib.naf_gemm_bundle(arr, 2, st()) == 0
    torch.cuda.synchronize()
    assert int(err[0]) == 0
    np.testing.assert_allclose(c1.cpu().numpy(), dz @ W, rtol=2e-4, atol=2e-4 * np.sqrt(H))
    np.testing.assert_allclose(slabs.sum(0).cpu().numpy(), dz.T @ A1, rtol=2e-4, atol=2e-4 * np.sqrt(B))
    np.testing.assert_allclose(dg.cpu().numpy(), sdx, rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(db.cpu().numpy(), sdy, rtol=1e-4, atol=1e-3)
    # a second product with records of its own that no workgroup folds (only the launch's first prologue is folded): its threads
    # wait 20 us, then fold for themselves — and arrive at the very bits of the folded product
    cst2 = torch.zeros(H, 4, device="cuda")
    pro2 = _lib.GemmBn2Bwd(t["z"].data_ptr(), t["parts"].data_ptr(), t["gamma"].data_ptr(), t["mean"].data_ptr(), t["inv"].data_ptr(),
                           dg.data_ptr(), db.data_ptr(), npb, B, H, cst2.data_ptr(), epoch.data_ptr(), err.data_ptr())
    c1b = torch.zeros(B, N1, device="cuda")
    c2 = torch.zeros(B, N1, device="cuda")
    arr2 = (D * 2)(D(t["dy"].data_ptr(), t["W"].data_ptr(), c1b.data_ptr(), None, B, N1, H, H, N1, N1, 0, 1, 1, 0, None, C.addressof(pro)),
                   D(t["dy"].data_ptr(), t["W"].data_ptr(), c2.data_ptr(), None, B, N1, H, H, N1, N1, 0, 1, 1, 0, None, C.addressof(pro2)))
    assert lib.naf_gemm_bundle(arr2, 2, st()) == 0
    torch.cuda.synchronize()
    assert int(err[0]) > 0                                # the fallbacks are counted on the host ...
    assert torch.equal(c1b, c1)                           # ... the folded product is what it was ...
    assert torch.equal(c2, c1)                            # ... and the self-folded one has the same bits
    assert (cst2 == 0).all()                              # (nobody published its records)


def test_replay_edge_cases(lib):
    """Boundary behaviour the reference has by construction: population == batch (random.sample returns a permutation),
    host add() beyond the pinned staging size, n = 0 appends, sample() on a too-small buffer."""
    from robotic_manipulator_rloa_amd.utils.replay_buffer import ReplayBuffer
    from synth_data import make_transitions
    S, A = 21, 6
    # population == batch: every stored transition exactly once
    buf = ReplayBuffer(64, 64, "cuda", 5, state_size=S, action_size=A)
    buf.add_rows_device(torch.arange(64 * 64, device="cuda", dtype=torch.float32).view(64, 64), 64)
    idx = torch.zeros(3, 64, dtype=torch.int32, device="cuda")
    buf.sample_indices(idx, 3)
    for b in range(3):
        assert sorted(idx[b].cpu().tolist()) == list(range(64))
    np.testing.assert_array_equal(idx.cpu().numpy(), O.replay_sample_indices(5, 0, 64, 64, 3, True))
    # n = 0 is a no-op
    buf.add_rows_device(torch.zeros(1, 64, device="cuda"), 0)
    assert len(buf) == 64 and int(buf.meta[2].item()) == 64
    # host add(): 2500 transitions through a 1024-row staging area into a 2000-row ring
    st_, ac, rw, ns, dn = make_transitions(2500, S, A, seed=1)
    hb = ReplayBuffer(2000, 32, "cuda", 0)
    with pytest.raises(ValueError):
        hb.add(st_[0], ac[0], float(rw[0]), ns[0], int(dn[0]))
        hb.sample()                                                   # 1 < batch_size: same error class as random.sample
    for i in range(1, 2500):
        hb.add(st_[i], ac[i], float(rw[i]), ns[i], int(dn[i]))
    assert len(hb) == 2000
    hb.flush()
    expect = O.pack_rows(st_, ac, rw, ns, dn, 64)[-2000:]
    out = torch.empty(2000, 64, device="cuda")
    hb.action_mode = 1
    hb.gather_rows(torch.arange(2000, dtype=torch.int32, device="cuda"), out, 2000)
    np.testing.assert_array_equal(out.cpu().numpy(), expect)          # FIFO: the newest 2000, oldest first
    s, a, r, s2, d = hb.sample()
    assert s.shape == (32, S) and a.dtype == torch.float32           # action_mode FLOAT keeps the continuous action
    # handle misuse is reported, not executed
    assert lib.naf_replay_add_batch(None, out.data_ptr(), 1, st()) == -2
    assert lib.naf_replay_add_batch(hb.handle, out.data_ptr(), 2001, st()) == -1


@pytest.mark.parametrize("B,A,H", [(256, 6, 256), (512, 7, 256), (100, 8, 128), (64, 1, 128)])
@pytest.mark.parametrize("mode", [0, 1])
def test_bn_relu_fwd_heads_partial_and_splitk_head(lib, B, A, H, mode):
    """S3: layer-2 BN+ReLU with the heads GEMM split over K inside it, and the head kernel that adds the slabs.
    BN outputs and statistics must equal naf_bn_relu_fwd_train bit for bit (same tile, same arithmetic); the summed
    slabs must equal A2 @ Wh^T (+ bias) to f32 rounding; Q / d_heads / loss must match the unsplit head on that sum."""
    rng = np.random.default_rng(B + A + H)
    T = A * (A + 1) // 2
    NH, NHP, HP = A + T + 1, (A + T + 1 + 15) // 16 * 16, H + 16
    g = dev(rng.standard_normal((2, B, H)) * 2)
    bias, gamma, beta = (dev(rng.standard_normal((2, H))) for _ in range(3))
    Wh = np.zeros((2, NHP, HP))
    Wh[:, :NH, :H + 1] = rng.standard_normal((2, NH, H + 1)) / 8.0
    Wh = dev(Wh)
    outs = []
    for which in range(2):
        rm, rv = torch.zeros(2, H, device="cuda"), torch.ones(2, H, device="cuda")
        out = torch.zeros(2, B, HP, device="cuda")
        out[:, :, H] = 1.0
        sm, si = torch.empty(2, H, device="cuda"), torch.empty(2, H, device="cuda")
        if which == 0:
            assert lib.naf_bn_relu_fwd_train(g.data_ptr(), B * H, H, bias.data_ptr(), gamma.data_ptr(), beta.data_ptr(), H,
                                             rm.data_ptr(), rv.data_ptr(), H, out.data_ptr(), B * HP, HP, sm.data_ptr(),
                                             si.data_ptr(), B, H, 2, 0.1, 1e-5, st()) == 0
        else:
            hp = torch.full((H // 8, B * NHP + 64), 7.0, device="cuda")        # slabs padded by 64 floats
            vp = torch.full((H // 8, B), 7.0, device="cuda")
            assert lib.naf_bn_relu_fwd_heads_partial(
                g.data_ptr(), B * H, H, bias.data_ptr(), gamma.data_ptr(), beta.data_ptr(), H, rm.data_ptr(), rv.data_ptr(),
                H, out.data_ptr(), B * HP, HP, sm.data_ptr(), si.data_ptr(), Wh.data_ptr(), NHP * HP, HP, NHP, A + T,
                hp.data_ptr(), B * NHP + 64, vp.data_ptr(), B, H, 0.1, 1e-5, st()) == 0
        torch.cuda.synchronize()
        outs.append((out, rm, rv, sm, si))
    for a, b in zip(outs[0], outs[1]):     # same tile and arithmetic; the compiler may contract an fma differently
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-6, atol=1e-6)
    assert (hp[:, B * NHP:] == 7.0).all()                                # nothing written into the padding
    slab_stride, hp = B * NHP + 64, hp[:, :B * NHP].reshape(H // 8, B, NHP)
    hp_flat = torch.zeros(H // 8, slab_stride, device="cuda")
    hp_flat[:, :B * NHP] = hp.reshape(H // 8, -1)
    A2 = outs[1][0]
    want = (A2.double() @ Wh.double().transpose(1, 2))                   # [2, B, NHP] incl. the ones column = bias
    np.testing.assert_allclose(hp.double().sum(0).cpu().numpy(), want[0].cpu().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(vp.double().sum(0).cpu().numpy(), want[1, :, A + T].cpu().numpy(), rtol=1e-4, atol=1e-4)
    # the head on the slabs == the head on their (index-ordered f32) sum
    heads = hp[0].clone()
    vn = vp[0].clone()
    for w in range(1, H // 8):
        heads += hp[w]
        vn += vp[w]
    u = dev(np.trunc(rng.uniform(-1.5, 1.5, (B, A))))
    r = dev(rng.uniform(-1.5, 0, B))
    res = []
    for split in (False, True):
        q, dH = torch.empty(B, device="cuda"), torch.empty(B, NHP, device="cuda")
        lp = torch.zeros((B + 7) // 8, device="cuda")
        if split:
            assert lib.naf_head_fwd_bwd_mse_splitk(hp_flat.data_ptr(), slab_stride, vp.data_ptr(), H // 8, NHP, u.data_ptr(), A, r.data_ptr(), 1,
                                                   0.99, q.data_ptr(), dH.data_ptr(), lp.data_ptr(), B, A, mode, st()) == 0
        else:
            assert lib.naf_head_fwd_bwd_mse(heads.data_ptr(), NHP, u.data_ptr(), A, r.data_ptr(), 1, vn.data_ptr(), 1, 0.99,
                                            q.data_ptr(), dH.data_ptr(), lp.data_ptr(), B, A, mode, st()) == 0
        torch.cuda.synchronize()
        res.append((q, dH, lp))
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)
    assert lib.naf_bn_relu_fwd_heads_partial(
        g.data_ptr(), B * H, H, bias.data_ptr(), gamma.data_ptr(), beta.data_ptr(), H, rm.data_ptr(), rv.data_ptr(), H,
        out.data_ptr(), B * HP, HP, sm.data_ptr(), si.data_ptr(), Wh.data_ptr(), NHP * HP, HP, 40, A + T, hp_flat.data_ptr(),
        slab_stride, vp.data_ptr(), B, H, 0.1, 1e-5, st()) == -1


@pytest.mark.gpu
def test_lane_ops_against_their_definition(tmp_path):
    """csrc/common.h's cross-lane helpers (DPP adds, v_permlane16/32_swap sums) lane by lane on exact integers: the probe
    that caught hipcc folding the two results of a swap into one (every reduction built on them would be wrong)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not on this box")
    exe = str(tmp_path / "lane_ops")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "benchmarks", "probe", "lane_ops_probe.hip")
    inc = os.path.join(root, "robotic_manipulator_rloa_amd", "csrc")
    for opt in ("-O2", "-O3"):                  # the library is built -O3
        r = subprocess.run([hipcc, "--offload-arch=gfx950", opt, "-w", "-I", inc, src, "-o", exe], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and r.stdout.count(" ok ") == 8, r.stdout
