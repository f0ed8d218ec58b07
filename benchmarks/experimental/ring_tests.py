"""The tests of round 4's LDS-DMA ring form of the backward GEMM launch (gemm_ring.h in this directory), as they stood in
tests/test_kernels_gpu.py and tests/test_learner_gpu.py while the form was compiled into the library (`naf_gemm_bundle_ex(form = 2)`,
`NAF_GEMM_FORM=2`; git history of round 4). Not collected by pytest from here. To run them again: restore the `#include
"gemm_ring.h"` + the `form` argument in csrc/gemm_bundle.hip and `Learner.gemm_ring` (commit 94303aa has both), then copy these
functions back into the two test files."""

# ---- tests/test_kernels_gpu.py -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,KP", [(1024, 24), (2048, 32), (1536, 24)])
def test_gemm_ring_form_of_the_backward_bundle(lib, B, KP):
    """naf_gemm_bundle_ex(form = 2), csrc/gemm_ring.h: the backward GEMMs of a large-batch update (autograd of
    naf_neural_network.py:76-87) on 64 x 64 tiles fed by an LDS-DMA ring, as the row-split chain launches them — dA1 = dZ2 W2 with the
    BatchNorm-backward prologue on its k-contiguous A panels and the layer-1 backward pass as its epilogue (block sums + P = dY^T X
    per 32 rows, from the kept xhat), dW2 = dZ2^T A1 with the prologue on k-major panels and split K, dWh = dH^T A2 (M = 32 < tile,
    N = 272: edge tiles) — against numpy in double, against the 32 x 32 form (form = 1) on the same inputs, and bitwise against
    itself (fixed summation order)."""
    from robotic_manipulator_rloa_amd import _lib
    rng = np.random.default_rng(B + KP)
    H, NHP, HP, rows = 256, 32, 272, 16
    S = 21 if KP == 24 else 26
    ldx = 52 if KP == 24 else 56
    npb = B // rows
    dy = rng.standard_normal((B, H)) * (rng.random((B, H)) > 0.4)
    z = rng.standard_normal((B, H)) * 1.5 + 0.3
    W2 = rng.standard_normal((H, H)) * 0.1               # [K = H][N = H]: k-major B of dA1
    A1 = rng.standard_normal((B, H))
    dH = rng.standard_normal((B, NHP))
    A2 = rng.standard_normal((B, HP))
    X = rng.standard_normal((B, ldx))
    xhat1 = rng.standard_normal((B, H))
    g1, be1 = rng.standard_normal(H) + 1.0, rng.standard_normal(H) * 0.5
    gamma = rng.standard_normal(H) + 1.5
    mean, var = z.mean(0), z.var(0)
    invstd = 1.0 / np.sqrt(var + 1e-5)
    xhat = (z - mean) * invstd
    sdy, sdx = dy.sum(0), (dy * xhat).sum(0)
    k1 = gamma * invstd
    dz = k1 * dy - k1 * (sdy / B) - (z - mean) * (invstd * k1 * (sdx / B))
    parts = np.stack([np.stack([dy[i * rows:(i + 1) * rows].sum(0), (dy * xhat)[i * rows:(i + 1) * rows].sum(0)], -1)
                      for i in range(npb)])
    dA1 = dz @ W2
    dy1 = np.where(xhat1 * g1 + be1 > 0, dA1, 0.0)
    want_part = np.stack([np.stack([dy1[i * 32:(i + 1) * 32].sum(0), (dy1 * xhat1)[i * 32:(i + 1) * 32].sum(0)], -1)
                          for i in range(B // 32)])                                       # [B/32][H][2]
    want_P = np.stack([dy1[i * 32:(i + 1) * 32].T @ X[i * 32:(i + 1) * 32, :KP] for i in range(B // 32)])   # [B/32][H][KP]
    t = dict(dy=dev(dy), z=dev(z), W2=dev(W2), A1=dev(A1), dH=dev(dH), A2=dev(A2), X=dev(X), xh=dev(xhat1), g1=dev(g1), be1=dev(be1),
             gamma=dev(gamma), mean=dev(mean), inv=dev(invstd), parts=dev(parts))
    ks = max(d for d in range(1, 9) if B % d == 0 and (B // d) % 32 == 0 and B // d >= 256)
    D = _lib.GemmDesc

    def run(form, with_c):
        dg, db = torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda")
        cst = torch.zeros(H, 4, device="cuda")
        epoch = torch.full((1,), 11 + form, dtype=torch.int32, device="cuda")
        err = torch.zeros(8, dtype=torch.int64).pin_memory()
        pro = _lib.GemmBn2Bwd(t["z"].data_ptr(), t["parts"].data_ptr(), t["gamma"].data_ptr(), t["mean"].data_ptr(), t["inv"].data_ptr(),
                              dg.data_ptr(), db.data_ptr(), npb, B, H, cst.data_ptr(), epoch.data_ptr(), err.data_ptr())
        part = torch.full((B // 32, H, 2), -7.0, device="cuda")
        P = torch.full((B // 32, H, KP), -7.0, device="cuda")
        epi = _lib.GemmL1Bwd(t["X"].data_ptr(), t["W2"].data_ptr(), t["g1"].data_ptr(), t["A1"].data_ptr(), t["mean"].data_ptr(),
                             t["inv"].data_ptr(), part.data_ptr(), P.data_ptr(), ldx, S, KP, H, t["xh"].data_ptr(), t["g1"].data_ptr(),
                             t["be1"].data_ptr())
        c1 = torch.full((B, H), -7.0, device="cuda")
        s2 = torch.full((ks, H, H), -7.0, device="cuda")
        sh = torch.full((ks, NHP, HP), -7.0, device="cuda")
        arr = (D * 3)(
            D(t["dy"].data_ptr(), t["W2"].data_ptr(), c1.data_ptr() if with_c else None, None, B, H, H, H, H, H, 0, 1, 1, 0,
              C.addressof(epi), C.addressof(pro)),
            D(t["dy"].data_ptr(), t["A1"].data_ptr(), s2.data_ptr(), None, H, H, B, H, H, H, 1, 1, ks, H * H, None, C.addressof(pro)),
            D(t["dH"].data_ptr(), t["A2"].data_ptr(), sh.data_ptr(), None, NHP, HP, B, NHP, HP, HP, 1, 1, ks, NHP * HP))
        assert lib.naf_gemm_bundle_ex(arr, 3, form, st()) == 0
        torch.cuda.synchronize()
        assert int(err[0]) == 0
        return c1, s2, sh, part, P, dg, db

    c1, s2, sh, part, P, dg, db = run(2, True)
    tol = dict(rtol=2e-4, atol=2e-4 * np.sqrt(H))
    np.testing.assert_allclose(c1.cpu().numpy(), dA1, **tol)
    np.testing.assert_allclose(s2.sum(0).cpu().numpy(), dz.T @ A1, rtol=2e-4, atol=2e-4 * np.sqrt(B))
    np.testing.assert_allclose(sh.sum(0).cpu().numpy(), dH.T @ A2, rtol=2e-4, atol=2e-4 * np.sqrt(B))
    np.testing.assert_allclose(dg.cpu().numpy(), sdx, rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(db.cpu().numpy(), sdy, rtol=1e-4, atol=1e-3)
    # the epilogue's ReLU decision sits on xhat g + b > 0 of given inputs: identical on both sides; its sums run over 32 rows
    np.testing.assert_allclose(part.cpu().numpy(), want_part, rtol=1e-3, atol=2e-3 * np.sqrt(H))
    np.testing.assert_allclose(P.cpu().numpy()[:, :, :S], want_P[:, :, :S], rtol=1e-3, atol=2e-3 * np.sqrt(H))
    # against the 32 x 32 form, and without the dA1 store (the chain never stores it)
    o1 = run(1, True)
    for a, b_ in zip((c1, s2.sum(0), sh.sum(0), part, P[:, :, :S]), (o1[0], o1[1].sum(0), o1[2].sum(0), o1[3], o1[4][:, :, :S])):
        np.testing.assert_allclose(a.cpu().numpy(), b_.cpu().numpy(), rtol=1e-3, atol=1e-3)
    again = run(2, False)
    assert (again[0] == -7.0).all()
    for a, b_ in zip((s2, sh, part, P), again[1:5]):
        assert torch.equal(a, b_)                             # bitwise reproducible
    # a bundle that does not fit the ring form is refused by form = 2 and taken by the library's choice
    bad = (D * 1)(D(t["dy"].data_ptr(), t["A1"].data_ptr(), s2.data_ptr(), None, H, H, B, H, H, H, 1, 0, 1, 0))
    assert lib.naf_gemm_bundle_ex(bad, 1, 2, st()) == -1



# ---- tests/test_learner_gpu.py -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["xarm1024", "panda2048"])
def test_learn_with_the_ring_form_of_the_backward_gemms_g3(tag, monkeypatch):
    """NAF_GEMM_FORM=2: the row-split chain with its backward GEMM launch on the LDS-DMA ring (csrc/gemm_ring.h, round 4's
    experiment — callable, not the default) against the unmodified reference's learn() at configs[3] / [4]'s batch sizes: the five
    losses, and bitwise the same run twice."""
    monkeypatch.delenv("NAF_FUSE", raising=False)
    monkeypatch.setenv("NAF_GEMM_FORM", "2")
    from synth_data import make_transitions
    g, main0, target0 = g3_case(tag)
    S, A, B = [int(x) for x in g[f"{tag}/dims"]]
    st, ac, rw, ns, dn = make_transitions(5 * B, S, A, seed=7)
    out = []
    for rep in range(2):
        L = make_learner(S, A, B, main0, target0)
        assert L.fuse == ROWS and L.gemm_ring
        rows = rows_device(L, st, ac, rw, ns, dn)
        lp = torch.zeros(5, L.n_loss_wg, device="cuda")
        for k in range(5):
            L.learn_rows(rows[k * B:(k + 1) * B], lp[k])
        torch.cuda.synchronize()
        out.append((lp.sum(1).cpu().numpy(), L.theta2.clone()))
    np.testing.assert_allclose(out[0][0], g[f"{tag}/losses5"], rtol=5e-3)
    assert np.array_equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])


