"""
Generates tests/golden/*.npz by importing and running the UNMODIFIED reference
(/root/reference, JavierMtz5/robotic_manipulator_rloa) on CPU. Runs only in the build container — the
GPU box has no /root/reference; the fixtures (numbers only) are what travels.

Usage (from anywhere):  python tests/golden/make_golden.py [--g5-updates N]

The reference imports pybullet at package import; pybullet is not installed, so two stub modules are put
on sys.path first (a MagicMock-attribute `pybullet`, a one-function `pybullet_data`). Import side effects
(`training_logs.log`, `checkpoints/`) land in a scratch CWD under /tmp.
"""
import argparse
import os
import sys
import tempfile
import types
from unittest.mock import MagicMock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from synth_data import make_transitions, batch_indices  # noqa: E402

REF = "/root/reference"


def _install_stubs():
    pb = types.ModuleType("pybullet")
    pb.error = type("error", (Exception,), {})
    pb.GUI, pb.DIRECT, pb.POSITION_CONTROL, pb.VELOCITY_CONTROL = 1, 2, 2, 0
    pb.__getattr__ = lambda name: MagicMock()
    pbd = types.ModuleType("pybullet_data")
    pbd.getDataPath = lambda: "/nonexistent/pybullet_data"
    sys.modules["pybullet"] = pb
    sys.modules["pybullet_data"] = pbd
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True


def sd_np(sd):
    return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def flat(prefix, d):
    return {f"{prefix}/{k}": v for k, v in d.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--g5-updates", type=int, default=20000)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    only = set(args.only.split(",")) if args.only else None

    _install_stubs()
    scratch = tempfile.mkdtemp(prefix="naf_golden_")
    os.chdir(scratch)
    import random
    import torch
    import torch.nn.functional as F
    from robotic_manipulator_rloa.naf_components import naf_algorithm as ref_alg
    from robotic_manipulator_rloa.naf_components.naf_algorithm import NAFAgent
    from robotic_manipulator_rloa.naf_components.naf_neural_network import NAF
    from robotic_manipulator_rloa.utils.replay_buffer import ReplayBuffer

    torch.set_num_threads(8)
    cpu = torch.device("cpu")
    meta = {"torch": torch.__version__, "numpy": np.__version__}

    def want(name):
        return only is None or name in only

    # ---------------- G1: the reference's own known-answer test -------------------------------------
    if want("g1"):
        net = NAF(10, 5, 256, 0, cpu)
        s = torch.arange(20).reshape(2, 10).float()
        a = torch.tensor([[0, 1, 2, 3, 4], [10, 11, 12, 13, 14]]).long()
        sd0 = sd_np(net.state_dict())
        _, q, v = net(s, a)
        torch.set_default_dtype(torch.float64)  # the reference allocates L with the default dtype (:95)
        net64 = NAF(10, 5, 256, 0, cpu)
        net64.load_state_dict({k: (v.double() if v.dtype.is_floating_point else v) for k, v in net.state_dict().items()})
        net64.train()
        _, q64, v64 = net64(s.double(), a)
        torch.set_default_dtype(torch.float32)
        np.savez_compressed(os.path.join(HERE, "g1_known_answer.npz"),
                            states=s.numpy(), actions=a.numpy(), q=q.detach().numpy(), v=v.detach().numpy(),
                            q_f64=q64.detach().numpy(), v_f64=v64.detach().numpy(),
                            q_test_literal=np.array([[-35.50931930541992], [-638.494873046875]]),
                            v_test_literal=np.array([[0.5665180683135986], [-0.08311141282320023]]),
                            **flat("sd", sd0))
        print("g1", q.detach().numpy().ravel(), v.detach().numpy().ravel())

    # ---------------- G2: head only, through the reference forward + autograd -----------------------
    if want("g2"):
        out = {}
        for A in (5, 6, 7):
            S = 9 + 2 * A
            for B in (2, 256):
                for tag, seed in (("rand", 1), ("wide", 2)):
                    torch.manual_seed(100 + seed)
                    net = NAF(S, A, 256, seed, cpu)
                    if tag == "wide":  # spread the pre-activations so tanh/exp are exercised off the linear zone
                        with torch.no_grad():
                            for lin in (net.action_values, net.matrix_entries, net.value):
                                lin.weight.mul_(8.0)
                                lin.bias.uniform_(-1.0, 1.0)
                    rng = np.random.Generator(np.random.PCG64(1000 * A + B + seed))
                    x = torch.from_numpy(rng.standard_normal((B, S)).astype(np.float32))
                    u_float = rng.uniform(-1, 1, (B, A)).astype(np.float32)
                    u_float[rng.random((B, A)) < 0.1] = 1.0
                    u_long = torch.from_numpy(u_float).long()
                    cap = {}

                    def mk(name):
                        def hook(mod, inp, outp):
                            outp.retain_grad()
                            cap[name] = outp
                        return hook
                    hs = [net.action_values.register_forward_hook(mk("mu_pre")),
                          net.matrix_entries.register_forward_hook(mk("l_pre")),
                          net.value.register_forward_hook(mk("V"))]
                    _, q, v = net(x, u_long)
                    wq = torch.from_numpy(rng.standard_normal((B, 1)).astype(np.float32))
                    (q * wq).sum().backward()
                    for h in hs:
                        h.remove()
                    key = f"A{A}_B{B}_{tag}"
                    out[f"{key}/mu_pre"] = cap["mu_pre"].detach().numpy().copy()
                    out[f"{key}/l_pre"] = cap["l_pre"].detach().numpy().copy()
                    out[f"{key}/V"] = cap["V"].detach().numpy().copy()
                    out[f"{key}/u_float"] = u_float
                    out[f"{key}/u_trunc"] = u_long.numpy().astype(np.float32)
                    out[f"{key}/q"] = q.detach().numpy().copy()
                    out[f"{key}/dq"] = wq.numpy()
                    out[f"{key}/d_mu_pre"] = cap["mu_pre"].grad.numpy().copy()
                    out[f"{key}/d_l_pre"] = cap["l_pre"].grad.numpy().copy()
                    out[f"{key}/d_V"] = cap["V"].grad.numpy().copy()
        np.savez_compressed(os.path.join(HERE, "g2_head.npz"), **out)
        print("g2", len(out), "arrays")

    # ---------------- G3: one full learn() + 5-step trace --------------------------------------------
    def g3_case(S, A, B, tag, compact, out, H=256):
        """compact: the initial weights are those of a smaller case with the same (S, A) and seed (asserted by the caller),
        so only the results of the update are stored."""
        agent = NAFAgent(object(), S, A, H, B, 100000, 1e-3, 1e-3, 0.99, 1, 1, 500, cpu, 0)
        st, ac, rw, ns, dn = make_transitions(5 * B, S, A, seed=7)
        losses, pre_clip = [], {}
        real_mse, real_clip = ref_alg.F.mse_loss, ref_alg.clip_grad_norm_

        def tap_mse(a_, b_):
            l_ = real_mse(a_, b_)
            losses.append(float(l_.detach()))
            tap_mse.q, tap_mse.y = a_.detach().numpy().copy(), b_.detach().numpy().copy()
            return l_

        def tap_clip(params, max_norm):
            params = list(params)
            if "grads" not in pre_clip:
                names = [n for n, _ in agent.qnetwork_main.named_parameters()]
                pre_clip["grads"] = {n: p.grad.detach().numpy().copy() for n, p in zip(names, params)}
            tn = real_clip(params, max_norm)
            pre_clip.setdefault("norm", float(tn))
            return tn

        ref_alg.F.mse_loss = tap_mse
        ref_alg.clip_grad_norm_ = tap_clip
        main0 = sd_np(agent.qnetwork_main.state_dict())
        if not compact:
            out.update(flat(f"{tag}/main0", main0))
            out.update(flat(f"{tag}/target0", sd_np(agent.qnetwork_target.state_dict())))
        for k in range(5):
            sl = slice(k * B, (k + 1) * B)
            ex = (torch.from_numpy(st[sl]), torch.from_numpy(ac[sl]).long(), torch.from_numpy(rw[sl, None]),
                  torch.from_numpy(ns[sl]), torch.from_numpy(dn[sl, None]))
            agent.learn(ex)
            if k == 0:
                out[f"{tag}/q1"], out[f"{tag}/y1"] = tap_mse.q, tap_mse.y
                out.update(flat(f"{tag}/grads1", pre_clip["grads"]))
                out[f"{tag}/grad_norm1"] = np.array(pre_clip["norm"])
                out.update(flat(f"{tag}/main1", sd_np(agent.qnetwork_main.state_dict())))
                out.update(flat(f"{tag}/target1", sd_np(agent.qnetwork_target.state_dict())))
                if not compact:
                    opt = agent.optimizer.state_dict()["state"]
                    names = [n for n, _ in agent.qnetwork_main.named_parameters()]
                    for i, n in enumerate(names):
                        out[f"{tag}/adam_m1/{n}"] = opt[i]["exp_avg"].numpy().copy()
                        out[f"{tag}/adam_v1/{n}"] = opt[i]["exp_avg_sq"].numpy().copy()
        ref_alg.F.mse_loss, ref_alg.clip_grad_norm_ = real_mse, real_clip
        out[f"{tag}/losses5"] = np.array(losses)
        if not compact:
            out.update(flat(f"{tag}/main5", sd_np(agent.qnetwork_main.state_dict())))
            out.update(flat(f"{tag}/target5", sd_np(agent.qnetwork_target.state_dict())))
        out[f"{tag}/dims"] = np.array([S, A, B]) if H == 256 else np.array([S, A, B, H])
        print("g3", tag, losses)
        return main0

    if want("g3"):
        out = {}
        for (S, A, B, tag) in ((21, 6, 256, "kuka"), (23, 7, 64, "panda")):
            g3_case(S, A, B, tag, False, out)
        np.savez_compressed(os.path.join(HERE, "g3_learn.npz"), **out)

    # ---------------- G3 at the network of the reference's own agent test: NAF(10, 5, 128), batch 64 -----------------------------
    # (tests/robotic_manipulator_rloa/naf_components/test_naf_algorithm.py:74 builds NAF(10, 5, 128, 0, 'cpu'); layer_size is a
    #  hyper-parameter of the framework, rl_framework.py:68-74)
    if want("g3h128"):
        out = {}
        g3_case(10, 5, 64, "h128", False, out, H=128)
        np.savez_compressed(os.path.join(HERE, "g3_learn_h128.npz"), **out)

    # ---------------- G3 at layer sizes beyond 256 (round 6: the row-split chain runs widths up to 512 on two 256-column halves; 384 is
    # stored zero-padded to 512): NAF(21, 6, 512) at batch 256 and NAF(21, 6, 384) at batch 64. SLIM: the 512 x 512 matrix is kept
    # as its first 16 rows (main0 / main1 / target1 / grads1) + a (sum, sum of squares) pair per full tensor — the initial weights are
    # the constructor's at seed 0, which the build reproduces (reference_init_state_dict; the test checks slices and sums).  --only g3wide
    if only is not None and "g3wide" in only:
        out = {}
        for (S, A, B, H, tag) in ((21, 6, 256, 512, "h512"), (21, 6, 64, 384, "h384")):
            full = {}
            main0 = g3_case(S, A, B, tag, True, full, H=H)
            full.update(flat(f"{tag}/main0", main0))
            for k_, v_ in full.items():
                v_ = np.asarray(v_)
                if v_.ndim == 2 and v_.shape[0] == H and v_.shape[1] == H:
                    out[k_ + "@rows16"] = v_[:16].copy()
                    out[k_ + "@sums"] = np.array([v_.astype(np.float64).sum(), (v_.astype(np.float64) ** 2).sum()])
                else:
                    out[k_] = v_
        np.savez_compressed(os.path.join(HERE, "g3_learn_wide.npz"), **out)

    # ---------------- G3 at 9 and 11 joints (round 6: the row-split chain's fused layer-2 launch holds one sample per 16-lane group there):
    # NAF(27, 9, 256) at batch 256 and NAF(31, 11, 256) at batch 64 — the reference's state is 9 + 2 A floats (environment.py:261).
    # SLIM like g3wide: the 256 x 256 matrix as its first 16 rows + (sum, sum of squares); the initial weights are the constructor's at
    # seed 0 (reference_init_state_dict reproduces them; the tests check slices and sums).  --only g3joints
    if only is not None and "g3joints" in only:
        out = {}
        for (S, A, B, H, tag) in ((27, 9, 256, 256, "j9"), (31, 11, 64, 256, "j11")):
            full = {}
            main0 = g3_case(S, A, B, tag, True, full, H=H)
            full.update(flat(f"{tag}/main0", main0))
            full[f"{tag}/dims"] = np.array([S, A, B, H])
            for k_, v_ in full.items():
                v_ = np.asarray(v_)
                if v_.ndim == 2 and v_.shape[0] == H and v_.shape[1] == H:
                    out[k_ + "@rows16"] = v_[:16].copy()
                    out[k_ + "@sums"] = np.array([v_.astype(np.float64).sum(), (v_.astype(np.float64) ** 2).sum()])
                else:
                    out[k_] = v_
        np.savez_compressed(os.path.join(HERE, "g3_learn_joints.npz"), **out)

    # ---------------- G3 at the batch sizes of BASELINE configs[3] and [4] (one reference learn() trace each) ---------
    if want("g3big"):
        out = {}
        small = np.load(os.path.join(HERE, "g3_learn.npz"))
        for (S, A, B, tag, init_of) in ((21, 6, 1024, "xarm1024", "kuka"), (23, 7, 2048, "panda2048", "panda")):
            main0 = g3_case(S, A, B, tag, True, out)
            for k_, v_ in main0.items():       # same (S, A, seed) => same initial weights as the small case: not stored twice
                assert np.array_equal(v_, small[f"{init_of}/main0/{k_}"]), k_
            out[f"{tag}/init_of"] = np.array(init_of)
        np.savez_compressed(os.path.join(HERE, "g3_learn_big.npz"), **out)

    # ---------------- G3 at the batch sizes the reference itself defaults to: configs[0]'s 64 and HyperParameters' 128
    # (rl_framework.py:68-74), KUKA 21/6, and 192 = three 64-row blocks (the row-split chain's first odd block count) ----
    if want("g3def"):
        out = {}
        small = np.load(os.path.join(HERE, "g3_learn.npz"))
        for (S, A, B, tag, init_of) in ((21, 6, 64, "kuka64", "kuka"), (21, 6, 128, "kuka128", "kuka"),
                                        (21, 6, 192, "kuka192", "kuka")):
            main0 = g3_case(S, A, B, tag, True, out)
            for k_, v_ in main0.items():
                assert np.array_equal(v_, small[f"{init_of}/main0/{k_}"]), k_
            out[f"{tag}/init_of"] = np.array(init_of)
        np.savez_compressed(os.path.join(HERE, "g3_learn_default.npz"), **out)

    # ---------------- G4: replay contract ------------------------------------------------------------
    if want("g4"):
        S, A, cap, B = 21, 6, 300, 64
        buf = ReplayBuffer(cap, B, cpu, 0)
        st, ac, rw, ns, dn = make_transitions(500, S, A, seed=11)
        st64 = st.astype(np.float64)  # env states are float64 (environment.py:451)
        for i in range(500):
            s = st64[i].copy()
            s[0] = float(i)  # tag: transition id in state[0]
            buf.add(s, ac[i], float(rw[i]), ns[i].astype(np.float64), int(dn[i]))
        assert len(buf) == cap
        ids_in_order = np.array([e.state[0] for e in buf.memory])
        random.seed(0)
        draws = []
        outs = None
        for k in range(3):
            o = buf.sample()
            draws.append(o[0][:, 0].numpy().copy())
            if k == 0:
                outs = o
        # does random.sample(deque) draw the same positions as random.sample(range(len))? pin it
        random.seed(0)
        pos = [random.sample(range(len(buf)), B) for _ in range(3)]
        np.savez_compressed(os.path.join(HERE, "g4_replay.npz"),
                            dims=np.array([S, A, cap, B]), ids_in_order=ids_in_order,
                            sampled_ids=np.stack(draws), positions_from_range=np.array(pos),
                            s=outs[0].numpy(), a=outs[1].numpy(), r=outs[2].numpy(), s2=outs[3].numpy(), d=outs[4].numpy(),
                            dtypes=np.array([str(t.dtype) for t in outs]))
        print("g4", [str(t.dtype) for t in outs], [tuple(t.shape) for t in outs])
        # step gating trace (naf_algorithm.py:144-156): at which timesteps does learn fire?
        gate = {}
        for (uf, nu) in ((1, 1), (4, 1), (4, 2), (3, 2)):
            agent = NAFAgent(object(), S, A, 256, 8, 1000, 1e-3, 1e-3, 0.99, uf, nu, 500, cpu, 0)
            calls = []
            agent.learn = lambda ex, _c=calls: _c.append(1)
            fired = []
            for t in range(40):
                before = len(calls)
                agent.step(st64[t], ac[t], float(rw[t]), st64[t + 1], 0)
                fired.append(len(calls) - before)
            gate[f"uf{uf}_nu{nu}"] = np.array(fired)
        np.savez_compressed(os.path.join(HERE, "g4_gating.npz"), **gate)

    # ---------------- G6: act() in eval mode with the shipped demo weights ---------------------------
    if want("g6"):
        out = {}
        for name in ("kuka", "xarm6"):
            path = os.path.join(REF, "robotic_manipulator_rloa/naf_components/demo_weights", f"weights_{name}.p")
            sd = torch.load(path, map_location="cpu")
            net = NAF(21, 6, 256, 0, cpu)
            net.load_state_dict(sd)
            net.eval()
            rng = np.random.Generator(np.random.PCG64(5))
            x = rng.standard_normal((64, 21)).astype(np.float32)
            with torch.no_grad():
                xt = torch.from_numpy(x)
                h = torch.relu(net.bn1(net.input_layer(xt)))
                h = torch.relu(net.bn2(net.hidden_layer(h)))
                mu = torch.tanh(net.action_values(h))
                l_pre = net.matrix_entries(h)
                V = net.value(h)
                torch.manual_seed(0)
                acts = torch.stack([net(xt)[0] for _ in range(256)])  # noise samples for distributional check
            out[f"{name}/x"] = x
            out[f"{name}/mu"] = mu.numpy()
            out[f"{name}/l_pre"] = l_pre.numpy()
            out[f"{name}/V"] = V.numpy()
            out[f"{name}/act_mean"] = acts.mean(0).numpy()
            out[f"{name}/act_std"] = acts.std(0).numpy()
            # a small slice of the weights so the test can rebuild the net: all of it (328 KB) is data under MIT
            out.update(flat(f"{name}/sd", sd_np(sd)))
        np.savez_compressed(os.path.join(HERE, "g6_act.npz"), **out)
        print("g6 done")

    # ---------------- G5: teacher-forced loss curve --------------------------------------------------
    if want("g5"):
        S, A, B, NROWS = 21, 6, 256, 200000
        n_upd = args.g5_updates
        st, ac, rw, ns, dn = make_transitions(NROWS, S, A, seed=2024, rare_events=False, structured_reward=True)
        idx = batch_indices(NROWS, B, n_upd, seed=99)
        agent = NAFAgent(object(), S, A, 256, B, NROWS, 1e-3, 1e-3, 0.99, 1, 1, 500, cpu, 0)
        losses = []
        real_mse = ref_alg.F.mse_loss

        def tap(a_, b_):
            l_ = real_mse(a_, b_)
            losses.append(float(l_.detach()))
            return l_
        ref_alg.F.mse_loss = tap
        tst, tac, trw, tns, tdn = (torch.from_numpy(st), torch.from_numpy(ac).long(), torch.from_numpy(rw[:, None]),
                                   torch.from_numpy(ns), torch.from_numpy(dn[:, None]))
        import time
        t0 = time.time()
        checks = {}
        for k in range(n_upd):
            ii = torch.from_numpy(idx[k].astype(np.int64))
            agent.learn((tst[ii], tac[ii], trw[ii], tns[ii], tdn[ii]))
            if (k + 1) in (1000, 10000, 100000, n_upd):
                sd = agent.qnetwork_main.state_dict()
                checks[f"theta_l2_{k + 1}"] = np.array(float(sum((v.double() ** 2).sum() for n, v in sd.items()
                                                                if v.dtype.is_floating_point and 'running' not in n) ** 0.5))
        ref_alg.F.mse_loss = real_mse
        dt = time.time() - t0
        print(f"g5: {n_upd} updates in {dt:.1f}s = {n_upd / dt:.1f} updates/s (reference learn(), CPU, "
              f"{torch.get_num_threads()} threads)")
        np.savez_compressed(os.path.join(HERE, "g5_curve.npz"), losses=np.array(losses, dtype=np.float32),
                            dims=np.array([S, A, B, NROWS, n_upd]), data_seed=np.array(2024), idx_seed=np.array(99),
                            ref_updates_per_s=np.array(n_upd / dt), **checks,
                            **flat("main_end", sd_np(agent.qnetwork_main.state_dict())))

    # ---------------- G7: teacher-forced loss curves at batch sizes that are NOT whole 64-row blocks / lie beyond 2048 ------------
    # (round 6: the partial-block and beyond-2048 variants of the row-split kernels were held to the numpy oracle over thousands of
    #  updates, computed live in the GPU suite — now to the unmodified reference's learn() itself, as G5 holds the whole-block ones.
    #  Same rows, positions and initial weights as tests/test_learner_gpu.py::test_long_teacher_forced_run_*: 40000 rows of seed
    #  2024, batch_indices(seed 99), the constructor's own weights at seed 0 = G3's kuka/main0.)   --only g7
    if only is not None and "g7" in only:
        import time
        S, A, NROWS = 21, 6, 40000
        st, ac, rw, ns, dn = make_transitions(NROWS, S, A, seed=2024, rare_events=False, structured_reward=True)
        tst, tac, trw, tns, tdn = (torch.from_numpy(st), torch.from_numpy(ac).long(), torch.from_numpy(rw[:, None]),
                                   torch.from_numpy(ns), torch.from_numpy(dn[:, None]))
        out = {}
        for B, n_upd in ((100, 5000), (1000, 5000), (2560, 600), (4096, 2000)):
            idx = batch_indices(NROWS, B, n_upd, seed=99)
            agent = NAFAgent(object(), S, A, 256, B, NROWS, 1e-3, 1e-3, 0.99, 1, 1, 500, cpu, 0)
            losses = []
            real_mse = ref_alg.F.mse_loss

            def tap(a_, b_):
                l_ = real_mse(a_, b_)
                losses.append(float(l_.detach()))
                return l_
            ref_alg.F.mse_loss = tap
            t0 = time.time()
            for k in range(n_upd):
                ii = torch.from_numpy(idx[k].astype(np.int64))
                agent.learn((tst[ii], tac[ii], trw[ii], tns[ii], tdn[ii]))
            ref_alg.F.mse_loss = real_mse
            dt = time.time() - t0
            sd = agent.qnetwork_main.state_dict()
            out[f"b{B}/losses"] = np.array(losses, dtype=np.float32)
            out[f"b{B}/theta_l2"] = np.array(float(sum((v.double() ** 2).sum() for n, v in sd.items()
                                                       if v.dtype.is_floating_point and 'running' not in n) ** 0.5))
            out[f"b{B}/dims"] = np.array([S, A, B, NROWS, n_upd])
            print(f"g7: B = {B}: {n_upd} updates in {dt:.1f}s = {n_upd / dt:.1f} updates/s (reference learn(), CPU)", flush=True)
        np.savez_compressed(os.path.join(HERE, "g7_curves.npz"), data_seed=np.array(2024), idx_seed=np.array(99), **out)

    # ---------------- G8: teacher-forced loss curves at 9 and 11 joints and at layer size 512 (round 6: the shapes whose fused kernels are
    # new this round — one sample per 16-lane group in the fused layer-2 launch, two 256-column halves — held to the unmodified
    # reference's learn() over thousands of updates as G5 / G7 hold the others). Rows of seed 2024, batch_indices(seed 99), the
    # constructor's own weights at seed 0 (reference_init_state_dict reproduces them).   --only g8
    if only is not None and "g8" in only:
        import time
        NROWS = 40000
        out = {}
        for tag, S, A, H, B, n_upd in (("j9", 27, 9, 256, 256, 3000), ("j11", 31, 11, 256, 1000, 2000), ("h512", 21, 6, 512, 256, 2000),
                                       ("j10big", 29, 10, 256, 2560, 600)):
            st, ac, rw, ns, dn = make_transitions(NROWS, S, A, seed=2024, rare_events=False, structured_reward=True)
            tst, tac, trw, tns, tdn = (torch.from_numpy(st), torch.from_numpy(ac).long(), torch.from_numpy(rw[:, None]),
                                       torch.from_numpy(ns), torch.from_numpy(dn[:, None]))
            idx = batch_indices(NROWS, B, n_upd, seed=99)
            agent = NAFAgent(object(), S, A, H, B, NROWS, 1e-3, 1e-3, 0.99, 1, 1, 500, cpu, 0)
            losses = []
            real_mse = ref_alg.F.mse_loss

            def tap(a_, b_):
                l_ = real_mse(a_, b_)
                losses.append(float(l_.detach()))
                return l_
            ref_alg.F.mse_loss = tap
            t0 = time.time()
            for k in range(n_upd):
                ii = torch.from_numpy(idx[k].astype(np.int64))
                agent.learn((tst[ii], tac[ii], trw[ii], tns[ii], tdn[ii]))
            ref_alg.F.mse_loss = real_mse
            dt = time.time() - t0
            sd = agent.qnetwork_main.state_dict()
            out[f"{tag}/losses"] = np.array(losses, dtype=np.float32)
            out[f"{tag}/theta_l2"] = np.array(float(sum((v.double() ** 2).sum() for n, v in sd.items()
                                                        if v.dtype.is_floating_point and 'running' not in n) ** 0.5))
            out[f"{tag}/dims"] = np.array([S, A, H, B, NROWS, n_upd])
            print(f"g8: {tag}: {n_upd} updates in {dt:.1f}s = {n_upd / dt:.1f} updates/s (reference learn(), CPU)", flush=True)
        np.savez_compressed(os.path.join(HERE, "g8_curves.npz"), data_seed=np.array(2024), idx_seed=np.array(99), **out)

    with open(os.path.join(HERE, "VERSIONS.txt"), "w") as f:
        f.write(f"generated by make_golden.py with torch {meta['torch']}, numpy {meta['numpy']} (CPU)\n")


if __name__ == "__main__":
    main()
