"""GPU: the drop-in boundary — NAF / NAFAgent / ReplayBuffer / ManipulatorFramework with the reference's
signatures — against the reference's own known-answer test (G1), its act() statistics with the shipped demo
weights (G6), its step() gating trace (G4) and its checkpoint file formats."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT, load_group
from oracle import naf_oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.fixture()
def scratch_cwd(tmp_path):
    old = os.getcwd()
    os.chdir(tmp_path)
    yield tmp_path
    os.chdir(old)


def test_naf_forward_reference_known_answer_g1():
    """The reference's tests/.../test_naf_neural_network.py:53-67, same constructor call and inputs; the literals
    there are pinned to rtol 2e-5 (they drift 4.5e-6 across torch versions on the reference itself)."""
    from robotic_manipulator_rloa_amd.naf_components.naf_neural_network import NAF
    g = np.load(os.path.join(GOLDEN, "g1_known_answer.npz"))
    net = NAF(10, 5, 256, 0, DEV)
    sd = net.state_dict()
    ref = load_group(g, "sd")
    assert list(sd.keys()) == list(ref.keys())
    for k in ref:
        assert tuple(sd[k].shape) == tuple(ref[k].shape), k
        np.testing.assert_array_equal(sd[k].cpu().numpy(), ref[k], err_msg=k)      # same seed -> same weights, bit for bit
    assert [tuple(p.shape) for p in net.parameters()] == [tuple(ref[k].shape) for k in O.PARAM_ORDER]
    states = torch.arange(20).reshape(2, 10).float()
    actions = torch.tensor([[0, 1, 2, 3, 4], [10, 11, 12, 13, 14]]).long()
    a, q, v = net(states.to(DEV), actions.to(DEV))
    assert a.shape == (2, 5) and q.shape == (2, 1) and v.shape == (2, 1) and a.abs().max() <= 1
    assert q.requires_grad                                                   # differentiable, like the reference's forward
    q, v = q.detach(), v.detach()
    np.testing.assert_allclose(q.cpu().numpy(), g["q_test_literal"], rtol=2e-5)
    np.testing.assert_allclose(v.cpu().numpy(), g["v_test_literal"], rtol=5e-5)
    np.testing.assert_allclose(q.cpu().numpy(), g["q_f64"], rtol=2e-5)
    _, q_none, _ = net(states.to(DEV))
    assert q_none is None
    assert int(net.state_dict()["bn1.num_batches_tracked"]) == 2                   # two train-mode forwards
    # 'matmul' (textbook NAF) must differ: nobody silently "fixed" the reference's Hadamard P
    net2 = NAF(10, 5, 256, 0, DEV, p_mode="matmul")
    _, q2, _ = net2(states.to(DEV), actions.to(DEV))
    assert (q2.detach() - q).abs().max() > 1.0


def test_act_eval_mode_with_reference_demo_weights_g6(scratch_cwd):
    """Reference-format weight files load unchanged; act() = eval-mode mu + N(0, inverse(P)) noise, clamped."""
    from robotic_manipulator_rloa_amd.naf_components.naf_algorithm import NAFAgent
    g = np.load(os.path.join(GOLDEN, "g6_act.npz"))
    sd = {k: torch.from_numpy(v) for k, v in load_group(g, "kuka/sd").items()}
    torch.save(sd, "weights_kuka.p")
    agent = NAFAgent(object(), 21, 6, 256, 64, 1000, 1e-3, 1e-3, 0.99, 1, 1, 500, DEV, 0)
    agent.initialize_pretrained_agent_from_weights_file("weights_kuka.p")
    assert int(agent.qnetwork_main.state_dict()["bn1.num_batches_tracked"]) == int(sd["bn1.num_batches_tracked"])
    x = g["kuka/x"]
    agent.qnetwork_main.eval()
    gh = agent.qnetwork_main.heads(torch.from_numpy(x).to(DEV)).cpu().numpy()
    agent.qnetwork_main.train()
    np.testing.assert_allclose(np.tanh(gh[:, :6]), g["kuka/mu"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(gh[:, 6:27], g["kuka/l_pre"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(gh[:, 27], g["kuka/V"].ravel(), rtol=1e-4, atol=1e-4)
    draws = np.stack([agent.act(x[0]) for _ in range(600)])
    assert draws.shape == (600, 6) and draws.dtype == np.float32 and np.abs(draws).max() <= 1.0
    sigma = O.noise_std_hadamard(g["kuka/l_pre"][:1], 6)[0]
    mu = g["kuka/mu"][0]
    free = np.abs(mu) + 3.5 * sigma < 1.0
    if free.any():
        np.testing.assert_allclose(draws.mean(0)[free], mu[free], atol=5 * sigma[free].max() / np.sqrt(600))
        np.testing.assert_allclose(draws.std(0)[free], sigma[free], rtol=0.15)
    # against the reference's own sampled statistics (256 draws/state there)
    np.testing.assert_allclose(draws.mean(0), g["kuka/act_mean"][0], atol=0.2)
    with pytest.raises(Exception):
        agent.initialize_pretrained_agent_from_weights_file("missing.p")


def test_step_gating_trace_matches_reference_g4(scratch_cwd):
    """Which timesteps fire learn(), for update_freq/num_updates combinations: the reference's trace
    (naf_algorithm.py:144-156: strict len > batch_size, (t+1) % update_freq == 0)."""
    from robotic_manipulator_rloa_amd.naf_components.naf_algorithm import NAFAgent
    from synth_data import make_transitions
    gate = np.load(os.path.join(GOLDEN, "g4_gating.npz"))
    st, ac, rw, ns, dn = make_transitions(500, 21, 6, seed=11)
    for key in gate.files:
        uf, nu = [int(x[2:]) for x in key.split("_")]
        agent = NAFAgent(object(), 21, 6, 256, 8, 1000, 1e-3, 1e-3, 0.99, uf, nu, 500, DEV, 0)
        fired, prev = [], 0
        for t in range(40):
            agent.step(st[t].astype(np.float64), ac[t], float(rw[t]), st[t + 1].astype(np.float64), 0)
            now = int(agent.learner.step_dev.item())
            fired.append(now - prev)
            prev = now
        np.testing.assert_array_equal(fired, gate[key], err_msg=key)
        assert len(agent.memory) == 40 and torch.isfinite(agent.learner.theta2).all()


def test_learn_api_with_reference_sample_tuple(scratch_cwd):
    """NAFAgent.learn((states, actions int64, rewards, next_states, dones)) == the reference's losses (G3)."""
    from robotic_manipulator_rloa_amd.naf_components.naf_algorithm import NAFAgent
    from synth_data import make_transitions
    g = np.load(os.path.join(GOLDEN, "g3_learn.npz"))
    S, A, B = 21, 6, 256
    st, ac, rw, ns, dn = make_transitions(5 * B, S, A, seed=7)
    agent = NAFAgent(object(), S, A, 256, B, 100000, 1e-3, 1e-3, 0.99, 1, 1, 500, DEV, 0)
    losses = []
    for k in range(5):
        sl = slice(k * B, (k + 1) * B)
        agent.learn((torch.from_numpy(st[sl]), torch.from_numpy(ac[sl]).long(), torch.from_numpy(rw[sl, None]),
                     torch.from_numpy(ns[sl]), torch.from_numpy(dn[sl, None])))
        losses.append(agent.last_loss())
    np.testing.assert_allclose(losses, g["kuka/losses5"], rtol=5e-3)
    assert agent.optimizer.state_dict()["step"] == 5
    with pytest.raises(ValueError):
        agent.learn((torch.zeros(3, S), torch.zeros(3, A), torch.zeros(3, 1), torch.zeros(3, S), torch.zeros(3, 1)))


def test_soft_update_api(scratch_cwd):
    from robotic_manipulator_rloa_amd.naf_components.naf_algorithm import NAFAgent
    agent = NAFAgent(object(), 21, 6, 256, 8, 100, 1e-3, 0.25, 0.99, 1, 1, 500, DEV, 0)
    with torch.no_grad():
        agent.learner.theta2[0].add_(1.0)
    main0 = {k: v.cpu().numpy() for k, v in agent.qnetwork_main.state_dict().items()}
    tgt0 = {k: v.cpu().numpy() for k, v in agent.qnetwork_target.state_dict().items()}
    agent.soft_update(agent.qnetwork_main, agent.qnetwork_target)
    tgt1 = agent.qnetwork_target.state_dict()
    for k in O.PARAM_ORDER:
        np.testing.assert_array_equal(tgt1[k].cpu().numpy(), O.polyak(tgt0[k], main0[k], 0.25), err_msg=k)   # bit-exact
    np.testing.assert_array_equal(tgt1["bn1.running_mean"].cpu().numpy(), tgt0["bn1.running_mean"])          # buffers untouched
    # foreign modules (e.g. a user's torch nets) go through the same kernel per tensor
    a, b = torch.nn.Linear(16, 16).to(DEV), torch.nn.Linear(16, 16).to(DEV)
    wa, wb = a.weight.detach().cpu().numpy().copy(), b.weight.detach().cpu().numpy().copy()
    agent.soft_update(a, b)
    np.testing.assert_array_equal(b.weight.detach().cpu().numpy(), O.polyak(wb, wa, 0.25))


def test_run_writes_reference_format_checkpoints(scratch_cwd):
    """NAFAgent.run with a host env: scores dict, checkpoints/{ep}/weights.p + scores.txt, model.p
    (naf_algorithm.py:273-289; pinned by the reference's tests/.../test_naf_algorithm.py:316-328)."""
    from robotic_manipulator_rloa_amd.environment.synthetic import SyntheticEnvironment
    from robotic_manipulator_rloa_amd.naf_components.naf_algorithm import NAFAgent
    from oracle.torch_cpu_port import Net
    env = SyntheticEnvironment(6)
    agent = NAFAgent(env, 21, 6, 256, 16, 1000, 1e-3, 1e-3, 0.99, 1, 1, 2, DEV, 0)
    assert os.path.isdir("checkpoints")
    scores = agent.run(frames=15, episodes=4, verbose=False)
    assert set(scores.keys()) == {1, 2, 3, 4} and all(len(v) == 2 for v in scores.values())
    assert scores[4][1] == 15 and scores[1][0] < 0
    assert int(agent.learner.step_dev.item()) == 60 - 16                           # learning starts when len > batch_size
    for ep in (2, 4):
        assert os.path.isfile(f"checkpoints/{ep}/weights.p") and os.path.isfile(f"checkpoints/{ep}/scores.txt")
    saved_scores = json.loads(open("checkpoints/4/scores.txt").read())
    assert saved_scores["4"] == [scores[4][0], scores[4][1]]
    sd = torch.load("model.p", map_location="cpu")
    g = np.load(os.path.join(GOLDEN, "g3_learn.npz"))
    ref = load_group(g, "kuka/main0")
    assert list(sd.keys()) == list(ref.keys())
    for k in ref:
        assert tuple(sd[k].shape) == tuple(ref[k].shape) and sd[k].device.type == "cpu"
    assert int(sd["bn1.num_batches_tracked"]) == 44
    Net(sd, 6)                                                                      # loads into a reference-layout net
    agent2 = NAFAgent(env, 21, 6, 256, 16, 1000, 1e-3, 1e-3, 0.99, 1, 1, 2, DEV, 1)
    agent2.initialize_pretrained_agent_from_episode(4)
    for k, v in agent2.qnetwork_target.state_dict().items():
        np.testing.assert_array_equal(v.cpu().numpy(), torch.load("checkpoints/4/weights.p")[k].numpy(), err_msg=k)


def test_framework_end_to_end_synthetic(scratch_cwd):
    from robotic_manipulator_rloa_amd import ManipulatorFramework
    f = ManipulatorFramework()
    f.set_hyperparameter("batch_size", 64)          # BASELINE configs[0]: demo training at batch 64
    f.set_hyperparameter("buffer_size", 5000)
    f.run_demo_training("kuka_training", interactive=False, environment="synthetic", episodes=2, frames=40)
    assert f.env is None and f.naf_agent is None and os.path.isfile("model.p")
    f.initialize_synthetic_environment(6)
    f.initialize_naf_agent(checkpoint_frequency=1, seed=3)
    f.load_pretrained_parameters_from_weights_file("model.p")
    out = f.test_trained_model(2, 10)
    assert out["episodes"] == 2
    stats = f.run_vectorized_training(vector_steps=6, n_envs=32, max_frames=50)
    assert stats["env_steps"] == 192 and stats["updates"] == 32 * 4 and np.isfinite(stats["last_loss"])
    f.get_nafagent_configuration()
    f.get_environment_configuration()
    f.delete_naf_agent()
    f.delete_environment()


_DP1 = r'''
import os, sys
sys.path.insert(0, os.environ["NAF_ROOT"]); sys.path.insert(0, os.path.join(os.environ["NAF_ROOT"], "tests")); sys.path.insert(0, os.path.join(os.environ["NAF_ROOT"], "tests", "golden"))
import numpy as np, torch, torch.distributed as dist
from robotic_manipulator_rloa_amd import parallel
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ["NAF_TEST_PORT"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from test_learner_gpu import _kuka_learner_and_replay
from robotic_manipulator_rloa_amd.engine import TrainChunk
res = []
os.environ["NAF_NO_FOLD_NORM"] = "1"      # both runs take the gradient norm with the same (post-all-reduce) kernel
for force in ("0", "1"):
    os.environ["NAF_FORCE_ALLREDUCE"] = force
    L, buf = _kuka_learner_and_replay(5000, 256, seed_data=5)
    chunk = TrainChunk(L, buf, 4)
    chunk.capture()                        # with force=1 the RCCL all-reduce is a node of the captured graph
    for _ in range(3):
        chunk.run()
    torch.cuda.synchronize()
    res.append(L.theta2.clone())
assert torch.equal(res[0], res[1])
dist.destroy_process_group()
print("RCCL_CAPTURE_OK")
'''


def test_rccl_allreduce_inside_captured_graph_world1(tmp_path):
    """The N > 1 launch structure on the one GPU available here: an RCCL all-reduce of the flat gradient captured
    inside the learn() graph (world_size 1, so the sum is the identity and the result must be bit-identical)."""
    import socket
    script = tmp_path / "dp1.py"
    script.write_text(_DP1)
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = str(sock.getsockname()[1])
    r = subprocess.run([sys.executable, str(script)],
                       env=dict(os.environ, NAF_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", NAF_TEST_PORT=port),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_CAPTURE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_device_env_kernel_matches_numpy_environment():
    """csrc/synth_env.hip against its numpy twin (environment/synthetic.py) on the same actions."""
    from robotic_manipulator_rloa_amd import _lib
    from robotic_manipulator_rloa_amd.environment.synthetic import SyntheticEnvironment
    lib = _lib.load()
    E, A, S = 8, 6, 21
    nst = lib.naf_synth_env_state_floats(A)
    st = torch.zeros(E, nst, device=DEV)
    obs = torch.zeros(E, S, device=DEV)
    rows = torch.zeros(E, 64, device=DEV)
    stream = torch.cuda.current_stream().cuda_stream
    assert lib.naf_synth_env_reset(st.data_ptr(), obs.data_ptr(), E, A, 5, 0, None, stream) == 0
    envs = [SyntheticEnvironment(A) for _ in range(E)]
    q0 = st[:, :A].cpu().numpy()
    for e, env in enumerate(envs):
        env.reset(False)
        env.q = q0[e].copy()                       # same randomised start as the device env
        np.testing.assert_allclose(obs[e].cpu().numpy(), env.get_state(), atol=2e-6)
    rng = np.random.default_rng(0)
    for t in range(30):
        act = rng.uniform(-1, 1, (E, A)).astype(np.float32)
        a_d = torch.from_numpy(act).to(DEV)
        assert lib.naf_synth_env_step(st.data_ptr(), a_d.data_ptr(), rows.data_ptr(), obs.data_ptr(), E, A, 5, None, 0, stream) == 0
        r = rows.cpu().numpy()
        for e, env in enumerate(envs):
            s_before = env.get_state()
            s2, rew, done = env.step(act[e])
            np.testing.assert_allclose(r[e, :S], s_before, atol=3e-6)
            np.testing.assert_array_equal(r[e, S:S + A], act[e])
            _, off_r, off_s2, off_d = O.row_offsets(S, A)
            np.testing.assert_allclose(r[e, off_s2:off_s2 + S], s2, atol=3e-6)
            np.testing.assert_allclose(r[e, off_r], rew, rtol=1e-4, atol=1e-5)
            assert r[e, off_d] == done
            assert (r[e, off_d + 1:] == 0).all() and (r[e, off_r + 1:off_s2] == 0).all()


def test_naf_forward_is_differentiable_like_the_reference():
    """Q.backward() through NAF.forward fills .grad of the 14 parameters: compared with the numpy oracle's analytic
    gradients (the same oracle that is pinned to the reference's autograd in G3)."""
    from robotic_manipulator_rloa_amd.naf_components.naf_neural_network import NAF
    from synth_data import make_transitions
    S, A, B = 21, 6, 64
    net = NAF(S, A, 256, 0, DEV)
    st, ac, rw, ns, dn = make_transitions(B, S, A, seed=3)
    u = torch.from_numpy(ac).long()
    sd0 = {k: v.cpu().numpy() for k, v in net.state_dict().items()}
    wq = torch.from_numpy(np.random.default_rng(0).standard_normal((B, 1)).astype(np.float32)).to(DEV)
    _, q, v = net(torch.from_numpy(st).to(DEV), u.to(DEV))
    assert q.requires_grad and v.requires_grad
    (q * wq).sum().backward()
    p = O.cast_params(sd0, np.float64)
    fwd, _ = O.net_forward_train(p, st, np.trunc(ac))
    np.testing.assert_allclose(q.detach().cpu().numpy().ravel(), fwd["Q"], rtol=2e-4, atol=2e-4)
    grads = O.net_backward(p, fwd, np.trunc(ac), wq.cpu().numpy().ravel().astype(np.float64))
    named = dict(net.named_parameters())
    assert list(named.keys()) == O.PARAM_ORDER
    for name in O.PARAM_ORDER:
        if name in ("input_layer.bias", "hidden_layer.bias"):
            continue
        g = named[name].grad
        assert g is not None, name
        scale = max(1e-3, np.abs(grads[name]).max())
        np.testing.assert_allclose(g.cpu().numpy().reshape(grads[name].shape), grads[name], rtol=5e-3, atol=2e-4 * scale,
                                   err_msg=name)
    # eval mode / no_grad: plain tensors, no graph
    with torch.no_grad():
        _, q2, _ = net(torch.from_numpy(st).to(DEV), u.to(DEV))
    assert not q2.requires_grad


def test_device_env_presets_and_per_env_obstacles():
    """BASELINE configs[3]: per-env randomised obstacle positions (seeded by (seed, env)); xArm6 / Panda presets."""
    import ctypes
    from robotic_manipulator_rloa_amd import _lib
    from robotic_manipulator_rloa_amd.engine import DeviceEnvLoop
    lib = _lib.load()
    E, A, S = 16, 6, 21
    nst = lib.naf_synth_env_state_floats(A)
    stream = torch.cuda.current_stream().cuda_stream
    outs = []
    for rep in range(2):
        st = torch.zeros(E, nst, device=DEV)
        obs = torch.zeros(E, S, device=DEV)
        preset = (ctypes.c_float * 15)(*(DeviceEnvLoop.PRESETS["xarm6"] + [0.1]))
        assert lib.naf_synth_env_reset(st.data_ptr(), obs.data_ptr(), E, A, 77, 0, preset, stream) == 0
        outs.append(obs.cpu().numpy())
    np.testing.assert_array_equal(outs[0], outs[1])                               # deterministic in (seed, env)
    o = outs[0]
    np.testing.assert_allclose(o[:, 15:18], np.tile([0.3, 0.47, 0.61], (E, 1)), atol=1e-6)     # shared target
    obst = o[:, 18:21]
    assert np.abs(obst - np.array([0.25, 0.27, 0.5])).max() <= 0.1 + 1e-6 and len({tuple(x) for x in obst.round(5)}) == E
    np.testing.assert_allclose(o[:, 1], 1.0, atol=0.1 + 1e-6)                        # xArm6 initial joints (+-0.1 variation)
    np.testing.assert_allclose(o[:, 3], -2.3, atol=0.1 + 1e-6)


def _make_env6():
    from robotic_manipulator_rloa_amd.environment.synthetic import SyntheticEnvironment
    return SyntheticEnvironment(6, initial_positions_variation_range=[0.1] * 6)


@pytest.mark.parametrize("async_policy", [False, True])
def test_host_vector_env_training_end_to_end(scratch_cwd, async_policy):
    """E environments in worker processes feed the HBM replay ring through pinned staging; batched act on the GPU;
    E learn() per vector step once len(memory) > batch_size."""
    from robotic_manipulator_rloa_amd.environment.vector_env import HostVectorEnv
    from robotic_manipulator_rloa_amd.naf_components.naf_algorithm import NAFAgent
    E = 8
    vec = HostVectorEnv(_make_env6, E, 21, 6, envs_per_worker=4, max_frames=20, seed=1)
    try:
        agent = NAFAgent(None, 21, 6, 256, 32, 1000, 1e-3, 1e-3, 0.99, 1, 1, 500, DEV, 0)
        out = agent.run_host_vectorized(vec, vector_steps=12, async_policy=async_policy)
        assert out["env_steps"] == 96 and len(agent.memory) == 96 and int(agent.memory.meta[1].item()) == 96
        # learning starts at the first vector step that leaves len(memory) > 32, i.e. after step 5 (40 rows): 8 steps x 8
        assert out["updates"] == 64 and int(agent.learner.step_dev.item()) == 64
        assert out["episodes_finished"] == 0 and np.isfinite(out["last_loss"]) and out["mean_reward"] < 0
        rows = agent.memory.rows[:96].cpu().numpy()
        s, a, r, s2, d = O.unpack_rows(rows, 21, 6)
        np.testing.assert_allclose(s2[:, :6], s[:, :6] + a / 240.0, atol=1e-6)      # the worker envs' transitions, intact
        assert np.abs(a).max() <= 1.0 and (d == 0).all()
        e0 = rows[0::E]
        np.testing.assert_array_equal(e0[1:, :21], O.unpack_rows(e0[:-1], 21, 6)[3])   # env 0's states chain step to step
    finally:
        vec.close()


@pytest.mark.parametrize("world,fuse", [(2, "l1,b2,gb,s3"), (4, None)])
def test_xgmi_oneshot_allreduce_ranks_sharing_one_gpu(world, fuse):
    """csrc/xgmi_reduce.hip with W > 1 on the one GPU available: W processes on cuda:0 (tests/xgmi_worker.py) map each
    other's receive slabs through hipIpc and run the one-shot all-reduce eagerly, inside a captured graph and under
    Learner.learn_rows — results bit-exact against the rank-ordered sum, replicas in lock-step, no timed-out wait.
    Two ranks on the column-tile chain (part of the gradient pushed ahead from inside its last kernel), four on the default
    chain of B = 256 (the row-split one: the vector travels whole)."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = str(sock.getsockname()[1])
    env = dict(os.environ, NAF_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
               OMP_NUM_THREADS="2")
    env.pop("NAF_FUSE", None)
    if fuse:
        env["NAF_FUSE"] = fuse
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", port, os.path.join(ROOT, "tests", "xgmi_worker.py")],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-4000:]
    for k in range(world):
        assert f"XGMI_OK_{k};" in r.stdout, r.stdout[-2000:]


def _check_bench_line_n2(r, steps, rehearsal):
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == steps and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["parallelism"] == "dp2"
    assert ("REHEARSAL" in out["config"]["launch"]) == rehearsal
    assert out["sanity"]["params_finite"] and out["sanity"]["replicas_identical"] is True
    assert out["sanity"]["optimizer_steps"] == (steps + 2) * 64   # warm-up inside capture leaves no trace
    assert abs(out["value"] - 2 * 64 * steps / (out["ms_per_step"] * steps * 1e-3)) < 1e-3 * out["value"]
    assert "cpu_baseline" not in out                               # timed at N = 1 only
    return out


def test_bench_two_ranks_rehearsal_on_one_gpu():
    """bench.py's N > 1 code path (torch.distributed.run launch, rank-0-only JSON line, barrier-bracketed timing, MAX
    over ranks, whole-job aggregate, gradient exchange inside the captured graphs) with both ranks on cuda:0
    (NAF_BENCH_REHEARSAL=1: gloo control plane, the peer-memory all-reduce as the exchange)."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = str(sock.getsockname()[1])
    env = dict(os.environ, NAF_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=port, OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", port, os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "6", "--warmup", "2", "--buffer", "100000", "--roofline-ring", "0"],
                       env=env, capture_output=True, text=True, timeout=900)
    out = _check_bench_line_n2(r, 6, rehearsal=True)
    assert "one-shot" in out["config"]["grad_exchange"] and out["sanity"]["xgmi_timed_out_waits"] == 0


def test_bench_gpus2_launches_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher around it (how the driver invokes N = 1): the parent starts
    torch.distributed.run as a child before touching the GPU and relays rank 0's line. On the 1-GPU box the two ranks
    share cuda:0 (rehearsal, labelled); without the rehearsal switch the same command is refused."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--buffer",
           "100000", "--roofline-ring", "0"]
    if torch.cuda.device_count() < 2:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 2 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        env["NAF_BENCH_REHEARSAL"] = "1"
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    _check_bench_line_n2(r, 5, rehearsal=torch.cuda.device_count() < 2)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two MI355X (the gpurun box has one)")
@pytest.mark.parametrize("xgmi", ["1", "0"])
def test_bench_two_real_gpus_rccl_and_oneshot(xgmi):
    """Two ranks on two DISTINCT devices: RCCL (NAF_XGMI=0) and the one-shot peer-memory exchange (default) each move
    the gradient between devices; replicas stay bit-identical, no wait times out, RCCL saw two ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT",
                                                            "NAF_BENCH_REHEARSAL")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", NAF_XGMI=xgmi)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3",
                        "--buffer", "100000", "--roofline-ring", "0"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and "REHEARSAL" not in out["config"]["launch"]
    assert out["sanity"]["replicas_identical"] is True and out["sanity"]["params_finite"]
    if xgmi == "1" and "one-shot" in out["config"]["grad_exchange"]:
        assert out["sanity"]["xgmi_timed_out_waits"] == 0 and out["sanity"]["xgmi_allreduces"] >= 23 * 64
    else:
        assert out["config"]["grad_exchange"] == "RCCL all-reduce"
