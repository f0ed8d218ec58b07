"""CPU (`-m "not gpu"`): host logic, the C-ABI library's exported symbols, loud failure without a GPU, and the
world_size-2 data-parallel path over gloo."""
import ctypes
import os
import re
import subprocess
import sys

import time

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT, load_group
from oracle import naf_oracle as O


def test_library_loads_and_exports_every_declared_symbol():
    """Every function include/naf_hip.h declares must be exported by the built .so (no compute calls here)."""
    from robotic_manipulator_rloa_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "naf_hip.h")).read()
    declared = set(re.findall(r"\b(naf_[a-z0-9_]+)\s*\(", header))
    declared -= {"naf_replay_t"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/naf_hip.h but not exported"
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    assert lib.naf_hip_arch() == b"gfx950" and lib.naf_hip_abi_version() == _lib.header_abi_version() >= 3
    assert lib.naf_replay_batch_row_floats(21, 6) == 52 and lib.naf_replay_batch_row_floats(23, 7) == 56
    assert lib.naf_replay_row_floats(21, 6) == 64 and lib.naf_replay_row_floats(23, 7) == 64
    assert lib.naf_replay_row_floats(5, 1) == 32 and lib.naf_replay_row_floats(0, 6) == -1


def test_product_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from robotic_manipulator_rloa_amd import _lib, ManipulatorFramework
    from robotic_manipulator_rloa_amd.naf_components.naf_algorithm import NAFAgent
    from robotic_manipulator_rloa_amd.naf_components.naf_neural_network import NAF
    from robotic_manipulator_rloa_amd.utils.replay_buffer import ReplayBuffer
    with pytest.raises(_lib.NafHipError):
        NAF(10, 5, 256, 0, torch.device("cpu"))
    with pytest.raises(_lib.NafHipError):
        ReplayBuffer(100, 8, "cpu", 0)
    with pytest.raises(_lib.NafHipError):
        NAFAgent(object(), 21, 6, 256, 8, 100, 1e-3, 1e-3, 0.99, 1, 1, 500, torch.device("cpu"), 0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "robotic_manipulator_rloa_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{f} imports the oracle"


def test_reference_init_bit_exact_and_layout_roundtrip():
    from robotic_manipulator_rloa_amd.learner import NetLayout, PARAM_ORDER
    from robotic_manipulator_rloa_amd.naf_components.naf_neural_network import reference_init_state_dict
    g = np.load(os.path.join(GOLDEN, "g3_learn.npz"))
    for tag, (S, A) in (("kuka", (21, 6)), ("panda", (23, 7))):
        sd = reference_init_state_dict(S, A, 256, 0)
        for k, v in sd.items():
            np.testing.assert_array_equal(v.numpy(), g[f"{tag}/main0/{k}"], err_msg=k)   # same draws as the reference ctor
        lay = NetLayout(S, A, 256)
        assert lay.n_ref_params() == sum(v.numel() for k, v in sd.items() if k in PARAM_ORDER)
        flat = torch.zeros(lay.P)
        views = lay.param_views(flat)
        for k in PARAM_ORDER:
            views[k].copy_(sd[k].reshape(views[k].shape))
        used = torch.zeros(lay.P, dtype=torch.bool)
        probe = lay.param_views(torch.arange(lay.P, dtype=torch.float32))
        total = 0
        for k in PARAM_ORDER:
            ids = probe[k].reshape(-1).long()
            assert not used[ids].any(), f"{k} overlaps another parameter"
            used[ids] = True
            total += ids.numel()
            np.testing.assert_array_equal(views[k].reshape(sd[k].shape).numpy(), sd[k].numpy())
        assert total == lay.n_ref_params() and (flat[~used] == 0).all()
        assert lay.P % 64 == 0 and lay.NHP % 8 == 0 and all(s.offset % 64 == 0 for s in lay.seg.values())
    assert NetLayout(21, 6, 256).n_ref_params() == 79644 and NetLayout(23, 7, 256).n_ref_params() == 82212
    # 9 .. 64 joints: a layout like any other (one sample per 16- / 32- / 64-lane group in the stand-alone head kernels: a wavefront
    # has 64 lanes); beyond: refused
    wide = NetLayout(27, 9, 256)
    assert (wide.T, wide.NH, wide.NHP, wide.row_floats) == (45, 55, 64, 128) and wide.n_ref_params() == 256 * 27 + 256 * 256 + 6 * 256 + 55 * 257
    big = NetLayout(137, 64, 256)
    assert (big.T, big.NH, big.NHP, big.row_floats) == (2080, 2145, 2160, 512)
    with pytest.raises(ValueError):
        NetLayout(139, 65, 256)
    # a layer_size below 256 is STORED zero-padded to 256 (the kernels see H = 256), shown in the reference's shapes (H_ref): the
    # reference's own agent test builds NAF(10, 5, 128)
    for h in (128, 64, 200, 4):
        sd = reference_init_state_dict(10, 5, h, 0)
        lay = NetLayout(10, 5, h)
        plain = NetLayout(10, 5, h, pad_layer=False)
        assert (lay.H, lay.H_ref, lay.HP) == (256, h, 272) and (plain.H, plain.H_ref) == (h, h)
        assert lay.P == NetLayout(10, 5, 256).P and lay.n_ref_params() == plain.n_ref_params() == sum(v.numel() for k, v in sd.items() if k in PARAM_ORDER)
        flat = torch.zeros(lay.P)
        views = lay.param_views(flat)
        probe = lay.param_views(torch.arange(lay.P, dtype=torch.float32))
        used = torch.zeros(lay.P, dtype=torch.bool)
        for k in PARAM_ORDER:
            assert tuple(views[k].shape) == tuple(plain.param_views(torch.zeros(plain.P))[k].shape) == tuple(sd[k].shape) or \
                views[k].numel() == sd[k].numel(), k
            views[k].copy_(sd[k].reshape(views[k].shape))
            ids = probe[k].reshape(-1).long()
            assert not used[ids].any(), k
            used[ids] = True
        assert (flat[~used] == 0).all() and int(used.sum()) == lay.n_ref_params()
        # the padded units: rows h.. of W1 / b1 / g1 / be1 / W2 / b2 / g2 / be2, columns h.. of W2, columns h..255 of Wh — never a view's
        assert not used[lay.seg["g1"].offset + h:lay.seg["g1"].offset + 256].any()
        assert float(lay.view(flat, "Wh")[:, h:256].abs().max()) == 0.0 and float(lay.view(flat, "W2")[:, h:].abs().max()) == 0.0
    # (round 6: a width in (256, 512) is stored as 512 the same way — the row-split chain runs 512 columns as two 256-column halves;
    #  beyond 512, and wherever pad_layer = False, as it is)
    mid = NetLayout(21, 6, 300)
    assert (mid.H, mid.H_ref, mid.HP) == (512, 300, 528) and mid.P == NetLayout(21, 6, 512).P and \
        mid.n_ref_params() == NetLayout(21, 6, 300, pad_layer=False).n_ref_params()
    assert NetLayout(21, 6, 300, pad_layer=False).H == 300 and NetLayout(21, 6, 512).H == 512 and NetLayout(21, 6, 600).H == 600


def test_hyperparameter_rules_match_reference_contract():
    """Aliases, ranges and messages of set_hyperparameter (reference rl_framework.py:159-230; its tests :124-204)."""
    from robotic_manipulator_rloa_amd import ManipulatorFramework
    from robotic_manipulator_rloa_amd.utils.exceptions import InvalidHyperParameter
    f = ManipulatorFramework()
    hp = f._hyperparameters
    assert (hp.buffer_size, hp.batch_size, hp.gamma, hp.tau, hp.learning_rate, hp.update_freq, hp.num_updates) == \
        (100000, 128, 0.99, 0.001, 0.001, 1, 1)
    ok = {"buffer_size": ("buffer_size", 5), "buffersize": ("buffer_size", 6), "BUFFER_SIZE": ("buffer_size", 7),
          "BUFFERSIZE": ("buffer_size", 8), "batch_size": ("batch_size", 64), "BATCHSIZE": ("batch_size", 32),
          "gamma": ("gamma", 0.5), "GAMMA": ("gamma", 0.9), "tau": ("tau", 0), "TAU": ("tau", 1),
          "learning_rate": ("learning_rate", 0.1), "LEARNINGRATE": ("learning_rate", 2), "update_freq": ("update_freq", 4),
          "UPDATEFREQ": ("update_freq", 2), "num_update": ("num_updates", 3), "NUMUPDATE": ("num_updates", 2),
          "NUM_UPDATE": ("num_updates", 5)}
    for name, (field, value) in ok.items():
        f.set_hyperparameter(name, value)
        assert getattr(f._hyperparameters, field) == value
    bad = [("buffer_size", 0), ("buffer_size", 1.5), ("batch_size", -1), ("gamma", 0), ("gamma", 1), ("gamma", "x"),
           ("tau", -0.1), ("tau", 1.1), ("learning_rate", 0), ("update_freq", 0), ("update_freq", 2.0), ("num_update", 0),
           ("Buffer_Size", 5), ("num_updates", 2), ("lr", 0.1)]
    for name, value in bad:
        with pytest.raises(InvalidHyperParameter):
            f.set_hyperparameter(name, value)
    with pytest.raises(InvalidHyperParameter, match="Gamma is not a float or its value is out of range"):
        f.set_hyperparameter("gamma", 2)


def test_framework_guards_and_exception_messages():
    from robotic_manipulator_rloa_amd import ManipulatorFramework
    from robotic_manipulator_rloa_amd.utils import exceptions as E
    f = ManipulatorFramework()
    with pytest.raises(E.EnvironmentNotInitialized):
        f.initialize_naf_agent()
    with pytest.raises(E.ConfigurationIncomplete):
        f.run_training(1, 1)
    with pytest.raises(E.ConfigurationIncomplete):
        f.test_trained_model(1, 1)
    with pytest.raises(E.EnvironmentNotInitialized):
        f.load_pretrained_parameters_from_weights_file("x.p")
    f.initialize_synthetic_environment()
    with pytest.raises(E.NAFAgentNotInitialized):
        f.load_pretrained_parameters_from_episode(3)
    with pytest.raises(E.InvalidNAFAgentParameter, match="Checkpoint Frequency or Seed received is not an integer"):
        f.initialize_naf_agent(checkpoint_frequency=1.5)
    with pytest.raises(E.InvalidManipulatorFile):       # pybullet is not installed in this image
        f.initialize_environment("kuka.sdf", 13, [6], [0, 1, 2, 3, 4, 5], [0, 0, 0], [1, 1, 1])
    # messages and str() format of the reference's exception types
    assert str(E.MissingWeightsFile()) == "MissingWeightsFile: The weight file provided does not exist"
    assert str(E.InvalidHyperParameter("boom")) == "InvalidHyperParameter: boom"
    assert E.ConfigurationIncomplete().message.startswith("The configuration for the training is incomplete.")
    assert issubclass(E.InvalidEnvironmentParameter, E.FrameworkException) and issubclass(E.FrameworkException, Exception)
    assert E.EnvironmentNotInitialized().set_message("m").message == "m"
    f.delete_environment()
    assert f.env is None


def test_synthetic_environment_protocol_and_rewards():
    from robotic_manipulator_rloa_amd.environment.synthetic import SyntheticEnvironment, forward_kinematics
    env = SyntheticEnvironment(6)
    s = env.reset(False)
    assert s.shape == (21,) and s.dtype == np.float64 and env.observation_space.shape == (21,) and env.action_space.shape == (6,)
    np.testing.assert_allclose(s[:6], [0.9, 0.45, 0, 0, 0, 0], atol=1e-7)
    np.testing.assert_array_equal(s[6:12], 0)
    np.testing.assert_allclose(s[15:18], [0.4, 0.85, 0.71], atol=1e-7)
    np.testing.assert_allclose(s[18:21], [0.45, 0.55, 0.55], atol=1e-7)
    a = np.array([1, -1, 0.5, 0, 0, 0], np.float32)
    s2, r, d = env.step(a)
    np.testing.assert_allclose(s2[:6], s[:6] + a / 240.0, atol=1e-6)
    np.testing.assert_allclose(s2[6:12], a)
    dist = np.linalg.norm(s2[12:15] - s2[15:18])
    assert d == 0 and abs(r + (dist - 0.05)) < 1e-5
    # chain of zero angles points straight up: ee = (0, 0, sum of the first 6 links)
    ee, hit = forward_kinematics(np.zeros(6, np.float32), np.array([5, 5, 5], np.float32))
    np.testing.assert_allclose(ee, [0, 0, 0.34 + 0.02 + 0.40 + 0.02 + 0.40 + 0.13], atol=1e-6)
    # terminal events
    env2 = SyntheticEnvironment(6, target_position=list(ee))
    env2.q = np.zeros(6, np.float32)
    _, r, d = env2.step(np.zeros(6, np.float32))
    assert (r, d) == (250, 1)
    env3 = SyntheticEnvironment(6, obstacle_position=[0, 0, 0.35])
    env3.q = np.zeros(6, np.float32)
    _, r, d = env3.step(np.zeros(6, np.float32))
    assert (r, d) == (-1000, 1)


_DP_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["NAF_ROOT"]); sys.path.insert(0, os.path.join(os.environ["NAF_ROOT"], "tests", "golden"))
import numpy as np, torch, torch.distributed as dist
from oracle import naf_oracle as O
from robotic_manipulator_rloa_amd.learner import NetLayout, PARAM_ORDER
from robotic_manipulator_rloa_amd import parallel
from synth_data import make_transitions
rank, local_rank, world = parallel.init_distributed("gloo")
assert world == 2 and dist.get_backend() == "gloo"
S, A, B = 21, 6, 32
g = np.load(os.path.join(os.environ["NAF_ROOT"], "tests", "golden", "g3_learn.npz"))
sd = {k[len("kuka/main0/"):]: g[k] for k in g.files if k.startswith("kuka/main0/")}
lay = NetLayout(S, A, 256)
# replicas start identical; rank 1 deliberately perturbs, then the broadcast repairs it
theta2 = torch.zeros(2, lay.P)
for net in range(2):
    for k, v in lay.param_views(theta2[net]).items():
        v.copy_(torch.from_numpy(sd[k]).reshape(v.shape))
if rank == 1:
    theta2 += 1.0
parallel.broadcast_parameters(theta2)
ref = torch.zeros(2, lay.P)
for net in range(2):
    for k, v in lay.param_views(ref[net]).items():
        v.copy_(torch.from_numpy(sd[k]).reshape(v.shape))
assert torch.equal(theta2, ref)
# each rank: its own shard of transitions (independent units, no data-path collective) -> local gradient
st, ac, rw, ns, dn = make_transitions(2 * B, S, A, seed=5 + 0)
sl = slice(rank * B, (rank + 1) * B)
L = O.LearnerOracle(sd, dtype=np.float32)
u = np.trunc(ac[sl]).astype(np.float32)
tf, _ = O.net_forward_train(L.target, ns[sl])
y = rw[sl] + np.float32(0.99) * tf["V"]
mf, _ = O.net_forward_train(L.main, st[sl], u)
dQ = (2.0 * (mf["Q"] - y) / B).astype(np.float32)
grads = O.net_backward(L.main, mf, u, dQ)
flat = torch.zeros(lay.P)
for k, v in lay.param_views(flat).items():
    v.copy_(torch.from_numpy(grads[k]).reshape(v.shape))
local = flat.clone()
parallel.all_reduce_flat_grad(flat)                 # the one exchange per update
gathered = [torch.zeros(lay.P) for _ in range(world)]
dist.all_gather(gathered, local)
assert torch.allclose(flat, gathered[0] + gathered[1], rtol=0, atol=0)
# identical optimizer step on every rank from the summed gradient with inv_world folded into the clip
avg = {k: (v.numpy().reshape(sd[k].shape) / world).astype(np.float32) for k, v in lay.param_views(flat).items()}
clipped, norm = O.clip_grad_norm(avg, 1.0)
new = {k: O.adam_step(L.main[k], clipped[k], L.m[k], L.v[k], 1, 1e-3)[0] for k in PARAM_ORDER}
chk = torch.tensor([float(sum(np.abs(v).sum() for v in new.values())), norm], dtype=torch.float64)
both = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
dist.all_gather(both, chk)
assert torch.equal(both[0], both[1]), both        # replicas stay in lock-step
assert parallel.rank_seed(0, 0) == 0 and parallel.rank_seed(0, 1) != parallel.rank_seed(0, 0)
assert [parallel.units_per_rank(130, 4, r) for r in range(4)] == [33, 33, 32, 32]
dist.barrier(); dist.destroy_process_group()
os.write(1, f"DP_OK_{rank};".encode())      # one atomic write per rank: the two ranks share stdout
'''


def test_data_parallel_world2_gloo(tmp_path):
    """N > 1 path on CPU: 2 ranks over gloo — parameter broadcast, per-rank shards, ONE sum all-reduce of the flat
    gradient laid out by NetLayout, identical clip+Adam on both ranks."""
    import socket
    script = tmp_path / "dp_worker.py"
    script.write_text(_DP_WORKER)
    with socket.socket() as sock:                      # a port nobody is using right now
        sock.bind(("127.0.0.1", 0))
        port = str(sock.getsockname()[1])
    env = dict(os.environ, NAF_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "DP_OK_0;" in r.stdout and "DP_OK_1;" in r.stdout, r.stdout[-2000:]


_TICK_WORKER = r'''
import os, sys, random, time
sys.path.insert(0, os.environ["NAF_ROOT"])
import torch.distributed as dist
from robotic_manipulator_rloa_amd import parallel
rank, local_rank, world = parallel.init_distributed("gloo")
ag = parallel.TickAgreement.try_create(None, timeout_s=30.0)
assert ag is not None and ag.world == world == 3 and ag.rank == rank
# every rank's votes are known to every rank (seeded by tick and rank): the AND must come out the same everywhere, tick after tick,
# whatever the ranks' relative speed (rank 1 dawdles at random; rank 2 races ahead)
n, got = 4000, []
rnd = random.Random(99)
for t in range(n):
    vote = [random.Random(1000 * t + r).random() < 0.8 for r in range(world)]
    if rank == 1 and rnd.random() < 0.02:
        time.sleep(0.0005)
    assert ag.all_ok(vote[rank]) == all(vote), (t, rank)
ag.close()
dist.barrier(); dist.destroy_process_group()
os.write(1, f"TICK_OK_{rank};".encode())
'''


def test_tick_agreement_three_ranks_gloo(tmp_path):
    """parallel.TickAgreement — the per-tick vote that keeps the ranks of a data-parallel job on the SAME graph of the pipelined
    per-timestep path (an exchange pairs with an exchange) — at world 3 over gloo on the CPU: 4000 ticks, every rank computes the
    AND of all votes, with one rank dawdling at random (the two slots per rank, by the tick's parity, are what keeps a fast peer's
    next vote from overwriting the one a slow rank still has to read)."""
    import socket
    script = tmp_path / "tick_worker.py"
    script.write_text(_TICK_WORKER)
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = str(sock.getsockname()[1])
    env = dict(os.environ, NAF_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3",
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert all(f"TICK_OK_{k};" in r.stdout for k in range(3)), r.stdout[-2000:]


def test_bench_cli_refuses_to_fake_a_multi_gpu_run():
    """bench.py --gpus N on a host with fewer GPUs: an error, never a silent 1-rank line labelled n_gpus=1;
    a launcher world that disagrees with --gpus: an error too. (No GPU is touched: the checks run before any HIP call.)"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NAF_BENCH_REHEARSAL")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "--gpus 2" in r.stderr and "GPU(s)" in r.stderr, r.stderr[-2000:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"],
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
    # the argument surface the driver uses
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse_args(["--gpus", "8", "--steps", "20", "--warmup", "5"])
    assert (a.gpus, a.steps, a.warmup) == (8, 20, 5)
    d = bench.parse_args([])
    assert d.gpus == 1 and d.batch == 256 and d.buffer == 1_000_000 and d.envs == 64 and d.robot == "kuka"
    assert json.dumps({"roofline": None}) and bench.pmc_traffic(-1, -1) == (None, None)


def test_pybullet_environment_adapter_with_fake_simulator(monkeypatch):
    """environment/environment.py (the PyBullet adapter) driven through a kinematic test double of `pybullet`
    (tests/fake_pybullet.py): validation messages, load errors, reset/step protocol, reward / terminal logic; and it
    must agree with the built-in synthetic environment, which models the same chain."""
    import importlib
    import fake_pybullet
    pb, pbd = fake_pybullet.make_modules(n_joints=6)
    monkeypatch.setitem(sys.modules, "pybullet", pb)
    monkeypatch.setitem(sys.modules, "pybullet_data", pbd)
    for m in ("robotic_manipulator_rloa_amd.utils.collision_detector", "robotic_manipulator_rloa_amd.environment.environment"):
        sys.modules.pop(m, None)
    envmod = importlib.import_module("robotic_manipulator_rloa_amd.environment.environment")
    from robotic_manipulator_rloa_amd.environment.synthetic import SyntheticEnvironment
    from robotic_manipulator_rloa_amd.utils.exceptions import InvalidEnvironmentParameter, InvalidManipulatorFile
    C = envmod.EnvironmentConfiguration
    good = dict(endeffector_index=5, fixed_joints=[], involved_joints=[0, 1, 2, 3, 4, 5], target_position=[0.4, 0.85, 0.71],
                obstacle_position=[0.45, 0.55, 0.55], initial_joint_positions=[0.9, 0.45, 0, 0, 0, 0],
                initial_positions_variation_range=None, max_force=200., visualize=False)
    for key, bad, msg in (("endeffector_index", 1.5, "End Effector index received is not an integer"),
                          ("fixed_joints", (1, 2), "Fixed Joints received is not a list"),
                          ("involved_joints", [0, "a"], "An item inside the Involved Joints list is not an integer"),
                          ("target_position", "x", "Target Position received is not a list"),
                          ("obstacle_position", [0, None, 1], "An item inside the Obstacle Position list is not a float"),
                          ("initial_joint_positions", 3, "Initial Joint Positions received is not a list"),
                          ("initial_positions_variation_range", ["a"], "An item inside the Initial Positions Variation Range list is not a float"),
                          ("max_force", "strong", "Maximum Force value received is not a float"),
                          ("visualize", 1, "Visualize value received is not a boolean")):
        with pytest.raises(InvalidEnvironmentParameter, match=msg):
            C(**dict(good, **{key: bad}))
    cfg = C(**good)
    with pytest.raises(InvalidManipulatorFile, match="neither .sdf nor .urdf"):
        envmod.Environment("arm.obj", cfg)
    with pytest.raises(InvalidManipulatorFile):
        envmod.Environment("broken.urdf", cfg)
    with pytest.raises(InvalidManipulatorFile, match="not a string"):
        envmod.Environment(5, cfg)
    env = envmod.Environment("kuka_iiwa/model.sdf", cfg)
    assert ("connect", pb.DIRECT) in pb._state["calls"]
    twin = SyntheticEnvironment(6)
    s, t = env.reset(False), twin.reset(False)
    assert s.shape == (21,) and env.observation_space.shape == (21,) and env.action_space.shape == (6,)
    np.testing.assert_allclose(s, t, atol=1e-6)
    rng = np.random.default_rng(0)
    for _ in range(20):
        a = rng.uniform(-1, 1, 6).astype(np.float32)
        (s2, r, d), (t2, tr, td) = env.step(a), twin.step(a)
        np.testing.assert_allclose(s2, t2, atol=2e-6)
        assert d == td == 0 and abs(r - tr) < 1e-5
    assert env.is_terminal_state() == 0 and env.get_reward() < 0
    # target reached / obstacle hit
    pb._state["q"][:] = 0
    top = [0.0, 0.0, float(0.34 + 0.02 + 0.40 + 0.02 + 0.40 + 0.13)]
    env2 = envmod.Environment("arm.urdf", C(**dict(good, target_position=top, initial_joint_positions=None)))
    env2.reset(False)
    _, r, d = env2.step(np.zeros(6, np.float32))
    assert (r, d) == (250, 1) and env2.get_reward() == 250 and env2.is_terminal_state() == 1
    env3 = envmod.Environment("arm.urdf", C(**dict(good, obstacle_position=[0.0, 0.0, 0.35], initial_joint_positions=None)))
    env3.reset(False)
    _, r, d = env3.step(np.zeros(6, np.float32))
    assert (r, d) == (-1000, 1)
    assert set(env3.get_manipulator_collisions_with_itself().keys()) == {f"joint_{j}" for j in range(6)}
    env3.close()
    assert ("disconnect", 7) in pb._state["calls"]
    # the framework facade builds it when pybullet is importable
    from robotic_manipulator_rloa_amd import ManipulatorFramework
    f = ManipulatorFramework()
    f.initialize_environment("arm.urdf", 5, [], [0, 1, 2, 3, 4, 5], [0.4, 0.85, 0.71], [0.45, 0.55, 0.55], visualize=False)
    assert isinstance(f.env, envmod.Environment)
    f.delete_environment()
    for m in ("robotic_manipulator_rloa_amd.utils.collision_detector", "robotic_manipulator_rloa_amd.environment.environment"):
        sys.modules.pop(m, None)


@pytest.mark.parametrize("robot,n_joints,A", [("kuka", 14, 6), ("xarm6", 14, 6), ("xarm6_robot", 7, 6), ("panda", 12, 7)])
def test_robot_presets_one_table_and_the_joint_index_quirk(robot, n_joints, A, monkeypatch):
    """presets.py is the one table behind the framework's demo environments, the device env presets and bench.py's
    --robot (BASELINE configs[3]: xarm/xarm6_robot.urdf, configs[4]: franka_panda/panda.urdf). Driven through the
    PyBullet adapter on the test double: state size 9 + 2A, actions go to `involved_joints`, and get_state reports joints
    0 .. A-1 whatever `involved_joints` says — the reference's quirk (environment/environment.py:442-444)."""
    import importlib
    import fake_pybullet
    from robotic_manipulator_rloa_amd import presets
    from robotic_manipulator_rloa_amd.rl_framework import _DEMO_ENVS
    assert set(presets.ROBOT_PRESETS) == {"kuka", "xarm6", "xarm6_robot", "panda"} == set(_DEMO_ENVS)
    assert presets.ROBOT_PRESETS["xarm6_robot"]["manipulator_file"] == "xarm/xarm6_robot.urdf"
    assert presets.ROBOT_PRESETS["panda"]["manipulator_file"] == "franka_panda/panda.urdf"
    kw = presets.pybullet_arguments(robot)
    assert kw == _DEMO_ENVS[robot] and len(kw["involved_joints"]) == A == presets.action_size(robot)
    assert not set(kw["involved_joints"]) & set(kw["fixed_joints"]) and max(kw["involved_joints"] + kw["fixed_joints"] +
                                                                           [kw["endeffector_index"]]) < n_joints
    dp = presets.device_env_preset(robot)
    assert len(dp) == 14 and dp[8:11] == kw["target_position"] and dp[11:14] == kw["obstacle_position"]
    assert dp[:A] == [float(x) for x in kw["initial_joint_positions"][:A]]
    pb, pbd = fake_pybullet.make_modules(n_joints=n_joints)
    monkeypatch.setitem(sys.modules, "pybullet", pb)
    monkeypatch.setitem(sys.modules, "pybullet_data", pbd)
    for m in ("robotic_manipulator_rloa_amd.utils.collision_detector", "robotic_manipulator_rloa_amd.environment.environment"):
        sys.modules.pop(m, None)
    try:
        importlib.import_module("robotic_manipulator_rloa_amd.environment.environment")
        from robotic_manipulator_rloa_amd import ManipulatorFramework
        f = ManipulatorFramework()
        f._demo_environment(robot, "pybullet", None, visualize=False)
        env = f.env
        s = env.reset(False)
        S = 9 + 2 * A
        assert s.shape == (S,) and env.observation_space.shape == (S,) and env.action_space.shape == (A,)
        init = [float(x) for x in kw["initial_joint_positions"]]
        np.testing.assert_allclose(s[:A], init[:A], atol=1e-6)        # value k went to joint index k; joints 0..A-1 reported
        np.testing.assert_allclose(s[2 * A + 3:2 * A + 6], kw["target_position"])
        np.testing.assert_allclose(s[2 * A + 6:], kw["obstacle_position"])
        a = np.linspace(-1, 1, A).astype(np.float32)
        s2, r, d = env.step(a)
        moved = np.zeros(n_joints, np.float32)
        moved[kw["involved_joints"]] = a / 240.0                        # velocity control on the INVOLVED joints, one tick
        full0 = np.zeros(n_joints, np.float32)
        full0[:len(init)] = init
        full0[kw["fixed_joints"]] = 0.0                                  # fixed joints are held at 0 (environment.py:470)
        np.testing.assert_allclose(s2[:A], (full0 + moved)[:A], atol=1e-6)   # ... but joints 0..A-1 are what is observed
        assert np.isfinite(r) and d in (0, 1)
        f.delete_environment()
    finally:
        for m in ("robotic_manipulator_rloa_amd.utils.collision_detector", "robotic_manipulator_rloa_amd.environment.environment"):
            sys.modules.pop(m, None)


def test_framework_misc_api_contract(tmp_path, monkeypatch):
    """Small pieces of the ManipulatorFramework surface the reference's tests pin (tests/.../test_rl_framework.py):
    log-level setter, plot_training_rewards' missing-file error and its block means, delete_* on empty state."""
    import json
    import logging
    from robotic_manipulator_rloa_amd import ManipulatorFramework
    from robotic_manipulator_rloa_amd.utils.logger import get_global_logger
    monkeypatch.chdir(tmp_path)
    f = ManipulatorFramework()
    f.set_log_level(10)
    assert get_global_logger().level == 10
    f.get_required_hyperparameters()
    f.set_log_level(15)                                    # invalid: level unchanged
    assert get_global_logger().level == 10
    f.set_log_level(20)
    with pytest.raises(FileNotFoundError):
        f.plot_training_rewards(7)
    os.makedirs("checkpoints/4")
    with open("checkpoints/4/scores.txt", "w") as fh:
        fh.write(json.dumps({str(i): (float(i), 10) for i in range(1, 9)}))
    import types
    shown = {}
    fake_plt = types.ModuleType("matplotlib.pyplot")
    fake_plt.plot = lambda x, y: shown.update(x=list(x), y=list(y))
    fake_plt.show = lambda: None
    monkeypatch.setitem(sys.modules, "matplotlib", types.ModuleType("matplotlib"))
    monkeypatch.setitem(sys.modules, "matplotlib.pyplot", fake_plt)
    f.plot_training_rewards(4, mean_range=4)
    assert shown["y"] == [2.5, 6.5] and shown["x"] == [0, 1]   # means of episodes 1-4 and 5-8
    f.delete_environment()
    f.delete_naf_agent()
    f.get_environment_configuration()
    f.get_nafagent_configuration()
    assert f.env is None and f.naf_agent is None


def _make_synth_env():
    from robotic_manipulator_rloa_amd.environment.synthetic import SyntheticEnvironment
    return SyntheticEnvironment(6, initial_positions_variation_range=[0.1] * 6)


def test_host_vector_env_replaces_a_dead_and_a_hung_worker(tmp_path):
    """SURVEY.md section 5's build note ("env worker crash -> respawn & mark transitions invalid"), VERDICT r05 item 6: a worker
    killed in front of a step (a simulator that segfaults) and a worker whose env.step never returns are both replaced by a fresh
    process with fresh environments — the step reports their envs' transitions as invalid, pack_rows packs the others only, the
    next observation of the replaced env is a reset state, every other env is untouched, and the vector env goes on."""
    import functools
    from robotic_manipulator_rloa_amd.environment.vector_env import HostVectorEnv
    from scripted_env import ScriptedEnvironment, make_hanging, make_scripted
    E, S, A = 4, 21, 6
    acts = np.zeros((E, A), np.float32)
    vec = HostVectorEnv(functools.partial(make_scripted, 0), E, S, A, max_frames=50, seed=0)
    try:
        obs = vec.reset().copy()
        assert (obs[:, 1] == 0).all() and (obs[:, 2] == 0).all()         # (offset, k, t) = (0, 0, 0) everywhere
        for t in range(1, 3):
            _, _, rw, ns, dn, obs = vec.step(acts)
            assert (ns[:, 2] == t).all() and vec.arr["valid"].all()
        vec._fault_at = (vec.steps, 2)                                   # worker 2 (env 2) dies in front of the third step
        st, _, rw, ns, dn, obs = vec.step(acts)
        assert list(vec.arr["valid"]) == [1, 1, 0, 1] and vec.respawned_envs == [2] and vec.respawns == 1 and vec.dropped_transitions == 1
        assert (ns[[0, 1, 3], 2] == 3).all() and (dn[[0, 1, 3]] == 1).all()      # the others: episode 0 ends at its third step
        assert tuple(obs[2][:3]) == (0, 0, 0) and vec.arr["episode_end"][2] == 0   # env 2: a fresh environment's reset state
        rows = np.zeros((E, 64), np.float32)
        assert vec.pack_rows(rows, 28) == 3 and list(rows[:3, 28 + 2]) == [3, 3, 3] and (rows[3] == 0).all()
        _, _, rw, ns, dn, obs = vec.step(acts)
        assert vec.arr["valid"].all() and vec.respawned_envs == [] and vec.pack_rows(rows, 28) == E
        assert tuple(ns[2][:3]) == (0, 0, 1) and tuple(ns[0][:3]) == (0, 1, 1)     # env 2 is one episode behind from here on
    finally:
        vec.close()
    marker = str(tmp_path / "hung_once")
    vec = HostVectorEnv(functools.partial(make_hanging, 0, 3, marker), 2, S, A, max_frames=50, seed=0, step_timeout_s=1.5)
    try:
        vec.reset()
        acts2 = np.zeros((2, A), np.float32)
        vec.step(acts2), vec.step(acts2)
        t0 = time.time()
        vec.step(acts2)                                                  # the first-built env hangs in its third step
        assert 1.0 < time.time() - t0 < 30 and vec.respawns == 1 and len(vec.respawned_envs) == 1
        lost = vec.respawned_envs[0]
        assert vec.arr["valid"][lost] == 0 and vec.arr["valid"][1 - lost] == 1
        for _ in range(4):
            vec.step(acts2)                                              # ... and its replacement does not
        assert vec.respawns == 1 and vec.arr["valid"].all()
    finally:
        vec.close()


def test_host_vector_env_workers_match_serial_envs():
    """HostVectorEnv (worker processes + shared memory) against the same envs stepped serially in this process: same
    transitions, per-env auto-reset on the frame budget, row packing in the HBM layout."""
    import random
    from robotic_manipulator_rloa_amd.environment.vector_env import HostVectorEnv
    E, S, A = 6, 21, 6
    vec = HostVectorEnv(_make_synth_env, E, S, A, envs_per_worker=2, max_frames=5, seed=3)
    try:
        obs = vec.reset().copy()
        # serial twin: worker w seeds Python's RNG with seed + first env index, then resets its envs in order
        twins = []
        for first in range(0, E, 2):
            random.seed(3 + first)
            envs = [_make_synth_env() for _ in range(2)]
            states = [e.reset(False) for e in envs]
            twins.append((first, envs, states))
        for first, envs, states in twins:
            for j in range(2):
                np.testing.assert_allclose(obs[first + j], states[j], atol=1e-12)
        rng = np.random.default_rng(0)
        cur = obs.copy()
        for t in range(7):
            actions = rng.uniform(-1, 1, (E, A)).astype(np.float32)
            st, ac, rw, ns, dn, nxt = vec.step(actions)
            np.testing.assert_array_equal(st, cur)
            np.testing.assert_array_equal(ac, actions)
            # the env's own dynamics (velocity control for one tick)
            np.testing.assert_allclose(ns[:, :A], st[:, :A] + actions / 240.0, atol=1e-6)
            rows = np.zeros((E, 64), np.float32)
            vec.pack_rows(rows, 28)
            exp = O.pack_rows(st.astype(np.float32), ac, rw.astype(np.float32), ns.astype(np.float32), dn.astype(np.float32), 64)
            np.testing.assert_array_equal(rows, exp)
            if t == 4:                                       # frame budget of 5 reached: every env was reset
                assert vec.arr["episode_end"].sum() == E and vec.episodes_finished == E
                assert not np.allclose(nxt, ns)               # next observation = reset state, stored next_state = true one
            elif t < 4:
                np.testing.assert_array_equal(nxt, ns)
            cur = nxt.copy()
    finally:
        vec.close()
    assert all(not p.is_alive() for p in vec._procs)


def test_episode_ledger_writes_what_run_writes(tmp_path, monkeypatch):
    """engine.EpisodeLedger = the per-episode bookkeeping of NAFAgent.run (naf_algorithm.py:241, :266, :273-289) for loops
    whose episodes finish in any order: dict pre-filled with (0, 0), numbered in completion order, checkpoint every
    checkpoint_frequency episodes (weights.p + the WHOLE dict as scores.txt), model.p at the end, rank 0 only; episodes
    beyond the budget are counted, not recorded."""
    import json
    from robotic_manipulator_rloa_amd.engine import EPISODE_RECORD, EpisodeLedger
    from robotic_manipulator_rloa_amd.naf_components.naf_algorithm import NAFAgent
    monkeypatch.chdir(tmp_path)
    assert EPISODE_RECORD.itemsize == 32 and EPISODE_RECORD.fields["frames"][1] == 8 and EPISODE_RECORD.fields["env"][1] == 28
    calls = []

    def sd():
        calls.append(1)
        return {"w": torch.full((2,), float(len(calls)))}
    led = EpisodeLedger(5, 2, sd)
    assert led.scores == {e: (0, 0) for e in range(1, 6)} and not led.complete
    for k in range(7):
        led.add(-1.5 * k, 10 + k)
    assert led.complete and led.count == 5 and led.extra == 2 and led.checkpoints == [2, 4]
    assert led.scores[5] == (-6.0, 14) and led.scores[1] == (0.0, 10)
    saved = json.loads(open("checkpoints/2/scores.txt").read())
    assert saved == {"1": [0.0, 10], "2": [-1.5, 11], "3": [0, 0], "4": [0, 0], "5": [0, 0]}
    assert torch.load("checkpoints/4/weights.p")["w"].tolist() == [2.0, 2.0]
    assert led.finish() is led.scores and torch.load("model.p")["w"].tolist() == [3.0, 3.0]
    # a rank other than 0 keeps the dict and writes nothing; open-ended ledgers grow
    os.chdir(tmp_path / "checkpoints")
    led2 = EpisodeLedger(None, 1, sd, write=False)
    led2.add(1.0, 3)
    led2.add(2.0, 4)
    led2.finish()
    assert led2.scores == {1: (1.0, 3), 2: (2.0, 4)} and not os.path.exists("checkpoints") and not os.path.exists("model.p")
    # evaluation quotas: every env contributes its FIRST q_e episodes, sum = n_episodes
    assert NAFAgent._episode_quota(70, 64).tolist() == [2] * 6 + [1] * 58
    assert NAFAgent._episode_quota(3, 8).tolist() == [1, 1, 1, 0, 0, 0, 0, 0]
    assert NAFAgent._episode_quota(128, 64).sum() == 128


def test_framework_many_env_arguments_without_a_gpu(monkeypatch):
    """n_envs plumbing of ManipulatorFramework that needs no device: validation, the synthetic env's configuration in
    the device kernel's preset layout, a picklable factory of copies of the configured environment."""
    import pickle
    from robotic_manipulator_rloa_amd import ManipulatorFramework
    from robotic_manipulator_rloa_amd.utils.exceptions import ConfigurationIncomplete, InvalidNAFAgentParameter
    f = ManipulatorFramework()
    f.initialize_synthetic_environment(6, [0.3, 0.47, 0.61], [0.25, 0.27, 0.5], [0., 1., 0., -2.3, 0., 0.], [0, 0, 0, 0.3, 1, 1])
    with pytest.raises(InvalidNAFAgentParameter):
        f.initialize_naf_agent(n_envs=0)
    with pytest.raises(InvalidNAFAgentParameter):
        f.initialize_naf_agent(n_envs=2.5)
    kw = f._device_env_arguments()
    np.testing.assert_allclose(kw["preset"], [0., 1., 0., -2.3, 0., 0., 0., 0., 0.3, 0.47, 0.61, 0.25, 0.27, 0.5], rtol=1e-6)
    np.testing.assert_allclose(kw["variation"], [0, 0, 0, 0.3, 1, 1, 0, 0], rtol=1e-6)
    env2 = pickle.loads(pickle.dumps(f._env_factory))()
    np.testing.assert_array_equal(env2.target_pos, f.env.target_pos)
    np.testing.assert_array_equal(env2.initial_positions_variation_range, f.env.initial_positions_variation_range)
    with pytest.raises(ConfigurationIncomplete):
        f.run_training(3, 10, n_envs=4)          # no agent yet
    f.delete_environment()
    assert f._env_factory is None


def test_engine_classes_stay_small():
    """engine.TrainChunk was five machines in one class (chunked, teacher-forced, fused, prefetching,
    pipelined) and the round's three state-handling bugs were all in it. Since round 6: UpdateChunk (chunked / teacher-forced),
    TimestepGraph (the per-timestep forms), _Pipeline (the pipelined form's sets, verdicts and graphs), HostStoreRow — none above
    300 lines, and the three A/B switches are ONE (NAF_STEP_FORM, tests only)."""
    import ast
    src = open(os.path.join(ROOT, "robotic_manipulator_rloa_amd", "engine.py")).read()
    sizes = {n.name: n.end_lineno - n.lineno + 1 for n in ast.parse(src).body if isinstance(n, ast.ClassDef)}
    assert {"UpdateChunk", "TimestepGraph", "_Pipeline", "HostStoreRow", "DeviceEnvLoop", "EpisodeLedger"} <= set(sizes), sizes
    assert max(sizes.values()) <= 300, sizes
    assert "NAF_STEP_FUSED" not in src and "NAF_STEP_PREFETCH" not in src and "NAF_STEP_PIPELINE" not in src


def test_no_kernel_spills_to_scratch_and_the_switch_list_is_short():
    """nothing in libnaf_hip.so may keep values in scratch memory (private_segment_fixed_size, as hipcc's
    -Rpass-analysis=kernel-resource-usage reports it for every kernel at build time -> csrc/libnaf_hip.so.usage.json), and the
    product reads at most 13 NAF_* environment switches — the ones DESIGN.md section 4c documents."""
    import json
    from robotic_manipulator_rloa_amd import _lib
    _lib.load()
    if not os.path.exists(_lib.USAGE_PATH):
        _lib.build_library(force=True)
    usage = json.load(open(_lib.USAGE_PATH))
    assert len(usage) >= 60, "resource report looks truncated"
    spilling = {k: v for k, v in usage.items() if v.get("scratch_bytes_per_lane", 0) or v.get("vgpr_spills", 0)}
    assert not spilling, spilling
    # the kernels a timestep's HOST waits on, and the launches of every update's chain, keep their scalars in
    # scalar registers — at most 32 SGPRs spilled to VGPR lanes (round 5's adam_act_kernel with the prefetch riding in it: 171 - 196;
    # since round 6 the pipelined timestep's first launch is the variant without that workgroup: 4 / 12). The prefetch body itself
    # (step_prep_kernel, step_prefetch_kernel, adam_act_kernel<.., 1..4>: ~200 spilled SGPRs, no scratch) runs beside the graph.
    hot = {k: v["sgpr_spills"] for k, v in usage.items()
           if (("adam_act_kernelILi0ELi0E" in k or "adam_act_kernelILi1ELi0E" in k) or
               any(n in k for n in ("bb_layer1_kernel", "bb_linear_stats", "bb_layer2_head_kernel", "gemm_bundle_kernel",
                                    "bb_layer1_bwd_finish_kernel", "replay_gather_rows_kernel", "policy_act_kernel",
                                    "adam_polyak_kernel", "xgmi_allreduce_kernel")))}
    assert len(hot) >= 20
    # (the launches the presets run — Hadamard head, up to 8 joints, layer size 256, with or without layer 1 riding — stay within 32;
    #  the variants with the 16-lane noise group, the 512-wide rows or the L L^T head AND the riding layer 1's 25 scalar arguments reach
    #  46, spilled to VGPR lanes — never to scratch, asserted above)
    preset = {k: v for k, v in hot.items() if not re.search(r"adam_act_kernelILi\dELi0ELi\d+ELi\d+ELi[68]E", k) or
              re.search(r"adam_act_kernelILi0ELi0ELi8ELi256ELi[68]E", k)}
    assert max(preset.values()) <= 32, {k: v for k, v in preset.items() if v > 32}
    assert max(hot.values()) <= 48, {k: v for k, v in hot.items() if v > 48}
    # the kernels that take the draw's table / the minibatch as DYNAMIC LDS (csrc/step_path.hip, SP_MAX_DYN_LDS) must fit a CU's 160 KB with
    # their own arrays beside it: hipFuncSetAttribute refuses the limit otherwise — at run time, on the first per-timestep launch
    src_sp = open(os.path.join(ROOT, "robotic_manipulator_rloa_amd", "csrc", "step_path.hip")).read()
    dyn = 1024 * int(re.search(r"#define SP_MAX_DYN_LDS \((\d+) \* 1024\)", src_sp).group(1))
    taking = {k: v["lds_bytes"] for k, v in usage.items()
              if "step_prep_kernel" in k or "step_prefetch_kernel" in k or re.search(r"adam_act_kernelILi\dELi[1-4]E", k)}
    assert len(taking) >= 24 and dyn >= 82176 + 1024 + 65536 and max(taking.values()) + dyn <= 160 * 1024, (dyn, max(taking.values()))
    for need in ("gemm_bundle_kernel", "bb_layer2_head_kernel", "replay_gather_rows_kernel", "naf_head_kernel", "adam_polyak_kernel",
                 "synth_env_step_kernel", "policy_act_kernel", "xgmi_allreduce_kernel"):
        assert any(need in k for k in usage), need
    pkg = os.path.join(ROOT, "robotic_manipulator_rloa_amd")
    names = set()
    for dirpath, _, files in os.walk(pkg):
        if os.sep + "build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                names |= set(re.findall(r'(?:environ\.get\(|getenv\(|environ\[|NAF_ENV_INT\()\s*["\'](NAF_[A-Z0-9_]+)["\']', src))
    documented = {"NAF_FUSE", "NAF_DEFER_ADAM", "NAF_XGMI", "NAF_XGMI_FREE_SLAB", "NAF_BLAS_DEFAULT", "NAF_BLAS_TUNING_FILE",
                  "NAF_BUILD_DEFINES", "NAF_LOG_FILE", "NAF_DP_SHARE_GPU", "NAF_DP_EXCHANGE", "NAF_STEP_FORM", "NAF_HOST_STORE",
                  "NAF_STEP_L1_RIDE"}
    assert names <= documented, names - documented
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    for n in names:
        assert n in design, f"{n} is read by the product but not documented in DESIGN.md"
