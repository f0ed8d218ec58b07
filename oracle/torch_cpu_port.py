"""
oracle/torch_cpu_port.py — TEST / BASELINE INFRASTRUCTURE ONLY (never imported by the product).

A torch-CPU restatement of the reference's per-timestep path, issuing the same torch ops in the same order as
the reference does, so that its wall time on the GPU box's host cores is a fair stand-in for the reference's
(whose files cannot travel to the GPU box). bench.py times it as `cpu_baseline` (kind "port").

Restated, with the reference lines each block follows:
  ReplayBuffer add/sample .......... utils/replay_buffer.py:32-67   (deque + random.sample + np.stack + .long())
  NAF.forward (train + eval) ....... naf_components/naf_neural_network.py:76-123 (incl. the MultivariateNormal
                                     draw that every forward performs, even inside learn())
  NAFAgent.act / step / learn ...... naf_components/naf_algorithm.py:129-215
  NAFAgent.soft_update ............. naf_components/naf_algorithm.py:217-226

Pinned by tests/test_oracle_golden.py::test_torch_port_matches_reference_golden against G3 (losses of 5
consecutive learn() calls of the unmodified reference, rtol 1e-5), and its timing against the reference itself
is recorded in DESIGN.md.
"""
from __future__ import annotations

import math
import random
from collections import deque, namedtuple

import numpy as np
import torch
import torch.nn.functional as F
from torch.distributions import MultivariateNormal

KEYS = ["input_layer", "hidden_layer", "action_values", "value", "matrix_entries"]


def init_state_dict(S: int, A: int, H: int, seed: int) -> dict:
    """Same draw order as the reference constructor (naf_neural_network.py:33-54): manual_seed, then the five
    Linear layers in declaration order (BatchNorm draws nothing)."""
    torch.manual_seed(seed)
    sd = {}

    def linear(name, fan_in, fan_out):
        w = torch.empty(fan_out, fan_in)
        torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        bound = 1 / math.sqrt(fan_in)
        b = torch.empty(fan_out).uniform_(-bound, bound)
        sd[f"{name}.weight"], sd[f"{name}.bias"] = w, b

    def bn(name):
        sd[f"{name}.weight"], sd[f"{name}.bias"] = torch.ones(H), torch.zeros(H)
        sd[f"{name}.running_mean"], sd[f"{name}.running_var"] = torch.zeros(H), torch.ones(H)
        sd[f"{name}.num_batches_tracked"] = torch.tensor(0)

    linear("input_layer", S, H)
    bn("bn1")
    linear("hidden_layer", H, H)
    bn("bn2")
    linear("action_values", H, A)
    linear("value", H, 1)
    linear("matrix_entries", H, A * (A + 1) // 2)
    return sd


class Net:
    PARAMS = ["input_layer.weight", "input_layer.bias", "bn1.weight", "bn1.bias", "hidden_layer.weight",
              "hidden_layer.bias", "bn2.weight", "bn2.bias", "action_values.weight", "action_values.bias",
              "value.weight", "value.bias", "matrix_entries.weight", "matrix_entries.bias"]

    def __init__(self, sd: dict, A: int):
        self.A = A
        self.p = {k: torch.as_tensor(np.asarray(sd[k]), dtype=torch.float32).clone().requires_grad_(True) for k in self.PARAMS}
        self.buf = {k: torch.as_tensor(np.asarray(sd[k]), dtype=torch.float32).clone()
                    for k in ("bn1.running_mean", "bn1.running_var", "bn2.running_mean", "bn2.running_var")}
        self.training = True

    def parameters(self):
        return [self.p[k] for k in self.PARAMS]

    def forward(self, x, action=None):
        p, A = self.p, self.A
        h = F.linear(x, p["input_layer.weight"], p["input_layer.bias"])
        h = torch.relu(F.batch_norm(h, self.buf["bn1.running_mean"], self.buf["bn1.running_var"], p["bn1.weight"],
                                    p["bn1.bias"], self.training, 0.1, 1e-5))
        h = F.linear(h, p["hidden_layer.weight"], p["hidden_layer.bias"])
        h = torch.relu(F.batch_norm(h, self.buf["bn2.running_mean"], self.buf["bn2.running_var"], p["bn2.weight"],
                                    p["bn2.bias"], self.training, 0.1, 1e-5))
        mu = torch.tanh(F.linear(h, p["action_values.weight"], p["action_values.bias"]))
        entries = torch.tanh(F.linear(h, p["matrix_entries.weight"], p["matrix_entries.bias"]))
        V = F.linear(h, p["value.weight"], p["value.bias"])
        mu = mu.unsqueeze(-1)
        L = torch.zeros((x.shape[0], A, A))
        tri = torch.tril_indices(row=A, col=A, offset=0)
        L[:, tri[0], tri[1]] = entries
        L.diagonal(dim1=1, dim2=2).exp_()
        P = L * L.transpose(2, 1)                      # elementwise, as in the reference (:104)
        Q = None
        if action is not None:
            d = action.unsqueeze(-1) - mu
            Q = (-0.5 * torch.matmul(torch.matmul(d.transpose(2, 1), P), d)).squeeze(-1) + V
        noisy = MultivariateNormal(mu.squeeze(-1), torch.inverse(P)).sample()
        return torch.clamp(noisy, min=-1, max=1), Q, V


class TorchCpuAgent:
    def __init__(self, S, A, H, batch_size, buffer_size, lr=1e-3, tau=1e-3, gamma=0.99, update_freq=1, num_updates=1,
                 seed=0, state_dict=None):
        random.seed(seed)
        sd = state_dict if state_dict is not None else init_state_dict(S, A, H, seed)
        self.main, self.target = Net(sd, A), Net(sd, A)
        self.opt = torch.optim.Adam(self.main.parameters(), lr=lr)
        self.memory = deque(maxlen=buffer_size)
        # namedtuple records with the reference's field names (utils/replay_buffer.py:27, :43-45): attribute access and
        # record size are what its sample() pays for
        self.experience = namedtuple("Experience", field_names=["state", "action", "reward", "next_state", "done"])
        self.batch_size, self.tau, self.gamma = batch_size, tau, gamma
        self.update_freq, self.num_updates, self.t = update_freq, num_updates, 0
        self.losses = []

    def add(self, s, a, r, s2, d):
        self.memory.append(self.experience(s, a, r, s2, d))

    def sample(self):
        ex = random.sample(self.memory, k=self.batch_size)
        dev = torch.device("cpu")
        states = torch.from_numpy(
            np.stack([e.state if not isinstance(e.state, tuple) else e.state[0] for e in ex])).float().to(dev)
        actions = torch.from_numpy(np.vstack([e.action for e in ex if e is not None])).long().to(dev)
        rewards = torch.from_numpy(np.vstack([e.reward for e in ex if e is not None])).float().to(dev)
        next_states = torch.from_numpy(np.stack([e.next_state for e in ex if e is not None])).float().to(dev)
        dones = torch.from_numpy(np.vstack([e.done for e in ex if e is not None]).astype(np.uint8)).float().to(dev)
        return states, actions, rewards, next_states, dones

    def act(self, state):
        x = torch.from_numpy(state).float()
        self.main.training = False
        with torch.no_grad():
            a, _, _ = self.main.forward(x.unsqueeze(0))
        self.main.training = True
        return a.cpu().squeeze().numpy()

    def learn(self, ex):
        self.opt.zero_grad()
        states, actions, rewards, next_states, _dones = ex
        with torch.no_grad():
            _, _, v_next = self.target.forward(next_states)       # target net stays in training mode
        y = rewards + self.gamma * v_next
        _, q, _ = self.main.forward(states, actions)
        loss = F.mse_loss(q, y)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(self.main.parameters(), 1)
        self.opt.step()
        for tp, mp in zip(self.target.parameters(), self.main.parameters()):
            tp.data.copy_(self.tau * mp.data + (1. - self.tau) * tp.data)
        self.losses.append(float(loss.detach()))

    def step(self, s, a, r, s2, d):
        self.add(s, a, r, s2, d)
        self.t = (self.t + 1) % self.update_freq
        if self.t == 0 and len(self.memory) > self.batch_size:
            for _ in range(self.num_updates):
                self.learn(self.sample())


def time_baseline(S=21, A=6, H=256, B=256, N=1_000_000, fill=None, budget_s=15.0, seed=0, threads=None):
    """Times the restated per-timestep path (act + add + sample + learn) and learn() alone on this host.
    Returns a dict; `fill` rows are pre-loaded into the deque (default: full buffer)."""
    import os
    import time
    if threads:
        torch.set_num_threads(threads)
    rng = np.random.default_rng(seed)
    agent = TorchCpuAgent(S, A, H, B, N, seed=seed)
    fill = N if fill is None else fill
    # every stored transition owns its arrays (views of two big blocks), as in a real run where each env state is a
    # fresh ndarray: deque + random.sample then see the same pointer-chasing the reference does
    st = rng.standard_normal((fill + 4097, S))
    ac = rng.uniform(-1, 1, (fill + 4097, A)).astype(np.float32)
    for i in range(fill):
        agent.add(st[i], ac[i], -0.5, st[i + 1], 0)
    # learn() only
    ex = agent.sample()
    for _ in range(3):
        agent.learn(ex)
    t0, n_learn = time.perf_counter(), 0
    while time.perf_counter() - t0 < budget_s * 0.3:
        agent.learn(ex)
        n_learn += 1
    t_learn = (time.perf_counter() - t0) / n_learn
    # sample() only (replay_buffer.py:47-67): random.sample over a deque walks O(N) pointers per draw — memory latency, the
    # part of the timestep that differs most from host to host (a 256-cpu server's L3 swallows most of a 1e6-entry deque)
    t0, n_sample = time.perf_counter(), 0
    while time.perf_counter() - t0 < min(2.0, budget_s * 0.1) or n_sample < 3:
        agent.sample()
        n_sample += 1
    t_sample = (time.perf_counter() - t0) / n_sample
    # full timestep
    s = st[fill]
    t0, n_step = time.perf_counter(), 0
    while time.perf_counter() - t0 < budget_s * 0.7:
        a = agent.act(s)
        s2 = st[fill + ((n_step + 1) & 4095)]
        agent.step(s, a, -0.5, s2, 0)
        s = s2
        n_step += 1
    t_step = (time.perf_counter() - t0) / n_step
    return {"steps_per_s": 1.0 / t_step, "learn_updates_per_s": 1.0 / t_learn, "sample_ms": 1e3 * t_sample, "n_steps": n_step, "n_learn": n_learn,
            "threads": torch.get_num_threads(), "host_cpus": os.cpu_count(), "B": B, "N": N, "fill": fill}
