"""Deterministic synthetic transitions shared by make_golden.py (reference side) and the parity tests
(build side). Only numpy PCG64 with fixed call order — never stored, regenerated on both sides.
Value ranges follow SURVEY.md §8(d): states ~ N(0,1) clipped to +-pi, actions U(-1,1) with 10 % saturated
to +-1 (the clamp of naf_neural_network.py:121), rewards -|.| in [-1.5, 0] with 0.5 % of +250 / -1000,
done with p = 0.005."""
import numpy as np


def make_transitions(n: int, S: int, A: int, seed: int, rare_events: bool = True, structured_reward: bool = False):
    rng = np.random.Generator(np.random.PCG64(seed))
    states = np.clip(rng.standard_normal((n, S)), -np.pi, np.pi).astype(np.float32)
    next_states = np.clip(states + 0.05 * rng.standard_normal((n, S)), -np.pi, np.pi).astype(np.float32)
    actions = rng.uniform(-1.0, 1.0, (n, A)).astype(np.float32)
    sat = rng.random((n, A)) < 0.10
    actions = np.where(sat, np.sign(actions), actions).astype(np.float32)
    rewards = (-1.5 * rng.random(n)).astype(np.float32)
    ev = rng.random(n)
    if structured_reward:
        # learnable signal, like the real env: -(distance between the next state's end-effector and target slots)
        # (environment/environment.py:366-371 with the state layout of :449-451)
        ee, tgt = next_states[:, 2 * A:2 * A + 3], next_states[:, 2 * A + 3:2 * A + 6]
        rewards = (-0.5 * np.linalg.norm(ee - tgt, axis=1)).astype(np.float32)
    if not rare_events:
        return states, actions, rewards, next_states, np.zeros(n, np.float32)
    rewards = np.where(ev < 0.0025, np.float32(250.0), rewards)
    rewards = np.where((ev >= 0.0025) & (ev < 0.005), np.float32(-1000.0), rewards).astype(np.float32)
    dones = (ev < 0.005).astype(np.float32)
    return states, actions, rewards, next_states, dones


def batch_indices(n_rows: int, B: int, n_updates: int, seed: int) -> np.ndarray:
    """Teacher-forced minibatch positions: without replacement inside a minibatch (random.sample semantics)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = np.empty((n_updates, B), dtype=np.int32)
    for k in range(n_updates):
        out[k] = rng.choice(n_rows, size=B, replace=False)
    return out
