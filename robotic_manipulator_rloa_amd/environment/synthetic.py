"""Host-side synthetic manipulator environment with the reference Environment's protocol
(environment/environment.py: reset(verbose) -> state[S]; step(action) -> (state, reward, done);
observation_space / action_space as zero arrays, :259-262, :264-309, :453-485).

It is the numpy twin of csrc/synth_env.hip (same kinematic chain, same constants), NOT a PyBullet port: PyBullet
is absent from the image, its dynamics have no pinned version upstream, and the simulator is outside the
accelerated hot path. It exists so NAFAgent.run()/test_trained_model() can be exercised end to end, and so the
device env kernel has a CPU statement to be checked against.
  state  = [q(A), qdot(A), end-effector xyz, target xyz, obstacle xyz]        (environment.py:449-451)
  reward = +250 reached (dist < 0.05) | -1000 obstacle contact | -(dist - 0.05)   (environment.py:345-371, :419-429)
"""
from __future__ import annotations

import random
from typing import List, Optional, Tuple

import numpy as np

LINKS = np.array([0.34, 0.02, 0.40, 0.02, 0.40, 0.13, 0.05, 0.05], dtype=np.float32)
DT = np.float32(1.0 / 240.0)
OBSTACLE_RADIUS = np.float32(0.06)
TARGET_THRESHOLD = np.float32(0.05)


def forward_kinematics(q: np.ndarray, obstacle: np.ndarray) -> Tuple[np.ndarray, bool]:
    """End-effector position of the alternating z/y revolute chain and whether any joint frame origin lies inside
    the obstacle sphere. float32 throughout, same operation order as fk_chain() in csrc/synth_env.hip."""
    R = np.eye(3, dtype=np.float32)
    p = np.zeros(3, dtype=np.float32)
    hit = False
    for k in range(len(q)):
        c, s = np.float32(np.cos(np.float32(q[k]))), np.float32(np.sin(np.float32(q[k])))
        N = np.empty_like(R)
        if k % 2 == 0:
            N[:, 0] = R[:, 0] * c + R[:, 1] * s
            N[:, 1] = -R[:, 0] * s + R[:, 1] * c
            N[:, 2] = R[:, 2]
        else:
            N[:, 0] = R[:, 0] * c - R[:, 2] * s
            N[:, 1] = R[:, 1]
            N[:, 2] = R[:, 0] * s + R[:, 2] * c
        R = N
        p = p + R[:, 2] * LINKS[k]
        d = p - obstacle
        hit = hit or bool(np.dot(d, d) < OBSTACLE_RADIUS * OBSTACLE_RADIUS)
    return p.astype(np.float32), hit


_LINKS_F = [float(x) for x in LINKS]
_OBST_R2 = float(OBSTACLE_RADIUS) ** 2


def _fk_fast(q, obstacle):
    """forward_kinematics() on Python floats (double precision, one pass, no array temporaries): the same chain, 5 x faster
    for the 6-8 joints of an arm than ~100 small numpy calls — env.step was the largest single cost of a timestep of the
    reference-API path (75 of 165 us, benchmarks/host_api_breakdown.py). Agrees with the float32 forms (numpy above, device
    kernel) to their rounding: tests compare them at 2e-6."""
    from math import cos, sin
    r00, r01, r02, r10, r11, r12, r20, r21, r22 = 1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0
    px = py = pz = 0.0
    ox, oy, oz = obstacle
    hit = False
    for k, qk in enumerate(q):
        c, s = cos(qk), sin(qk)
        if k % 2 == 0:      # about z: columns 0, 1 mix
            r00, r01 = r00 * c + r01 * s, -r00 * s + r01 * c
            r10, r11 = r10 * c + r11 * s, -r10 * s + r11 * c
            r20, r21 = r20 * c + r21 * s, -r20 * s + r21 * c
        else:               # about y: columns 0, 2 mix
            r00, r02 = r00 * c - r02 * s, r00 * s + r02 * c
            r10, r12 = r10 * c - r12 * s, r10 * s + r12 * c
            r20, r22 = r20 * c - r22 * s, r20 * s + r22 * c
        l = _LINKS_F[k]
        px += r02 * l
        py += r12 * l
        pz += r22 * l
        dx, dy, dz = px - ox, py - oy, pz - oz
        if dx * dx + dy * dy + dz * dz < _OBST_R2:
            hit = True
    return (px, py, pz), hit


class SyntheticEnvironment:

    def __init__(self, n_joints: int = 6, target_position: Optional[List[float]] = None,
                 obstacle_position: Optional[List[float]] = None, initial_joint_positions: Optional[List[float]] = None,
                 initial_positions_variation_range: Optional[List[float]] = None):
        self.n = int(n_joints)
        self.involved_joints = list(range(self.n))
        self.target_pos = np.array(target_position if target_position is not None else [0.4, 0.85, 0.71], np.float32)
        self.obstacle_pos = np.array(obstacle_position if obstacle_position is not None else [0.45, 0.55, 0.55], np.float32)
        init = initial_joint_positions if initial_joint_positions is not None else [0.9, 0.45] + [0.0] * 6
        self.initial_joint_positions = np.array(list(init)[:self.n] + [0.0] * max(0, self.n - len(init)), np.float32)
        var = initial_positions_variation_range
        self.initial_positions_variation_range = None if var is None else np.array(list(var)[:self.n], np.float32)
        self._observation_space = np.zeros((9 + 2 * self.n,))
        self._action_space = np.zeros((self.n,))
        self._target_f = tuple(float(x) for x in self.target_pos)
        self._obstacle_f = tuple(float(x) for x in self.obstacle_pos)
        self.q = self.initial_joint_positions.copy()
        self.qd = np.zeros(self.n, np.float32)

    @property
    def observation_space(self) -> np.ndarray:
        return self._observation_space

    @property
    def action_space(self) -> np.ndarray:
        return self._action_space

    def _state_from(self, ee) -> np.ndarray:
        n = self.n
        out = np.empty(9 + 2 * n)
        out[:n] = self.q
        out[n:2 * n] = self.qd
        out[2 * n:2 * n + 3] = np.asarray(ee, np.float32)          # (the observation carries float32 values, as the device env's)
        out[2 * n + 3:2 * n + 6] = self.target_pos
        out[2 * n + 6:] = self.obstacle_pos
        return out

    def get_state(self) -> np.ndarray:
        ee, _ = _fk_fast(self.q.tolist(), self._obstacle_f)
        return self._state_from(ee)

    def reset(self, verbose: bool = True) -> np.ndarray:
        """Initial joint positions (+ uniform variation drawn from Python's global RNG, as environment.py:284-293)."""
        if self.initial_positions_variation_range is None:
            self.q = self.initial_joint_positions.copy()
        else:
            self.q = np.array([random.uniform(p - v, p + v) for p, v in
                               zip(self.initial_joint_positions, self.initial_positions_variation_range)], np.float32)
        self.qd = np.zeros(self.n, np.float32)
        return self.get_state()

    def step(self, action) -> Tuple[np.ndarray, float, int]:
        a = np.asarray(action, np.float32).reshape(self.n)
        self.q = (self.q + DT * a).astype(np.float32)     # velocity control: commanded velocity held for one tick
        self.qd = a.copy()
        ee, hit = _fk_fast(self.q.tolist(), self._obstacle_f)
        ee32 = np.asarray(ee, np.float32)
        diff = ee32 - self.target_pos
        dist = np.float32(np.sqrt(np.float32(diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2])))
        reached = bool(dist < TARGET_THRESHOLD)
        reward = 250 if reached else (-1000 if hit else -1 * float(dist - TARGET_THRESHOLD))
        done = 1 if (reached or hit) else 0
        return self._state_from(ee32), reward, done
