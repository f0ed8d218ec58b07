"""Host-side vector environment: N independent environments in worker processes feeding one GPU learner — the
"N independent PyBullet DIRECT envs per GPU feed a per-GPU buffer" shape of BASELINE.json's north_star.

The reference steps ONE environment in the training process (naf_algorithm.py:246-270). Its PyBullet Environment
talks to the implicit global physics client (environment.py:208-210, 229, 296, 443), so there can be only one per
process: each worker process here owns `envs_per_worker` environments (1 for PyBullet; more for light envs) built by a
picklable factory, and exchanges data with the trainer through shared-memory arrays — per vector step the trainer
writes E x A actions and reads E transitions [state | action | reward | next_state | done]; no pickling of
observations, one small pipe message per worker per step.

Episode bookkeeping is per env: when an env reports done, or has run `max_frames` steps (the frame budget of
NAFAgent.run, naf_algorithm.py:249), the worker resets it and the NEXT state handed to the policy is the reset state
(the stored transition keeps the true next_state, as the reference does).
"""
from __future__ import annotations

import multiprocessing as mp
from multiprocessing import shared_memory
from typing import Callable, List, Optional

import numpy as np


def _attach(name: str, shape, dtype):
    shm = shared_memory.SharedMemory(name=name)
    return shm, np.ndarray(shape, dtype=dtype, buffer=shm.buf)


def _worker(conn, env_fn, first: int, count: int, names: dict, E: int, S: int, A: int, max_frames: int, seed: int):
    import random
    random.seed(seed + first)            # Environment.reset draws its joint variation from Python's global RNG
    np.random.seed((seed + first) % (2 ** 32))
    handles = {k: _attach(v, shp, dt) for k, (v, shp, dt) in names.items()}
    arr = {k: h[1] for k, h in handles.items()}
    envs = [env_fn() for _ in range(count)]
    frames = [0] * count
    try:
        while True:
            cmd = conn.recv()
            if cmd == "reset":
                for j, env in enumerate(envs):
                    arr["obs"][first + j] = env.reset(False)
                    frames[j] = 0
                conn.send(True)
            elif cmd == "step":
                for j, env in enumerate(envs):
                    e = first + j
                    action = arr["actions"][e].copy()
                    nxt, reward, done = env.step(action)
                    arr["state"][e] = arr["obs"][e]
                    arr["next_state"][e] = nxt
                    arr["reward"][e] = reward
                    arr["done"][e] = done
                    frames[j] += 1
                    finished = bool(done) or (max_frames > 0 and frames[j] >= max_frames)
                    arr["episode_end"][e] = 1 if finished else 0
                    if finished:
                        nxt = env.reset(False)
                        frames[j] = 0
                    arr["obs"][e] = nxt
                conn.send(True)
            elif cmd == "close":
                for env in envs:
                    close = getattr(env, "close", None)
                    if close:
                        close()
                conn.send(True)
                break
    finally:
        for shm, _ in handles.values():
            shm.close()


class HostVectorEnv:

    def __init__(self, env_fn: Callable[[], object], n_envs: int, state_size: int, action_size: int,
                 envs_per_worker: int = 1, max_frames: int = 400, seed: int = 0, start_method: str = "spawn"):
        """env_fn: picklable zero-argument factory returning an object with reset(verbose) -> state[S] and
        step(action) -> (state[S], reward, done) (the reference Environment protocol).
        start_method 'spawn' keeps the workers free of the trainer's GPU context."""
        self.E, self.S, self.A = int(n_envs), int(state_size), int(action_size)
        self.max_frames = int(max_frames)
        spec = {"obs": ((self.E, self.S), np.float64), "state": ((self.E, self.S), np.float64),
                "next_state": ((self.E, self.S), np.float64), "actions": ((self.E, self.A), np.float32),
                "reward": ((self.E,), np.float64), "done": ((self.E,), np.int64), "episode_end": ((self.E,), np.int64)}
        self._shm, self.arr, names = [], {}, {}
        for k, (shape, dt) in spec.items():
            shm = shared_memory.SharedMemory(create=True, size=max(8, int(np.prod(shape)) * np.dtype(dt).itemsize))
            self._shm.append(shm)
            self.arr[k] = np.ndarray(shape, dtype=dt, buffer=shm.buf)
            self.arr[k][...] = 0
            names[k] = (shm.name, shape, dt)
        ctx = mp.get_context(start_method)
        self._conns, self._procs = [], []
        for first in range(0, self.E, envs_per_worker):
            count = min(envs_per_worker, self.E - first)
            parent, child = ctx.Pipe()
            p = ctx.Process(target=_worker, args=(child, env_fn, first, count, names, self.E, self.S, self.A,
                                                  self.max_frames, seed), daemon=True)
            p.start()
            child.close()
            self._conns.append(parent)
            self._procs.append(p)
        self.episodes_finished = 0
        self.closed = False

    def _all(self, cmd: str) -> None:
        for c in self._conns:
            c.send(cmd)
        for c in self._conns:
            c.recv()

    def reset(self) -> np.ndarray:
        """All envs to their initial state; returns the E x S observation array (a view of shared memory)."""
        self._all("reset")
        return self.arr["obs"]

    def step(self, actions: np.ndarray):
        """actions: E x A float32. Returns views (state, action, reward, next_state, done, obs_for_next_act)."""
        self.arr["actions"][...] = actions
        self._all("step")
        self.episodes_finished += int(self.arr["episode_end"].sum())
        a = self.arr
        return a["state"], a["actions"], a["reward"], a["next_state"], a["done"], a["obs"]

    def pack_rows(self, out: np.ndarray, off_next_state: int) -> None:
        """Last step's E transitions in the HBM row layout (include/naf_hip.h) into `out` [E, row_floats] f32."""
        a, S, A = self.arr, self.S, self.A
        out[:, :S] = a["state"]
        out[:, S:S + A] = a["actions"]
        out[:, S + A] = a["reward"]
        out[:, off_next_state:off_next_state + S] = a["next_state"]
        out[:, off_next_state + S] = a["done"]

    def close(self) -> None:
        if self.closed:
            return
        self.closed = True
        try:
            self._all("close")
        except (BrokenPipeError, EOFError, OSError):
            pass
        for p in self._procs:
            p.join(timeout=5)
            if p.is_alive():
                p.terminate()
        for shm in self._shm:
            shm.close()
            try:
                shm.unlink()
            except FileNotFoundError:
                pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
