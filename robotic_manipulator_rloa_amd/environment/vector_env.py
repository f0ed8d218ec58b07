"""Host-side vector environment: N independent environments in worker processes feeding one GPU learner — the
"N independent PyBullet DIRECT envs per GPU feed a per-GPU buffer" shape of BASELINE.json's north_star.

The reference steps ONE environment in the training process (naf_algorithm.py:246-270). Its PyBullet Environment
talks to the implicit global physics client (environment.py:208-210, 229, 296, 443), so there can be only one per
process: each worker process here owns `envs_per_worker` environments (1 for PyBullet; more for light envs) built by a
picklable factory, and exchanges data with the trainer through shared-memory arrays — per vector step the trainer
writes E x A actions and reads E transitions [state | action | reward | next_state | done]; no pickling of
observations.

Commands travel the same way (round 6): a 64-byte line of shared memory per worker holds {command ordinal, command, done ordinal}
— the trainer rings the doorbell by bumping the ordinal, the worker answers by copying it — instead of two pipe messages per worker
and step. Every wait of the trainer is bounded: a worker that has DIED (a simulator that segfaults takes its process along) or that
does not answer within `step_timeout_s` is killed and RESPAWNED with fresh environments, reset; the transitions its envs had in
flight are dropped — `arr["valid"]` says which of a step's E transitions exist, `pack_rows` packs those only — and counted
(`respawns`, `dropped_transitions`): training goes on ("env worker crash -> respawn & mark transitions invalid").

Episode bookkeeping is per env: when an env reports done, or has run `max_frames` steps (the frame budget of
NAFAgent.run, naf_algorithm.py:249), the worker resets it and the NEXT state handed to the policy is the reset state
(the stored transition keeps the true next_state, as the reference does).
"""
from __future__ import annotations

import multiprocessing as mp
import os
import time
from multiprocessing import shared_memory
from typing import Callable, List, Optional

import numpy as np

_CMD_RESET, _CMD_STEP, _CMD_CLOSE = 1, 2, 3
_LINE = 8                                  # uint64 words per worker: [command ordinal, command, done ordinal, pid, 0 ...]


def usable_cpus() -> int:
    """CPUs this process may actually use: the scheduler affinity mask and the cgroup's CPU quota, whichever is smaller —
    os.cpu_count() counts the machine's (256 on a GPU box whose container gets 16)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:                      # noqa: BLE001
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1") and float(quota) > 0:
                n = min(n, max(1, int(float(quota) / period)))
            break
        except Exception:                  # noqa: BLE001
            continue
    return max(1, n)


def _attach(name: str, shape, dtype):
    shm = shared_memory.SharedMemory(name=name)      # (a spawned worker shares the trainer's resource tracker: nothing to unregister)
    return shm, np.ndarray(shape, dtype=dtype, buffer=shm.buf)


def _worker(w: int, env_fn, first: int, count: int, names: dict, max_frames: int, seed: int, spin_s: float):
    import random
    random.seed(seed + first)            # Environment.reset draws its joint variation from Python's global RNG
    np.random.seed((seed + first) % (2 ** 32))
    handles = {k: _attach(v, shp, dt) for k, (v, shp, dt) in names.items()}
    arr = {k: h[1] for k, h in handles.items()}
    ctrl = arr["ctrl"]
    base = w * _LINE
    envs = [env_fn() for _ in range(count)]
    frames = [0] * count
    seen = 0
    parent = os.getppid()
    try:
        while True:
            # the doorbell: spin while a step is likely to follow the last one (the trainer's GPU work between two steps is a few
            # milliseconds), then sleep in short naps — and leave if the trainer is gone
            t_idle = time.perf_counter()
            while int(ctrl[base]) == seen:
                idle = time.perf_counter() - t_idle
                if idle > spin_s:
                    # (more workers than cores: give the core away at once — a yield while a step may still be near, naps after)
                    time.sleep(0.0 if idle < spin_s + 0.002 else 0.0002)
                    if os.getppid() != parent:
                        return
            seen = int(ctrl[base])
            cmd = int(ctrl[base + 1])
            if cmd == _CMD_RESET:
                for j, env in enumerate(envs):
                    arr["obs"][first + j] = env.reset(False)
                    frames[j] = 0
            elif cmd == _CMD_STEP:
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
            elif cmd == _CMD_CLOSE:
                for env in envs:
                    close = getattr(env, "close", None)
                    if close:
                        close()
                ctrl[base + 2] = seen
                return
            ctrl[base + 2] = seen        # done (x86: the stores above are visible before this one)
    finally:
        arr.clear()
        ctrl = None
        for shm, _ in handles.values():
            try:
                shm.close()
            except Exception:              # noqa: BLE001  (a view still alive: the process is leaving anyway)
                pass


class HostVectorEnv:

    def __init__(self, env_fn: Callable[[], object], n_envs: int, state_size: int, action_size: int,
                 envs_per_worker: int = 1, max_frames: int = 400, seed: int = 0, start_method: str = "spawn",
                 step_timeout_s: float = 60.0, start_timeout_s: float = 300.0, spin_ms: float = 5.0):
        """env_fn: picklable zero-argument factory returning an object with reset(verbose) -> state[S] and
        step(action) -> (state[S], reward, done) (the reference Environment protocol).
        start_method 'spawn' keeps the workers free of the trainer's GPU context.
        step_timeout_s: a worker that has not answered a command after that long is taken for hung, killed and respawned;
        start_timeout_s: the same bound for a worker's first answer (process start + imports + environment construction).
        spin_ms: how long an idle worker spins on its doorbell before it starts napping (0.2 ms at a time)."""
        self.E, self.S, self.A = int(n_envs), int(state_size), int(action_size)
        self.max_frames = int(max_frames)
        self.env_fn, self.seed = env_fn, int(seed)
        self.step_timeout_s, self.start_timeout_s, self.spin_s = float(step_timeout_s), float(start_timeout_s), float(spin_ms) * 1e-3
        self._slices = [(first, min(envs_per_worker, self.E - first)) for first in range(0, self.E, envs_per_worker)]
        W = len(self._slices)
        if W + 1 > usable_cpus():
            self.spin_s = 0.0              # more workers than cores: a spinning worker would keep a working one off its core
        spec = {"obs": ((self.E, self.S), np.float64), "state": ((self.E, self.S), np.float64),
                "next_state": ((self.E, self.S), np.float64), "actions": ((self.E, self.A), np.float32),
                "reward": ((self.E,), np.float64), "done": ((self.E,), np.int64), "episode_end": ((self.E,), np.int64),
                "valid": ((self.E,), np.int64), "ctrl": ((W * _LINE,), np.uint64)}
        self._shm, self.arr, self._names = [], {}, {}
        for k, (shape, dt) in spec.items():
            shm = shared_memory.SharedMemory(create=True, size=max(8, int(np.prod(shape)) * np.dtype(dt).itemsize))
            self._shm.append(shm)
            self.arr[k] = np.ndarray(shape, dtype=dt, buffer=shm.buf)
            self.arr[k][...] = 0
            self._names[k] = (shm.name, shape, dt)
        self.arr["valid"][...] = 1
        self._ctx = mp.get_context(start_method)
        self._procs: List[Optional[mp.Process]] = [None] * W
        self._fresh = [True] * W           # no command answered yet: the first wait gets the start-up bound
        for w in range(W):
            self._spawn(w)
        self.episodes_finished = 0
        self.steps = 0
        self.respawns = 0                  # workers found dead or hung, replaced
        self.dropped_transitions = 0       # transitions their envs had in flight
        self.respawned_envs: List[int] = []    # env indices replaced during the LAST reset() / step()
        self._fault_at = None              # tests: (step ordinal, worker) — the worker is killed in front of that step's doorbell
        self.closed = False

    # ---- workers -------------------------------------------------------------------------------------------------------------
    def _spawn(self, w: int) -> None:
        first, count = self._slices[w]
        self.arr["ctrl"][w * _LINE:(w + 1) * _LINE] = 0
        p = self._ctx.Process(target=_worker, args=(w, self.env_fn, first, count, self._names, self.max_frames, self.seed, self.spin_s),
                              daemon=True)
        p.start()
        self._procs[w] = p
        self._fresh[w] = True

    def _ring(self, w: int, cmd: int) -> None:
        c = self.arr["ctrl"]
        c[w * _LINE + 1] = cmd
        c[w * _LINE] = int(c[w * _LINE]) + 1

    def _answered(self, w: int, bound_s: float) -> bool:
        """spin until worker w has answered its last command; False if it died or the bound ran out"""
        c, b = self.arr["ctrl"], w * _LINE
        want = int(c[b])
        n, t0 = 0, None
        while int(c[b + 2]) != want:
            n += 1
            if n & 0xFF == 0:
                now = time.perf_counter()
                if t0 is None:
                    t0 = now
                if not self._procs[w].is_alive() or now - t0 > bound_s:
                    return int(c[b + 2]) == want
                if now - t0 > 0.002:
                    time.sleep(0.0002)         # (a long wait — a worker starting up: do not burn a core on it)
                elif self.spin_s == 0.0:
                    time.sleep(0.0)            # (more workers than cores: let them run)
        return True

    def _replace(self, w: int) -> None:
        """worker w is dead or hung: a fresh process with fresh environments, reset; what its envs had in flight is gone"""
        p = self._procs[w]
        if p is not None:
            try:
                p.kill()
                p.join(timeout=5)
            except Exception:                  # noqa: BLE001
                pass
        first, count = self._slices[w]
        self._spawn(w)
        self._ring(w, _CMD_RESET)
        if not self._answered(w, self.start_timeout_s):
            raise RuntimeError(f"HostVectorEnv: the replacement of worker {w} (envs {first} .. {first + count - 1}) did not come up "
                               f"within {self.start_timeout_s:.0f} s")
        self._fresh[w] = False
        self.respawns += 1
        self.respawned_envs += list(range(first, first + count))

    def _all(self, cmd: int) -> None:
        W = len(self._slices)
        for w in range(W):
            self._ring(w, cmd)
        for w in range(W):
            if self._answered(w, self.start_timeout_s if self._fresh[w] else self.step_timeout_s):
                self._fresh[w] = False
            else:
                self._replace(w)

    def kill_worker(self, w: int) -> None:
        """fault injection (tests): SIGKILL worker w, as a crashing simulator would"""
        self._procs[w].kill()
        self._procs[w].join(timeout=5)

    # ---- the environment protocol, E at a time ----------------------------------------------------------------------------------
    def reset(self) -> np.ndarray:
        """All envs to their initial state; returns the E x S observation array (a view of shared memory)."""
        self.respawned_envs = []
        self._all(_CMD_RESET)
        self.arr["valid"][...] = 1
        return self.arr["obs"]

    def step(self, actions: np.ndarray):
        """actions: E x A float32. Returns views (state, action, reward, next_state, done, obs_for_next_act). arr["valid"][e] == 0:
        env e's worker was replaced during this step — its transition does not exist (state .. done hold stale values), its
        observation is a fresh environment's reset state, its episode in flight is gone."""
        a = self.arr
        a["actions"][...] = actions
        self.respawned_envs = []
        if self._fault_at is not None and self._fault_at[0] == self.steps:
            self.kill_worker(self._fault_at[1])
        self._all(_CMD_STEP)
        self.steps += 1
        if self.respawned_envs:
            a["valid"][...] = 1
            a["valid"][self.respawned_envs] = 0
            a["episode_end"][self.respawned_envs] = 0
            self.dropped_transitions += len(self.respawned_envs)
        elif not a["valid"].all():
            a["valid"][...] = 1
        self.episodes_finished += int(a["episode_end"].sum())
        return a["state"], a["actions"], a["reward"], a["next_state"], a["done"], a["obs"]

    def pack_rows(self, out: np.ndarray, off_next_state: int) -> int:
        """Last step's VALID transitions in the HBM row layout (include/naf_hip.h) into the leading rows of `out` [E, row_floats]
        f32, in env order; returns how many (E unless a worker was replaced during the step)."""
        a, S, A = self.arr, self.S, self.A
        if not self.respawned_envs:
            out[:, :S] = a["state"]
            out[:, S:S + A] = a["actions"]
            out[:, S + A] = a["reward"]
            out[:, off_next_state:off_next_state + S] = a["next_state"]
            out[:, off_next_state + S] = a["done"]
            return self.E
        keep = np.nonzero(a["valid"])[0]
        n = len(keep)
        out[:n, :S] = a["state"][keep]
        out[:n, S:S + A] = a["actions"][keep]
        out[:n, S + A] = a["reward"][keep]
        out[:n, off_next_state:off_next_state + S] = a["next_state"][keep]
        out[:n, off_next_state + S] = a["done"][keep]
        return n

    def close(self) -> None:
        if self.closed:
            return
        self.closed = True
        W = len(self._slices)
        try:
            for w in range(W):
                if self._procs[w] is not None and self._procs[w].is_alive():
                    self._ring(w, _CMD_CLOSE)
            for w in range(W):
                if self._procs[w] is not None and self._procs[w].is_alive():
                    self._answered(w, 5.0)
        except Exception:                      # noqa: BLE001
            pass
        for p in self._procs:
            if p is None:
                continue
            p.join(timeout=5)
            if p.is_alive():
                p.kill()
        self.arr = {}
        for shm in self._shm:
            try:
                shm.close()
            except Exception:                  # noqa: BLE001
                pass
            try:
                shm.unlink()
            except FileNotFoundError:
                pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
