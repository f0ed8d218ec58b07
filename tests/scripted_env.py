"""A deterministic environment with the reference Environment's protocol (reset(verbose) -> state, step(action) ->
(state, reward, done)) whose rewards and episode lengths are known in closed form: test double for the per-env episode
bookkeeping of the many-env training / evaluation loops. Importable by spawned worker processes (tests/ is on sys.path)."""
import numpy as np


class ScriptedEnvironment:
    """Episode k (0-based) of an env built with `offset` lasts L = 3 + (offset + k) % 4 steps; step t (1-based) pays
    -(t + 0.125 * offset), the last one +250 (a "reached") when (offset + k) is even and -1000 (a "collision") otherwise."""

    S, A = 21, 6

    def __init__(self, offset: int = 0, scale: int = 1):
        """scale: every episode `scale` times as long (L = scale * (3 + (offset + k) % 4))"""
        self.offset = int(offset)
        self.scale = int(scale)
        self.k = -1
        self.t = 0
        self.observation_space = np.zeros((self.S,))
        self.action_space = np.zeros((self.A,))

    def _state(self):
        s = np.zeros(self.S)
        s[0], s[1], s[2] = self.offset, self.k, self.t
        return s

    def reset(self, verbose: bool = True):
        self.k += 1
        self.t = 0
        return self._state()

    def length(self, k: int) -> int:
        return self.scale * (3 + (self.offset + k) % 4)

    def step(self, action):
        self.t += 1
        last = self.t >= self.length(self.k)
        if last:
            reward = 250 if (self.offset + self.k) % 2 == 0 else -1000
        else:
            reward = -(self.t + 0.125 * self.offset)
        return self._state(), reward, int(last)

    def expected(self, k: int):
        """(score, frames) of episode k"""
        L = self.length(k)
        score = 0
        for t in range(1, L):
            score += -(t + 0.125 * self.offset)
        score += 250 if (self.offset + k) % 2 == 0 else -1000
        return score, L


class HangingEnvironment(ScriptedEnvironment):
    """a simulator that hangs: step number `hang_at` (counted over the instance's life) never returns — but only in the first
    process that builds one with this marker file absent (the replacement must work)"""

    def __init__(self, offset: int = 0, hang_at: int = 4, marker: str = ""):
        super().__init__(offset)
        import os
        self._n, self._hang_at = 0, hang_at
        self._armed = bool(marker) and not os.path.exists(marker)
        if self._armed:
            open(marker, "w").close()

    def step(self, action):
        self._n += 1
        if self._armed and self._n == self._hang_at:
            import time
            while True:
                time.sleep(1.0)
        return super().step(action)


def make_hanging(offset: int = 0, hang_at: int = 4, marker: str = ""):
    return HangingEnvironment(offset, hang_at, marker)


_NEXT = [0]


def make_scripted(offset: int = 0):
    return ScriptedEnvironment(offset)
