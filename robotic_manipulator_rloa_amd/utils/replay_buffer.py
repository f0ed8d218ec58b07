"""ReplayBuffer with the reference's interface (utils/replay_buffer.py:14-75) backed by an HBM ring of 256-byte
transition rows and the libnaf_hip.so sample/gather kernels. Differences that are deliberate and documented
(DESIGN.md): sampling is drawn on the device from Philox4x32-10 instead of Python's Mersenne Twister (same
distribution: uniform, without replacement inside a minibatch), and rows are float32 at insertion.
"""
from __future__ import annotations

import random
from typing import Optional, Tuple

import numpy as np
import torch

from .. import _lib
from .._lib import check, ptr, stream_ptr

_DIRECT_ROWS = 16      # up to this many staged rows are read by the append kernel from pinned host memory
_STAGE_ROWS = 1024


class ReplayBuffer:

    def __init__(self, buffer_size: int, batch_size: int, device, seed: int, state_size: Optional[int] = None,
                 action_size: Optional[int] = None, action_mode: int = _lib.ACTION_TRUNC_INT,
                 without_replacement: bool = True):
        """
        Args mirror the reference (replay_buffer.py:16-30). state_size/action_size may be given up front; otherwise
        the HBM ring is allocated at the first add() from the shapes of that transition.
        action_mode: ACTION_TRUNC_INT reproduces the reference's `.long()` cast of sampled actions (:60).
        """
        _lib.require_gpu()
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.buffer_size = int(buffer_size)
        self.batch_size = int(batch_size)
        self.seed = int(seed)
        self.action_mode = int(action_mode)
        self.without_replacement = bool(without_replacement)
        random.seed(seed)   # the reference seeds Python's GLOBAL RNG here (:30); Environment.reset draws from it
        self._handle = None
        self._total_added = 0      # host mirror of the device counters (adds are host-initiated: always known)
        self._pending = 0
        self._gen = 0              # bumped by everything here that changes the ring or the sampler's stream position (a pipelined
                                   # per-timestep chunk, engine.TrainChunk, trusts its prefetch only while this stands still)
        self._side_join = None     # set by a pipelined per-timestep graph: makes the current stream wait for the stream its prefetch
                                   # launches run on — they append to the ring and move the counters (engine._Pipeline.join)
        self.S = self.A = None
        if state_size is not None and action_size is not None:
            self._allocate(int(state_size), int(action_size))

    # ---- storage ------------------------------------------------------------------------------------------
    def _allocate(self, S: int, A: int) -> None:
        self.S, self.A = S, A
        self.row_floats = self.lib.naf_replay_row_floats(S, A)
        self.batch_row_floats = self.lib.naf_replay_batch_row_floats(S, A)    # gathered minibatch rows: no tail padding
        self.off_s2 = self.lib.naf_replay_row_off_next_state(S, A)
        self.rows = torch.zeros(self.buffer_size, self.row_floats, dtype=torch.float32, device=self.device)
        self.meta = torch.zeros(8, dtype=torch.int64, device=self.device)
        h = _lib.C.c_void_p()
        check(self.lib.naf_replay_create(self.buffer_size, S, A, ptr(self.rows), ptr(self.meta), _lib.C.byref(h)),
              "naf_replay_create")
        self._handle = h
        # two pinned staging areas used alternately: the H2D copy of one is in flight while add() fills the other, so a
        # flush does not have to wait for its own copy (one event wait, normally already satisfied, before reuse)
        self._stage_hosts = [torch.zeros(_STAGE_ROWS, self.row_floats, dtype=torch.float32).pin_memory() for _ in range(2)]
        self._stage_events = [torch.cuda.Event(), torch.cuda.Event()]
        self._stage_devs = [torch.zeros(_STAGE_ROWS, self.row_floats, dtype=torch.float32, device=self.device) for _ in range(2)]
        self._stage_cur = 0
        self._stage_np = self._stage_hosts[0].numpy()
        self._idx = torch.zeros(self.batch_size, dtype=torch.int32, device=self.device)
        self._sample_ctr = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._sample_scratch = None        # (batch sizes beyond 4096: the sampler's table, allocated at the first draw)

    def __del__(self):
        try:
            if self._handle is not None:
                self.lib.naf_replay_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    @property
    def handle(self):
        if self._handle is None:
            raise _lib.NafHipError("ReplayBuffer storage is not allocated yet (no transition added)")
        return self._handle

    def _join_side(self) -> None:
        if self._side_join is not None:
            self._side_join()

    # ---- add ------------------------------------------------------------------------------------------------
    def add(self, state, action, reward: float, next_state, done: int) -> None:
        """Append one experience (replay_buffer.py:32-45). Staged in pinned host memory; reaches the HBM ring at the
        next flush() (sample() and the agent's update path flush first), so FIFO order and eviction are exact."""
        self._gen += 1
        if self._handle is None:
            self._allocate(int(np.asarray(state).shape[-1]), int(np.asarray(action).shape[-1]))
        S, A = self.S, self.A
        row = self._stage_np[self._pending]
        row[:S] = state[0] if isinstance(state, tuple) else state     # same tuple guard as replay_buffer.py:58
        row[S:S + A] = action
        row[S + A] = reward
        row[self.off_s2:self.off_s2 + S] = next_state
        row[self.off_s2 + S] = done
        self._pending += 1
        self._total_added += 1
        if self._pending == _STAGE_ROWS:
            self.flush()

    def flush(self) -> None:
        self._join_side()
        n = self._pending
        if n == 0:
            return
        self._gen += 1
        cur = self._stage_cur
        host, dev = self._stage_hosts[cur], self._stage_devs[cur]
        if n <= _DIRECT_ROWS:
            src = host                 # pinned and device-mapped: the append kernel reads the few rows straight from it
        else:
            dev[:n].copy_(host[:n], non_blocking=True)
            src = dev
        for lo in range(0, n, self.buffer_size):      # a ring smaller than the staging area: append in ring-sized pieces
            k = min(self.buffer_size, n - lo)
            self.add_rows_device(src[lo:lo + k], k, _count=False)
        self._stage_events[cur].record()
        # switch to the other staging area; its previous copy (two flushes ago) has long completed
        self._stage_cur = cur ^ 1
        self._stage_events[self._stage_cur].synchronize()
        self._stage_np = self._stage_hosts[self._stage_cur].numpy()
        self._pending = 0

    def add_rows_device(self, rows_dev: torch.Tensor, n: int, _count: bool = True) -> None:
        """Append n packed transition rows that already live on the device (vector-env path)."""
        self._join_side()
        self._gen += 1
        if self._handle is None:
            raise _lib.NafHipError("allocate the ReplayBuffer with state_size/action_size before add_rows_device")
        if n > self.buffer_size:
            raise ValueError("cannot add more rows than the buffer holds in one call")
        check(self.lib.naf_replay_add_batch(self.handle, ptr(rows_dev), int(n), stream_ptr()), "naf_replay_add_batch")
        if _count:
            self._total_added += int(n)

    # ---- sample -----------------------------------------------------------------------------------------------
    def sample_indices(self, idx_out: torch.Tensor, n_batches: int = 1) -> None:
        self._join_side()
        self._gen += 1
        if self.batch_size > 4096:
            # beyond one workgroup's LDS: the duplicate check's table in device memory (csrc/replay.hip, replay_sample_big_kernel)
            per = self.lib.naf_replay_sample_scratch_ints(self.batch_size)
            if per < 0:
                raise ValueError(f"batch_size {self.batch_size}: the replay sampler draws minibatches of at most 1,048,576")
            if self._sample_scratch is None or self._sample_scratch.numel() < per * int(n_batches):
                self._sample_scratch = torch.zeros(per * int(n_batches), dtype=torch.int32, device=self.device)
            check(self.lib.naf_replay_sample_indices_big(self.handle, self.seed, ptr(self._sample_ctr), 0, ptr(idx_out),
                                                         self.batch_size, int(n_batches), int(self.without_replacement),
                                                         ptr(self._sample_scratch), stream_ptr()), "naf_replay_sample_indices_big")
        else:
            check(self.lib.naf_replay_sample_indices(self.handle, self.seed, ptr(self._sample_ctr), 0, ptr(idx_out),
                                                     self.batch_size, int(n_batches), int(self.without_replacement),
                                                     stream_ptr()), "naf_replay_sample_indices")
        check(self.lib.naf_counter_add(ptr(self._sample_ctr), int(n_batches), stream_ptr()), "naf_counter_add")

    def gather_rows(self, idx: torch.Tensor, out_rows: torch.Tensor, n: int) -> None:
        """out_rows[..., ld] (contiguous) receives the leading ld floats of the n indexed rows; ld = out_rows.shape[-1]
        between batch_row_floats (what the learner reads) and row_floats (the whole padded ring row)."""
        self._join_side()
        if not out_rows.is_contiguous() or out_rows.numel() < int(n) * out_rows.shape[-1]:
            raise ValueError("gather_rows: out_rows must be contiguous and hold n rows")
        check(self.lib.naf_replay_gather_rows(self.handle, ptr(idx), ptr(out_rows), int(n), int(out_rows.shape[-1]),
                                              self.action_mode, stream_ptr()), "naf_replay_gather_rows")

    def sample(self, idx: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, ...]:
        """(states, actions, rewards, next_states, dones) with the reference's shapes and dtypes
        (replay_buffer.py:47-67): f32 (B,S), int64 (B,A) truncated, f32 (B,1), f32 (B,S), f32 (B,1).
        idx: optional int32 deque positions (0 = oldest) to take instead of drawing."""
        self._join_side()
        self.flush()
        B, dev = self.batch_size, self.device
        if idx is None and len(self) < B:
            # random.sample(deque, k) of the reference raises exactly this (replay_buffer.py:55)
            raise ValueError("Sample larger than population or is negative")
        if idx is None:
            self.sample_indices(self._idx, 1)
            idx = self._idx
        else:
            idx = idx.to(dev, torch.int32).contiguous()
            B = idx.numel()
        s = torch.empty(B, self.S, dtype=torch.float32, device=dev)
        u = torch.empty(B, self.A, dtype=torch.float32, device=dev)
        r = torch.empty(B, 1, dtype=torch.float32, device=dev)
        s2 = torch.empty(B, self.S, dtype=torch.float32, device=dev)
        d = torch.empty(B, 1, dtype=torch.float32, device=dev)
        check(self.lib.naf_replay_gather_soa(self.handle, ptr(idx), ptr(s), ptr(u), ptr(r), ptr(s2), ptr(d), B,
                                             self.action_mode, stream_ptr()), "naf_replay_gather_soa")
        actions = u.long() if self.action_mode == _lib.ACTION_TRUNC_INT else u
        return s, actions, r, s2, d

    def device_len(self) -> int:
        """Fill level as the device sees it (blocking; __len__ is the host-side count and never synchronises)."""
        self._join_side()
        out = _lib.C.c_uint64()
        check(self.lib.naf_replay_size(self.handle, _lib.C.byref(out), stream_ptr()), "naf_replay_size")
        return int(out.value)

    def bad_index_count(self) -> int:
        self._join_side()
        return int(self.meta[7].item())

    def __len__(self) -> int:
        """Current number of stored experiences (replay_buffer.py:69-75), pending staged rows included."""
        return min(self._total_added, self.buffer_size)
