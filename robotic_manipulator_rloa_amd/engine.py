"""HIP-graph engines around the Learner: the steady-state loop of NAFAgent.step()/run()
(reference naf_components/naf_algorithm.py:129-156, :228-270) with zero host<->device synchronisation.

  TrainChunk     : [sample U minibatches] -> [gather U*B rows, one launch] -> U x learn()  as ONE graph.
  DeviceEnvLoop  : E synthetic arms stepped on the device: act (eval-mode policy + noise) -> env step ->
                   append E transitions to the HBM ring, as ONE graph; followed by a TrainChunk with
                   U = E * num_updates / update_freq so the reference's update-to-data ratio is kept.

Graph capture goes through torch.cuda.CUDAGraph (= hipGraph on ROCm): the ctypes kernel launches use the
stream torch reports as current, which inside the capture context is the capturing stream.
"""
from __future__ import annotations

import json
import os
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr
from .learner import ActPath, Learner
from .presets import ROBOT_PRESETS, device_env_preset
from .utils.replay_buffer import ReplayBuffer


class _StateSnapshot:
    """Warm-up before capture must not leave a trace: save/restore every piece of mutable learner state."""

    def __init__(self, learner: Learner, replay: Optional[ReplayBuffer], extra=()):
        self.pairs = [(t, t.clone()) for t in (learner.theta2, learner.grad, learner.adam_m, learner.adam_v,
                                               learner.bn_stats, learner.step_dev, learner.partials)]
        if replay is not None and replay._handle is not None:
            self.pairs += [(t, t.clone()) for t in (replay.meta, replay._sample_ctr)]
        self.pairs += [(t, t.clone()) for t in extra]

    def restore(self) -> None:
        for live, saved in self.pairs:
            live.copy_(saved)


def _capture(body, snapshot: _StateSnapshot, warmup: int = 2, after_warmup=None) -> torch.cuda.CUDAGraph:
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warmup):        # lets hipBLASLt pick kernels / allocate workspaces outside the capture
            body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    snapshot.restore()
    if after_warmup is not None:
        after_warmup()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    snapshot.restore()                 # capture does not execute, but keep the contract explicit
    torch.cuda.synchronize()
    return g


class TrainChunk:
    """U consecutive learn() updates on U freshly sampled minibatches."""

    def __init__(self, learner: Learner, replay: ReplayBuffer, n_updates: int, teacher_forced: bool = False,
                 use_graph: bool = True, gather_outside_graph: bool = False, tail=None, tail_state=(), head_row=None):
        """gather_outside_graph: launch sample+gather eagerly in front of the graph of U updates, so the caller can
        bracket the gather launch with events (bench.py's live roofline measurement).
        tail: optional callable enqueued behind the last update, inside the same graph (NAFAgent puts the NEXT timestep's
        act() there); tail_state: device tensors it changes, so that the capture's warm-up leaves no trace in them.
        head_row: optional pinned [1, row_floats] tensor: the graph STARTS by appending that one transition to the ring
        (the append kernel reads pinned host memory itself) — ReplayBuffer.add of the timestep inside the graph of its
        update instead of a launch, two event calls and a staging switch of its own. The caller fills the row and counts
        the transition (replay._total_added) before every run(head_rows=1); a run() without a new transition (an idle
        tick of a data-parallel run(), a timestep whose row went through the staging area) appends nothing: the node
        reads its row count from a pinned word that run() sets (naf_replay_add_counted)."""
        self.tail, self._tail_state = tail, tuple(tail_state)
        self.head_row = head_row
        self.head_count = torch.zeros(1, dtype=torch.int32).pin_memory() if head_row is not None else None
        self._ran = None                   # event behind the last run(): the pinned words are free again once it has passed
        self.L, self.replay, self.U = learner, replay, int(n_updates)
        self.teacher_forced = teacher_forced
        self.gather_outside_graph = gather_outside_graph
        self.gather_events = None          # optional (start, end) torch.cuda.Event pair recorded around the gather
        self.empty_events = None           # optional pair recorded back to back right after it (what a bracket costs)
        B, dev = learner.B, learner.dev
        if replay.batch_size != B:
            raise ValueError("ReplayBuffer.batch_size must equal the learner's batch size")
        self.idx = torch.zeros(self.U, B, dtype=torch.int32, device=dev)
        # gathered minibatches: packed rows (no padding to whole 128-B lines on the output side of the gather); the
        # zero tail keeps the layer-1 kernels' whole-float4 reads of the last row inside the allocation for any S
        brf = learner.lay.batch_row_floats
        self._batch_store = torch.zeros(self.U * B * brf + 64, dtype=torch.float32, device=dev)
        self.batch = self._batch_store[:self.U * B * brf].view(self.U, B, brf)
        self.loss_parts = torch.zeros(self.U, learner.n_loss_wg, dtype=torch.float32, device=dev)
        # large-batch chain: the moments records of the U minibatches' layer-1 inputs, one launch behind the gather
        self.moments = (torch.zeros(self.U, 2, learner.mom_floats, dtype=torch.float32, device=dev)
                        if "bb" in learner.fuse else None)
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.use_graph = use_graph

    def _sample_gather(self) -> None:
        if not self.teacher_forced:
            self.replay.sample_indices(self.idx, self.U)
        ev = self.gather_events
        if ev is not None:
            ev[0].record()
        self.replay.gather_rows(self.idx, self.batch, self.U * self.L.B)
        if ev is not None:
            ev[1].record()
            if self.empty_events is not None:
                self.empty_events[0].record()
                self.empty_events[1].record()

    def _updates(self) -> None:
        if self.moments is not None:
            self.L.moments(self.batch.view(self.U * self.L.B, -1), self.moments, self.U)
        # a chain of updates: the optimizer step of update k rides on the first two launches of update k + 1 (one launch
        # less per update, Learner.defer_ok); the last one of the chunk takes its step as a launch of its own, so the
        # parameter buffers are current whenever anything outside the chunk looks at them
        d = self.L.defer_ok
        for k in range(self.U):
            self.L.learn_rows(self.batch[k], self.loss_parts[k], None if self.moments is None else self.moments[k],
                              pending=d and k > 0, defer=d and k < self.U - 1)
        if self.tail is not None:
            self.tail()

    def _body(self) -> None:
        if self.head_row is not None:
            check(self.L.lib.naf_replay_add_counted(self.replay.handle, ptr(self.head_row), ptr(self.head_count), 1, stream_ptr()),
                  "naf_replay_add_counted")
        self._sample_gather()
        self._updates()

    def capture(self) -> None:
        self.replay.flush()
        snap = _StateSnapshot(self.L, self.replay, (self.idx, self.batch, self.loss_parts) + self._tail_state)
        if self.gather_outside_graph:
            self._sample_gather()          # the updates need a valid batch to warm up on
            self.graph = _capture(self._updates, snap)
        elif self.head_row is not None:
            # the warm-up appends rows to the ring: {head, size} come back with the snapshot, the ring slots it wrote
            # (live rows, if the ring is full) are saved and put back
            warmup = 2
            head = int(self.replay.meta[0].item())
            pos = (head + torch.arange(warmup, device=self.L.dev)) % self.replay.buffer_size
            saved = self.replay.rows[pos].clone()
            count = int(self.head_count[0])
            self.head_count[0] = 1         # the warm-up runs really append (and are undone)

            def put_back():
                self.replay.rows[pos] = saved
                self.head_count[0] = count
            self.graph = _capture(self._body, snap, warmup=warmup, after_warmup=put_back)
        else:
            self.graph = _capture(self._body, snap)

    def wait_pinned_free(self) -> None:
        """Block until the last run() has passed: what it reads from pinned host memory (the head row, its count, a tail's
        observation) may be rewritten afterwards. In NAFAgent.run's loop act() has waited for it already."""
        if self._ran is not None:
            self._ran.synchronize()

    def run(self, head_rows: int = 0) -> None:
        """Enqueue the chunk (asynchronous). With teacher forcing, fill self.idx first. head_rows (chunks built with
        head_row): 1 = the pinned row holds a new transition that the chunk's first node appends, 0 = it appends nothing.
        The caller has waited (wait_pinned_free) before it rewrote the row."""
        self.L.raise_on_device_error()         # pinned host words written by the kernels: costs two loads, never a sync
        if self.head_row is not None:
            if self.use_graph and self.graph is None:
                self.capture()
            # the previous run's append node must have read its count before the word changes (an idle tick right behind a
            # step() would otherwise zero the count of a row the GPU has not appended yet); in run()'s loop act() has waited
            self.wait_pinned_free()
            self.head_count[0] = 1 if head_rows else 0
        if self.use_graph:
            if self.graph is None:
                self.capture()
            if self.gather_outside_graph:
                self._sample_gather()
            self.graph.replay()
        else:
            self._body()
        if self.head_row is not None or self.tail is not None:
            if self._ran is None:
                self._ran = torch.cuda.Event()
            self._ran.record()

    def losses(self) -> torch.Tensor:
        """[U] MSE losses of the last run (device tensor; summing the per-workgroup parts in index order)."""
        return self.loss_parts.sum(dim=1)


# naf_episode_record_t (include/naf_hip.h), 32 bytes
EPISODE_RECORD = np.dtype([("score", "<f8"), ("frames", "<i4"), ("done", "<i4"), ("last_reward", "<f4"),
                           ("episode", "<i4"), ("step_lo", "<u4"), ("env", "<u4")])


class EpisodeLedger:
    """What NAFAgent.run keeps and writes per finished episode (naf_algorithm.py:241-289), for loops in which E
    environments finish episodes in any order: `scores` = {episode: (score, last_frame)} numbered in COMPLETION order,
    `checkpoints/{episode}/weights.p` + `scores.txt` whenever the count reaches a multiple of checkpoint_frequency, and
    `model.p` at the end — same file names, JSON shape and state-dict keys as the reference, written by rank 0 only."""

    def __init__(self, episodes: Optional[int], checkpoint_frequency: int, state_dict_fn: Callable[[], dict],
                 write: bool = True, model_path: str = "model.p"):
        """episodes: the budget (`episodes` of run()); the dict is pre-filled with (0, 0) as the reference's is (:241) and
        episodes beyond it are counted but not recorded. None: open-ended."""
        self.limit = episodes
        self.scores: Dict[int, Tuple[float, int]] = {} if episodes is None else {e: (0, 0) for e in range(1, episodes + 1)}
        self.count = 0                      # episodes recorded
        self.extra = 0                      # episodes that finished after the budget was met (not recorded)
        self.every = int(checkpoint_frequency)
        self._state_dict_fn, self.write, self.model_path = state_dict_fn, bool(write), model_path
        self.checkpoints: List[int] = []

    @property
    def complete(self) -> bool:
        return self.limit is not None and self.count >= self.limit

    def add(self, score: float, frames: int) -> None:
        if self.complete:
            self.extra += 1
            return
        self.count += 1
        self.scores[self.count] = (float(score), int(frames))
        if self.every > 0 and self.count % self.every == 0 and self.write:
            d = f"checkpoints/{self.count}/"
            os.makedirs(d, exist_ok=True)
            torch.save(self._state_dict_fn(), d + "weights.p")
            with open(d + "scores.txt", "w") as f:
                f.write(json.dumps(self.scores))
            self.checkpoints.append(self.count)

    def finish(self) -> Dict[int, Tuple[float, int]]:
        if self.write:
            torch.save(self._state_dict_fn(), self.model_path)
        return self.scores


class DeviceEnvLoop:
    """E synthetic manipulator envs living on the GPU (csrc/synth_env.hip) driven by the agent's policy.

    Episode records (`records=True`): the step kernel keeps every env's running score and frame count and writes one
    naf_episode_record_t per (vector step, env) into a ring of `drain_every` slots; every `drain_every` steps the ring is
    copied to pinned host memory behind the step that filled it (asynchronous, an event marks it) and the PREVIOUS copy
    — long finished by then — is parsed: finished episodes reach the host in (step, env) order with a lag of at most
    2 x drain_every vector steps and without a synchronisation per step. `drain(final=True)` flushes the rest."""

    # [initial joint positions(8) | target | obstacle] per robot, from the one preset table (presets.py)
    PRESETS = {name: device_env_preset(name) for name in ROBOT_PRESETS}

    def __init__(self, learner: Learner, replay: Optional[ReplayBuffer], n_envs: int, seed: int, max_frames: int = 400,
                 noise_scale: float = 1.0, use_graph: bool = True, robot: str = "kuka", obstacle_jitter: float = 0.0,
                 preset: Optional[List[float]] = None, variation: Optional[List[float]] = None, records: bool = False,
                 drain_every: int = 64):
        """replay=None: no transitions are appended (evaluation). preset: 14 floats [initial joint positions(8) | target |
        obstacle] instead of a named robot's. variation: per-joint half-width of the reset range (None: 0.1 everywhere,
        the stand-in's historical value)."""
        self.L, self.replay, self.E = learner, replay, int(n_envs)
        lay, dev = learner.lay, learner.dev
        self.lib = learner.lib
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.max_frames = int(max_frames)
        self.noise_scale = float(noise_scale)
        self.actor = ActPath(learner, self.E, seed=self.seed ^ 0xA5A5A5A5)
        nst = self.lib.naf_synth_env_state_floats(lay.A)
        self.env_state = torch.zeros(self.E, nst, dtype=torch.float32, device=dev)
        self.rows = torch.zeros(self.E, lay.row_floats, dtype=torch.float32, device=dev)
        self.step_ctr = torch.zeros(1, dtype=torch.int64, device=dev)
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.use_graph = use_graph
        self.env_steps = 0
        import ctypes
        base = list(preset) if preset is not None else list(self.PRESETS[robot])
        if len(base) != 14:
            raise ValueError("DeviceEnvLoop: preset = [initial joint positions(8) | target xyz | obstacle xyz]")
        var = [0.1] * 8 if variation is None else ([float(v) for v in variation] + [0.0] * 8)[:8]
        self._preset = (ctypes.c_float * 23)(*(base + [float(obstacle_jitter)] + var))
        self.drain_every = max(1, int(drain_every))
        self.records = None
        self._finished: List[tuple] = []
        self._steps = 0              # vector steps enqueued since reset
        self._copied = 0             # ... of which this many have had their record slots copied out
        self._inflight = None        # (pinned copy, event, first step, n steps) not parsed yet
        if records:
            self.records = torch.zeros(self.drain_every, self.E, 8, dtype=torch.int32, device=dev)
            self._pins = [torch.zeros(self.drain_every, self.E, 8, dtype=torch.int32).pin_memory() for _ in range(2)]
            self._pin_i = 0
        self.reset()

    def reset(self) -> None:
        check(self.lib.naf_synth_env_reset(ptr(self.env_state), ptr(self.actor.obs), self.E, self.L.lay.A, self.seed, 0,
                                           self._preset, 23, stream_ptr()), "synth_env_reset")
        self.step_ctr.zero_()
        self._steps = self._copied = 0
        self._inflight = None
        self._finished = []

    def _body(self) -> None:
        st = stream_ptr()
        self.actor.act(self.noise_scale)                                     # NAFAgent.act for E states
        check(self.lib.naf_synth_env_step(ptr(self.env_state), ptr(self.actor.actions), ptr(self.rows),
                                          ptr(self.actor.obs), self.E, self.L.lay.A, self.seed, ptr(self.step_ctr),
                                          self.max_frames, ptr(self.records), self.drain_every if self.records is not None else 0,
                                          st), "synth_env_step")   # environment.step
        check(self.lib.naf_counter_add(ptr(self.step_ctr), 1, st), "counter_add")
        if self.replay is not None:
            check(self.lib.naf_replay_add_batch(self.replay.handle, ptr(self.rows), self.E, st), "replay_add_batch")

    def capture(self) -> None:
        warmup = 2
        extra = (self.env_state, self.rows, self.step_ctr, self.actor.obs, self.actor.counter, self.actor.actions)
        if self.records is not None:
            extra += (self.records,)
        snap = _StateSnapshot(self.L, self.replay, extra)
        put_back = None
        if self.replay is not None:
            # the warm-up appends rows to the ring: {head,size} come back with the snapshot, the ring slots it wrote
            # (possibly live rows of a full ring) are saved and put back here
            head = int(self.replay.meta[0].item())
            pos = (head + torch.arange(warmup * self.E, device=self.L.dev)) % self.replay.buffer_size
            saved = self.replay.rows[pos].clone()

            def put_back():
                self.replay.rows[pos] = saved
        self.graph = _capture(self._body, snap, warmup=warmup, after_warmup=put_back)

    def step(self) -> None:
        """One vector step: E env transitions appended to the replay ring (asynchronous)."""
        if self.use_graph:
            if self.graph is None:
                self.capture()
            self.graph.replay()
        else:
            self._body()
        if self.replay is not None:
            self.replay._total_added += self.E
        self.env_steps += self.E
        self._steps += 1
        if self.records is not None and self._steps - self._copied == self.drain_every:
            self._copy_out()

    # ---- episode records ----------------------------------------------------------------------------------------------
    def _copy_out(self) -> None:
        """Enqueue the copy of the record slots of steps [_copied, _steps) behind them; parse the copy before it."""
        self._parse_inflight()
        n = self._steps - self._copied
        pin = self._pins[self._pin_i]
        self._pin_i ^= 1
        pin.copy_(self.records, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._inflight = (pin, ev, self._copied, n)
        self._copied = self._steps

    def _parse_inflight(self) -> None:
        if self._inflight is None:
            return
        pin, ev, first, n = self._inflight
        self._inflight = None
        ev.synchronize()
        rec = pin.numpy().view(EPISODE_RECORD).reshape(self.drain_every, self.E)
        # steps first .. first+n-1 sit in slots (first + j) % drain_every; first is a multiple of drain_every
        for j in range(n):
            row = rec[(first + j) % self.drain_every]
            for e in np.nonzero(row["frames"] > 0)[0]:
                r = row[e]
                self._finished.append((float(r["score"]), int(r["frames"]), int(r["done"]), float(r["last_reward"]),
                                       int(e), int(r["episode"]), first + j))

    def drain(self, final: bool = False) -> List[tuple]:
        """Finished episodes the host has seen since the last call, in (step, env) order:
        (score, frames, done, last_reward, env, env's episode ordinal, vector step). final=True waits for the GPU and
        returns everything up to the last enqueued step."""
        if self.records is None:
            raise _lib.NafHipError("DeviceEnvLoop(records=False) keeps no episode records")
        if final:
            if self._steps > self._copied:
                self._copy_out()
            self._parse_inflight()
        out, self._finished = self._finished, []
        return out
