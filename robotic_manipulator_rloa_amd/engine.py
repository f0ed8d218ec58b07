"""HIP-graph engines around the Learner: the steady-state loop of NAFAgent.step()/run()
(reference naf_components/naf_algorithm.py:129-156, :228-270) with zero host<->device synchronisation.

  UpdateChunk    : [sample U minibatches] -> [gather U*B rows, one launch] -> U x learn()  as ONE graph (TrainChunk: its old name).
  TimestepGraph  : one tick of NAFAgent.step(): [append the transition] -> the tick's updates -> [act() on the next state]; with one
                   update per tick in four forms up to the PIPELINED one (_Pipeline: six launches, the host waits for the first,
                   the prefetch of a later minibatch on a stream of its own).
  DeviceEnvLoop  : E synthetic arms stepped on the device: act (eval-mode policy + noise) -> env step ->
                   append E transitions to the HBM ring, as ONE graph; followed by an UpdateChunk with
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


class UpdateChunk:
    """U consecutive learn() updates on U freshly sampled (or teacher-forced) minibatches, as one graph:
    [sample U minibatches] -> [gather U*B rows, one launch] -> [moments of all U] -> U x learn(), the optimizer step of update k
    riding on the first two launches of update k + 1. The many-env loops (DeviceEnvLoop + this), bench.py's headline and the
    long parity runs use it; the reference's own per-timestep loop is TimestepGraph below."""

    def __init__(self, learner: Learner, replay: ReplayBuffer, n_updates: int, teacher_forced: bool = False,
                 use_graph: bool = True, gather_outside_graph: bool = False):
        """gather_outside_graph: launch sample+gather eagerly in front of the graph of U updates, so the caller can
        bracket the gather launch with events (bench.py's live roofline measurement)."""
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

    def _updates(self, moments_ready: bool = False, defer_last: bool = False) -> None:
        if self.moments is not None and not moments_ready:
            self.L.moments(self.batch.view(self.U * self.L.B, -1), self.moments, self.U)
        # a chain of updates: the optimizer step of update k rides on the first two launches of update k + 1 (one launch
        # less per update, Learner.defer_ok); the last one of the chunk takes its step as a launch of its own (defer_last: the
        # caller's next launch carries it), so the parameter buffers are current whenever anything outside the chunk looks
        d = self.L.defer_ok
        for k in range(self.U):
            self.L.learn_rows(self.batch[k], self.loss_parts[k], None if self.moments is None else self.moments[k],
                              pending=d and k > 0, defer=d and (k < self.U - 1 or defer_last))

    def _body(self) -> None:
        self._sample_gather()
        self._updates()

    def _snapshot(self, extra=()) -> _StateSnapshot:
        return _StateSnapshot(self.L, self.replay, (self.idx, self.batch, self.loss_parts) + tuple(extra))

    def capture(self) -> None:
        self.replay.flush()
        snap = self._snapshot()
        if self.gather_outside_graph:
            self._sample_gather()          # the updates need a valid batch to warm up on
            self.graph = _capture(self._updates, snap)
        else:
            self.graph = _capture(self._body, snap)

    def run(self) -> None:
        """Enqueue the chunk (asynchronous). With teacher forcing, fill self.idx first."""
        self.L.raise_on_device_error()         # pinned host words written by the kernels: costs two loads, never a sync
        if not self.use_graph:
            self._body()
            return
        if self.graph is None:
            self.capture()
        if self.gather_outside_graph:
            self._sample_gather()
        self.graph.replay()

    def losses(self) -> torch.Tensor:
        """[U] MSE losses of the last run (device tensor; summing the per-workgroup parts in index order)."""
        return self.loss_parts.sum(dim=1)


TrainChunk = UpdateChunk                   # (the name rounds 1 - 5 used; tests and benchmarks still say it)

STEP_FORMS = ("separate", "fused", "prefetch", "pipelined")


def step_form() -> str:
    """NAF_STEP_FORM (tests and A/B measurements only): which form of the per-timestep path TimestepGraph builds where the shape
    allows every one of them — `separate` (twelve launches), `fused` (naf_step_prep + chain + naf_adam_polyak_act), `prefetch`
    (+ the last launch draws the next minibatch), `pipelined` (default)."""
    f = os.environ.get("NAF_STEP_FORM", "pipelined")
    if f not in STEP_FORMS:
        raise ValueError(f"NAF_STEP_FORM={f!r}: one of {STEP_FORMS}")
    return f


class TimestepGraph(UpdateChunk):
    """One tick of NAFAgent.step()'s update schedule (naf_algorithm.py:144-156) as one graph:
    [append this timestep's transition] -> sample U minibatches -> gather -> U x learn() -> [tail: act() on the next state].

    head_row: pinned [1, row_floats (+ 4)] tensor: the graph STARTS by appending that one transition to the ring
        (ReplayBuffer.add of the timestep inside the graph of its update). The caller fills the row and counts the transition
        (replay._total_added) before every run(head_rows=1); a run(head_rows=0) (an idle tick of a data-parallel run(), a timestep
        whose row went through the staging area) appends nothing: the node reads its row count from a word run() sets.
    tail: callable enqueued behind the last update (NAFAgent puts the NEXT timestep's act() there); tail_state: device tensors it
        changes, so that the capture's warm-up leaves no trace in them.

    With ONE update per tick on the row-split chain (the reference's own loop) the tick has four forms, each bit-equal to the one
    before (`form`):
      separate   counted append, draw, counter, gather, moments, the chain's five, optimizer step, act(): twelve launches
      fused      naf_step_prep (append + draw + gather + moments) + chain + naf_adam_polyak_act (optimizer step + act()): seven
      prefetch   ... whose last launch also draws the NEXT tick's minibatch; the next naf_step_prep then only appends
      pipelined  (one GPU, graphs) the chain of the next update runs BEFORE its transition exists, while the host steps the
                 environment: _Pipeline below — six launches per tick of which the host waits for the first
    """

    def __init__(self, learner: Learner, replay: ReplayBuffer, n_updates: int, use_graph: bool = True, tail=None, tail_state=(),
                 head_row=None):
        super().__init__(learner, replay, n_updates, use_graph=use_graph)
        self.tail, self._tail_state = tail, tuple(tail_state)
        self.head_row = head_row
        self.head_count = torch.zeros(1, dtype=torch.int32).pin_memory() if head_row is not None else None
        rf, dev, B = learner.lay.row_floats, learner.dev, learner.B
        row_with_count = head_row is not None and head_row.numel() >= rf + 4
        if row_with_count:
            # [row (rf floats) | count (int32) | 0 0 0] — one contiguous piece of host memory, so the per-timestep path can hand
            # both to the GPU in one store (head_dev below)
            self.head_count = head_row.view(-1)[rf:rf + 1].view(torch.int32)
        self._head_count_np = self.head_count.numpy() if self.head_count is not None else None
        self._ran = None                   # event behind the last run(): the pinned words are free again once it has passed
        want = STEP_FORMS.index(step_form())
        actor = getattr(tail, "__self__", None)
        # (naf_step_prep holds the appended row as 32 float4 of LDS: ring rows of up to 128 floats — every shape the row-split chain takes)
        can_prep = self.U == 1 and self.moments is not None and learner.lay.row_floats <= 128
        can_tail = isinstance(actor, ActPath) and actor.can_ride and learner.defer_ok
        self.fused_prep = want >= 1 and can_prep
        self.fused_tail = want >= 1 and can_tail
        self._tail_actor = actor if self.fused_tail else None
        both = self.fused_prep and self.fused_tail
        # with both: the first launch leaves a device copy of the pinned row, and the last takes the policy's observation from its
        # next_state columns (the state the loop asks about next IS the transition's next state) — one read of host memory per
        # timestep instead of two, and none on the last launch's critical path
        self.row_dev = None
        if both and head_row is not None:
            self.row_dev = torch.zeros(rf, dtype=torch.float32, device=dev)
            self._obs_ptr = self.row_dev.data_ptr() + 4 * learner.lay.off_s2
        # fused_prep with a [row | count] head row: the launch reads both from DEVICE memory that the host stores into directly
        # (HostStoreRow: a library allocation the CPU can reach, proven by a self-test at construction) — its first dependent load
        # is then a local-memory latency (~0.8 us) instead of a PCIe round trip to pinned host memory (~2.8 us)
        self.head_dev: Optional[HostStoreRow] = None
        if self.fused_prep and row_with_count:
            self.head_dev = HostStoreRow.try_create(learner.lib, dev, head_row, 4 * (rf + 1))
        # prefetch: the last launch of a tick also draws, gathers and takes the moments of the NEXT tick's minibatch (one more
        # workgroup, beside its own work and behind the action's announcement: what a tick draws depends on the row it appends only
        # through the fill level — and through the row itself if the draw picks it, which the record says); the next tick's first
        # launch then only appends its row (csrc/step_path.hip, step_prep_body)
        self.spec_rec = self.idx_spec = self._prefetch = None
        if both and want >= 2:
            self.spec_rec = torch.zeros(12, dtype=torch.int32, device=dev)
            self.idx_spec = torch.zeros(B, dtype=torch.int32, device=dev)
            r = replay
            self._prefetch = _lib.StepPrefetch(r.handle, r.seed, ptr(r._sample_ctr), ptr(self.idx_spec), ptr(self.batch),
                                               self.batch.shape[-1], r.action_mode, ptr(self.moments), B,
                                               int(r.without_replacement), ptr(self.spec_rec), 1)
        self.pipe: Optional[_Pipeline] = None
        pipe_ok = self._prefetch is not None and want >= 3 and self.head_dev is not None and use_graph
        if learner.world_size == 1:
            if pipe_ok and learner.fold_norm and not learner._force_allreduce:
                self.pipe = _Pipeline(self)
        elif want >= 3 and use_graph and row_with_count and can_prep and can_tail:
            # Data parallel (round 6): the ranks must run the SAME graph every tick — one learn() chain, one gradient exchange, when
            # the tick's prefetches hold, two when it starts over — so they vote per tick (parallel.TickAgreement: host to host
            # through shared memory). Collective: every rank builds this object at the same tick (the gate under data parallel is
            # the tick count) and on rank-independent conditions; what may differ per rank is agreed before anybody pipelines.
            from .parallel import TickAgreement, _agree
            agree = TickAgreement.try_create(learner.pg)
            if _agree(pipe_ok and agree is not None, dev, learner.pg):
                self.pipe = _Pipeline(self, agree)
            elif agree is not None:
                agree.close()
        # fused tail: the launch hands its action to the host as self-validating 16-byte chunks {three components, ordinal}
        # (ActPath.act_rec) — the host learns that a run() has passed by polling the chunks' ordinals instead of synchronising an
        # event / the stream, and takes the action from the chunks (wait_tail). _seq_np = chunk 0's ordinal.
        self._seq_np = actor.ordinal_np if self.fused_tail else None
        self._seq_prev = 0
        self._inflight = False             # a run() whose ordinal the host has not seen yet
        self._err_np = learner.err_host.numpy()

    # ---- what tests and the agent ask -----------------------------------------------------------------------------------------
    @property
    def pipelined(self) -> bool:
        return self.pipe is not None

    @property
    def form(self) -> str:
        return ("pipelined" if self.pipe is not None else "prefetch" if self._prefetch is not None else
                "fused" if (self.fused_prep or self.fused_tail) else "separate")

    @property
    def fast_runs(self) -> int:
        return self.pipe.fast_runs if self.pipe is not None else 0

    @property
    def slow_runs(self) -> int:
        return self.pipe.slow_runs if self.pipe is not None else 0

    def prefetch_stats(self) -> Tuple[int, int]:
        """(ticks that took a prefetched minibatch, ticks that drew for themselves) since the graph was built; (0, 0) without the
        prefetch. Synchronises (the pipelined form counts on the host and does not)."""
        if self.pipe is not None:
            return self.pipe.fast_runs, self.pipe.slow_runs
        if self.spec_rec is None:
            return 0, 0
        r = self.spec_rec.cpu()
        return int(r[8]), int(r[9])

    def error_words(self) -> Dict[str, int]:
        """The path's pinned error counters (never a synchronisation): polls inside naf_adam_polyak_act that ran into their bound,
        pipelined ticks whose record did not hold on the device, host-side waits for a verdict that had to synchronise."""
        return {"act_poll_timeouts": int(self._err_np[1]), "pipe_errors": int(self._err_np[2]),
                "verdict_waits_synchronised": self.pipe.sync_waits if self.pipe is not None else 0}

    # ---- the graph's body ---------------------------------------------------------------------------------------------------------
    def _finish(self) -> None:
        if self.fused_tail:
            # the last update's clip + Adam + Polyak and the tail's act(): one launch
            self._tail_actor.act_with_optimizer_step(obs_ptr=self._obs_ptr if self.row_dev is not None else None,
                                                     prefetch=self._prefetch)
        elif self.tail is not None:
            self.tail()

    def _body(self) -> None:
        if self.pipe is not None:
            self.pipe.body_slow()
            return
        if self.fused_prep:
            r = self.replay
            src, cnt = ptr(self.head_row), ptr(self.head_count)
            if self.head_dev is not None:
                src, cnt = self.head_dev.ptr, self.head_dev.ptr + 4 * self.L.lay.row_floats
            check(self.L.lib.naf_step_prep(r.handle, src, cnt, ptr(self.row_dev), r.seed,
                                           ptr(r._sample_ctr), ptr(self.idx), ptr(self.batch), self.batch.shape[-1], r.action_mode,
                                           ptr(self.moments), self.L.B, int(r.without_replacement), ptr(self.spec_rec),
                                           ptr(self.idx_spec), None, stream_ptr()), "naf_step_prep")
            self._updates(moments_ready=True, defer_last=self.fused_tail)
        else:
            if self.head_row is not None:
                check(self.L.lib.naf_replay_add_counted(self.replay.handle, ptr(self.head_row), ptr(self.head_count), 1, stream_ptr()),
                      "naf_replay_add_counted")
            self._sample_gather()
            self._updates(defer_last=self.fused_tail)
        self._finish()

    def capture(self) -> None:
        self.replay.flush()
        pipe = self.pipe
        if pipe is not None:
            pipe.prepare()
        # (the prefetch's record among them: the warm-up's last run leaves a valid one, for a ring that is put back)
        snap = self._snapshot(self._tail_state + ((self.spec_rec,) if self.spec_rec is not None else ()) +
                              (pipe.state() if pipe is not None else ()))
        if self.head_row is None:
            self.graph = _capture(self._body, snap)
            return
        # the warm-up appends rows to the ring: {head, size} come back with the snapshot, the ring slots it wrote
        # (live rows, if the ring is full) are saved and put back
        warmup = 2
        head = int(self.replay.meta[0].item())
        pos = (head + torch.arange(warmup, device=self.L.dev)) % self.replay.buffer_size
        saved = self.replay.rows[pos].clone()
        count = int(self.head_count[0])
        self.head_count[0] = 1             # the warm-up runs really append (and are undone)
        if self.head_dev is not None:
            self.head_dev.publish()

        def put_back():
            self.replay.rows[pos] = saved
            self.head_count[0] = count
        self.graph = _capture(self._body, snap, warmup=warmup, after_warmup=put_back)
        if pipe is not None:
            pipe.capture_fast(snap)

    # ---- hand-overs with the host -----------------------------------------------------------------------------------------------
    def wait_pinned_free(self) -> None:
        """Block until the last run() has passed: what it reads from pinned host memory (the head row, its count, a tail's
        observation) may be rewritten afterwards. In NAFAgent.run's loop act() has waited for it already."""
        if self._seq_np is not None:
            if self._inflight:
                self.wait_tail()
        elif self._ran is not None:
            self._ran.synchronize()

    def wait_tail(self) -> None:
        """Block until the last run()'s tail has written its action to pinned host memory. Fused tail: a spin on the ordinals of
        the action's chunks (no stream synchronisation: the hipStreamSynchronize round trip was a tenth of a timestep); other
        tails: the stream."""
        if self._seq_np is None:
            torch.cuda.current_stream().synchronize()
            return
        if not self._inflight:
            return
        sq, prev, n = self._seq_np, self._seq_prev, 0
        a = self._tail_actor
        rec, ords = a.rec_np, a._rec_ords
        while True:
            # every chunk the action needs carries the new ordinal (the chunks are stores of their own: they may land in any order)
            want = sq[0]
            if want != prev and all(rec[w] == want for w in ords):
                break
            n += 1
            if n > 4000000:                 # (seconds: something is wrong — let the runtime say what)
                torch.cuda.current_stream().synchronize()
                want = sq[0]
                if want == prev or not all(rec[w] == want for w in ords):
                    raise _lib.NafHipError("the update graph finished without its tail launch writing an action")
                break
        a.actions_np[0, :] = a.rec_f[a._rec_words]
        self._inflight = False
        first = a.actions_np[0, 0]
        if first != first:
            # a NaN action is how the launch says that one of its bounded polls ran out; the counter that says so travels as a
            # store of its own and may still be on its way: let the stream drain before reading it
            torch.cuda.synchronize()
        if self._err_np[2]:
            raise _lib.NafHipError("the pipelined timestep found its prefetched minibatch not to hold although the host had read that "
                                   "it does (engine._Pipeline.decide): the update is not valid")
        if self._err_np[1]:
            raise _lib.NafHipError(f"naf_adam_polyak_act: {int(self.L.err_host[1])} polls inside the launch ran into their 2-ms bound "
                                   "(the GPU is over-subscribed or a workgroup died): the action is not valid")

    def run(self, head_rows: int = 0) -> None:
        """Enqueue the tick (asynchronous). head_rows (graphs built with head_row): 1 = the pinned row holds a new transition
        that the graph's first node appends, 0 = it appends nothing. The caller has waited (wait_pinned_free) before it rewrote
        the row."""
        self.L.raise_on_device_error()         # pinned host words written by the kernels: costs two loads, never a sync
        if self.use_graph and self.graph is None:
            self.capture()
        if self.head_row is not None:
            # the previous run's append node must have read its count before the word changes (an idle tick right behind a
            # step() would otherwise zero the count of a row the GPU has not appended yet); in run()'s loop act() has waited
            self.wait_pinned_free()
            self.head_count[0] = 1 if head_rows else 0
        if self._seq_np is not None:
            if self._inflight:
                self.wait_tail()               # (its ordinal must be in before the next launch's "before" value is read)
            self._seq_prev = int(self._seq_np[0])
        if self.pipe is not None:
            self.pipe.launch(bool(head_rows))
        else:
            if self.head_dev is not None:
                self.head_dev.publish()
            if self.use_graph:
                self.graph.replay()
            else:
                self._body()
        if self._seq_np is not None:
            self._inflight = True
        elif self.head_row is not None or self.tail is not None:
            if self._ran is None:
                self._ran = torch.cuda.Event()
            self._ran.record()

    def run_row(self) -> None:
        """run(head_rows=1) for the caller that has done run()'s checks itself (NAFAgent.step's per-timestep path: the graph
        exists, the previous run has passed, the pinned row is filled)."""
        self._head_count_np[0] = 1
        self._seq_prev = int(self._seq_np[0])
        if self.pipe is not None:
            self.pipe.launch(True)
        else:
            if self.head_dev is not None:
                self.head_dev.publish()
            self.graph.replay()
        self._inflight = True


class HostStoreRow:
    """A few hundred bytes of DEVICE memory the host stores into directly (large BAR): the hand-over of a timestep's
    [transition row | count] to the graph's first launch without a PCIe read on the launch's critical path (csrc/lib.hip,
    naf_host_publish; the reading kernels use system-scope loads).

    The memory is the library's own hipMalloc — not the framework allocator's, whose segments need not be CPU-mapped
    (expandable segments, pools) — and the hand-over is PROVEN at construction: 1000 distinct patterns are stored through the very
    call the path uses and read back by a kernel with the path's own loads (naf_host_store_selftest). A device without a large BAR,
    an allocation the CPU cannot reach, or one mismatch -> try_create() returns None with a warning, and the launch reads the
    pinned row across PCIe instead (NAF_HOST_STORE=0 forces that)."""

    def __init__(self, lib, dev_ptr: int, src: torch.Tensor, n_bytes: int):
        self.lib, self.ptr, self._src, self.n_bytes = lib, int(dev_ptr), src.data_ptr(), int(n_bytes)
        self._keep = src

    @classmethod
    def try_create(cls, lib, device, src: torch.Tensor, n_bytes: int) -> Optional["HostStoreRow"]:
        if os.environ.get("NAF_HOST_STORE", "1") == "0" or lib.naf_host_store_supported(device.index or 0) != 1:
            return None
        p = _lib.C.c_void_p()
        if lib.naf_host_store_alloc(4 * ((n_bytes + 3) // 4 + 4), _lib.C.byref(p)) != 0 or not p.value:
            return None
        bad = lib.naf_host_store_selftest(p.value, 1000, stream_ptr())
        if bad != 0:
            import warnings
            warnings.warn(f"naf_host_store_selftest: {bad} (mismatching patterns, or an error if negative) — the per-timestep path "
                          "hands its transition row over through pinned host memory instead of device memory", RuntimeWarning)
            lib.naf_host_store_free(p.value)
            return None
        return cls(lib, p.value, src, n_bytes)

    def publish(self) -> None:
        self.lib.naf_host_publish(self.ptr, self._src, self.n_bytes)

    def __del__(self):
        try:
            if self.ptr:
                torch.cuda.synchronize()
                self.lib.naf_host_store_free(self.ptr)
                self.ptr = 0
        except Exception:
            pass


class _Pipeline:
    """The pipelined form of TimestepGraph (one GPU). Timestep t's graph is
        naf_adam_polyak_act  [clip + Adam + Polyak with the gradient that is WAITING (taken from minibatch t while the host stepped
                              the environment), act(s_t+1), commit working -> public state]      -> the action: what the host waits for
        the chain's five launches on minibatch t + 1                                             -> the gradient that waits next
    and BESIDE it, on a stream of its own (round 6; round 5 ran it as the first launch's extra workgroup and the chain waited
    4 - 9 us for it):
        step_prefetch_kernel [append row t, hand over minibatch t's indices, draw + gather + moments of minibatch t + 2 on the ring
                              as two more appends will leave it, verdict -> pinned host word]
    Three sets of {minibatch, moments, indices, record, verdict} rotate: timestep t consumes set p (`phase`), its chain reads set
    p + 1, its prefetch fills set p + 2 — which the chain of timestep t - 2's graph read last, and that graph had finished before
    the host saw the action of timestep t - 1. The host takes this graph iff the tick brings a row, both sets it needs are valid
    (verdicts: the rows to come were not among the positions drawn) and nothing has touched the ring, the sampler's stream or the
    learner since the last tick (call counters and torch's version counters). Otherwise the OTHER graph starts the timestep over
    on the public state: naf_step_prep (reset working <- public, append, draw minibatch t into set 0) + chain + naf_adam_polyak_act
    (+ depth-1 prefetch of t + 1 into set 1) + depth-2 prefetch of t + 2 into set 2 + chain — and leaves phase 1.
    Ordering between the two streams is the host's: a launch is only made after every verdict owed by earlier launches has been
    read, and a verdict is stored behind a release of everything its workgroup wrote."""

    def __init__(self, tg: TimestepGraph, agree=None):
        self.tg = tg
        self.agree = agree                         # data parallel: the ranks' per-tick vote on which graph runs (parallel.TickAgreement)
        L, r = tg.L, tg.replay
        B, dev, brf = L.B, L.dev, L.lay.batch_row_floats
        # set 0 = the graph's own buffers (what a reader of tg.batch / tg.moments finds after a timestep that started over)
        self._stores = [tg._batch_store] + [torch.zeros(B * brf + 64, dtype=torch.float32, device=dev) for _ in range(2)]
        self.batches = [s[:B * brf].view(B, brf) for s in self._stores]
        self.moments = [tg.moments[0]] + [torch.zeros_like(tg.moments[0]) for _ in range(2)]
        self.spec_rec = torch.zeros(3, 12, dtype=torch.int32, device=dev)
        self.idx_spec = torch.zeros(3, B, dtype=torch.int32, device=dev)
        self.pf_seq = torch.zeros(4, dtype=torch.int32, device=dev)
        self.host_spec = torch.zeros(3, 2, dtype=torch.int32).pin_memory()
        self._hs = self.host_spec.numpy().view(np.uint64).reshape(3)       # {ordinal, valid}: ONE 8-byte store of the launch, one load here
        self.side = torch.cuda.Stream(device=dev)
        self._side_raw = self.side.cuda_stream
        self._dev_index = dev.index if dev.index is not None else torch.cuda.current_device()
        self.phase = 0
        self.valid = [False, False, False]
        self.owed: List[Tuple[int, int]] = []      # (set, ordinal) of verdicts launched and not read yet
        self.n_verdicts = 0                        # host mirror of *pf_seq
        self.armed = False                         # the last launch was one of this object's graphs
        self._sig, self._r_total = None, -1
        self.fast_runs = self.slow_runs = self.sync_waits = 0
        self.graph_fast: List[Optional[torch.cuda.CUDAGraph]] = [None, None, None]
        self._exec = self._exec_fast = None
        r._side_join = self.join                   # (ReplayBuffer's own launches read what the side stream writes)
        self.l1_ride = os.environ.get("NAF_STEP_L1_RIDE", "1") != "0"      # (tests / A-B only: layer 1 as a launch of its own)
        # The loop host -> launch -> action -> env.step -> verdict -> launch has TWO stable states at B = 256, where the side launch
        # runs 17 us: one in which the verdict is in when the host asks (35 us per timestep), and one in which every tick waits 10 us
        # for a verdict that left late because the tick before waited too (43 us; a host hiccup is enough to fall into it, and whole
        # processes stayed in it: benchmarks/debug/api_step_histogram.py). A tick that had to wait launches the prefetch BEFORE the
        # graph (naf_step_launch, prefetch_first): its verdict leaves 8 us sooner, the host gets ahead again.
        self.waited = False                        # the last collect() found a verdict missing
        self.side_first_runs = 0
        self._l1_args = [None, None, None]

    def state(self) -> tuple:
        """device tensors a warm-up run changes"""
        return (self.spec_rec, self.idx_spec, self.pf_seq, self.bn_work, self.step_work, self.loss_work) + tuple(self._stores[1:])

    def prepare(self) -> None:
        """working copies and argument structures (at capture time: the public tensors' addresses are final by then)"""
        tg = self.tg
        L, r, a = tg.L, tg.replay, tg._tail_actor
        H, rf, B = L.lay.H, L.lay.row_floats, L.B
        self.bn_work = L.bn_stats.clone()
        self.step_work = L.step_dev.clone()
        self.loss_work = torch.zeros_like(tg.loss_parts[0])
        bw = self.bn_work.data_ptr()
        self._net_work = _lib.ActNet.from_buffer_copy(a._net)
        self._net_work.running_mean1, self._net_work.running_var1 = bw, bw + 4 * H
        self._net_work.running_mean2, self._net_work.running_var2 = bw + 8 * H, bw + 12 * H
        self._adam_work = _lib.AdamArgs.from_buffer_copy(L._adam_args)
        self._adam_work.step_dev = self.step_work.data_ptr()
        commit = _lib.StepCopies.of((bw, L.bn_stats.data_ptr(), 8 * H),
                                    (self.loss_work.data_ptr(), tg.loss_parts.data_ptr(), self.loss_work.numel()),
                                    (self.step_work.data_ptr(), L.step_dev.data_ptr(), 1))
        self._reset = _lib.StepCopies.of((L.bn_stats.data_ptr(), bw, 8 * H), (L.step_dev.data_ptr(), self.step_work.data_ptr(), 1))
        rowp = tg.head_dev.ptr
        errp = L.err_host.data_ptr() + 16
        seq = self.pf_seq.data_ptr()

        def pf(k, mode, depth, **kw):
            s = _lib.StepPrefetch(r.handle, r.seed, ptr(r._sample_ctr), self.idx_spec[k].data_ptr(), ptr(self.batches[k]),
                                  self.batches[k].shape[-1], r.action_mode, ptr(self.moments[k]), B, int(r.without_replacement),
                                  self.spec_rec[k].data_ptr(), mode)
            s.host_spec, s.pipe_errors, s.depth, s.pf_seq = self.host_spec[k].data_ptr(), errp, depth, seq
            for name, v in kw.items():
                setattr(s, name, v)
            return s
        self._pf_commit = _lib.StepPrefetch()                      # mode 0: the first launch of the fast graph commits, nothing else
        self._pf_commit.copies = commit
        self._pf_mid = pf(1, 1, 1, copies=commit)                  # started-over graph: the optimizer launch's workgroup, set 1
        self._pf_far = pf(2, 1, 2)                                 # ... and a launch of its own behind it, set 2
        # timestep in phase p: consumes set p, fills set p + 2 (a launch of its own on the side stream)
        self._pf_side = [pf((p + 2) % 3, 2, 2, src_row=rowp, n_word=rowp + 4 * rf, idx_out=ptr(tg.idx),
                            spec_rec_in=self.spec_rec[p].data_ptr(), idx_spec_in=self.idx_spec[p].data_ptr()) for p in range(3)]

    def _chain(self, k: int, l1_done: bool = False) -> None:
        """learn() on set k with the learner's working state; the optimizer step is left to the next graph's first launch.
        l1_done: the chain's first launch (layer 1) rode on the launch in front (body_fast)."""
        L = self.tg.L
        L.bn_live, L.step_live = self.bn_work, self.step_work
        try:
            L.learn_rows(self.batches[k], self.loss_work, self.moments[k], pending=False, defer=True, l1_done=l1_done)
        finally:
            L.bn_live, L.step_live = L.bn_stats, L.step_dev

    def body_fast(self, p: int) -> None:
        """phase p, the prefetches held: apply the waiting gradient + act() + commit in ONE launch, then the chain for the next
        update (six launches; the host waits for the first)"""
        tg = self.tg
        k = (p + 1) % 3
        # (round 6: layer 1 of the chain rides on the first launch — extra workgroups behind its optimizer step — so the chain proper
        #  starts at GEMM 2 without waiting for the act() tail or a launch boundary; NAF_STEP_L1_RIDE=0: as a launch of its own)
        l1 = None
        if self.l1_ride:
            tg.L.bn_live = self.bn_work
            try:
                l1 = self._l1_args[k] = tg.L.layer1_args(self.batches[k], self.moments[k])      # (kept alive: the capture reads it now)
            finally:
                tg.L.bn_live = tg.L.bn_stats
        tg._tail_actor.act_with_optimizer_step(obs_ptr=tg.head_dev.ptr + 4 * tg.L.lay.off_s2, prefetch=self._pf_commit,
                                               obs_system_scope=True, adam_args=self._adam_work, net=self._net_work, layer1=l1)
        self._chain(k, l1_done=l1 is not None)

    def body_slow(self) -> None:
        """they did not (or nothing was prefetched): reset the working state, draw, chain — and from there as the other graph"""
        tg = self.tg
        r, L = tg.replay, tg.L
        src = tg.head_dev.ptr
        check(L.lib.naf_step_prep(r.handle, src, src + 4 * L.lay.row_floats, ptr(tg.row_dev), r.seed, ptr(r._sample_ctr), ptr(tg.idx),
                                  ptr(self.batches[0]), self.batches[0].shape[-1], r.action_mode, ptr(self.moments[0]), L.B,
                                  int(r.without_replacement), None, None, _lib.C.byref(self._reset), stream_ptr()), "naf_step_prep")
        self._chain(0)
        tg._tail_actor.act_with_optimizer_step(obs_ptr=tg._obs_ptr, prefetch=self._pf_mid, adam_args=self._adam_work,
                                               net=self._net_work)
        check(L.lib.naf_step_prefetch(_lib.C.byref(self._pf_far), stream_ptr()), "naf_step_prefetch")
        self._chain(1)

    def capture_fast(self, snap: _StateSnapshot) -> None:
        """(behind the other graph's capture: its kernels are warm — these are the same but for a template argument of the first —
        and a run of them needs prefetches that hold: captured without a warm-up)"""
        tg = self.tg
        for p in range(3):
            self.graph_fast[p] = _capture(lambda p=p: self.body_fast(p), snap, warmup=0)
        torch.cuda.synchronize()
        self.host_spec.zero_()             # (the warm-up's verdicts: ordinals start over with the restored counter)
        self.armed, self.owed, self.n_verdicts, self.valid = False, [], 0, [False, False, False]
        self._exec = self._exec_fast = None
        self._launch = tg.L.lib.naf_step_launch
        if hasattr(tg.graph, "raw_cuda_graph_exec"):
            try:
                self._exec = int(tg.graph.raw_cuda_graph_exec())
                self._exec_fast = [int(g.raw_cuda_graph_exec()) for g in self.graph_fast]
            except Exception:              # (a torch build that keeps the handle to itself)
                self._exec = self._exec_fast = None

    # ---- per timestep ---------------------------------------------------------------------------------------------------------------
    def collect(self) -> None:
        """read every verdict the launches so far owe (a few microseconds behind the action of the launch they rode beside)"""
        hs = self._hs
        self.waited = False
        for k, want in self.owed:
            n = 0
            v = int(hs[k])
            self.waited = self.waited or (v & 0xFFFFFFFF) != want
            while (v & 0xFFFFFFFF) != want:
                n += 1
                if n > 2000000:                 # (something is wrong: let the runtime say what)
                    torch.cuda.synchronize()
                    self.sync_waits += 1
                    v = int(hs[k])
                    if (v & 0xFFFFFFFF) != want:
                        raise _lib.NafHipError(f"the prefetch launch of the pipelined timestep never reported (set {k}: ordinal "
                                               f"{v & 0xFFFFFFFF}, expected {want})")
                    break
                v = int(hs[k])
            self.valid[k] = (v >> 32) == 1
        self.owed = []

    def join(self) -> None:
        """the current stream waits for whatever the side stream still runs (ReplayBuffer calls this before it reads or writes the
        ring from the current stream: a user's memory.sample() between two timesteps)"""
        if self.owed and not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream().wait_stream(self.side)

    def decide(self, brings_row: bool) -> bool:
        """About to launch: may this tick run the graph that starts with the waiting gradient? Also the bookkeeping for the
        launch that follows."""
        r, L = self.tg.replay, self.tg.L
        self.collect()                     # (always: the launches that owe them read the row's device copy the caller is about to rewrite)
        # (the call counters, and torch's version counters of the buffers: an in-place write through ANY view of them — the NAF
        #  modules' load_state_dict, a snapshot's restore — moves those; graph replays and this path's own launches do not.
        #  What stays invisible is a write through `.data` or a raw pointer: INTEGRATION.md says so.)
        sig = (r._gen, L._gen, L.theta2._version, L.bn_stats._version, L.adam_m._version, L.adam_v._version, L.step_dev._version,
               L.grad._version, L.partials._version, r.rows._version, r.meta._version, r._sample_ctr._version)
        p = self.phase
        ok = (brings_row and self.armed and sig == self._sig and r._total_added == self._r_total + 1 and r._pending == 0 and
              self.valid[p] and self.valid[(p + 1) % 3])
        self.armed = True
        self._sig, self._r_total = sig, r._total_added
        return ok

    def launch(self, brings_row: bool) -> None:
        tg = self.tg
        fast = self.decide(brings_row)
        if self.agree is not None:
            fast = self.agree.all_ok(fast)     # (every rank, every tick: an exchange pairs with an exchange)
        hd = tg.head_dev
        if fast:
            p = self.phase
            k = (p + 2) % 3
            if self._exec_fast is not None:
                # the row into device memory, the graph and the side launch in ONE foreign call
                # (the current stream as its raw handle: torch.cuda.current_stream() is 2.6 us of Python per timestep — it asks for the
                #  device count on its way — in a loop whose host side and GPU side are 35 and 35.5 us: benchmarks/debug/torch_call_costs.py)
                rc = self._launch(hd.ptr, hd._src, hd.n_bytes, self._exec_fast[p], torch._C._cuda_getCurrentRawStream(self._dev_index),
                                  _lib.C.byref(self._pf_side[p]), self._side_raw, 1 if self.waited else 0)
                if rc:
                    check(rc, "naf_step_launch")
            else:
                hd.publish()
                if self.waited:
                    check(tg.L.lib.naf_step_prefetch(_lib.C.byref(self._pf_side[p]), self._side_raw), "naf_step_prefetch")
                self.graph_fast[p].replay()
                if not self.waited:
                    check(tg.L.lib.naf_step_prefetch(_lib.C.byref(self._pf_side[p]), self._side_raw), "naf_step_prefetch")
            self.side_first_runs += self.waited
            self.n_verdicts += 1
            self.owed = [(k, self.n_verdicts & 0xFFFFFFFF)]
            self.valid[k] = False
            self.phase = (p + 1) % 3
            self.fast_runs += 1
            return
        cur = torch.cuda.current_stream()
        # start over, everything on the current stream (the side stream is idle: its verdicts have been read — the wait is for
        # the order of the memory operations, not for time)
        cur.wait_stream(self.side)
        if self._exec is not None:
            rc = self._launch(hd.ptr, hd._src, hd.n_bytes, self._exec, cur.cuda_stream, None, None, 0)
            if rc:
                check(rc, "naf_step_launch")
        else:
            hd.publish()
            tg.graph.replay()
        self.owed = [(1, (self.n_verdicts + 1) & 0xFFFFFFFF), (2, (self.n_verdicts + 2) & 0xFFFFFFFF)]
        self.n_verdicts += 2
        self.valid = [False, False, False]
        self.phase = 1
        self.slow_runs += 1


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
        if lay.A > 8:
            raise _lib.NafHipError(f"the on-device stand-in environment models arms of up to 8 joints (action_size {lay.A}): drive "
                                   "such an agent through run() / run_host_vectorized with a host environment")
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
