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
        rf = learner.lay.row_floats
        if head_row is not None and head_row.numel() >= rf + 4:
            # (a row with room for its count behind it: [row (rf floats) | count (int32) | 0 0 0] — one contiguous piece of host
            #  memory, so the per-timestep path can hand both to the GPU in one store, see head_dev below)
            self.head_count = head_row.view(-1)[rf:rf + 1].view(torch.int32)
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
        # The per-timestep shape (ONE update per chunk on the row-split chain: the reference's own loop, NAFAgent.step with
        # num_updates = 1) runs two fused launches of csrc/step_path.hip around the chain's five:
        #   fused_prep: [counted append +] sample + gather + moments in one launch (naf_step_prep) instead of five;
        #   fused_tail: the update's optimizer step and the tail's act() in one launch (naf_adam_polyak_act) instead of two —
        #               `tail` must then be the bound `act` of an ActPath that can take the step along (ActPath.can_ride).
        # NAF_STEP_FUSED=0 keeps the separate launches (A/B measurements; the bit-equality test runs both).
        fused = os.environ.get("NAF_STEP_FUSED", "1") != "0"
        self.fused_prep = (fused and self.U == 1 and self.moments is not None and not teacher_forced and not gather_outside_graph)
        actor = getattr(tail, "__self__", None)
        self.fused_tail = (fused and isinstance(actor, ActPath) and actor.can_ride and learner.defer_ok)
        self._tail_actor = actor if self.fused_tail else None
        # with both: the first launch leaves a device copy of the pinned row, and the last takes the policy's observation from its
        # next_state columns (the state the loop asks about next IS the transition's next state) — one read of host memory per
        # timestep instead of two, and none on the last launch's critical path
        self.row_dev = None
        if self.fused_prep and self.fused_tail and head_row is not None:
            self.row_dev = torch.zeros(learner.lay.row_floats, dtype=torch.float32, device=dev)
            self._obs_ptr = self.row_dev.data_ptr() + 4 * learner.lay.off_s2
        # fused_prep with a [row | count] head row: the launch reads both from DEVICE memory that the host stores into directly
        # (naf_host_publish: every device allocation is CPU-mapped here) — its first dependent load is then a local-memory
        # latency (~0.8 us) instead of a PCIe round trip to pinned host memory (~2.8 us, measured inside the kernel)
        self.head_dev = None
        if self.fused_prep and head_row is not None and head_row.numel() >= rf + 4 and os.environ.get("NAF_HOST_STORE", "1") != "0" and \
                learner.lib.naf_host_store_supported(dev.index or 0) == 1:       # (no large BAR: the kernel reads the pinned row itself)
            self.head_dev = torch.zeros(rf + 4, dtype=torch.float32, device=dev)
            self._head_src, self._head_dst, self._head_bytes = head_row.data_ptr(), self.head_dev.data_ptr(), 4 * (rf + 1)
            self._publish = learner.lib.naf_host_publish
        # with both fused launches: the last launch of a timestep also draws, gathers and takes the moments of the NEXT timestep's
        # minibatch (one more workgroup, beside its own work and behind the action's announcement to the host: what a timestep draws
        # depends on the row it appends only through the fill level — and through the row itself if the draw picks it, which the
        # record says); the next timestep's first launch then only appends its row (csrc/step_path.hip, step_prep_body).
        # NAF_STEP_PREFETCH=0: every timestep draws for itself.
        self.spec_rec = self.idx_spec = self._prefetch = None
        if self.fused_prep and self.fused_tail and os.environ.get("NAF_STEP_PREFETCH", "1") != "0":
            self.spec_rec = torch.zeros(12, dtype=torch.int32, device=dev)
            self.idx_spec = torch.zeros(B, dtype=torch.int32, device=dev)
            r = replay
            self._prefetch = _lib.StepPrefetch(r.handle, r.seed, ptr(r._sample_ctr), ptr(self.idx_spec), ptr(self.batch),
                                               self.batch.shape[-1], r.action_mode, ptr(self.moments), B,
                                               int(r.without_replacement), ptr(self.spec_rec), 1)
        # ... and PIPELINED (one GPU, graphs): with the minibatch of timestep t + 1 in place before its transition exists, its whole
        # learn() chain can run before it too — the gradient depends on the parameters update t leaves and on that minibatch, not
        # on the new row. A timestep's graph is then [naf_adam_polyak_act: append the row, apply the gradient that is waiting, act(),
        # prefetch] -> [the chain on the prefetched minibatch], and what the host waits for is the first launch: the chain runs
        # while it steps the environment. The chain works on copies of the state it advances besides the gradient (BatchNorm
        # running statistics, step count, loss partials); the first launch of the next graph commits them with the update, so the
        # learner's public buffers are exactly "after update t" between graphs. If the prefetch does not hold (the host reads the
        # verdict from pinned memory before it launches) the other graph runs: reset the working copies, draw, chain, and from
        # there as above. NAF_STEP_PIPELINE=0: the prefetch only. (_init_pipeline, at capture time.)
        self.pipelined = (self._prefetch is not None and head_row is not None and self.head_dev is not None and use_graph and
                          learner.world_size == 1 and learner.fold_norm and not learner._force_allreduce and
                          os.environ.get("NAF_STEP_PIPELINE", "1") != "0")
        self.graph_fast = None
        self._exec_fast = None
        self._spec_armed = False           # the last launch was one of the pipelined graphs: a verdict on its prefetch will come
        self._sig, self._r_total = None, -1
        self.fast_runs = self.slow_runs = 0
        # fused tail: the launch hands its action to the host as self-validating 16-byte chunks {three components, ordinal}
        # (ActPath.act_rec) — the host learns that a run() has passed by polling the chunks' ordinals instead of synchronising an
        # event / the stream, and takes the action from the chunks (wait_tail). _seq_np = chunk 0's ordinal.
        self._seq_np = actor.ordinal_np if self.fused_tail else None
        self._seq_prev = 0
        self._inflight = False             # a run() whose ordinal the host has not seen yet
        self._exec = None
        self._err_np = learner.err_host.numpy()
        self._head_count_np = self.head_count.numpy() if self.head_count is not None else None

    def prefetch_stats(self) -> Tuple[int, int]:
        """(timesteps that took the minibatch the previous timestep's last launch had prefetched, timesteps that drew for
        themselves) since the chunk was built; (0, 0) without the prefetch. Synchronises."""
        if self.spec_rec is None:
            return 0, 0
        r = self.spec_rec.cpu()
        return int(r[8]), int(r[9])

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

    def _updates(self, moments_ready: bool = False) -> None:
        if self.moments is not None and not moments_ready:
            self.L.moments(self.batch.view(self.U * self.L.B, -1), self.moments, self.U)
        # a chain of updates: the optimizer step of update k rides on the first two launches of update k + 1 (one launch
        # less per update, Learner.defer_ok); the last one of the chunk takes its step as a launch of its own, so the
        # parameter buffers are current whenever anything outside the chunk looks at them
        d = self.L.defer_ok
        for k in range(self.U):
            self.L.learn_rows(self.batch[k], self.loss_parts[k], None if self.moments is None else self.moments[k],
                              pending=d and k > 0, defer=d and (k < self.U - 1 or self.fused_tail))
        if self.fused_tail:
            # the last update's clip + Adam + Polyak and the tail's act(): one launch
            self._tail_actor.act_with_optimizer_step(obs_ptr=self._obs_ptr if self.row_dev is not None else None,
                                                     prefetch=self._prefetch)
        elif self.tail is not None:
            self.tail()

    def _init_pipeline(self) -> None:
        """working copies, argument structures and the verdict word of the pipelined graphs (at capture time: the public tensors'
        addresses are final by then)"""
        L, r, a = self.L, self.replay, self._tail_actor
        H, rf, B = L.lay.H, L.lay.row_floats, L.B
        self.bn_work = L.bn_stats.clone()
        self.step_work = L.step_dev.clone()
        self.loss_work = torch.zeros_like(self.loss_parts[0])
        self.host_spec = torch.zeros(2, dtype=torch.int32).pin_memory()
        self._host_spec_np = self.host_spec.numpy().view(np.uint64)      # {ordinal, valid}: ONE 8-byte store of the launch, one load here
        bw = self.bn_work.data_ptr()
        self._net_work = _lib.ActNet.from_buffer_copy(a._net)
        self._net_work.running_mean1, self._net_work.running_var1 = bw, bw + 4 * H
        self._net_work.running_mean2, self._net_work.running_var2 = bw + 8 * H, bw + 12 * H
        self._adam_work = _lib.AdamArgs.from_buffer_copy(L._adam_args)
        self._adam_work.step_dev = self.step_work.data_ptr()
        commit = _lib.StepCopies.of((bw, L.bn_stats.data_ptr(), 8 * H),
                                    (self.loss_work.data_ptr(), self.loss_parts.data_ptr(), self.loss_work.numel()),
                                    (self.step_work.data_ptr(), L.step_dev.data_ptr(), 1))
        self._reset = _lib.StepCopies.of((L.bn_stats.data_ptr(), bw, 8 * H), (L.step_dev.data_ptr(), self.step_work.data_ptr(), 1))
        rowp = self.head_dev.data_ptr()
        common = (r.handle, r.seed, ptr(r._sample_ctr), ptr(self.idx_spec), ptr(self.batch), self.batch.shape[-1], r.action_mode,
                  ptr(self.moments), B, int(r.without_replacement), ptr(self.spec_rec))
        errp = L.err_host.data_ptr() + 16
        self._pf_fast = _lib.StepPrefetch(*common, 2, rowp, rowp + 4 * rf, ptr(self.row_dev), ptr(self.idx), ptr(self.host_spec),
                                          errp, commit)
        self._pf_slow = _lib.StepPrefetch(*common, 1, None, None, None, None, ptr(self.host_spec), errp, commit)

    def _chain_on_working_state(self) -> None:
        L = self.L
        L.bn_live, L.step_live = self.bn_work, self.step_work
        try:
            L.learn_rows(self.batch[0], self.loss_work, self.moments[0], pending=False, defer=True)
        finally:
            L.bn_live, L.step_live = L.bn_stats, L.step_dev

    def _body_fast(self) -> None:
        """the prefetch held: append + apply the waiting gradient + act() + prefetch in ONE launch, then the chain for the next
        update (six launches; the host waits for the first)"""
        self._tail_actor.act_with_optimizer_step(obs_ptr=self.head_dev.data_ptr() + 4 * self.L.lay.off_s2, prefetch=self._pf_fast,
                                                 obs_system_scope=True, adam_args=self._adam_work, net=self._net_work)
        self._chain_on_working_state()

    def _body_slow(self) -> None:
        """it did not (or nothing was prefetched): reset the working state, draw, chain — and from there as the other graph"""
        r = self.replay
        src = self.head_dev.data_ptr()
        check(self.L.lib.naf_step_prep(r.handle, src, src + 4 * self.L.lay.row_floats, ptr(self.row_dev), r.seed,
                                       ptr(r._sample_ctr), ptr(self.idx), ptr(self.batch), self.batch.shape[-1], r.action_mode,
                                       ptr(self.moments), self.L.B, int(r.without_replacement), ptr(self.spec_rec),
                                       ptr(self.idx_spec), _lib.C.byref(self._reset), stream_ptr()), "naf_step_prep")
        self._chain_on_working_state()
        self._tail_actor.act_with_optimizer_step(obs_ptr=self._obs_ptr, prefetch=self._pf_slow, adam_args=self._adam_work,
                                                 net=self._net_work)
        self._chain_on_working_state()

    def _body(self) -> None:
        if self.pipelined:
            self._body_slow()
            return
        if self.fused_prep:
            r = self.replay
            src, cnt = ptr(self.head_row), ptr(self.head_count)
            if self.head_dev is not None:
                src, cnt = self.head_dev.data_ptr(), self.head_dev.data_ptr() + 4 * self.L.lay.row_floats
            check(self.L.lib.naf_step_prep(r.handle, src, cnt, ptr(self.row_dev), r.seed,
                                           ptr(r._sample_ctr), ptr(self.idx), ptr(self.batch), self.batch.shape[-1], r.action_mode,
                                           ptr(self.moments), self.L.B, int(r.without_replacement), ptr(self.spec_rec),
                                           ptr(self.idx_spec), None, stream_ptr()), "naf_step_prep")
            self._updates(moments_ready=True)
            return
        if self.head_row is not None:
            check(self.L.lib.naf_replay_add_counted(self.replay.handle, ptr(self.head_row), ptr(self.head_count), 1, stream_ptr()),
                  "naf_replay_add_counted")
        self._sample_gather()
        self._updates()

    def capture(self) -> None:
        self.replay.flush()
        if self.pipelined:
            self._init_pipeline()
        # (the prefetch's record among them: the warm-up's last run leaves a valid one, for a ring that is put back)
        snap = _StateSnapshot(self.L, self.replay, (self.idx, self.batch, self.loss_parts) + self._tail_state +
                              ((self.spec_rec,) if self.spec_rec is not None else ()) +
                              ((self.bn_work, self.step_work, self.loss_work) if self.pipelined else ()))
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
            if self.head_dev is not None:
                self._publish(self._head_dst, self._head_src, self._head_bytes)

            def put_back():
                self.replay.rows[pos] = saved
                self.head_count[0] = count
            self.graph = _capture(self._body, snap, warmup=warmup, after_warmup=put_back)
            if self.pipelined:
                # (its kernels are warm — the other graph's are the same but for a template argument of the first — and a run of it
                #  needs a prefetch that holds: captured without one)
                self.graph_fast = _capture(self._body_fast, snap, warmup=0)
                self._spec_armed = False
        else:
            self.graph = _capture(self._body, snap)
        self._raw_exec()

    def _raw_exec(self) -> None:
        """the instantiated graph's handle, for run_row()'s direct launch (None: torch's replay())"""
        self._exec = None
        self._launch = self.L.lib.naf_host_publish_launch
        if self.graph is not None and self.head_dev is not None and hasattr(self.graph, "raw_cuda_graph_exec"):
            try:
                self._exec = int(self.graph.raw_cuda_graph_exec())
                if self.graph_fast is not None:
                    self._exec_fast = int(self.graph_fast.raw_cuda_graph_exec())
            except Exception:                      # (a torch build that keeps the handle to itself)
                self._exec = self._exec_fast = None

    def wait_pinned_free(self) -> None:
        """Block until the last run() has passed: what it reads from pinned host memory (the head row, its count, a tail's
        observation) may be rewritten afterwards. In NAFAgent.run's loop act() has waited for it already."""
        if self._seq_np is not None:
            if self._inflight:
                self.wait_tail()
        elif self._ran is not None:
            self._ran.synchronize()

    def wait_tail(self) -> None:
        """Block until the last run()'s tail has written its action to pinned host memory. Fused tail: a spin on the pinned
        ordinal the launch writes behind the action (no stream synchronisation: the hipStreamSynchronize round trip was a tenth
        of a timestep); other tails: the stream."""
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
        if self._err_np[2]:
            raise _lib.NafHipError("the pipelined timestep found its prefetched minibatch not to hold although the host had read that "
                                   "it does (engine.TrainChunk._holds): the update is not valid")
        if self._err_np[1]:
            raise _lib.NafHipError(f"naf_adam_polyak_act: {int(self.L.err_host[1])} polls inside the launch ran into their 2-ms bound "
                                   "(the GPU is over-subscribed or a workgroup died): the action is not valid")

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
        if self.use_graph and self.graph is None:
            self.capture()
        if self._seq_np is not None:
            if self._inflight:
                self.wait_tail()               # (its ordinal must be in before the next launch's "before" value is read)
            self._seq_prev = int(self._seq_np[0])
        if self.head_dev is not None:
            self._publish(self._head_dst, self._head_src, self._head_bytes)
        if self.use_graph:
            if self.gather_outside_graph:
                self._sample_gather()
            if self.pipelined and self._holds(bool(head_rows)):
                self.graph_fast.replay()
            else:
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
        fast = self.pipelined and self._holds(True)
        if self._exec is not None:
            # the row into device memory and the graph's launch in ONE foreign call (torch's replay() is that launch plus
            # device guards and generator bookkeeping this graph does not need)
            rc = self._launch(self._head_dst, self._head_src, self._head_bytes, self._exec_fast if fast else self._exec,
                              torch.cuda.current_stream().cuda_stream)
            if rc:
                check(rc, "naf_host_publish_launch")
        else:
            if self.head_dev is not None:
                self._publish(self._head_dst, self._head_src, self._head_bytes)
            (self.graph_fast if fast else self.graph).replay()
        self._inflight = True

    def _holds(self, brings_row: bool) -> bool:
        """Pipelined chunk, about to launch: may this timestep run the graph that starts with the append and the waiting gradient?
        Yes if it brings a row, the last launch was one of this chunk's graphs whose prefetch says it holds (pinned words its extra
        workgroup wrote a few microseconds behind the action; _seq_prev is that launch's ordinal: the caller has seen its action),
        and nobody has touched the ring, the sampler's stream or the learner in between. Also does the bookkeeping for the launch
        that follows."""
        r, L = self.replay, self.L
        # (the call counters, and torch's version counters of the buffers: an in-place write through ANY view of them — the NAF
        #  modules' load_state_dict, a snapshot's restore — moves those; graph replays and this path's own launches do not.
        #  What stays invisible is a write through `.data` or a raw pointer: INTEGRATION.md says so.)
        sig = (r._gen, L._gen, L.theta2._version, L.bn_stats._version, L.adam_m._version, L.adam_v._version, L.step_dev._version,
               L.grad._version, L.partials._version, r.rows._version, r.meta._version, r._sample_ctr._version)
        ok = False
        if brings_row and self._spec_armed and sig == self._sig and r._total_added == self._r_total + 1 and r._pending == 0:
            hs, want, n = self._host_spec_np, int(self._seq_prev) & 0xFFFFFFFF, 0
            v = int(hs[0])
            while (v & 0xFFFFFFFF) != want:
                n += 1
                if n > 2000000:                 # (the verdict is a few microseconds behind the action: something is wrong)
                    torch.cuda.current_stream().synchronize()
                    v = int(hs[0])
                    break
                v = int(hs[0])
            ok = (v & 0xFFFFFFFF) == want and (v >> 32) == 1
        self._spec_armed = True
        self._sig, self._r_total = sig, r._total_added
        if ok:
            self.fast_runs += 1
        else:
            self.slow_runs += 1
        return ok

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
