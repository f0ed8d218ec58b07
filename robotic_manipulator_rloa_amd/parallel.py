"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).

The reference has no distributed code at all; the scheme here is the natural sharding of its hot path
(SURVEY.md §8e): every rank owns its envs, its HBM replay shard and its minibatch; parameters and optimizer
state are replicated; the ONLY exchange is one sum all-reduce of the flat f32 gradient per learn()
(4*P bytes = 333 KB at A=6), after which every rank applies the identical clip+Adam+Polyak with the 1/W folded
into the clip scale, so replicas stay in lock-step. BatchNorm statistics stay per rank (the reference's
per-learner semantics at per-GPU batch size).
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


def share_one_gpu() -> bool:
    """NAF_DP_SHARE_GPU=1: every rank of the launch uses cuda:0 and gloo is the control plane (RCCL refuses two ranks on
    one device). A REHEARSAL of the data-parallel code on a 1-GPU box — hipIpc mappings, the one-shot all-reduce, the
    training loops' lock-step logic are the multi-GPU code; only the wire is local HBM. Never a performance mode."""
    return os.environ.get("NAF_DP_SHARE_GPU") == "1"


def local_device() -> torch.device:
    """The device of this rank: cuda:LOCAL_RANK (one process per GPU), cuda:0 for every rank in the rehearsal mode."""
    return torch.device("cuda", 0 if share_one_gpu() else int(os.environ.get("LOCAL_RANK", "0")))


def init_distributed(backend: Optional[str] = None) -> tuple:
    """(rank, local_rank, world). Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run sets them."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() and not share_one_gpu() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        elif torch.cuda.is_available():
            torch.cuda.set_device(local_device())
        dist.init_process_group(backend, **kw)
    return rank, local_rank, world


def rank_seed(seed: int, rank: int, stream: int = 0) -> int:
    """Independent per-rank seeds for env / noise / sampler streams; rank 0, stream 0 keeps the user's seed so a
    1-GPU run is the reference-seeded run."""
    return (int(seed) + 7919 * int(rank) + 104729 * int(stream)) & 0xFFFFFFFFFFFFFFFF


def all_reduce_flat_grad(flat_grad: torch.Tensor, group=None) -> torch.Tensor:
    """In-place SUM all-reduce of the flat gradient. The mean is NOT taken here: naf_adam_polyak_fused multiplies by
    inv_world inside its clip scale, so the averaged gradient never makes a separate pass over memory."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return flat_grad


def broadcast_parameters(theta2: torch.Tensor, src: int = 0, group=None) -> None:
    """Same seed already gives identical initial weights on every rank; broadcast anyway so a loaded checkpoint or a
    stray RNG draw cannot desynchronise the replicas."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(theta2, src=src, group=group)


def units_per_rank(total_envs: int, world: int, rank: int) -> int:
    """Weak scaling keeps envs/GPU fixed; for a fixed global env count the remainder goes to the low ranks."""
    base, rem = divmod(int(total_envs), int(world))
    return base + (1 if rank < rem else 0)


def _agree(ok: bool, device: torch.device, group=None) -> bool:
    """True only when EVERY rank passes ok=True (MIN all-reduce on the backend's own device type)."""
    on_gpu = dist.get_backend(group) == "nccl"
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if on_gpu else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(t.item() == 1)


class TickAgreement:
    """A per-tick AND over the ranks of one node, host to host, through a page of shared memory — no collective, no GPU.

    The pipelined per-timestep path (engine._Pipeline) keeps a gradient waiting between ticks and has two graphs: one learn() chain
    — one gradient exchange — when the tick's prefetched minibatches hold, two when the tick starts over. Under data parallel every
    rank must run the SAME graph (an exchange pairs with an exchange), and whether a rank could take the short one is known on its
    host only when the tick is there (did the prefetch hold; does the tick bring a row at all, or is it one of run()'s idle ticks):
    so every rank writes its vote {tick ordinal, ok} into its own cache line of the page and spins until the other W - 1 lines carry
    the ordinal; all ok -> the short graph everywhere. The ranks are in lock-step anyway (each tick's action waits for the
    all-reduced gradient): the vote costs a cache line's trip between processes.

    try_create is collective (a name for the page and the hosts' names go through torch.distributed once): None unless every rank
    runs on the same host and mapped the page."""

    LINE = 8                                   # uint64 words per rank: one 64-byte line

    def __init__(self, shm, rank: int, world: int, timeout_s: float):
        import numpy as np
        self._shm, self.rank, self.world, self.timeout_s = shm, int(rank), int(world), float(timeout_s)
        self.words = np.ndarray((world * self.LINE,), dtype=np.uint64, buffer=shm.buf)
        self.tick = 0
        self.waited = 0                        # polls spent waiting for peers (a measure of skew, for the curious)

    @classmethod
    def try_create(cls, group=None, timeout_s: float = 120.0) -> Optional["TickAgreement"]:
        import socket
        from multiprocessing import shared_memory
        if not dist.is_initialized() or dist.get_world_size(group) < 2:
            return None
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        src = dist.get_global_rank(group, 0) if group is not None else 0
        shm, name = None, [None]
        if rank == 0:
            try:
                shm = shared_memory.SharedMemory(create=True, size=max(4096, world * cls.LINE * 8))
                shm.buf[:world * cls.LINE * 8] = bytes(world * cls.LINE * 8)
                name = [shm.name]
            except Exception:                  # noqa: BLE001
                shm = None
        dist.broadcast_object_list(name, src=src, group=group)
        ok = name[0] is not None
        if ok and rank != 0:
            try:
                shm = shared_memory.SharedMemory(name=name[0])
                try:                           # (attaching registers the segment with this process's resource tracker, which would
                    from multiprocessing import resource_tracker      # try to unlink it again at exit: rank 0 owns the name)
                    resource_tracker.unregister(shm._name, "shared_memory")
                except Exception:              # noqa: BLE001
                    pass
            except Exception:                  # noqa: BLE001
                ok = False
        facts = [None] * world
        dist.all_gather_object(facts, (socket.gethostname(), ok), group=group)     # (also: every rank has mapped before rank 0 unlinks)
        ok = all(f[1] for f in facts) and len({f[0] for f in facts}) == 1
        if rank == 0 and shm is not None:
            try:
                shm.unlink()                   # the mappings stay; the name goes (nothing to clean up after a crash)
            except Exception:                  # noqa: BLE001
                pass
        if not ok:
            if shm is not None:
                shm.close()
            return None
        return cls(shm, rank, world, timeout_s)

    def all_ok(self, ok: bool) -> bool:
        """This tick's vote; True iff every rank voted ok. Every rank calls it once per tick, in the same order of ticks.
        Two slots per rank, by the tick's parity: a peer that has read every vote of tick t may be writing its vote for t + 1 while a
        slower rank still reads the votes of t — into the other slot; it cannot reach t + 2 (the slot of t again) before it has
        read the slow rank's vote for t + 1, which that rank writes only after it is done with t."""
        import time
        self.tick += 1
        t, w, L = self.tick, self.words, self.LINE
        s = t & 1
        w[self.rank * L + s] = (t << 1) | (1 if ok else 0)         # one aligned 8-byte store: whole or not at all
        result, t0, n = bool(ok), None, 0
        for r in range(self.world):
            if r == self.rank:
                continue
            v = int(w[r * L + s])
            while (v >> 1) != t:
                n += 1
                if n & 0x3FF == 0:
                    if t0 is None:
                        t0 = time.perf_counter()
                    elif time.perf_counter() - t0 > self.timeout_s:
                        from ._lib import NafHipError
                        raise NafHipError(f"data parallel: rank {r} did not reach tick {t} of the per-timestep loop within "
                                          f"{self.timeout_s:.0f} s (rank {self.rank} is waiting for its vote, last seen: tick "
                                          f"{v >> 1}): the ranks are out of step")
                v = int(w[r * L + s])
            result = result and bool(v & 1)
        self.waited += n
        return result

    def close(self) -> None:
        try:
            self.words = None
            self._shm.close()
        except Exception:                      # noqa: BLE001
            pass


class XgmiAllReduce:
    """One-shot sum all-reduce of the flat gradient over peer-mapped device memory (csrc/xgmi_reduce.hip):
    every rank pushes its gradient into a slot on each peer over xGMI and sums the W contributions in rank order.
    One capturable launch, no host involvement per call; it also emits the grad-norm partials.

    `XgmiAllReduce.try_create` is collective: either every rank of the group gets a verified communicator, or every
    rank gets None (and the caller stays on the RCCL all-reduce). Verification = `self_test`: exact integer-valued
    patterns through the very same kernels, any mismatch or timed-out wait on any rank disables the path everywhere.
    """

    def __init__(self, handle, lib, n_floats: int, rank: int, world: int, device: torch.device, group):
        self.handle, self.lib, self.n, self.rank, self.world = handle, lib, int(n_floats), rank, world
        self.device, self.group = device, group
        self.chunk = lib.naf_xgmi_chunk_floats()
        self.n_partials = (self.n + self.chunk - 1) // self.chunk
        self.mem_kind = {2: "uncached", 1: "fine-grained"}.get(lib.naf_xgmi_mem_kind(handle), "?")

    @classmethod
    def try_create(cls, n_floats: int, device: torch.device, group=None, timeout_s: float = 30.0,
                   test_rounds: int = 128, test_timeout_s: float = 10.0) -> Optional["XgmiAllReduce"]:
        import ctypes as C
        from . import _lib
        if not dist.is_initialized() or dist.get_world_size(group) < 2:
            return None
        lib = _lib.load()
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        handle, blob, why = C.c_void_p(), None, ""
        if world <= 8 and n_floats % 4 == 0:
            with torch.cuda.device(device):
                rc = lib.naf_xgmi_create(rank, world, n_floats, float(test_timeout_s), C.byref(handle))
                if rc == 0:
                    buf = C.create_string_buffer(64)
                    rc = lib.naf_xgmi_export(handle, buf)
                    blob = (bytes(buf.raw), torch.device(device).index or 0) if rc == 0 else None
                why = "" if rc == 0 else f"create/export rc={rc}"
        blobs = [None] * world
        dist.all_gather_object(blobs, blob, group=group)       # also orders every rank's slab memset before any push
        ok = all(b is not None for b in blobs)
        if ok:
            with torch.cuda.device(device):
                devs = (C.c_int * world)(*[b[1] for b in blobs])
                rc = lib.naf_xgmi_connect(handle, b"".join(b[0] for b in blobs), devs)
            ok, why = rc == 0, (why or (f"connect rc={rc}" if rc else ""))
        # load the library's code object and warm the launch path BEFORE the barrier below, so that the ranks enter the
        # self-test together (its waits are bounded by test_timeout_s)
        with torch.cuda.device(device):
            warm = torch.zeros(1, dtype=torch.int64, device=device)
            lib.naf_counter_add(warm.data_ptr(), 0, torch.cuda.current_stream(device).cuda_stream)
            torch.cuda.synchronize(device)
        ok = _agree(ok, device, group)
        comm = cls(handle, lib, n_floats, rank, world, device, group) if handle.value else None
        if ok:
            ok = _agree(comm.self_test(test_rounds), device, group)
            why = why or ("" if ok else "self-test mismatch or time-out")
            lib.naf_xgmi_set_timeout(handle, float(timeout_s))
        if not ok:
            if rank == 0:
                import sys
                print(f"[xgmi] one-shot all-reduce disabled ({why or 'a peer failed'}); using RCCL", file=sys.stderr)
            # the teardown's barrier runs on EVERY rank, also on one whose create failed (null handle, no communicator):
            # a rank that skipped it would pair the others' barrier with its next collective — the RCCL gradient all-reduce
            # this very fallback leads to
            if comm is not None:
                lib.naf_xgmi_disconnect(handle)
            try:
                dist.barrier(group=group)
            except Exception:
                pass
            if comm is not None:
                lib.naf_xgmi_destroy(handle)
                comm.handle = None
            return None
        return comm

    @classmethod
    def local_group(cls, world: int, n_floats: int, device: torch.device, timeout_s: float = 10.0) -> list:
        """`world` communicators in THIS process, rank r's slab mapped into the others by plain pointers (naf_xgmi_connect_local):
        the kernels, the slot / flag protocol and the rank-ordered sum of the multi-GPU path at world sizes a one-GPU box cannot host
        as processes (six processes per card on this pool; north_star's world is eight). Each rank must launch on a stream of its
        own — a launch waits for its peers' launches. A rehearsal, never a measurement."""
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        handles = []
        with torch.cuda.device(device):
            for r in range(world):
                h = C.c_void_p()
                _lib.check(lib.naf_xgmi_create(r, world, n_floats, float(timeout_s), C.byref(h)), "naf_xgmi_create")
                handles.append(h)
            arr = (C.c_void_p * world)(*[h.value for h in handles])
            for h in handles:
                _lib.check(lib.naf_xgmi_connect_local(h, arr), "naf_xgmi_connect_local")
        return [cls(h, lib, n_floats, r, world, device, None) for r, h in enumerate(handles)]

    def push_desc(self):
        """naf_xgmi_push_t for kernels that push part of the gradient early (naf_bn_relu_bwd_wgrad_push)."""
        from ._lib import XgmiPushDesc, check
        import ctypes as C
        d = XgmiPushDesc()
        check(self.lib.naf_xgmi_push_desc(self.handle, C.byref(d)), "xgmi_push_desc")
        return d

    def push_early(self, grad_in: torch.Tensor, lo: int, hi: int) -> None:
        """grad_in[lo:hi] to the peers ahead of the all-reduce (which must then be called with pushed_lo=lo)."""
        from ._lib import check, ptr, stream_ptr
        check(self.lib.naf_xgmi_push_early(self.handle, ptr(grad_in), int(lo), int(hi), stream_ptr()), "xgmi_push_early")

    def all_reduce(self, grad_in: torch.Tensor, grad_out: torch.Tensor, partials: Optional[torch.Tensor] = None,
                   step_dev: Optional[torch.Tensor] = None, pushed_lo: Optional[int] = None,
                   pushed_also: Optional[tuple] = None) -> None:
        """grad_out = sum over ranks of grad_in on the current stream (in place allowed). pushed_lo: grad_in[pushed_lo:]
        has already gone to the peers (push_early, the layer-1 backward kernel's extra workgroups, the finish launch of the
        row-split chain); pushed_also = (lo, hi): so has that second range."""
        from ._lib import check, ptr, stream_ptr
        if grad_in.numel() != self.n or grad_out.numel() != self.n or grad_in.dtype != torch.float32 or \
                grad_out.dtype != torch.float32 or not grad_in.is_contiguous() or not grad_out.is_contiguous():
            raise ValueError("xgmi all_reduce: gradients must be contiguous f32 of the communicator's length")
        if partials is not None and partials.numel() < self.n_partials:
            raise ValueError("xgmi all_reduce: partials too short")
        lo = self.n if pushed_lo is None else int(pushed_lo)
        s_lo, s_hi = (0, 0) if pushed_also is None else (int(pushed_also[0]), int(pushed_also[1]))
        check(self.lib.naf_xgmi_allreduce_sum_from2(self.handle, ptr(grad_in), ptr(grad_out), ptr(partials), ptr(step_dev),
                                                    lo, s_lo, s_hi, stream_ptr()), "xgmi_allreduce")

    def status(self) -> tuple:
        """(all-reduces done, timed-out waits) — blocking."""
        import ctypes as C
        e, t = C.c_uint64(), C.c_uint64()
        self.lib.naf_xgmi_status(self.handle, C.byref(e), C.byref(t))
        return int(e.value), int(t.value)

    def timeouts_nowait(self) -> int:
        """Timed-out waits so far, read from pinned host memory the kernel writes: no synchronisation. The value lags the
        stream by whatever is still queued; a non-zero count never goes back to zero."""
        import ctypes as C
        t = C.c_uint64()
        self.lib.naf_xgmi_timeouts_nowait(self.handle, C.byref(t))
        return int(t.value)

    def raise_on_timeout(self) -> None:
        """A peer's gradient did not arrive within the time-out: the affected updates were skipped on this rank (poisoned
        norm partial -> naf_adam_polyak_fused leaves the buffers alone), the replicas can no longer be assumed identical."""
        n = self.timeouts_nowait()
        if n:
            from ._lib import NafHipError
            raise NafHipError(f"one-shot gradient all-reduce: {n} wait(s) on a peer timed out on rank {self.rank} — a rank is "
                              "slow, dead, or made a different number of learn() calls; the data-parallel replicas are out "
                              "of lock-step, stop and restart from a checkpoint")

    def self_test(self, rounds: int = 128) -> bool:
        """Exact check of this rank's results: rank r contributes (r+1) * ((i + 3*round) % 61), whose sum over ranks is
        an integer below 2^24 (exact in f32 whatever the order)."""
        i = torch.arange(self.n, device=self.device, dtype=torch.int64)
        out = torch.empty(self.n, device=self.device, dtype=torch.float32)
        part = torch.zeros(self.n_partials, device=self.device, dtype=torch.float32)
        bad = torch.zeros((), device=self.device, dtype=torch.int64)
        tri = self.world * (self.world + 1) // 2
        for k in range(rounds):
            if k in (1, 8):     # a broken mapping shows in the first round: do not sit through the others' time-outs
                torch.cuda.synchronize(self.device)
                if self.status()[1] or int(bad.item()):
                    return False
            base = ((i + 3 * k) % 61).to(torch.float32)
            g = base * float(self.rank + 1)
            lo = None
            if k % 3 == 2:                           # every third round: most of the vector goes ahead, as learn() does it
                lo = 4 * ((self.n // 13) // 4)
                self.push_early(g, lo, self.n)
            if k % 2:
                self.all_reduce(g, g, part, pushed_lo=lo)          # in place
                got = g
            else:
                self.all_reduce(g, out, part, pushed_lo=lo)
                got = out
            want = base * float(tri)
            bad += (got != want).sum()
            ss = (want.double() ** 2).sum()
            bad += ((part.double().sum() - ss).abs() > 1e-4 * ss).long()
        torch.cuda.synchronize(self.device)
        done, timeouts = self.status()
        return int(bad.item()) == 0 and timeouts == 0 and done >= rounds

    def close(self, collective: bool = True) -> None:
        """Collective by default (every rank of the group calls it): unmap the peers' slabs, barrier, release the own slab —
        no slab goes back to the allocator while a peer still maps it. collective=False (error paths where the peers may
        not follow): no barrier."""
        if self.handle is not None and self.handle.value:
            self.lib.naf_xgmi_disconnect(self.handle)
            if collective and dist.is_initialized():
                try:
                    dist.barrier(group=self.group)
                except Exception:
                    pass
            self.lib.naf_xgmi_destroy(self.handle)
        self.handle = None
