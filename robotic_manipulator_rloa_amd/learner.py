"""Flat-buffer NAF learner: the MI355X implementation of NAFAgent.learn() + soft_update
(reference naf_components/naf_algorithm.py:180-226) and of NAF.forward's training pass
(naf_components/naf_neural_network.py:76-115).

Design:
  * all learner state lives in a few flat f32 HBM allocations: theta2[2, P] (row 0 = main net, row 1 = target
    net), grad[P], adam m[P], v[P]; parameters of the nn.Module facades are views into them.
  * the big trunk GEMMs stay on PyTorch-ROCm (torch.bmm / torch.mm with out=; rocBLAS kernels picked by the shipped
    TunableOp results); main and target forward share every launch (batch-2 bmm, 2-net BN kernel).
  * everything else is libnaf_hip.so: bias+BatchNorm+ReLU fwd/bwd (with the K=21 and N=32 GEMMs folded in), the
    fused NAF head (fwd + TD target + MSE + bwd), the backward GEMM bundle on f32 MFMA, grad-norm partials,
    clip+Adam+Polyak in one pass.
  * the three head Linears are one GEMM against Wh[NHP, H+8]: column H of Wh is the bias and the activation
    buffer carries a constant-1 column there, so bias add and bias gradient ride inside the GEMMs.
  * no host sync anywhere: step count, clip factor, replay size and sampler counters live on the device, so a
    chunk of U updates (sample -> gather -> U x learn) is captured once as a HIP graph and replayed.
"""
from __future__ import annotations

import os

import numpy as np
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr
from .parallel import XgmiAllReduce, all_reduce_flat_grad

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
ADAM_BETA1, ADAM_BETA2, ADAM_EPS = 0.9, 0.999, 1e-8
MAX_GRAD_NORM = 1.0


_BLAS_CONFIGURED = False
TUNING_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning", "tunableop_gfx950.csv")


def configure_blas() -> str:
    """Pick the GEMM kernels PyTorch-ROCm will use for the (tiny) trunk GEMMs.
    Measured on MI355X (benchmarks/gemm_probe.py, B=256): hipBLASLt's default heuristic runs the 256x256x256 f32
    GEMMs as ONE 256x256 macro-tile on one CU (63 us each, 239 us for the 8 GEMMs of an update); rocBLAS takes
    42 us for the 8; with the shipped TunableOp results (best rocBLAS/hipBLASLt solution per shape) 26 us.
    Env: NAF_BLAS_DEFAULT=1 leaves torch's BLAS settings untouched; NAF_BLAS_TUNING_FILE overrides the results file
    ('none' disables it)."""
    global _BLAS_CONFIGURED
    if _BLAS_CONFIGURED or os.environ.get("NAF_BLAS_DEFAULT") == "1":
        _BLAS_CONFIGURED = True
        return "default"
    _BLAS_CONFIGURED = True
    torch.backends.cuda.preferred_blas_library("cublas")           # = rocBLAS on ROCm
    path = os.environ.get("NAF_BLAS_TUNING_FILE", TUNING_FILE)
    if path != "none" and os.path.exists(path):
        try:
            torch.cuda.tunable.enable(True)
            torch.cuda.tunable.tuning_enable(False)                # look up only: never tune inside a training run
            torch.cuda.tunable.read_file(path)
            return "rocblas+tunableop"
        except Exception:                                          # results from another library build: ignore them
            torch.cuda.tunable.enable(False)
    return "rocblas"


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


@dataclass(frozen=True)
class Segment:
    offset: int
    shape: Tuple[int, ...]

    @property
    def numel(self) -> int:
        n = 1
        for s in self.shape:
            n *= s
        return n


BB_MAX_STATE = 32      # the row-split chain: layer 1's K (csrc/big_batch.hip, BB_MAX_K4)
BB_MAX_JOINTS = 11     # ... and the fused layer-2 launch's heads tile (FK_MAX_A)
NATIVE_LAYER = 256     # the width the fused kernels are built for (rl_framework.py's presets; csrc/big_batch.hip, step_path.hip)


class NetLayout:
    """Where each parameter of one NAF network sits inside its flat buffer (units: floats).
    Segments start on 64-float (256-B) boundaries; the gaps and the pad rows/columns of Wh hold zeros, receive
    zero gradients and therefore stay zero under Adam.

    layer_size below 256 (the reference takes any; its own agent test builds 128): with pad_layer the network is STORED as a
    256-wide one whose units beyond layer_size have zero weights, zero biases and zero BatchNorm scale / shift — `H` (what every
    kernel sees) is 256, `H_ref` (what state_dict() shows and load_state_dict() takes) is layer_size. Such a unit's pre-activation is
    0 for every sample, its normalised value 0, its activation ReLU(0 * 0 + 0) = 0; nothing downstream multiplies it with anything
    but a zero weight, so every gradient it receives or passes on is 0 and Adam (m = v = 0) leaves it where it is: the padded network
    computes the narrow one's numbers, term for term, with exact zeros appended to its sums — and runs the kernels the presets run."""

    def __init__(self, state_size: int, action_size: int, layer_size: int, pad_layer: bool = True):
        if not (1 <= action_size <= 64):
            raise ValueError("action_size must be in 1..64 (one sample per 8- or 16-lane group in the fused head kernels, per 16-, "
                             "32- or 64-lane group in the stand-alone ones: a wavefront has 64 lanes)")
        self.H_ref = int(layer_size)
        self.S, self.A = state_size, action_size
        # (round 6: widths in (256, 512) likewise stored as 512 — the row-split chain runs 512 columns as two 256-column halves)
        self.H = (NATIVE_LAYER if (pad_layer and 0 < layer_size < NATIVE_LAYER) else
                  2 * NATIVE_LAYER if (pad_layer and NATIVE_LAYER < layer_size < 2 * NATIVE_LAYER) else int(layer_size))
        self.T = action_size * (action_size + 1) // 2
        self.NH = self.A + self.T + 1                  # [mu | l | V]
        self.NHP = _round_up(self.NH, 16)              # heads row stride (ldh): whole 16-wide MFMA tiles
        self.HP = self.H + 16                          # activations: H features | 1.0 | 15 zeros (K % 16 == 0)
        self.seg: Dict[str, Segment] = {}
        off = 0
        for name, shape in (("W1", (self.H, self.S)), ("b1", (self.H,)), ("g1", (self.H,)), ("be1", (self.H,)),
                            ("W2", (self.H, self.H)), ("b2", (self.H,)), ("g2", (self.H,)), ("be2", (self.H,)),
                            ("Wh", (self.NHP, self.HP))):
            self.seg[name] = Segment(off, shape)
            off = _round_up(off + self.seg[name].numel, 64)
        self.P = off
        self.row_floats = _lib.load().naf_replay_row_floats(self.S, self.A)              # ring rows (256 B)
        self.batch_row_floats = _lib.load().naf_replay_batch_row_floats(self.S, self.A)  # gathered minibatch rows
        # offsets inside a transition row
        self.off_u = self.S
        self.off_r = self.S + self.A
        self.off_s2 = _lib.load().naf_replay_row_off_next_state(self.S, self.A)   # 16-B aligned
        self.off_d = self.off_s2 + self.S

    def view(self, flat: torch.Tensor, name: str) -> torch.Tensor:
        s = self.seg[name]
        return flat[s.offset:s.offset + s.numel].view(*s.shape)

    def n_ref_params(self) -> int:
        """Number of parameters of the reference module (79,644 at S=21, A=6, H=256)."""
        h = self.H_ref
        return h * self.S + h * h + 6 * h + self.NH * (h + 1)

    # ---- mapping to the reference's state_dict keys (naf_neural_network.py:37-54) -------------------------
    def param_views(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        """the reference's tensors, in its shapes: views of the stored ones (of their first layer_size units where the layer is
        stored padded; the heads' bias column is column H of Wh)"""
        A, T, H, h = self.A, self.T, self.H, self.H_ref
        Wh = self.view(flat, "Wh")
        return {
            "input_layer.weight": self.view(flat, "W1")[:h], "input_layer.bias": self.view(flat, "b1")[:h],
            "bn1.weight": self.view(flat, "g1")[:h], "bn1.bias": self.view(flat, "be1")[:h],
            "hidden_layer.weight": self.view(flat, "W2")[:h, :h], "hidden_layer.bias": self.view(flat, "b2")[:h],
            "bn2.weight": self.view(flat, "g2")[:h], "bn2.bias": self.view(flat, "be2")[:h],
            "action_values.weight": Wh[0:A, 0:h], "action_values.bias": Wh[0:A, H],
            "value.weight": Wh[A + T:A + T + 1, 0:h], "value.bias": Wh[A + T:A + T + 1, H],
            "matrix_entries.weight": Wh[A:A + T, 0:h], "matrix_entries.bias": Wh[A:A + T, H],
        }


PARAM_ORDER = ["input_layer.weight", "input_layer.bias", "bn1.weight", "bn1.bias", "hidden_layer.weight",
               "hidden_layer.bias", "bn2.weight", "bn2.bias", "action_values.weight", "action_values.bias",
               "value.weight", "value.bias", "matrix_entries.weight", "matrix_entries.bias"]


class Learner:
    """Owns the flat learner state of one agent (main + target) and enqueues learn() updates on the current
    stream. Everything is asynchronous; nothing here calls .item()/.cpu()."""

    def __init__(self, state_size: int, action_size: int, layer_size: int, batch_size: int, learning_rate: float,
                 tau: float, gamma: float, device: torch.device, p_mode: int = _lib.P_HADAMARD,
                 world_size: int = 1, process_group=None, fuse: Optional[str] = None, _force_allreduce: bool = False,
                 _fold_norm: bool = True, pad_layer: bool = True, _xgmi=None):
        """fuse: "rows" | "columns" | "unfused" | None (= NAF_FUSE or the per-shape default, see below). The underscore
        arguments are for tests: run the gradient all-reduce at world size 1 / keep the separate grad-norm launch.
        pad_layer: a layer_size below 256 is stored zero-padded to 256 (NetLayout) and runs the kernels built for that width;
        False: its own width on the column-tile / unfused chains (tests compare the two)."""
        _lib.require_gpu()
        self.lib = _lib.load()
        self.blas = configure_blas()
        self.dev = torch.device(device)
        self.lay = NetLayout(state_size, action_size, layer_size, pad_layer=pad_layer)
        self.B = int(batch_size)
        self.lr, self.tau, self.gamma = float(learning_rate), float(tau), float(gamma)
        self.p_mode = int(p_mode)
        self.world_size = int(world_size)
        self.pg = process_group
        # Three chains of launches implement one learn(); `fuse` (a set of tags, read by forward_train / learn_rows) says which:
        #   "rows"     {bb, gb, hk, ep, s2}: the ROW-SPLIT chain of csrc/big_batch.hip — 64-row blocks over the whole chip, two-stage
        #              BatchNorm statistics, GEMM 2 on f32 MFMA, layer 2 + heads + NAF head + first backward stage in one launch
        #              (hk), the backward GEMMs as one launch (gb) that also carries the second stage of layer 2's BatchNorm
        #              backward as its prologue (s2) and the batch pass of layer 1's backward as its epilogue (ep), a finish
        #              launch — 5 launches per update in a chain of updates. Needs 16 <= B <= 4096 (any size in it), H = 256 | 512, S <= 32,
        #              A <= 8 (A <= 11 up to B = 2048).
        #              Default wherever the shape fits (measured at the end of round 3, updates/s, column-tile | row-split: B = 64:
        #              32.3k | 36.0k, 128: 30.7k | 35.4k, 192: 25.8k | 34.0k, 256: 25.7k | 34.7k, 512: 20.3k | 30.9k; until then the
        #              column-tile chain led below B = 256 — 32.5k | 29.3k at 64 in round 2).
        #   "columns"  {l1, b2, gb, s3}: the COLUMN-TILE chain of csrc/fused_layers.hip — a workgroup owns 8 feature columns x all
        #              B rows (ceil(B/64) <= 8 rows per thread in registers), the K = state-size and N = heads GEMMs folded into
        #              the BatchNorm kernels, 8 launches per update. B <= 512. Default below B = 16 (and up to 512 where H or S
        #              do not fit the row-split chain). (B = 16 ... 63 moved to the row-split chain in round 4: one partial
        #              block — 37.1k against 31.0k updates/s at B = 32, 36.5k against 21.1k at 63.)
        #   "unfused"  {gb} or {}: torch (rocBLAS) GEMMs + the BatchNorm / head kernels of csrc/bn_relu.hip, naf_head.hip, with
        #              the backward GEMM bundle where its shapes allow (B, H multiples of 16) — any shape up to B = 4096; 14
        #              launches per update (round 1's chain: 12.7k updates/s at B = 1024).
        # NAF_FUSE = rows | columns | unfused overrides the choice (a chain whose shape limits are not met falls to the next).
        lay0 = self.lay
        # The reference takes any positive batch_size (rl_framework.py:186-189). Here: every size up to 4096 trains — 16 ... 4096
        # on the row-split chain, smaller ones on the column-tile chain, other layer / state sizes on the unfused chain (beyond 2048
        # with the streamed BatchNorm kernels of csrc/bn_relu.hip) with a warning that names the row-split chain's range. 4096 is
        # the replay sampler's limit (one workgroup draws a minibatch without replacement in LDS, csrc/replay.hip).
        if self.B < 1 or self.B > (1 << 20):
            raise ValueError(f"batch_size {self.B}: 1 <= batch_size <= 1,048,576 (the replay sampler's table, csrc/replay.hip)")
        # (round 4: ANY batch size from 16 to 4096 — the last 64-row block of layer 1 / GEMM 2, the last 16-row workgroup of the fused
        #  layer-2 + head launch and the last block of the bundle's dA1 product may be partial: rows past the batch read as zeros, are
        #  never stored and stay out of every statistic and sum. The work buffers hold Bp = the next multiple of 16 rows (of 32 beyond
        #  B = 2048, where the fused layer-2 + head launch runs 32 rows per workgroup),
        #  zero-initialised: the weight-gradient products walk Bp rows as their K dimension, and a row past the batch is a zero in at
        #  least one operand of each — dH and dY2 rows the head body never writes, A1 rows layer 1 never stores.)
        # (round 6, late: state sizes up to 32 — the layer-1 kernels' K — and 9 .. 11 joints: the fused layer-2 launch then holds one
        #  sample per 16-lane group and a Wh tile of 64 | 80 rows, 16 rows per workgroup only, hence B <= 2048; csrc/big_batch.hip.
        #  The reference's state is 9 + 2 A floats, environment.py:261: 27 | 29 | 31 at 9 | 10 | 11 joints)
        #  (beyond 2048 — 32 rows per workgroup — with the reference's Hadamard head only: the matmul mode's L tiles do not fit there)
        self.bb_ok = (16 <= self.B <= 4096 and lay0.H in (256, 512) and lay0.S <= BB_MAX_STATE and
                      (lay0.A <= 8 or (lay0.A <= BB_MAX_JOINTS and (self.B <= 2048 or self.p_mode == _lib.P_HADAMARD))))
        want = (fuse or os.environ.get("NAF_FUSE", "default")).lower()
        if want not in ("default", "rows", "columns", "unfused"):
            raise ValueError(f"NAF_FUSE / fuse = {want!r}: one of default, rows, columns, unfused")
        if want == "default":
            want = "rows" if self.bb_ok else ("columns" if self.B <= 512 else "unfused")
        if want == "rows" and not self.bb_ok:
            want = "columns" if self.B <= 512 else "unfused"
        if lay0.A > 8 and want != "rows":
            want = "unfused"                   # (9 .. 16 joints: the column-tile kernels hold one sample per 8-lane group)
        if want == "rows":
            self.fuse = {"bb", "gb", "hk", "ep", "s2"}
        elif want == "columns" and self.B <= 512:
            self.fuse = {"l1", "b2", "gb", "s3"}
            if lay0.S > 32 or (lay0.S > 24 and self.B > 256):
                self.fuse -= {"l1"}            # (K = 25 .. 32: the layer-1 backward tile holds at most 4 rows per thread)
            if lay0.H not in (128, 256):
                self.fuse -= {"s3"}
        else:
            want = "unfused"
            self.fuse = {"gb"}
        if (self.B % 16 != 0 and want != "rows") or lay0.H % 16 != 0:
            self.fuse -= {"gb"}                # the MFMA kernels take whole 16 x 16 x 16 steps: M, N, K % 16 == 0
        self.chain = want
        if lay0.A > 8 and want != "rows":
            import warnings
            warnings.warn(f"action_size {lay0.A} at batch_size {self.B}, state_size {lay0.S} runs the unfused chain: the row-split "
                          f"chain takes up to {BB_MAX_JOINTS} joints at 16 <= batch_size <= 4096 (2048 with p_mode matmul), state_size <= {BB_MAX_STATE} "
                          f"(every arm the reference ships has 6 or 7: KUKA / xArm, Panda)", stacklevel=3)
        elif self.B > 512 and want != "rows":
            # (a performance cliff, not an error: say so once, with the sizes that avoid it)
            import warnings
            warnings.warn(f"batch_size {self.B} at H = {lay0.H}, S = {lay0.S} runs the unfused chain (about half the updates/s of the "
                          f"row-split chain): the row-split kernels need 16 <= batch_size <= 4096, layer_size <= 512 and "
                          f"state_size <= {BB_MAX_STATE}", stacklevel=3)
        # with every gradient element produced by one of our own kernels, those kernels also emit its sum-of-squares partial:
        # the separate grad-norm launch disappears. Data-parallel runs keep it (the norm is taken on the all-reduced gradient).
        self.fold_norm = ({"l1", "b2", "gb"} <= self.fuse or {"bb", "gb"} <= self.fuse) and self.world_size == 1 and \
            not _force_allreduce and _fold_norm
        self._force_allreduce = bool(_force_allreduce)
        lay, B, dev = self.lay, self.B, self.dev
        f32 = dict(dtype=torch.float32, device=dev)
        P, H, HP, NHP = lay.P, lay.H, lay.HP, lay.NHP
        # rows of the work buffers: whole 16-row groups on the row-split chain (see bb_ok above), B everywhere else
        # (beyond B = 256 whole 64-row blocks: the K ranges of the weight-gradient products then divide as they do for the next
        #  multiple of 64 — B = 1000 runs the products of B = 1024 with 24 zero rows, four 256-row ranges instead of three of 336)
        hk_rows = (self.lib.naf_bb_layer2_head_rows(B) if B <= 256 else 64) if "bb" in self.fuse else 1
        self.Bp = Bp = -(-B // hk_rows) * hk_rows

        # ---- persistent state --------------------------------------------------------------------------
        self.theta2 = torch.zeros(2, P, **f32)             # [main; target]
        self.grad = torch.zeros(P, **f32)
        self.adam_m = torch.zeros(P, **f32)
        self.adam_v = torch.zeros(P, **f32)
        # BN running statistics [net][rm1, rv1, rm2, rv2][H]; not in theta: soft_update copies parameters() only
        self.bn_stats = torch.zeros(2, 4, H, **f32)
        self.bn_stats[:, 1].fill_(1.0)
        self.bn_stats[:, 3].fill_(1.0)
        # what the row-split chain reads and advances: the public buffers themselves — except while a pipelined per-timestep chunk
        # (engine.TrainChunk) captures its graphs, whose chains work on copies of their own that the chunk's last launch commits
        self.bn_live = self.bn_stats
        self._gen = 0              # bumped by every method that changes parameters, optimizer state, statistics or the gradient
                                   # when CALLED (graph replays do not count): a pipelined chunk's pending gradient is void then
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)   # optimizer steps taken
        self.step_live = self.step_dev
        ft_tx = self.lib.naf_fused_tile_cols()
        ft_blocks = (H + ft_tx - 1) // ft_tx                               # workgroups of the column-tile kernels
        gb_blocks = ((NHP + 31) // 32) * ((HP + 31) // 32) + ((H + 31) // 32) ** 2   # dWh + dW2 blocks of the bundle
        self.n_partials_norm = (P + _lib.NORM_CHUNK - 1) // _lib.NORM_CHUNK
        # folded norm partials: the bundle's dWh + dW2 blocks, then the column-tile kernels (two launches of ft_blocks) or,
        # in the large-batch chain, the workgroups of the layer-1 finish kernel (two columns each)
        self.n_partials_fold = gb_blocks + 2 * ft_blocks
        if "bb" in self.fuse:
            self.n_partials_fold = self.lib.naf_bb_layer1_bwd_finish_blocks(H) + (H * H + 1023) // 1024 + (NHP * HP + 1023) // 1024
        self.n_partials = self.n_partials_fold if self.fold_norm else self.n_partials_norm
        # Data parallel inside one node — three forms of the one exchange (the sum of the flat gradient over the ranks, where
        # loss.backward() -> clip_grad_norm_ sit in the reference, naf_algorithm.py:207-210):
        #   "oneshot"  the one-shot peer-memory all-reduce as a launch of its own behind the finish launch (csrc/xgmi_reduce.hip; the
        #              finish launch pushes 91 % of the bytes early)
        #   "merged"   the whole exchange INSIDE the finish launch of the row-split chain (csrc/big_batch.hip, bb_finish_exchange):
        #              five launches per update as on one GPU
        #   "rccl"     torch.distributed's all-reduce (RCCL over xGMI) captured in the graph + a norm launch
        # The peer-memory forms exist when every rank could map every peer and the exact self-test passed on all of them
        # (XgmiAllReduce.try_create is collective; NAF_XGMI=0 skips it). WHICH form runs is measured where the job runs:
        # autotune_exchange() (end of this constructor) times them on this node and every rank takes the collectively fastest —
        # the one-GPU rehearsal ranks them oneshot < merged < collective, real xGMI may not (a peer's slab is not behind the
        # writer's L2 there). NAF_DP_EXCHANGE = oneshot | merged | rccl pins the form instead.
        self.xgmi = None
        self._xg = None                    # the communicator the CURRENT form uses (None: the collective)
        self.xgmi_merged = False
        self.exchange = "none" if self.world_size == 1 else "rccl"
        self.exchange_autotune = None      # {form: us per update ..., "chosen": form} once autotune_exchange() has run
        self._exchange_pick = "auto"
        self._push_desc = None
        n_partials_most = max(self.n_partials_fold + 1, self.n_partials_norm, self.n_partials)
        if _xgmi is not None:
            # a communicator handed in (XgmiAllReduce.local_group: the in-process rehearsal of world sizes beyond what a one-GPU
            # box hosts as processes); the exchange form must then be pinned — nothing here is collective
            if _xgmi.world != self.world_size or _xgmi.n != P or os.environ.get("NAF_DP_EXCHANGE", "auto") not in ("oneshot", "merged"):
                raise ValueError("Learner(_xgmi=...): a communicator of this world size and length, NAF_DP_EXCHANGE = oneshot | merged")
            self.xgmi = _xgmi
            n_partials_most = max(n_partials_most, self.xgmi.n_partials)
        elif self.world_size > 1 and os.environ.get("NAF_XGMI", "1") != "0":
            self.xgmi = XgmiAllReduce.try_create(P, dev, self.pg)
            if self.xgmi is not None:
                n_partials_most = max(n_partials_most, self.xgmi.n_partials)
        self.partials = torch.zeros(n_partials_most + 1, **f32)
        # the optimizer step of update k carried by the first two launches of update k + 1 (csrc/adam_body.h): the row-split
        # chain on one rank, gradient norm folded into the producers. NAF_DEFER_ADAM=0 keeps the launch of its own.
        # Data parallel over peer memory: the one-shot all-reduce launch leaves the norm partials and the step count exactly
        # as the folded producers do on one rank, so the step can ride there too (the RCCL path keeps its two launches).
        # The collective path (RCCL, or the test-only forced all-reduce): the norm must be taken on the REDUCED gradient, so a norm
        # launch stays behind the collective — and the optimizer step rides on the next update behind it all the same (round 4:
        # one launch and one boundary less per update there too; before, that path kept both launches).
        self.defer_ok = ("bb" in self.fuse and os.environ.get("NAF_DEFER_ADAM", "1") != "0" and
                         ((self.fold_norm and self.world_size == 1) or self.world_size > 1 or self._force_allreduce))
        self.adam_bc = torch.zeros(8, **f32)     # the next step's bias corrections, left by the riding optimizer workgroups
        self._adam_args = _lib.AdamArgs(
            ptr(self.theta2[0]), ptr(self.grad), ptr(self.adam_m), ptr(self.adam_v), ptr(self.theta2[1]), ptr(self.partials),
            self.n_partials, MAX_GRAD_NORM, self.lr, ADAM_BETA1, ADAM_BETA2, ADAM_EPS, self.tau, float(1.0 - self.tau),
            ptr(self.step_dev), 1.0 / self.world_size, P, lay.seg["W2"].offset, ptr(self.adam_bc))
        self._gb_wh_blocks = ((NHP + 31) // 32) * ((HP + 31) // 32)
        self._gb_blocks, self._ft_blocks = gb_blocks, ft_blocks
        # loss partials per update: one per workgroup of the head launch (8 samples each; 4 beyond 32 joints, csrc/naf_head_wide.hip)
        self.n_loss_wg = (B + 7) // 8 if lay.A <= 32 else (B + 3) // 4
        # a poll inside a fused launch that gave up on the folded records and folded for itself (csrc/bn2bwd_fold.h) bumps this
        # pinned HOST word; fold_fallbacks reads it without synchronising
        self.err_host = torch.zeros(8, dtype=torch.int64).pin_memory()

        # ---- work buffers for one minibatch ------------------------------------------------------------
        self.G1 = torch.zeros(2, Bp, H, **f32)
        self.A1 = torch.zeros(2, Bp, H, **f32)
        self.G2 = torch.zeros(2, Bp, H, **f32)
        self.A2 = torch.zeros(2, Bp, HP, **f32)
        self.A2[:, :, H] = 1.0                               # the constant-1 column that carries the head biases
        self.Gh = torch.zeros(2, Bp, NHP, **f32)
        self.dH = torch.zeros(Bp, NHP, **f32)
        self.dA2 = torch.zeros(Bp, HP, **f32)
        self.dZ2 = torch.zeros(Bp, H, **f32)
        self.dA1 = torch.zeros(Bp, H, **f32)
        self.dZ1 = torch.zeros(Bp, H, **f32)
        self.save_mean = torch.empty(2, 2, H, **f32)         # [layer][net][H]
        self.save_invstd = torch.empty(2, 2, H, **f32)
        self.q_out = torch.empty(B, **f32)
        if "bb" in self.fuse:
            NB = -(-B // 64)                                     # 64-row statistics blocks (the last may hold 16, 32 or 48 rows)
            NB1 = -(-Bp // 32)                                   # 32-row blocks of the bundle's dA1 product (M = Bp rows)
            kp = self.lib.naf_bb_layer1_bwd_kp(lay.S)
            # moments record of one minibatch's layer-1 inputs, [net][Sx | C]: everything layer 1's BatchNorm needs from
            # the batch dimension (TrainChunk computes the records of all its minibatches in one launch behind the gather)
            self.mom_floats = self.lib.naf_bb_moments_floats(lay.S)
            self.bb_mom = torch.zeros(2, self.mom_floats, **f32)
            self.bb_wc = torch.zeros(H, kp, **f32)               # w_c C of the main net, forward -> finish
            self.bb_st2 = torch.zeros(2, NB, H, 2, **f32)        # layer-2 statistics partials per 64-row block: (sum, M2)
            self.hk_rows = self.lib.naf_bb_layer2_head_rows(B)    # rows per block of the fused launch's backward partials
            self.bb_bw2 = torch.zeros(Bp // self.hk_rows, H, 2, **f32)   # backward partials of layer 2: (sum dy, sum dy*xhat)
            self.bb_bw1 = torch.zeros(NB1, H, 2, **f32)           # backward partials of layer 1, per 32-row block of the bundle
            self.bb_dw1 = torch.zeros(NB1, H, kp, **f32)          # per-block shares of P = dY1^T X
            self._nb1 = NB1
        if "s3" in self.fuse:
            # split-K heads: one [B, NHP] slab per 8-column workgroup of layer 2's BN kernel (+ the target's V column)
            # slabs 256 B further apart than their size: the H/8 pieces of one row, read together by the head kernel,
            # then sit in different L2 channels instead of one (32-KB stride: 6.6 us per head launch, padded: see DESIGN)
            self.n_slabs = H // 8
            self.slab_stride = B * NHP + 64
            self.heads_partial = torch.zeros(self.n_slabs * self.slab_stride, **f32)
            self.vnext_partial = torch.zeros(self.n_slabs, B, **f32)

        # ---- GEMM operand views (built once: no per-call tensor construction on the hot path) -------------
        seg = lay.seg
        t2 = self.theta2
        self.W1T2 = t2[:, seg["W1"].offset:seg["W1"].offset + seg["W1"].numel].view(2, H, lay.S).transpose(1, 2)
        self.W2T2 = t2[:, seg["W2"].offset:seg["W2"].offset + seg["W2"].numel].view(2, H, H).transpose(1, 2)
        self.WhT2 = t2[:, seg["Wh"].offset:seg["Wh"].offset + seg["Wh"].numel].view(2, NHP, HP).transpose(1, 2)
        # (running the weight-gradient GEMMs on a forked branch of the captured graph was measured SLOWER — 8.0k vs
        # 12.7k updates/s: hipGraph cross-stream edges cost more than the serial launches they hide — so the update is
        # one linear chain of launches)
        self.W2_main = lay.view(t2[0], "W2")
        self.Wh_main = lay.view(t2[0], "Wh")
        self.gW1 = lay.view(self.grad, "W1")
        self.gW2 = lay.view(self.grad, "W2")
        self.gWh = lay.view(self.grad, "Wh")
        self._f = self.lib  # shorthand
        D = _lib.GemmDesc
        pp = self.partials.data_ptr()
        sq_wh = pp if self.fold_norm else None
        sq_w2 = pp + 4 * self._gb_wh_blocks if self.fold_norm else None
        self._bundle = (D * 3)(
            D(ptr(self.dH), ptr(self.A2[0]), ptr(self.gWh), sq_wh, NHP, HP, B, NHP, HP, HP, 1, 1),          # dWh
            D(ptr(self.dZ2), ptr(self.A1[0]), ptr(self.gW2), sq_w2, H, H, B, H, H, H, 1, 1),                  # dW2
            D(ptr(self.dZ2), ptr(self.W2_main), ptr(self.dA1), None, B, H, H, H, H, H, 0, 1))                 # dA1
        self._bb_segs, self._bb_nsegs = None, 0
        if "bb" in self.fuse:
            # The weight gradients reduce over K = B. Cut K into ranges, one grid of blocks each, writing partial slabs that
            # the layer-1 finish launch adds in slab order (and takes the norm partials of): 256-row ranges — with every launch
            # of the chain working by eighths of the batch (csrc/big_batch.hip, bb_place_rows) a 256-row range is what one XCD
            # (or two) already holds of dY2, Z2 and A1. For dW2 twice as long (512 rows) where the launch would otherwise not fit
            # the chip at once — more than the 1024 blocks the four-wave form of the bundle has room for: B = 2048 (1096 blocks
            # with eight ranges, 840 with four). Updates/s with 256- | 512-row ranges, A/B/A/B on one box at the end of round 3:
            # B = 1024 28.0k | 27.0k, 1536 22.7k | 22.0k, 2048 20.9k | 21.3k. (Round 2 had it the other way round at 1024 — 420
            # blocks in one round of the eight-wave form against 548 — before the rows went to the XCDs by eighths.) Batch sizes
            # that are not multiples of 256: as many equal ranges (<= 8, whole 64-row blocks) as divide B / 64.
            def k_ranges(target, most):
                """largest number of equal K ranges <= most, each whole 16-k steps and at least `target` rows long"""
                return max([d for d in range(1, most + 1) if Bp % d == 0 and (Bp // d) % 16 == 0 and Bp // d >= target] or [1])

            def blocks(M, N, k_split):
                """32 x 32 blocks of one product of the bundle (csrc/gemm_bundle.hip)"""
                return ((M + 31) // 32) * ((N + 31) // 32) * k_split
            ks = Bp // 256 if (Bp % 256 == 0 and Bp <= 2048) else k_ranges(256, 8)      # (at most 8 slabs: csrc/big_batch.hip)
            ks_w2 = ks_wh = ks
            # (round 4's LDS-DMA ring form of this launch measured slower at every size: benchmarks/experimental/gemm_ring.h)
            if blocks(Bp, H, 1) + blocks(H, H, ks) + blocks(NHP, HP, ks) + 8 > 1024 and ks % 2 == 0 and \
                    (Bp // (ks // 2)) % 256 == 0:
                ks_w2 = ks // 2
            self.bb_slab_w2 = torch.zeros(ks_w2, H * H, **f32)
            self.bb_slab_wh = torch.zeros(ks_wh, NHP * HP, **f32)
            t2p_, seg_, gp_ = self.theta2.data_ptr(), lay.seg, self.grad.data_ptr()
            # ep: the batch pass of layer 1's backward as the epilogue of the dA1 blocks (x / ldx: set per minibatch)
            # (xhat of layer 1 kept by the forward pass for the epilogue up to B = 512 — one round of blocks, where recomputing it
            # sat on the critical path: updates/s 36.0k -> 36.8k at B = 64, 35.2k -> 36.0k at 128, 34.6k -> 35.1k at 256, 31.25k ->
            # 31.5k at 512; beyond that the recomputation hides and the extra B x H floats each way do not: 28.3k -> 27.7k at 1024)
            self.XH1 = torch.zeros(Bp, H, **f32) if B <= 512 else None
            self._epi = _lib.GemmL1Bwd(None, t2p_ + 4 * seg_["W1"].offset, t2p_ + 4 * seg_["b1"].offset, ptr(self.A1[0]),
                                       ptr(self.save_mean[0, 0]), ptr(self.save_invstd[0, 0]), ptr(self.bb_bw1), ptr(self.bb_dw1),
                                       0, lay.S, self.lib.naf_bb_layer1_bwd_kp(lay.S), H, ptr(self.XH1),
                                       t2p_ + 4 * seg_["g1"].offset, t2p_ + 4 * seg_["be1"].offset, B)
            # s2: dY2 -> dZ2 while the two products that read it stage their A panels; the block sums folded once per launch
            # by the bundle's first workgroups and handed on as tagged records (csrc/gemm_bundle.hip, gemm_bn2bwd_fold_block)
            self.bb_cst = torch.zeros(H, 4, **f32)                                        # per-column constants of the launch
            self.bb_fold_flag = torch.ones(1, dtype=torch.int32, device=dev)              # launch number: finish advances it; never restored
            self._pro = _lib.GemmBn2Bwd(ptr(self.G2[0]), ptr(self.bb_bw2), t2p_ + 4 * seg_["g2"].offset, ptr(self.save_mean[1, 0]),
                                        ptr(self.save_invstd[1, 0]), gp_ + 4 * seg_["g2"].offset, gp_ + 4 * seg_["be2"].offset,
                                        Bp // self.hk_rows, B, H, ptr(self.bb_cst), ptr(self.bb_fold_flag), self.err_host.data_ptr())
            pro_ = _lib.C.addressof(self._pro)
            # layer 2's forward statistics folded once per launch too where a workgroup would pull more than 16 blocks of them
            # (B > 1024; the library decides): records of their own, the same launch counter and error word
            self.bb_stat_rec = torch.zeros(2 * H, 4, **f32)
            # (H = 512: the fused layer-2 launch runs two workgroups per row block, one per 256-column half; their partial heads meet here)
            self.bb_exchange = torch.zeros(max(4, self.lib.naf_bb_layer2_head_exchange_floats(B, NHP)), **f32) if H > 256 else None
            self._stats_once_s = _lib.BbStatsOnce(ptr(self.bb_stat_rec), ptr(self.bb_fold_flag), self.err_host.data_ptr(),
                                                  ptr(self.bb_exchange))
            self._stats_once = _lib.C.byref(self._stats_once_s)
            # dA1 FIRST: its blocks carry the layer-1 epilogue and run longest; dispatched first, the short weight-gradient
            # blocks fill in behind them instead of the other way round
            self._bundle = (D * 3)(
                D(ptr(self.dZ2), ptr(self.W2_main), None, None, Bp, H, H, H, H, H, 0, 1, 1, 0, _lib.C.addressof(self._epi), pro_),
                D(ptr(self.dZ2), ptr(self.A1[0]), ptr(self.bb_slab_w2), None, H, H, Bp, H, H, H, 1, 1, ks_w2, H * H, None, pro_),
                D(ptr(self.dH), ptr(self.A2[0]), ptr(self.bb_slab_wh), None, NHP, HP, Bp, NHP, HP, HP, 1, 1, ks_wh, NHP * HP))
            SS = _lib.SlabSeg
            self._bb_segs = (SS * 2)(SS(ptr(self.bb_slab_w2), ptr(self.gW2), H * H, H * H, ks_w2),
                                     SS(ptr(self.bb_slab_wh), ptr(self.gWh), NHP * HP, NHP * HP, ks_wh))
            self._bb_nsegs = 2
        # ---- which form the gradient exchange takes (data parallel) -------------------------------------------------------
        if self.world_size > 1:
            want_x = os.environ.get("NAF_DP_EXCHANGE", "auto").lower()
            if want_x not in ("auto", "fastest", "oneshot", "merged", "rccl"):
                raise ValueError(f"NAF_DP_EXCHANGE = {want_x!r}: one of auto, fastest, oneshot, merged, rccl")
            forms = self.exchange_forms()
            self._exchange_pick = want_x
            if want_x in ("auto", "fastest"):
                self._set_exchange(forms[0])
                if len(forms) > 1:
                    self.autotune_exchange()
            else:
                if want_x not in forms:
                    raise _lib.NafHipError(f"NAF_DP_EXCHANGE={want_x}: not available here (forms: {forms})")
                self._set_exchange(want_x)

    # ---- data parallel: the form of the gradient exchange -------------------------------------------------------------
    def exchange_forms(self):
        """The forms this learner can run, fastest-in-the-rehearsal first: see the constructor."""
        if self.world_size == 1:
            return ["none"]
        seg, P = self.lay.seg, self.lay.P
        forms = []
        if self.xgmi is not None:
            forms.append("oneshot")
            if "bb" in self.fuse and seg["Wh"].offset + seg["Wh"].numel == P:
                forms.append("merged")
        return forms + ["rccl"]

    def _set_exchange(self, form: str) -> None:
        """Switch the exchange of every learn_rows() enqueued (or captured) from now on. Collective in effect: every rank must make
        the same switch between the same two updates."""
        if form not in self.exchange_forms():
            raise ValueError(f"exchange form {form!r} not available (forms: {self.exchange_forms()})")
        self.exchange = form
        self._xg = self.xgmi if form in ("oneshot", "merged") else None
        self.xgmi_merged = form == "merged"
        if self.world_size > 1:
            # who leaves the sum-of-squares partials of the REDUCED gradient, and how many
            self.n_partials = {"oneshot": self.xgmi.n_partials if self.xgmi is not None else 0,
                               "merged": self.n_partials_fold + 1,       # (+ the layer-1 / BatchNorm ranges' entry of the last workgroup)
                               "rccl": self.n_partials_norm}[form]
            self._adam_args.n_partials = self.n_partials

    def autotune_exchange(self, updates: int = 200, chunk: int = 8, verbose: bool = False) -> dict:
        """Time `updates` graph-replayed learn() calls under every available form of the exchange ON THIS NODE, agree on the
        fastest (every rank's time, MAX over the ranks, then the smallest: all ranks see the same three numbers and make the same
        choice) and switch to it. Collective: every rank of the group calls it at the same point — the constructor does. The
        learner's state is put back afterwards bit for bit. Returns {form: us per update, ..., "chosen": form} (also kept as
        self.exchange_autotune; bench.py prints it in its `preflight` object)."""
        import time
        import torch.distributed as dist
        forms = self.exchange_forms()
        lay, B, dev = self.lay, self.B, self.dev
        saved = [(t, t.clone()) for t in (self.theta2, self.grad, self.adam_m, self.adam_v, self.bn_stats, self.step_dev,
                                          self.partials, self.adam_bc)]
        g = torch.Generator(device=dev).manual_seed(1234)          # (the same rows on every rank: timing, not learning)
        rows = torch.randn(chunk, B, lay.batch_row_floats, generator=g, device=dev)
        rows[:, :, lay.S:lay.S + lay.A].clamp_(-1, 1).trunc_()
        store = torch.zeros(chunk * B * lay.batch_row_floats + 64, device=dev)
        store[:rows.numel()].copy_(rows.view(-1))
        rows = store[:rows.numel()].view(chunk, B, lay.batch_row_floats)
        d = self.defer_ok
        nccl = dist.get_backend(self.pg) == "nccl"

        def body():
            for k in range(chunk):
                self.learn_rows(rows[k], pending=d and k > 0, defer=d and k < chunk - 1)

        def restore():
            for live, keep in saved:
                live.copy_(keep)

        # two passes over the forms, the faster reading of each: the first launches of a process read five times too long
        # (clocks, first-touch, code objects) — with one pass the first form timed paid for all of that
        from .parallel import _agree
        times, errors = {}, {}
        for form in forms + forms:
            if form in errors:
                continue
            self._set_exchange(form)
            graph, err = None, None
            capturable = form != "rccl" or nccl              # (gloo, the one-GPU rehearsal's control plane, cannot be captured)
            # A form that cannot be warmed or captured HERE (a collective that refuses stream capture, a mapping that faults) must
            # cost this node that form, not the job: the attempt is fenced, and the ranks agree on its outcome before anyone times
            # anything (a rank that went on alone would wait at the barrier below for ever).
            try:
                body()                                       # warm the launch paths (and the collective's communicator)
                torch.cuda.synchronize(dev)
                if capturable:
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        body()
                    torch.cuda.synchronize(dev)
            except Exception as e:                           # noqa: BLE001
                err, graph = f"{type(e).__name__}: {e}"[:300], None
                try:
                    torch.cuda.synchronize(dev)
                except Exception:                            # noqa: BLE001
                    pass
            if not _agree(err is None, dev, self.pg):
                errors[form] = err or "failed on another rank"
                times[form] = float("inf")
                restore()
                continue
            run = graph.replay if graph is not None else body
            reps = max(1, updates // chunk) if graph is not None else max(1, updates // (4 * chunk))
            run()
            torch.cuda.synchronize(dev)
            dist.barrier(group=self.pg)
            t0 = time.perf_counter()
            for _ in range(reps):
                run()
            torch.cuda.synchronize(dev)
            took = (time.perf_counter() - t0) / (reps * chunk) * 1e6
            if verbose:
                import sys
                print(f"[autotune rank {dist.get_rank(self.pg)}] {form}: {took:.1f} us per update ({reps} x {chunk}, graph={graph is not None}, "
                      f"fold fallbacks {self.fold_fallbacks})", file=sys.stderr, flush=True)
            times[form] = min(times.get(form, 1e30), took)
            del graph
            restore()
            torch.cuda.synchronize(dev)
        # ONE reduction carries everything the ranks must agree on before they choose: each form's time (the slowest rank bounds a
        # lock-step update) AND the count of peer waits that timed out while the forms were being timed — that count is per rank (a
        # late rank A makes rank B give up while A then finds B's flags in place), and ranks that decided on their own counts would
        # part: one in an RCCL all-reduce without a partner, the other in peer waits
        my_timeouts = float(self.xgmi.status()[1]) if self.xgmi is not None else 0.0
        t = torch.tensor([times[f] for f in forms] + [my_timeouts], dtype=torch.float64, device=dev if nccl else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.pg)
        *tt, timeouts = t.tolist()
        timeouts = int(timeouts)
        agreed = {f: (round(float(v), 2) if v != float("inf") else None) for f, v in zip(forms, tt)}
        usable = [f for f in forms if agreed[f] is not None] or forms[:1]
        # the peer-memory forms sum in RANK order, RCCL in ring order: the same job gives the same bits from launch to launch only
        # if warm-up noise cannot move it between the two families — `auto` therefore chooses among the rank-ordered forms and
        # keeps RCCL as the fall-back (none of them usable, or one of them lost a wait while being timed);
        # NAF_DP_EXCHANGE=fastest lets the clock decide among all of them
        ordered = [f for f in usable if f != "rccl"]
        pool = usable if (self._exchange_pick == "fastest" or not ordered) else ordered
        best = min(pool, key=lambda f: (agreed[f] if agreed[f] is not None else 0.0, forms.index(f)))
        if timeouts and best != "rccl":                      # a peer-memory form that lost a wait while being timed is not trusted
            best = "rccl"
        self._set_exchange(best)
        self.exchange_autotune = dict(agreed, chosen=best, updates_timed=updates, xgmi_timed_out_waits=int(timeouts))
        if errors:
            self.exchange_autotune["errors"] = errors
        if dist.get_rank(self.pg) == 0:
            import sys
            print("[naf] gradient exchange on this node, us per update: " +
                  ", ".join(f"{f} {agreed[f]:.1f}" if agreed[f] is not None else f"{f} unavailable ({errors.get(f, '?')})" for f in forms) +
                  f" -> {best}", file=sys.stderr, flush=True)
        return self.exchange_autotune

    # ---- parameters in / out ----------------------------------------------------------------------------
    def main_views(self) -> Dict[str, torch.Tensor]:
        return self.lay.param_views(self.theta2[0])

    def target_views(self) -> Dict[str, torch.Tensor]:
        return self.lay.param_views(self.theta2[1])

    def bn_views(self, net: int) -> Dict[str, torch.Tensor]:
        s, h = self.bn_stats[net], self.lay.H_ref
        return {"bn1.running_mean": s[0][:h], "bn1.running_var": s[1][:h], "bn2.running_mean": s[2][:h], "bn2.running_var": s[3][:h]}

    def load_params(self, net: int, sd: Dict[str, torch.Tensor]) -> None:
        self._gen += 1
        views = self.lay.param_views(self.theta2[net])
        with torch.no_grad():
            for k, v in views.items():
                v.copy_(torch.as_tensor(sd[k]).to(self.dev, torch.float32).reshape(v.shape))
            for k, v in self.bn_views(net).items():
                if k in sd:
                    v.copy_(torch.as_tensor(sd[k]).to(self.dev, torch.float32))

    def reset_optimizer(self) -> None:
        self._gen += 1
        self.adam_m.zero_()
        self.adam_v.zero_()
        self.step_dev.zero_()

    def raise_on_device_error(self) -> None:
        """Host-side check of what the kernels can only flag: a bounded wait on a PEER that expired (the one-shot all-reduce).
        Reads a pinned host word the kernel bumps — a load, never a synchronisation; the value lags the stream by whatever is
        still queued. Called before every chunk of updates and every learn(). (The waits INSIDE a launch — the records of the
        BatchNorm folds — cannot fail: a thread that waits too long folds for itself, see fold_fallbacks.)"""
        if self._xg is not None:
            self._xg.raise_on_timeout()

    @property
    def fold_fallbacks(self) -> int:
        """How many polls inside the fused launches gave up on the folded records after 20 us and folded for themselves (same
        bits either way: csrc/bn2bwd_fold.h). Zero when the process has the GPU to itself; a count that grows means the GPU is
        shared or over-subscribed. Read from pinned host memory without a synchronisation."""
        return int(self.err_host[0])

    # ---- one learn() on the current stream --------------------------------------------------------------
    def _x2(self, rows: torch.Tensor) -> torch.Tensor:
        """[2, B, S] view of a [B, ld] minibatch: net 0 reads `state`, net 1 reads `next_state`."""
        lay = self.lay
        return rows.as_strided((2, self.B, lay.S), (lay.off_s2, rows.stride(0), 1), rows.storage_offset())

    def moments(self, rows: torch.Tensor, out: torch.Tensor, n_batches: int = 1) -> None:
        """Moments records of n_batches minibatches stored back to back in `rows` ([n_batches * B, ld]) -> out
        [n_batches, 2, mom_floats]: one launch (row-split chain only)."""
        lay = self.lay
        ld = rows.stride(-2)
        check(self._f.naf_bb_moments(rows.data_ptr(), self.B * ld, lay.off_s2, ld, lay.S, ptr(out), self.B, int(n_batches), 2,
                                     stream_ptr()), "bb_moments")

    def layer1_args(self, rows: torch.Tensor, moments: torch.Tensor) -> "_lib.BbLayer1":
        """The arguments of the row-split chain's first launch (layer 1 of both networks from the minibatch's moments record) as a
        structure: naf_adam_polyak_act_layer1 runs that launch's body in extra workgroups of the per-timestep path's first launch
        (csrc/step_path.hip), and learn_rows(..., l1_done=True) then starts the chain at GEMM 2. Reads the working BatchNorm
        statistics (bn_live) as they are set when it is called."""
        lay, H, P = self.lay, self.lay.H, self.lay.P
        seg, t2p, bnp = lay.seg, self.theta2.data_ptr(), self.bn_live.data_ptr()
        return _lib.BbLayer1(rows.data_ptr(), lay.off_s2, rows.stride(0), lay.S, t2p + 4 * seg["W1"].offset, t2p + 4 * seg["b1"].offset,
                             t2p + 4 * seg["g1"].offset, t2p + 4 * seg["be1"].offset, P, ptr(moments), bnp, bnp + 4 * H, 4 * H,
                             ptr(self.A1), self.Bp * H, H, ptr(self.save_mean[0]), ptr(self.save_invstd[0]), ptr(self.bb_wc),
                             ptr(self.XH1), self.B, H, 2, BN_MOMENTUM, BN_EPS)

    def forward_train(self, rows: torch.Tensor, heads_gemm: bool = True, moments: Optional[torch.Tensor] = None,
                      adam_pending: bool = False, l1_done: bool = False) -> None:
        """Both networks' training-mode forward up to the second hidden activation A2 (and, with heads_gemm, the
        heads pre-activations Gh). Main net sees `state`, target net sees `next_state` (naf_algorithm.py:194-202);
        both use batch statistics and both update their running statistics (the reference never calls .eval() on the
        target). adam_pending: the optimizer step of the previous update was deferred (learn_rows(defer=True)) and rides
        on the first two launches here. Row-split chain: stops behind GEMM 2 — layer 2 from Z2 on is inside
        naf_bb_layer2_head (learn_rows)."""
        lay, B, st = self.lay, self.B, stream_ptr()
        seg, P, H = lay.seg, lay.P, lay.H
        t2p = self.theta2.data_ptr()
        bnp = self.bn_stats.data_ptr()
        if "bb" in self.fuse:
            bnp = self.bn_live.data_ptr()      # (the row-split chain's statistics: bn_stats itself unless a pipelined chunk says otherwise)
            ld = rows.stride(0)
            if moments is None:      # a minibatch that came without its moments (learn_rows called directly): one launch more
                self.moments(rows, self.bb_mom)
                moments = self.bb_mom
            self._mom = moments
            # layer 1: batch statistics from the moments, z, normalise, ReLU — one launch for both nets
            # (adam_pending: the previous update's clip + Adam + Polyak ride on these two launches — extra workgroups of
            # the first step everything behind the layer-1 segment while its own workgroups evaluate the layer-1
            # parameters as the step will leave them, extra workgroups of the second step the layer-1 segment)
            adam = _lib.C.byref(self._adam_args) if adam_pending else None
            if l1_done:
                # (layer 1 of this minibatch rode on the per-timestep path's first launch, naf_adam_polyak_act_layer1: layer1_args)
                if adam_pending:
                    raise ValueError("forward_train: l1_done with a pending optimizer step (that step rides on layer 1's launch)")
            else:
                check(self._f.naf_bb_layer1_adam(
                    rows.data_ptr(), lay.off_s2, ld, lay.S, t2p + 4 * seg["W1"].offset, t2p + 4 * seg["b1"].offset,
                    t2p + 4 * seg["g1"].offset, t2p + 4 * seg["be1"].offset, P, ptr(moments), bnp, bnp + 4 * H, 4 * H,
                    ptr(self.A1), self.Bp * H, H, ptr(self.save_mean[0]), ptr(self.save_invstd[0]), ptr(self.bb_wc), ptr(self.XH1), B, H,
                    2, BN_MOMENTUM, BN_EPS, adam, st), "bb_layer1")
            # GEMM 2 of both nets on f32 MFMA, bias added, statistics partials from the epilogue
            check(self._f.naf_bb_linear_stats_adam(ptr(self.A1), self.Bp * H, H, t2p + 4 * seg["W2"].offset,
                                                   t2p + 4 * seg["b2"].offset, P, ptr(self.G2), self.Bp * H, H, ptr(self.bb_st2), B, H,
                                                   H, 2, adam, st), "bb_linear_stats")
            return
        if "l1" in self.fuse:
            # layer 1 (K = state size): GEMM + bias + BN + ReLU of both nets in one launch, straight off the rows
            check(self._f.naf_linear_bn_relu_fwd_train(
                rows.data_ptr(), lay.off_s2, rows.stride(0), lay.S, t2p + 4 * seg["W1"].offset, t2p + 4 * seg["b1"].offset,
                t2p + 4 * seg["g1"].offset, t2p + 4 * seg["be1"].offset, P, bnp, bnp + 4 * H, 4 * H, ptr(self.A1), B * H, H,
                ptr(self.save_mean[0]), ptr(self.save_invstd[0]), B, H, 2, BN_MOMENTUM, BN_EPS, st), "linear_bn_relu_fwd_train")
        else:
            torch.bmm(self._x2(rows), self.W1T2, out=self.G1)
            check(self._f.naf_bn_relu_fwd_train(
                ptr(self.G1), B * H, H, t2p + 4 * seg["b1"].offset, t2p + 4 * seg["g1"].offset, t2p + 4 * seg["be1"].offset, P,
                bnp, bnp + 4 * H, 4 * H, ptr(self.A1), B * H, H, ptr(self.save_mean[0]), ptr(self.save_invstd[0]),
                B, H, 2, BN_MOMENTUM, BN_EPS, st), "bn_relu_fwd_train(1)")
        torch.bmm(self.A1, self.W2T2, out=self.G2)
        if heads_gemm and "s3" in self.fuse:
            # layer-2 BN + ReLU and, on the same tile, this workgroup's K-slice of the heads GEMM (no heads launch)
            check(self._f.naf_bn_relu_fwd_heads_partial(
                ptr(self.G2), B * H, H, t2p + 4 * seg["b2"].offset, t2p + 4 * seg["g2"].offset, t2p + 4 * seg["be2"].offset,
                P, bnp + 8 * H, bnp + 12 * H, 4 * H, ptr(self.A2), B * lay.HP, lay.HP, ptr(self.save_mean[1]),
                ptr(self.save_invstd[1]), t2p + 4 * seg["Wh"].offset, P, lay.HP, lay.NHP, lay.A + lay.T,
                ptr(self.heads_partial), self.slab_stride, ptr(self.vnext_partial), B, H, BN_MOMENTUM, BN_EPS, st),
                "bn_relu_fwd_heads_partial")
            return
        check(self._f.naf_bn_relu_fwd_train(
            ptr(self.G2), B * H, H, t2p + 4 * seg["b2"].offset, t2p + 4 * seg["g2"].offset, t2p + 4 * seg["be2"].offset, P,
            bnp + 8 * H, bnp + 12 * H, 4 * H, ptr(self.A2), B * lay.HP, lay.HP, ptr(self.save_mean[1]),
            ptr(self.save_invstd[1]), B, H, 2, BN_MOMENTUM, BN_EPS, st), "bn_relu_fwd_train(2)")
        if heads_gemm:
            torch.bmm(self.A2, self.WhT2, out=self.Gh)

    def learn_rows(self, rows: torch.Tensor, loss_partials: Optional[torch.Tensor] = None,
                   moments: Optional[torch.Tensor] = None, pending: bool = False, defer: bool = False, l1_done: bool = False) -> None:
        """Enqueue one full NAFAgent.learn() (naf_algorithm.py:180-215) + soft_update (:217-226) on the minibatch
        `rows` [B, ld] in the transition-row layout, ld = rows.stride(0) >= lay.batch_row_floats (actions already
        truncated by the gather if the reference's `.long()` is mimicked).
        loss_partials: optional [n_loss_wg] f32 receiving the per-workgroup parts of the MSE loss.
        moments: optional [2, mom_floats] record of this minibatch (row-split chain; Learner.moments); computed here when
        missing.
        defer / pending (only where self.defer_ok; a chain of updates, engine.TrainChunk): defer = leave this update's
        optimizer step (clip + Adam + Polyak) to the NEXT learn_rows call, which must then say pending = True and whose
        first two launches carry it — one launch less per update. Between the two calls the parameter buffers still hold
        the values from before this update; the chain ends with a call that does not defer.
        l1_done (row-split chain): layer 1 of this minibatch has run already — in the extra workgroups of the per-timestep path's
        first launch (layer1_args, naf_adam_polyak_act_layer1) — and the chain starts at GEMM 2."""
        self._gen += 1
        if (pending or defer) and not self.defer_ok:
            raise ValueError("learn_rows: a deferred optimizer step needs the row-split chain with the gradient norm left by the "
                             "producers or by the one-shot all-reduce (Learner.defer_ok)")
        if not pending:
            self.raise_on_device_error()
        lay, B, st = self.lay, self.B, stream_ptr()
        seg, P, H, HP, NHP = lay.seg, lay.P, lay.H, lay.HP, lay.NHP
        f = self._f
        t2p, gp = self.theta2.data_ptr(), self.grad.data_ptr()
        rp, ld = rows.data_ptr(), rows.stride(0)
        if rows.shape[0] != B or rows.stride(1) != 1 or ld < lay.batch_row_floats or ld % 4 or rp % 16:
            raise ValueError(f"learn_rows: need [B={B}, >={lay.batch_row_floats}] f32 rows, 16-B aligned, row stride % 4 == 0")
        lp = ptr(loss_partials) if loss_partials is not None else None
        if l1_done and ("bb" not in self.fuse or pending or moments is None):
            raise ValueError("learn_rows: l1_done needs the row-split chain, the minibatch's moments and no pending optimizer step")
        if "bb" in self.fuse:
            self._learn_rows_split(rows, lp, moments, pending, l1_done)
        else:
            self._learn_rows_tiles(rows, lp)
        if self.world_size > 1 or self._force_allreduce:
            # data parallel: one sum all-reduce of the flat gradient over RCCL/xGMI; the 1/W is folded into the
            # clip scale of the optimizer kernel (the clip acts on the averaged gradient)
            if self._xg is not None:
                # push + rank-ordered reduce in one launch, which also leaves the sum-of-squares partials and the step count
                # (merged: the finish launch of the row-split chain has done all of that already)
                if not (self.xgmi_merged and "bb" in self.fuse):
                    self._xg.all_reduce(self.grad, self.grad, self.partials, self.step_live, pushed_lo=self._pushed_lo,
                                         pushed_also=self._pushed_also)
                if not defer:
                    self.optimizer_step(norm_ready=True)
                return
            all_reduce_flat_grad(self.grad, self.pg)
            if defer:
                # the step rides on the next update; the sum-of-squares partials of the reduced gradient (and the step count) it
                # reads are left by this launch
                check(self._f.naf_grad_norm_partials(ptr(self.grad), P, ptr(self.partials), ptr(self.step_live), st), "grad_norm")
                return
        if defer:
            return                       # the next learn_rows(pending=True) carries the step
        self.optimizer_step(norm_ready=self.fold_norm)

    def _learn_rows_split(self, rows: torch.Tensor, lp, moments, pending: bool, l1_done: bool = False) -> None:
        """The row-split chain (csrc/big_batch.hip + gemm_bundle.hip): layer 1 | GEMM 2 | layer 2 + heads + head + first
        backward stage | backward GEMM bundle with its prologue and epilogue | finish."""
        lay, B, st = self.lay, self.B, stream_ptr()
        seg, P, H, HP, NHP = lay.seg, lay.P, lay.H, lay.HP, lay.NHP
        f = self._f
        t2p, gp, bnp = self.theta2.data_ptr(), self.grad.data_ptr(), self.bn_live.data_ptr()
        rp, ld = rows.data_ptr(), rows.stride(0)
        self._pushed_lo = self._pushed_also = None
        self.forward_train(rows, moments=moments, adam_pending=pending, l1_done=l1_done)
        # BN2 + ReLU + heads (MFMA) + NAF head + dA2 (MFMA) + ReLU mask + backward block sums: one launch
        check(f.naf_bb_layer2_head(
            ptr(self.G2), self.Bp * H, H, t2p + 4 * seg["g2"].offset, t2p + 4 * seg["be2"].offset, P, ptr(self.bb_st2),
            bnp + 8 * H, bnp + 12 * H, 4 * H, ptr(self.A2[0]), HP, ptr(self.save_mean[1]), ptr(self.save_invstd[1]),
            t2p + 4 * seg["Wh"].offset, P, HP, NHP, rp + 4 * lay.off_u, ld, rp + 4 * lay.off_r, ld, self.gamma,
            ptr(self.q_out), ptr(self.dH), lp, ptr(self.dZ2), H, ptr(self.bb_bw2), B, H, lay.A, self.p_mode, BN_MOMENTUM,
            BN_EPS, self._stats_once, st), "bb_layer2_head")
        # dWh = dH^T A2, dW2 = dZ2^T A1, dA1 = dZ2 W2: one launch of MFMA tiles; dY2 becomes dZ2 while it is staged, the
        # dA1 blocks run layer 1's backward batch pass on their tile (this minibatch's rows: z recomputed from them)
        self._epi.x, self._epi.ldx = rp, ld
        check(f.naf_gemm_bundle(self._bundle, 3, st), "gemm_bundle")
        # finish: everything added in block order, the xhat term from the moments, the bundle's split-K slabs, the norm
        # partials (nb = 0: the layer-2 bias gradient is written as the 0 it identically is). Data parallel over peer memory:
        # the workgroups that add the slabs of dW2 and dWh (91 % of the flat gradient) store them to the peers as well, so the
        # all-reduce launch behind this one sends only the layer-1 / BatchNorm segments before it raises its flags
        push = None
        merged = False
        if self._xg is not None and seg["Wh"].offset + seg["Wh"].numel == P:      # (Wh ends the buffer: no pad behind it)
            if self._push_desc is None:
                self._push_desc = self._xg.push_desc()
            push = _lib.C.byref(self._push_desc)
            self._pushed_lo = seg["Wh"].offset                                         # Wh is the last segment: [Wh, P)
            self._pushed_also = (seg["W2"].offset, seg["W2"].offset + H * H)
            merged = self.xgmi_merged
        elif self.xgmi_merged:
            raise _lib.NafHipError("merged gradient exchange: the flat layout must end with the Wh segment")
        norm_here = self.fold_norm or merged
        check(f.naf_bb_layer1_bwd_finish(
            ptr(self.bb_dw1), lay.S, ptr(self.bb_bw1), self._nb1, None, 0,
            ptr(self._mom), ptr(self.bb_wc), t2p + 4 * seg["g1"].offset, ptr(self.save_invstd[0, 0]),
            gp + 4 * seg["W1"].offset, gp + 4 * seg["g1"].offset, gp + 4 * seg["be1"].offset, gp + 4 * seg["b1"].offset,
            gp + 4 * seg["b2"].offset, gp + 4 * seg["g2"].offset, gp + 4 * seg["be2"].offset,
            ptr(self.partials) if norm_here else None, ptr(self.step_live) if norm_here else None, B, H,
            self._bb_segs, self._bb_nsegs, ptr(self.bb_fold_flag), push, gp if push is not None else None, P if merged else 0, st),
            "bb_layer1_bwd_finish")

    def _learn_rows_tiles(self, rows: torch.Tensor, lp) -> None:
        """The column-tile chain (csrc/fused_layers.hip) and the unfused chain (torch GEMMs + csrc/bn_relu.hip), by `fuse`."""
        lay, B, st = self.lay, self.B, stream_ptr()
        seg, P, H, HP, NHP = lay.seg, lay.P, lay.H, lay.HP, lay.NHP
        f = self._f
        t2p, gp = self.theta2.data_ptr(), self.grad.data_ptr()
        rp, ld = rows.data_ptr(), rows.stride(0)
        self._pushed_lo = self._pushed_also = None
        self.forward_train(rows)
        if "s3" in self.fuse:
            # the head adds the split-K slabs (H/8 of them) while staging its rows
            check(f.naf_head_fwd_bwd_mse_splitk(
                ptr(self.heads_partial), self.slab_stride, ptr(self.vnext_partial), self.n_slabs, NHP, rp + 4 * lay.off_u,
                ld, rp + 4 * lay.off_r, ld, self.gamma, ptr(self.q_out), ptr(self.dH), lp, B, lay.A, self.p_mode,
                st), "head_fwd_bwd_mse_splitk")
        else:
            # y = r + gamma * V'(s') ; Q ; loss ; d loss / d heads_pre — one launch
            check(f.naf_head_fwd_bwd_mse(
                ptr(self.Gh[0]), NHP, rp + 4 * lay.off_u, ld, rp + 4 * lay.off_r, ld,
                self.Gh[1].data_ptr() + 4 * (lay.A + lay.T), NHP, self.gamma, ptr(self.q_out), ptr(self.dH), lp, B, lay.A,
                self.p_mode, st), "head_fwd_bwd_mse")
        # heads GEMM backward: weight+bias gradient in one GEMM thanks to the ones column
        gb = "gb" in self.fuse
        if not gb:
            torch.mm(self.dH.t(), self.A2[0], out=self.gWh)
        if "b2" in self.fuse:
            # dA2 = dH @ Wh (K = NHP) folded into the ReLU/BN backward of layer 2
            check(f.naf_heads_bwd_bn_relu_bwd(
                ptr(self.dH), NHP, t2p + 4 * seg["Wh"].offset, HP, ptr(self.G2[0]), H, t2p + 4 * seg["b2"].offset,
                ptr(self.A2[0]), HP, t2p + 4 * seg["g2"].offset, ptr(self.save_mean[1, 0]), ptr(self.save_invstd[1, 0]),
                ptr(self.dZ2), H, gp + 4 * seg["g2"].offset, gp + 4 * seg["be2"].offset, gp + 4 * seg["b2"].offset,
                self.partials.data_ptr() + 4 * (self._gb_blocks + self._ft_blocks) if self.fold_norm else None, B, H, st),
                "heads_bwd_bn_relu_bwd")
        else:
            torch.mm(self.dH, self.Wh_main, out=self.dA2)
            check(f.naf_bn_relu_bwd(
                ptr(self.dA2), HP, ptr(self.G2[0]), H, t2p + 4 * seg["b2"].offset, ptr(self.A2[0]), HP,
                t2p + 4 * seg["g2"].offset, ptr(self.save_mean[1, 0]), ptr(self.save_invstd[1, 0]), ptr(self.dZ2), H,
                gp + 4 * seg["g2"].offset, gp + 4 * seg["be2"].offset, gp + 4 * seg["b2"].offset, B, H, st), "bn_relu_bwd(2)")
        if gb:
            # dWh = dH^T A2, dW2 = dZ2^T A1, dA1 = dZ2 W2: one launch of MFMA tiles
            check(f.naf_gemm_bundle(self._bundle, 3, st), "gemm_bundle")
        else:
            torch.mm(self.dZ2.t(), self.A1[0], out=self.gW2)
            torch.mm(self.dZ2, self.W2_main, out=self.dA1)
        if "l1" in self.fuse:
            # ReLU/BN backward of layer 1 + dW1 = dZ1^T X in one launch (dZ1 never written). Data parallel over peer
            # memory: everything but layer 1's gradient is final by now (segments W2 .. Wh of the flat buffer) and goes
            # to the peers from extra workgroups of this very launch, so its wire time runs under the kernel
            push = None
            if self._xg is not None:
                if self._push_desc is None:
                    self._push_desc = self._xg.push_desc()
                push, self._pushed_lo = _lib.C.byref(self._push_desc), seg["W2"].offset
            check(f.naf_bn_relu_bwd_wgrad_push(
                ptr(self.dA1), H, rp, ld, lay.S, t2p + 4 * seg["W1"].offset, t2p + 4 * seg["b1"].offset,
                ptr(self.A1[0]), H, t2p + 4 * seg["g1"].offset, ptr(self.save_mean[0, 0]), ptr(self.save_invstd[0, 0]),
                gp + 4 * seg["g1"].offset, gp + 4 * seg["be1"].offset, gp + 4 * seg["b1"].offset, gp + 4 * seg["W1"].offset,
                self.partials.data_ptr() + 4 * self._gb_blocks if self.fold_norm else None,
                ptr(self.step_dev) if self.fold_norm else None, B, H, push, gp if push is not None else None,
                self._pushed_lo or 0, P if push is not None else 0, st), "bn_relu_bwd_wgrad")
        else:
            check(f.naf_bn_relu_bwd(
                ptr(self.dA1), H, ptr(self.G1[0]), H, t2p + 4 * seg["b1"].offset, ptr(self.A1[0]), H,
                t2p + 4 * seg["g1"].offset, ptr(self.save_mean[0, 0]), ptr(self.save_invstd[0, 0]), ptr(self.dZ1), H,
                gp + 4 * seg["g1"].offset, gp + 4 * seg["be1"].offset, gp + 4 * seg["b1"].offset, B, H, st), "bn_relu_bwd(1)")
            torch.mm(self.dZ1.t(), self._x2(rows)[0], out=self.gW1)

    def optimizer_step(self, norm_ready: Optional[bool] = None) -> None:
        """clip_grad_norm_(params, 1) + Adam.step() + soft_update on the flat buffers: 2 launches (1 when the
        producers of the gradient already left its sum-of-squares partials)."""
        self._gen += 1
        st, P = stream_ptr(), self.lay.P
        f = self._f
        if norm_ready is None:
            norm_ready = self.fold_norm
        if not norm_ready:
            check(f.naf_grad_norm_partials(ptr(self.grad), P, ptr(self.partials), ptr(self.step_dev), st), "grad_norm")
        check(f.naf_adam_polyak_fused(
            ptr(self.theta2[0]), ptr(self.grad), ptr(self.adam_m), ptr(self.adam_v), ptr(self.theta2[1]),
            ptr(self.partials), self.n_partials if norm_ready else self.n_partials_norm, MAX_GRAD_NORM, self.lr,
            ADAM_BETA1, ADAM_BETA2, ADAM_EPS, self.tau, float(1.0 - self.tau), ptr(self.step_dev), 1.0 / self.world_size, P, st), "adam_polyak")

    def soft_update(self) -> None:
        """Standalone NAFAgent.soft_update(main, target) (naf_algorithm.py:217-226) over the flat buffers."""
        self._gen += 1
        check(self._f.naf_polyak_update(ptr(self.theta2[1]), ptr(self.theta2[0]), self.tau, float(1.0 - self.tau),
                                        self.lay.P, stream_ptr()), "polyak")


class ActPath:
    """Eval-mode policy forward for E states at once (NAFAgent.act, naf_algorithm.py:158-178): running-stat BN
    (main net only), mu = tanh, exploration noise clamp(mu + P^-1/2 z) — noise is drawn on EVERY call exactly as
    the reference's forward does (naf_neural_network.py:119-121), unless noise_scale = 0."""

    def __init__(self, learner: Learner, n_states: int, seed: int, host_io: bool = False):
        """host_io: `obs` and `actions` live in pinned HOST memory that the one-launch kernel reads and writes directly
        (84 B in, 24 B out per state): the per-call H2D / D2H copies of NAFAgent.act() disappear, the caller fills
        `obs_np`, enqueues act(), synchronises the stream and reads `actions_np`. Only with the fused kernel."""
        self.L = learner
        lay, dev = learner.lay, learner.dev
        self.E = int(n_states)
        f32 = dict(dtype=torch.float32, device=dev)
        E, H, HP, NHP = self.E, lay.H, lay.HP, lay.NHP
        self.obs = torch.zeros(E, lay.S, **f32)
        self.G1 = torch.empty(E, H, **f32)
        self.A1 = torch.empty(E, H, **f32)
        self.G2 = torch.empty(E, H, **f32)
        self.A2 = torch.zeros(E, HP, **f32)
        self.A2[:, H] = 1.0
        self.Gh = torch.empty(E, NHP, **f32)
        self.actions = torch.zeros(E, lay.A, **f32)
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)   # noise stream position (uint64 on device)
        # one launch for the whole act() (csrc/policy_act.hip) when the shapes are the framework's (H = 256, S <= 32);
        # other shapes take the seven-launch path (3 GEMMs, 2 BN kernels, noise, counter)
        # (round 6: up to 11 joints — the state's group in the noise body is then 16 lanes wide)
        #  and layer sizes up to 512 — stored as 512: policy_act_512_kernel)
        self.fused = lay.H in (256, 512) and lay.S <= 32 and lay.A <= BB_MAX_JOINTS
        self._ticket = torch.zeros(1, dtype=torch.int32, device=dev)
        self.host_io = bool(host_io) and self.fused
        if self.host_io:
            self.obs = torch.zeros(E, lay.S, dtype=torch.float32).pin_memory()
            self.actions = torch.zeros(E, lay.A, dtype=torch.float32).pin_memory()
            self.obs_np, self.actions_np = self.obs.numpy(), self.actions.numpy()
        self.W1T = learner.W1T2[0]
        self.W2T = learner.W2T2[0]
        self.WhT = learner.WhT2[0]
        # ONE state on the row-split chain: the optimizer step of the update in front can take this act() along
        # (naf_adam_polyak_act, csrc/step_path.hip: the workgroups that step the weights multiply them with the policy's
        # activations) — NAFAgent.step's graph then ends with one launch instead of two. `seq` (pinned host): the launch's ordinal,
        # written behind the action: a host that polls it reads the action without synchronising the stream.
        seg = lay.seg
        self.can_ride = (self.fused and E == 1 and learner.defer_ok and lay.NHP <= 80 and
                         [seg[k].offset for k in ("W1", "b1", "g1", "be1", "W2", "b2", "g2", "be2", "Wh")] ==
                         [0, H * lay.S, H * lay.S + H, H * lay.S + 2 * H, H * lay.S + 3 * H, H * lay.S + 3 * H + H * H,
                          H * lay.S + 4 * H + H * H, H * lay.S + 5 * H + H * H, H * lay.S + 6 * H + H * H] and
                         lay.P == H * lay.S + 6 * H + H * H + lay.NHP * lay.HP)
        self.sync = self.seq = self.seq_np = self.ordinal_np = None
        if self.can_ride:
            self.sync = torch.zeros(learner.lib.naf_adam_polyak_act_sync_ints(), dtype=torch.int32, device=dev)
            self.seq = torch.zeros(2, dtype=torch.int32).pin_memory()
            self.seq_np = self.seq.numpy()
            # the action as the launch hands it to a host that does not synchronise the stream: three (9 .. 11 joints: four) 16-byte
            # chunks {a[3j], a[3j + 1], a[3j + 2], ordinal}, one store each (naf_adam_polyak_act: action_rec); `ordinal_np` = chunk 0's
            self.act_rec = torch.zeros(16, dtype=torch.int32).pin_memory()
            self.rec_np = self.act_rec.numpy()
            self.rec_f = self.rec_np.view(np.float32)
            self.ordinal_np = self.rec_np[3:4]
            self._rec_words = [w for w in (0, 1, 2, 4, 5, 6, 8, 9, 10, 12, 13)][:lay.A]
            self._rec_ords = [4 * j + 3 for j in range((lay.A + 2) // 3)]
            bnp = learner.bn_stats.data_ptr()
            self._net = _lib.ActNet(lay.S, lay.A, H, lay.NHP, lay.HP, *[seg[k].offset for k in
                                    ("W1", "b1", "g1", "be1", "W2", "b2", "g2", "be2", "Wh")],
                                    bnp, bnp + 4 * H, bnp + 8 * H, bnp + 12 * H, BN_EPS)

    def act_with_optimizer_step(self, noise_scale: float = 1.0, obs_ptr: Optional[int] = None, prefetch=None,
                                obs_system_scope: bool = False, adam_args=None, net=None, layer1=None) -> torch.Tensor:
        """The pending optimizer step of the learner (a learn_rows(defer=True) in front) and act() on the parameters it leaves, in
        one launch. Same parameters as Learner.optimizer_step() and the same action as act() behind it, bit for bit.
        obs_ptr: where the observation lies instead of self.obs (device-visible, S floats). prefetch: a _lib.StepPrefetch — one
        more workgroup of the launch draws the NEXT timestep's minibatch (engine.TrainChunk). obs_system_scope: the observation
        lies in device memory the host stores into. adam_args / net: the structures to pass instead of the learner's / this path's
        own (the pipelined chunk's: optimizer step count and BatchNorm statistics of its working state). layer1: a _lib.BbLayer1
        (Learner.layer1_args) — layer 1 of the NEXT update's chain rides on the launch (naf_adam_polyak_act_layer1); the caller then
        runs that chain with learn_rows(..., l1_done=True)."""
        L = self.L
        if layer1 is not None:
            check(L.lib.naf_adam_polyak_act_layer1(
                _lib.C.byref(adam_args if adam_args is not None else L._adam_args), _lib.C.byref(net if net is not None else self._net),
                obs_ptr or ptr(self.obs), ptr(self.Gh), ptr(self.actions), self.seed, ptr(self.counter), float(noise_scale), L.p_mode,
                # (no `host_seq` word here: the pipelined timestep's host takes the action from the self-validating chunks of act_rec
                #  alone, and the launch's last workgroup would wait a microsecond for the action's stores to cross PCIe before it
                #  stores that word — at the very end of the kernel the chain is queued behind)
                ptr(self.sync), L.err_host.data_ptr() + 8, None, ptr(self.act_rec),
                _lib.C.byref(prefetch) if prefetch is not None else None, int(bool(obs_system_scope)), _lib.C.byref(layer1),
                stream_ptr()), "adam_polyak_act_layer1")
            return self.actions
        check(L.lib.naf_adam_polyak_act(_lib.C.byref(adam_args if adam_args is not None else L._adam_args),
                                        _lib.C.byref(net if net is not None else self._net), obs_ptr or ptr(self.obs), ptr(self.Gh),
                                        ptr(self.actions), self.seed, ptr(self.counter), float(noise_scale), L.p_mode,
                                        ptr(self.sync), L.err_host.data_ptr() + 8, ptr(self.seq), ptr(self.act_rec),
                                        _lib.C.byref(prefetch) if prefetch is not None else None, int(bool(obs_system_scope)),
                                        stream_ptr()), "adam_polyak_act")
        return self.actions

    @property
    def act_timeouts(self) -> int:
        """Polls inside naf_adam_polyak_act whose 2-ms bound ran out (pinned host word; 0 on a healthy GPU): the action of such a
        launch is NaN."""
        return int(self.L.err_host[1])

    def heads(self) -> None:
        L, lay, E, st = self.L, self.L.lay, self.E, stream_ptr()
        seg, H = lay.seg, lay.H
        t2p, bnp = L.theta2.data_ptr(), L.bn_stats.data_ptr()
        f = L.lib
        torch.mm(self.obs, self.W1T, out=self.G1)
        check(f.naf_bn_relu_fwd_eval(ptr(self.G1), H, t2p + 4 * seg["b1"].offset, t2p + 4 * seg["g1"].offset,
                                     t2p + 4 * seg["be1"].offset, bnp, bnp + 4 * H, ptr(self.A1), H, E, H, BN_EPS, st),
              "bn_relu_fwd_eval(1)")
        torch.mm(self.A1, self.W2T, out=self.G2)
        check(f.naf_bn_relu_fwd_eval(ptr(self.G2), H, t2p + 4 * seg["b2"].offset, t2p + 4 * seg["g2"].offset,
                                     t2p + 4 * seg["be2"].offset, bnp + 8 * H, bnp + 12 * H, ptr(self.A2), lay.HP, E, H,
                                     BN_EPS, st), "bn_relu_fwd_eval(2)")
        torch.mm(self.A2, self.WhT, out=self.Gh)

    def act(self, noise_scale: float = 1.0) -> torch.Tensor:
        """obs (already in self.obs) -> self.actions; advances the noise counter on the device."""
        L, lay, st = self.L, self.L.lay, stream_ptr()
        if self.fused:
            seg, H = lay.seg, lay.H
            t2p, bnp = L.theta2.data_ptr(), L.bn_stats.data_ptr()
            off = lambda name: t2p + 4 * seg[name].offset          # noqa: E731
            check(L.lib.naf_policy_act(
                ptr(self.obs), lay.S, lay.S, off("W1"), off("b1"), off("g1"), off("be1"), off("W2"), off("b2"), off("g2"),
                off("be2"), off("Wh"), lay.HP, lay.NH, bnp, bnp + 4 * H, bnp + 8 * H, bnp + 12 * H, BN_EPS, H, ptr(self.Gh),
                lay.NHP, ptr(self.actions), self.seed, ptr(self.counter), ptr(self._ticket), float(noise_scale), self.E,
                lay.A, L.p_mode, st), "policy_act")
            return self.actions
        self.heads()
        check(L.lib.naf_act_noise(ptr(self.Gh), lay.NHP, ptr(self.actions), self.seed, ptr(self.counter), 0,
                                  float(noise_scale), self.E, lay.A, L.p_mode, st), "act_noise")
        check(L.lib.naf_counter_add(ptr(self.counter), 1, st), "counter_add")
        return self.actions
