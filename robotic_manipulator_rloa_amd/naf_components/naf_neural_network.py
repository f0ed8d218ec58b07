"""NAF network with the reference's interface (naf_components/naf_neural_network.py:8-123) whose parameters are
views into a flat HBM buffer and whose forward runs on libnaf_hip.so kernels + PyTorch-ROCm GEMMs.

    NAF(state_size, action_size, layer_size, seed, device)
    forward(input_, action=None) -> (noisy clamped action (B,A), Q (B,1) | None, V (B,1))

Same state_dict keys/shapes, same parameters() order, same seed -> bit-identical initial weights (the
constructor consumes torch's global CPU RNG exactly as the reference's does).
Reference quirks kept on purpose: P = L * L^T elementwise (p_mode='hadamard', the default;
'matmul' gives textbook NAF), integer actions are accepted and promoted, every forward draws exploration noise.
forward() is differentiable (like the reference's): when gradients are enabled and the parameters require them, the
BatchNorm+ReLU and head kernels are wrapped in torch.autograd.Function (backward = the same HIP kernels learn() uses)
and the Linears run as torch matmuls, so `Q.sum().backward()` fills `.grad` of the 14 parameters. NAFAgent.learn()
does not go through this path: its backward is fused into the flat-buffer kernels.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Any, Optional, Tuple

import torch
from torch import nn

from .. import _lib
from .._lib import check, ptr, stream_ptr
from ..learner import BN_EPS, BN_MOMENTUM, NetLayout, PARAM_ORDER

_P_MODES = {"hadamard": _lib.P_HADAMARD, "matmul": _lib.P_MATMUL, 0: 0, 1: 1}


def reference_init_state_dict(state_size: int, action_size: int, layer_size: int, seed: int) -> "OrderedDict[str, torch.Tensor]":
    """CPU state_dict with exactly the weights the reference constructor produces for `seed`: it calls
    torch.manual_seed(seed) and then builds Linear/BatchNorm1d layers in this order
    (naf_neural_network.py:33-54), so doing the same with stock torch modules reproduces the draws."""
    torch.manual_seed(seed)
    T = int(action_size * (action_size + 1) / 2)
    mods = OrderedDict([
        ("input_layer", nn.Linear(state_size, layer_size)), ("bn1", nn.BatchNorm1d(layer_size)),
        ("hidden_layer", nn.Linear(layer_size, layer_size)), ("bn2", nn.BatchNorm1d(layer_size)),
        ("action_values", nn.Linear(layer_size, action_size)), ("value", nn.Linear(layer_size, 1)),
        ("matrix_entries", nn.Linear(layer_size, T))])
    sd = OrderedDict()
    for name, m in mods.items():
        for k, v in m.state_dict().items():
            sd[f"{name}.{k}"] = v.detach().clone()
    return sd


class _BnReluTrain(torch.autograd.Function):
    """relu(batch_norm(g + bias)) in training mode for one network through naf_bn_relu_fwd_train / naf_bn_relu_bwd."""

    @staticmethod
    def forward(ctx, g, bias, gamma, beta, running_mean, running_var, lib):
        g = g.contiguous()
        B, H = g.shape
        out = torch.empty_like(g)
        sm, si = torch.empty(H, device=g.device), torch.empty(H, device=g.device)
        bias_c, gamma_c, beta_c = bias.contiguous(), gamma.contiguous(), beta.contiguous()
        check(lib.naf_bn_relu_fwd_train(ptr(g), 0, H, ptr(bias_c), ptr(gamma_c), ptr(beta_c), 0, ptr(running_mean),
                                        ptr(running_var), 0, ptr(out), 0, H, ptr(sm), ptr(si), B, H, 1, BN_MOMENTUM, BN_EPS,
                                        stream_ptr()), "bn_relu_fwd_train")
        ctx.save_for_backward(g, bias_c, gamma_c, out, sm, si)
        ctx.lib = lib
        return out

    @staticmethod
    def backward(ctx, d_out):
        g, bias, gamma, out, sm, si = ctx.saved_tensors
        B, H = g.shape
        d_out = d_out.contiguous()
        dz = torch.empty_like(g)
        dg, db, dbias = (torch.empty(H, device=g.device) for _ in range(3))
        check(ctx.lib.naf_bn_relu_bwd(ptr(d_out), H, ptr(g), H, ptr(bias), ptr(out), H, ptr(gamma), ptr(sm), ptr(si), ptr(dz), H,
                                      ptr(dg), ptr(db), ptr(dbias), B, H, stream_ptr()), "bn_relu_bwd")
        return dz, dbias, dg, db, None, None, None


class _NafHead(torch.autograd.Function):
    """Q from heads_pre = [mu_pre | l_pre | V | pad] and the action, through naf_head_fwd / naf_head_bwd."""

    @staticmethod
    def forward(ctx, heads, u, A, p_mode, lib):
        heads = heads.contiguous()
        B, ldh = heads.shape
        q = torch.empty(B, device=heads.device)
        check(lib.naf_head_fwd(ptr(heads), ldh, ptr(u), A, ptr(q), None, B, A, p_mode, stream_ptr()), "naf_head_fwd")
        ctx.save_for_backward(heads, u)
        ctx.meta = (A, p_mode, lib)
        return q

    @staticmethod
    def backward(ctx, dq):
        heads, u = ctx.saved_tensors
        A, p_mode, lib = ctx.meta
        B, ldh = heads.shape
        dq = dq.contiguous()
        dh = torch.empty_like(heads)
        check(lib.naf_head_bwd(ptr(heads), ldh, ptr(u), A, ptr(dq), ptr(dh), B, A, p_mode, stream_ptr()), "naf_head_bwd")
        return dh, None, None, None, None


class NAF(nn.Module):

    def __init__(self, state_size: int, action_size: int, layer_size: int, seed: int, device, *,
                 p_mode="hadamard", _flat: Optional[torch.Tensor] = None, _bn: Optional[torch.Tensor] = None,
                 _init: bool = True, pad_layer: bool = True) -> None:
        """
        Args (reference order, naf_neural_network.py:10): state_size, action_size, layer_size, seed, device.
        p_mode: 'hadamard' (reference parity) or 'matmul'.
        _flat/_bn: storage handed in by NAFAgent so main and target share one [2,P] allocation.
        """
        super().__init__()
        _lib.require_gpu()
        self.lib = _lib.load()
        self.seed = torch.manual_seed(seed)          # same global side effect as the reference (:33)
        self.state_size, self.action_size, self.layer_size = state_size, action_size, layer_size
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.NafHipError("NAF runs on the MI355X only (device must be cuda:N); there is no CPU path")
        self.p_mode = _P_MODES[p_mode]
        # (a layer_size below 256 is stored zero-padded to 256: NetLayout — lay.H is the stored width, lay.H_ref = layer_size)
        self.layout = NetLayout(state_size, action_size, layer_size, pad_layer=pad_layer)
        lay = self.layout
        self.flat = _flat if _flat is not None else torch.zeros(lay.P, dtype=torch.float32, device=self.device)
        self.bn_stats = _bn if _bn is not None else torch.zeros(4, lay.H, dtype=torch.float32, device=self.device)
        if tuple(self.bn_stats.shape) != (4, lay.H) or self.flat.numel() != lay.P:
            raise ValueError("NAF: the storage handed in does not have this layout's sizes (pad_layer differs from the owner's?)")
        if _bn is None:
            self.bn_stats[1].fill_(1.0)
            self.bn_stats[3].fill_(1.0)
        T = lay.T
        # nn.Module facade: stock layers (meta-device, nothing allocated) whose tensors are re-pointed at the flat buffer
        with torch.device("meta"):
            self.input_layer = nn.Linear(state_size, layer_size)
            self.bn1 = nn.BatchNorm1d(layer_size)
            self.hidden_layer = nn.Linear(layer_size, layer_size)
            self.bn2 = nn.BatchNorm1d(layer_size)
            self.action_values = nn.Linear(layer_size, action_size)
            self.value = nn.Linear(layer_size, 1)
            self.matrix_entries = nn.Linear(layer_size, T)
        views = lay.param_views(self.flat)
        for name in PARAM_ORDER:
            mod, attr = name.split(".")
            setattr(getattr(self, mod), attr, nn.Parameter(views[name], requires_grad=True))
        h = lay.H_ref
        self.bn1.running_mean, self.bn1.running_var = self.bn_stats[0][:h], self.bn_stats[1][:h]
        self.bn2.running_mean, self.bn2.running_var = self.bn_stats[2][:h], self.bn_stats[3][:h]
        self.bn1.num_batches_tracked = torch.zeros((), dtype=torch.long, device=self.device)
        self.bn2.num_batches_tracked = torch.zeros((), dtype=torch.long, device=self.device)
        self._tracked_base = 0            # batches tracked at load time
        self._tracked_eager = 0           # train-mode forwards through this facade
        self._tracked_hook = None         # NAFAgent: learn() count living on the device
        self._noise_counter = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._noise_seed = (int(seed) * 0x9E3779B97F4A7C15 + 0x5851F42D4C957F2D) & 0xFFFFFFFFFFFFFFFF
        if _init:
            self.load_state_dict(reference_init_state_dict(state_size, action_size, layer_size, seed))

    # ---- nn.Module plumbing ------------------------------------------------------------------------------------
    def to(self, *args, **kwargs):
        dev = args[0] if args else kwargs.get("device")
        if dev is None or torch.device(dev) == self.device or (torch.device(dev).type == "cuda" and torch.device(dev).index is None):
            return self
        raise _lib.NafHipError(f"NAF parameters live in one flat HBM buffer on {self.device}; cannot move to {dev}")

    def _tracked(self) -> int:
        extra = int(self._tracked_hook()) if self._tracked_hook is not None else 0
        return self._tracked_base + self._tracked_eager + extra

    def state_dict(self, *args, **kwargs):
        """Reference key set and order; tensors are contiguous clones (safe to torch.save: no flat-buffer aliasing)."""
        views = self.layout.param_views(self.flat)
        n = torch.tensor(self._tracked(), dtype=torch.long, device=self.device)
        sd = OrderedDict()
        for mod in ("input_layer", "bn1", "hidden_layer", "bn2", "action_values", "value", "matrix_entries"):
            sd[f"{mod}.weight"] = views[f"{mod}.weight"].detach().clone().contiguous()
            sd[f"{mod}.bias"] = views[f"{mod}.bias"].detach().clone().contiguous()
            if mod.startswith("bn"):
                k = 0 if mod == "bn1" else 2
                sd[f"{mod}.running_mean"] = self.bn_stats[k][:self.layout.H_ref].clone()
                sd[f"{mod}.running_var"] = self.bn_stats[k + 1][:self.layout.H_ref].clone()
                sd[f"{mod}.num_batches_tracked"] = n.clone()
        return sd

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        views = self.layout.param_views(self.flat)
        missing = [k for k in PARAM_ORDER if k not in state_dict]
        if missing and strict:
            raise RuntimeError(f"Error(s) in loading state_dict for NAF: missing keys {missing}")
        with torch.no_grad():
            for k, v in views.items():
                if k in state_dict:
                    src = torch.as_tensor(state_dict[k])
                    if tuple(src.shape) != tuple(v.shape):
                        raise RuntimeError(f"size mismatch for {k}: {tuple(src.shape)} vs {tuple(v.shape)}")
                    v.copy_(src.to(self.device, torch.float32))
            for i, k in enumerate(("bn1.running_mean", "bn1.running_var", "bn2.running_mean", "bn2.running_var")):
                if k in state_dict:
                    self.bn_stats[i][:self.layout.H_ref].copy_(torch.as_tensor(state_dict[k]).to(self.device, torch.float32))
        if "bn1.num_batches_tracked" in state_dict:
            self._tracked_base = int(torch.as_tensor(state_dict["bn1.num_batches_tracked"]).item())
            self._tracked_eager = 0
            if self._tracked_hook is not None:
                self._tracked_base -= int(self._tracked_hook())
        return torch.nn.modules.module._IncompatibleKeys(missing, [])

    # ---- forward -------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def heads(self, input_: torch.Tensor) -> torch.Tensor:
        """Trunk + merged heads GEMM: returns heads_pre [B, NHP] = [mu_pre | l_pre | V | 0-pad] (bias included)."""
        lay, lib, st = self.layout, self.lib, stream_ptr()
        seg, H, HP = lay.seg, lay.H, lay.HP
        x = input_.to(self.device, torch.float32)
        if x.dim() == 1:
            x = x.unsqueeze(0)
        B = x.shape[0]
        fp, bp = self.flat.data_ptr(), self.bn_stats.data_ptr()
        f32 = dict(dtype=torch.float32, device=self.device)
        a2 = torch.zeros(B, HP, **f32)
        a2[:, H] = 1.0
        a1 = torch.empty(B, H, **f32)
        g1 = torch.mm(x, lay.view(self.flat, "W1").t())
        g2 = torch.empty(B, H, **f32)
        if self.training:
            sm, si = torch.empty(H, **f32), torch.empty(H, **f32)
            check(lib.naf_bn_relu_fwd_train(ptr(g1), 0, H, fp + 4 * seg["b1"].offset, fp + 4 * seg["g1"].offset,
                                            fp + 4 * seg["be1"].offset, 0, bp, bp + 4 * H, 0, ptr(a1), 0, H, ptr(sm), ptr(si),
                                            B, H, 1, BN_MOMENTUM, BN_EPS, st), "bn_relu_fwd_train")
            torch.mm(a1, lay.view(self.flat, "W2").t(), out=g2)
            check(lib.naf_bn_relu_fwd_train(ptr(g2), 0, H, fp + 4 * seg["b2"].offset, fp + 4 * seg["g2"].offset,
                                            fp + 4 * seg["be2"].offset, 0, bp + 8 * H, bp + 12 * H, 0, ptr(a2), 0, HP, ptr(sm),
                                            ptr(si), B, H, 1, BN_MOMENTUM, BN_EPS, st), "bn_relu_fwd_train")
            self._tracked_eager += 1
        else:
            check(lib.naf_bn_relu_fwd_eval(ptr(g1), H, fp + 4 * seg["b1"].offset, fp + 4 * seg["g1"].offset,
                                           fp + 4 * seg["be1"].offset, bp, bp + 4 * H, ptr(a1), H, B, H, BN_EPS, st), "bn_eval")
            torch.mm(a1, lay.view(self.flat, "W2").t(), out=g2)
            check(lib.naf_bn_relu_fwd_eval(ptr(g2), H, fp + 4 * seg["b2"].offset, fp + 4 * seg["g2"].offset,
                                           fp + 4 * seg["be2"].offset, bp + 8 * H, bp + 12 * H, ptr(a2), HP, B, H, BN_EPS, st),
                  "bn_eval")
        return torch.mm(a2, lay.view(self.flat, "Wh").t())

    def _heads_autograd(self, input_: torch.Tensor) -> torch.Tensor:
        """Differentiable heads_pre [B, NHP] (training mode only): torch matmuls + the BN kernels as autograd Functions."""
        lay = self.layout
        x = input_.to(self.device, torch.float32)
        if x.dim() == 1:
            x = x.unsqueeze(0)
        h = _BnReluTrain.apply(x @ self.input_layer.weight.t(), self.input_layer.bias, self.bn1.weight, self.bn1.bias,
                               self.bn1.running_mean, self.bn1.running_var, self.lib)
        h = _BnReluTrain.apply(h @ self.hidden_layer.weight.t(), self.hidden_layer.bias, self.bn2.weight, self.bn2.bias,
                               self.bn2.running_mean, self.bn2.running_var, self.lib)
        self._tracked_eager += 1
        mu_pre = h @ self.action_values.weight.t() + self.action_values.bias
        l_pre = h @ self.matrix_entries.weight.t() + self.matrix_entries.bias
        v = h @ self.value.weight.t() + self.value.bias
        pad = x.new_zeros(x.shape[0], lay.NHP - lay.NH)
        return torch.cat([mu_pre, l_pre, v, pad], dim=1)

    def forward(self, input_: torch.Tensor, action: Optional[torch.Tensor] = None,
                noise_scale: float = 1.0) -> Tuple[torch.Tensor, Optional[Any], Any]:
        """(noisy action, Q | None, V), as naf_neural_network.py:56-123. `action` may be int64 (what the reference's
        ReplayBuffer.sample() yields) or float. Q and V carry an autograd graph when gradients are enabled in training
        mode; the noisy action never does (the reference samples it without reparameterisation too)."""
        lay, lib, st = self.layout, self.lib, stream_ptr()
        want_grad = torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.parameters())
        if want_grad:
            gh = self._heads_autograd(input_)
            B = gh.shape[0]
            V = gh[:, lay.A + lay.T].unsqueeze(-1)
            Q = None
            if action is not None:
                u = action.to(self.device, torch.float32).contiguous().view(B, lay.A)
                Q = _NafHead.apply(gh, u, lay.A, self.p_mode, lib).unsqueeze(-1)
            with torch.no_grad():
                ghd = gh.detach().contiguous()
                noisy = torch.empty(B, lay.A, dtype=torch.float32, device=self.device)
                check(lib.naf_act_noise(ptr(ghd), lay.NHP, ptr(noisy), self._noise_seed, ptr(self._noise_counter), 0,
                                        float(noise_scale), B, lay.A, self.p_mode, st), "naf_act_noise")
                check(lib.naf_counter_add(ptr(self._noise_counter), 1, st), "naf_counter_add")
            return noisy, Q, V
        with torch.no_grad():
            return self._forward_nograd(input_, action, noise_scale)

    @torch.no_grad()
    def _forward_nograd(self, input_, action, noise_scale):
        lay, lib, st = self.layout, self.lib, stream_ptr()
        gh = self.heads(input_)
        B = gh.shape[0]
        V = gh[:, lay.A + lay.T].clone().unsqueeze(-1)
        Q = None
        if action is not None:
            u = action.to(self.device, torch.float32).contiguous().view(B, lay.A)
            q = torch.empty(B, dtype=torch.float32, device=self.device)
            check(lib.naf_head_fwd(ptr(gh), lay.NHP, ptr(u), lay.A, ptr(q), None, B, lay.A, self.p_mode, st), "naf_head_fwd")
            Q = q.unsqueeze(-1)
        noisy = torch.empty(B, lay.A, dtype=torch.float32, device=self.device)
        check(lib.naf_act_noise(ptr(gh), lay.NHP, ptr(noisy), self._noise_seed, ptr(self._noise_counter), 0,
                                float(noise_scale), B, lay.A, self.p_mode, st), "naf_act_noise")
        check(lib.naf_counter_add(ptr(self._noise_counter), 1, st), "naf_counter_add")
        return noisy, Q, V
