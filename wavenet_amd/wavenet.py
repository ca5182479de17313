"""`WaveNet` / `Params` with the reference's Python face, backed by libwavenet_hip.so.

Drop-in for wavenet.py of musyoku/wavenet: same class, method and hyper-parameter names (including
``update_laerning_rate``), same logical tensor shapes ``(B, C, 1, T)`` and the same numerical
conventions (zero prefix of the reshape-trick convolution, activation before every head conv,
cross-entropy row order).  Nothing here computes on the CPU: every forward / backward FLOP is a HIP
kernel reached through the C ABI in include/wavenet_hip.h, and calling a forward method without a
GPU raises.  PyTorch supplies device memory, streams and the autograd tape only.

Memory layout: activations live time-major / channel-minor, ``x[b][t][c]``; the tensors handed back
to the caller are ``(B, C, 1, T)`` *views* of that storage (``torch.channels_last`` strides), so the
reference's indexing (``out[0, :, 0, -1]``) works unchanged and no transpose is ever materialised on
the hot path.  T-contiguous inputs (numpy one-hot images) are converted once at the boundary.
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, ptr_array, int_array, stream_ptr, ACT


# ----------------------------------------------------------------------------------------------
# hyper-parameters (wavenet.py:100-173): same fields, same defaults, same checks
# ----------------------------------------------------------------------------------------------
class Params(object):
    def __init__(self, dict=None):
        self.quantization_steps = 256
        self.sampling_rate = 8000
        self.causal_conv_no_bias = True
        self.causal_conv_filter_width = 2
        self.causal_conv_channels = [128]
        self.residual_conv_dilation_no_bias = True
        self.residual_conv_projection_no_bias = True
        self.residual_conv_filter_width = 2
        self.residual_conv_channels = [32, 32, 32, 32, 32, 32, 32, 32, 32]
        self.residual_num_blocks = 2
        self.softmax_conv_no_bias = False
        self.softmax_conv_channels = [128, 256]
        self.optimizer = "adam"
        self.weight_decay = 0
        self.momentum = 0.9
        self.gradient_clipping = 1.0
        if dict:
            self.from_dict(dict)

    def from_dict(self, dict):
        for attr, value in dict.items():
            if hasattr(self, attr):           # unknown keys are ignored, wavenet.py:151-154
                setattr(self, attr, value)

    def to_dict(self):
        return {attr: value for attr, value in self.__dict__.items()}

    def dump(self):
        print("params:")
        for attr, value in self.__dict__.items():
            print("	{}: {}".format(attr, value))

    def check(self):
        base = Params()
        for attr in self.__dict__:
            if not hasattr(base, attr):
                raise Exception("invalid parameter '{}'".format(attr))
        if self.quantization_steps != self.softmax_conv_channels[-1]:
            raise Exception("quantization_steps != softmax_conv_channels[-1]")


def zero_prefix(T: int, d: int, fw: int) -> int:
    """Columns the reference's reshape trick leaves at exactly 0 (wavenet.py:303-340)."""
    if d == 1:
        return 0
    pad = (-T) % d
    if (T + pad) // d < fw:
        pad += (fw - (T + pad) // d) * d
    return max(0, (fw - 1) * d - pad)


# ----------------------------------------------------------------------------------------------
# tensor plumbing
# ----------------------------------------------------------------------------------------------
def _need_gpu(t: torch.Tensor):
    if not t.is_cuda:
        raise _lib.WaveNetHipError(
            "this operation runs on the MI355X only: call to_gpu() first (there is no CPU path)")


def _as_view(btc: torch.Tensor) -> torch.Tensor:
    """(B,T,C) storage -> the reference's logical (B,C,1,T) shape, zero copy."""
    B, T, Cc = btc.shape
    return btc.view(B, T, 1, Cc).permute(0, 3, 2, 1)


def _to_btc(x: torch.Tensor) -> torch.Tensor:
    """Logical (B,C,1,T) tensor -> dense (B,T,C) storage; zero copy when x is one of our views."""
    if x.dim() != 4 or x.shape[2] != 1:
        raise Exception("expected a (B, C, 1, T) tensor, got %s" % (tuple(x.shape),))
    B, Cc, _, T = x.shape
    st = x.stride()
    if (st[1] == 1 or Cc == 1) and (st[3] == Cc or T == 1) and (st[0] == T * Cc or B == 1):
        return x.permute(0, 3, 2, 1).reshape(B, T, Cc)          # a view: memory already (B,T,C)
    if x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and not x.requires_grad:
        out = torch.empty((B, T, Cc), device=x.device, dtype=torch.float32)
        check(_lib.lib().wn_nchw_to_btc(ptr(x), ptr(out), B, Cc, T, stream_ptr()), "wn_nchw_to_btc")
        return out
    return x.permute(0, 3, 2, 1).reshape(B, T, Cc).contiguous()


class _Fn(torch.autograd.Function):
    """Base for the autograd nodes below.  Weight gradients are accumulated by the kernels straight
    into the model's flat gradient arena (``p.grad`` are views of it), so backward returns None for
    weights and tensors only for activations."""


class _EmbedFn(_Fn):
    @staticmethod
    def forward(ctx, idx, W, b, fw, net):
        B, T = idx.shape
        Cc, Q = W.shape[0], W.shape[1]
        out = torch.empty((B, T, Cc), device=idx.device, dtype=torch.float32)
        check(_lib.lib().wn_embed_fwd(ptr(idx), ptr(W), ptr(b), ptr(out), B, T, Q, Cc, fw, stream_ptr()),
              "wn_embed_fwd")
        ctx.save_for_backward(idx)
        ctx.W, ctx.b, ctx.fw, ctx.net = W, b, fw, net
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        W, b = ctx.W, ctx.b
        B, T = idx.shape
        dout = dout.contiguous()
        check(_lib.lib().wn_embed_bwd(ptr(idx), ptr(dout), ptr(W.grad), ptr(None if b is None else b.grad), B, T,
                                      W.shape[1], W.shape[0], ctx.fw, ctx.net._exec(B, T), stream_ptr()), "wn_embed_bwd")
        return None, None, None, None, None


class _ConvFn(_Fn):
    """Dense dilated causal convolution (DilatedConvolution1D.__call__)."""

    @staticmethod
    def forward(ctx, x, W, b, fw, d, Z, hook):
        B, T, Cin = x.shape
        Cout = W.shape[0]
        x = x.contiguous()
        out = torch.empty((B, T, Cout), device=x.device, dtype=torch.float32)
        check(_lib.lib().wn_conv_fwd(ptr(x), ptr(W), ptr(b), ptr(out), B, T, Cin, Cout, fw, d, Z, stream_ptr()),
              "wn_conv_fwd")
        ctx.save_for_backward(x)
        ctx.W, ctx.b, ctx.geom = W, b, (fw, d, Z)
        return out

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        W, b = ctx.W, ctx.b
        fw, d, Z = ctx.geom
        B, T, Cin = x.shape
        dout = dout.contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        check(_lib.lib().wn_conv_bwd(ptr(x), ptr(W), ptr(dout), ptr(dx), ptr(W.grad),
                                     ptr(None if b is None else b.grad), B, T, Cin, W.shape[0], fw, d, Z,
                                     stream_ptr()), "wn_conv_bwd")
        return dx, None, None, None, None, None, None


class _PointwiseFn(_Fn):
    """out = W act(x) + b  (head convs, projection convs)."""

    @staticmethod
    def forward(ctx, x, W, b, act, net):
        lead = x.shape[:-1]
        Cin, Cout = x.shape[-1], W.shape[0]
        x2 = x.reshape(-1, Cin).contiguous()
        out = torch.empty((x2.shape[0], Cout), device=x.device, dtype=torch.float32)
        check(_lib.lib().wn_pointwise_fwd(ptr(x2), ptr(W), ptr(b), ptr(out), x2.shape[0], Cin, Cout, act,
                                          net._exec(), stream_ptr()), "wn_pointwise_fwd")
        ctx.save_for_backward(x2)
        ctx.W, ctx.b, ctx.act, ctx.lead, ctx.net = W, b, act, lead, net
        return out.view(*lead, Cout)

    @staticmethod
    def backward(ctx, dout):
        (x2,) = ctx.saved_tensors
        W, b = ctx.W, ctx.b
        Cout, Cin = W.shape[0], x2.shape[1]
        dout2 = dout.reshape(-1, Cout).contiguous()
        dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        check(_lib.lib().wn_pointwise_bwd(ptr(x2), ptr(W), ptr(dout2), ptr(dx), ptr(W.grad),
                                          ptr(None if b is None else b.grad), x2.shape[0], Cin, Cout, ctx.act,
                                          ctx.net._exec(), stream_ptr()), "wn_pointwise_bwd")
        return (None if dx is None else dx.view(*ctx.lead, Cin)), None, None, None, None


class _SoftmaxFn(_Fn):
    @staticmethod
    def forward(ctx, x):
        Q = x.shape[-1]
        x2 = x.reshape(-1, Q).contiguous()
        out = torch.empty_like(x2)
        check(_lib.lib().wn_softmax_fwd(ptr(x2), ptr(out), x2.shape[0], Q, stream_ptr()), "wn_softmax_fwd")
        ctx.save_for_backward(out)
        return out.view(x.shape)

    @staticmethod
    def backward(ctx, dout):   # not on the reference's training path (train.py:76 passes apply_softmax=False)
        (p,) = ctx.saved_tensors
        g = dout.reshape(p.shape)
        return (p * (g - (g * p).sum(-1, keepdim=True))).view(dout.shape)


class _XentFn(_Fn):
    @staticmethod
    def forward(ctx, logits, target, n_norm=0):
        N, Q = logits.shape
        buf = torch.empty((_lib.XENT_LOSS_WORDS,), device=logits.device, dtype=torch.float32)
        dlog = torch.empty_like(logits) if ctx.needs_input_grad[0] else None
        check(_lib.lib().wn_softmax_xent(ptr(logits), ptr(target), ptr(buf), ptr(dlog), N, Q, int(n_norm), stream_ptr()),
              "wn_softmax_xent")
        ctx.dlog = dlog
        return buf[0]                                               # 0-dim view: buf[1:] are the per-workgroup sums

    @staticmethod
    def backward(ctx, dloss):
        # dlog was written by the forward for an upstream gradient of 1; scale it in place by the actual one, read from
        # device memory (no host sync, and no pass over the 8.4 M values when it is 1)
        dlog, ctx.dlog = ctx.dlog, None
        if dlog is None:
            raise RuntimeError("the cross-entropy node can be backpropagated once (its gradient buffer is scaled in place)")
        d = dloss.to(torch.float32).reshape(1).contiguous()
        check(_lib.lib().wn_scale_by_dev(ptr(dlog), ptr(d), dlog.numel(), stream_ptr()), "wn_scale_by_dev")
        return dlog, None, None


class _HeadXentFn(_Fn):
    """The last head convolution and the softmax cross-entropy as ONE node and one launch (wn_head_xent): the logits are formed
    and consumed on the chip.  forward_softmax_block(apply_softmax=False) followed by cross_entropy (wavenet.py:584-617) gives
    the same loss and the same gradients through two nodes and a (N, Q) logits tensor in memory."""

    @staticmethod
    def forward(ctx, x, W, b, target, act, n_norm, net):
        lead = x.shape[:-1]
        Cin, Cout = x.shape[-1], W.shape[0]
        x2 = x.reshape(-1, Cin).contiguous()
        N = x2.shape[0]
        buf = torch.empty((_lib.XENT_LOSS_WORDS,), device=x.device, dtype=torch.float32)
        dlog = torch.empty((N, Cout), device=x.device, dtype=torch.float32)
        check(_lib.lib().wn_head_xent(ptr(x2), ptr(W), ptr(b), ptr(target), ptr(buf), ptr(dlog), N, Cin, Cout, act,
                                      int(n_norm), net._exec(), stream_ptr()), "wn_head_xent")
        ctx.save_for_backward(x2)
        ctx.W, ctx.b, ctx.act, ctx.lead, ctx.net, ctx.dlog = W, b, act, lead, net, dlog
        return buf[0]

    @staticmethod
    def backward(ctx, dloss):
        (x2,) = ctx.saved_tensors
        dlog, ctx.dlog = ctx.dlog, None
        if dlog is None:
            raise RuntimeError("the fused head + cross-entropy node can be backpropagated once (its gradient buffer is scaled in place)")
        W, b = ctx.W, ctx.b
        Cout, Cin = W.shape[0], x2.shape[1]
        lib = _lib.lib()
        if not ctx.net._unit_upstream:       # (TrainStepGraph backpropagates a constant 1: not even the launch that would find that out)
            d = dloss.to(torch.float32).reshape(1).contiguous()
            check(lib.wn_scale_by_dev(ptr(dlog), ptr(d), dlog.numel(), stream_ptr()), "wn_scale_by_dev")
        dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        check(lib.wn_pointwise_bwd(ptr(x2), ptr(W), ptr(dlog), ptr(dx), ptr(W.grad), ptr(None if b is None else b.grad),
                                   x2.shape[0], Cin, Cout, ctx.act, ctx.net._exec(), stream_ptr()), "wn_pointwise_bwd")
        return (None if dx is None else dx.view(*ctx.lead, Cin)), None, None, None, None, None, None


class _StackFn(_Fn):
    """All residual layers + the deferred skip sum as ONE autograd node and ONE library call each way
    (WaveNet.forward_residual_block, wavenet.py:572-582)."""

    @staticmethod
    def forward(ctx, x, anchor, net, t_off, train, window_only=False):
        # `anchor` is a dummy leaf that requires grad: it keeps this node on the tape even when x
        # does not require grad, because the weights are not tensor inputs of the node.
        ctx.set_materialize_grads(False)
        B, T, Cr = x.shape
        x = x.contiguous()
        desc = net._stack_desc()
        L = len(net._flat_layers)
        ncd = sum(lay.cd for lay in net._flat_layers)
        dev_ = x.device
        xs = torch.empty((L, B, T, Cr), device=dev_, dtype=torch.float32)
        z = torch.empty((B * T * ncd,), device=dev_, dtype=torch.float32)
        # tanh is saved only when the backward cannot take the chained path (which recovers it as z / sigmoid)
        f = torch.empty_like(z) if train and _lib.lib().wn_stack_saves_tanh(desc, net._exec(B, T)) else None
        g = torch.empty_like(z) if train else None
        skip = torch.empty((B, T - t_off, net._Cs), device=dev_, dtype=torch.float32)
        check(_lib.lib().wn_stack_fwd(desc, ptr(x), ptr(xs), ptr(z), ptr(f), ptr(g), ptr(skip), B, T, t_off,
                                      1 if net.compat_zero_prefix else 0, 1 if window_only else 0, net._exec(B, T),
                                      stream_ptr()), "wn_stack_fwd")
        ctx.net, ctx.t_off, ctx.shape, ctx.window_only = net, t_off, (B, T, Cr), bool(window_only)
        ctx.saved = (x, xs, z, f, g) if train else None
        net._last_layer_inputs = [x] + [xs[l] for l in range(L - 1)]      # FasterWaveNet seeds its rings from these
        return xs[L - 1], skip

    @staticmethod
    def backward(ctx, dout, dskip):
        net, t_off = ctx.net, ctx.t_off
        B, T, Cr = ctx.shape
        if ctx.saved is None:
            raise _lib.WaveNetHipError("backward through a forward that ran without grad enabled")
        x, xs, z, f, g = ctx.saved
        if ctx.window_only and dout is not None:
            raise _lib.WaveNetHipError("forward_residual_block(window_only=True) computed only what the skip window needs: "
                                       "the residual output cannot carry a gradient")
        lib = _lib.lib()
        desc = net._stack_desc()
        gt = net._grad_tables()
        nbytes = lib.wn_stack_bwd_workspace_bytes(desc, B, T)
        ws = torch.empty((nbytes // 4,), device=x.device, dtype=torch.float32)
        dout = None if dout is None else dout.contiguous()
        dskip = None if dskip is None else dskip.contiguous()
        dx = torch.empty((B, T, Cr), device=x.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        check(lib.wn_stack_bwd(desc, ptr(x), ptr(xs), ptr(z), ptr(f), ptr(g), ptr(dout), ptr(dskip), ptr(dx),
                               gt["wf"], gt["bf"], gt["wg"], gt["bg"], gt["wp"], gt["bp"], gt["ws"], gt["bs"],
                               ptr(ws), nbytes, B, T, t_off, 1 if net.compat_zero_prefix else 0, net._exec(B, T),
                               stream_ptr()), "wn_stack_bwd")
        ctx.saved = None
        return dx, None, None, None, None, None


# ----------------------------------------------------------------------------------------------
# bf16-storage nodes (BASELINE config 5; include/wavenet_hip.h "bf16-storage path"): activations are torch.bfloat16
# (B, T, C) tensors, weights and their gradients stay fp32 in the flat arenas
# ----------------------------------------------------------------------------------------------
class _Embed16Fn(_Fn):
    @staticmethod
    def forward(ctx, idx, W, b, fw, net):
        B, T = idx.shape
        Cc, Q = W.shape[0], W.shape[1]
        out = torch.empty((B, T, Cc), device=idx.device, dtype=torch.bfloat16)
        check(_lib.lib().wn16_embed_fwd(ptr(idx), ptr(W), ptr(b), ptr(out), B, T, Q, Cc, fw, stream_ptr()),
              "wn16_embed_fwd")
        ctx.save_for_backward(idx)
        ctx.W, ctx.b, ctx.fw, ctx.net = W, b, fw, net
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        W, b = ctx.W, ctx.b
        B, T = idx.shape
        dout = dout.contiguous()
        lib = _lib.lib()
        if W.shape[0] == 128 and W.shape[1] == 256 and ctx.fw == 2:
            # one-hot contraction on the matrix cores, straight from the bf16 gradient
            nws = lib.wn16_embed_bwd_workspace_bytes(B, T)
            ws = torch.empty((nws,), device=dout.device, dtype=torch.uint8)
            check(lib.wn16_embed_bwd(ptr(idx), ptr(dout), ptr(W.grad), ptr(None if b is None else b.grad), B, T,
                                     W.shape[1], W.shape[0], ctx.fw, ptr(ws), nws, stream_ptr()), "wn16_embed_bwd")
            return None, None, None, None, None
        d32 = torch.empty(dout.shape, device=dout.device, dtype=torch.float32)
        check(lib.wn16_cvt_to_f32(ptr(dout), ptr(d32), dout.numel(), stream_ptr()), "wn16_cvt_to_f32")
        check(lib.wn_embed_bwd(ptr(idx), ptr(d32), ptr(W.grad), ptr(None if b is None else b.grad), B, T,
                               W.shape[1], W.shape[0], ctx.fw, ctx.net._exec(B, T), stream_ptr()), "wn_embed_bwd")
        return None, None, None, None, None


class _Stack16Fn(_Fn):
    """forward_residual_block in bf16 storage: one library call each way (wn16_stack_fwd / wn16_stack_bwd).  The forward
    keeps every layer's output and z only; the backward recomputes tanh / sigmoid."""

    @staticmethod
    def forward(ctx, x, anchor, net, t_off, train):
        ctx.set_materialize_grads(False)
        B, T, Cr = x.shape
        x = x.contiguous()
        desc = net._stack_desc()
        L = len(net._flat_layers)
        dev_ = x.device
        xs = torch.empty((L, B, T, Cr), device=dev_, dtype=torch.bfloat16)
        z = torch.empty((L, B, T, Cr), device=dev_, dtype=torch.bfloat16)
        skip = torch.empty((B, T - t_off, net._Cs), device=dev_, dtype=torch.bfloat16)
        check(_lib.lib().wn16_stack_fwd(desc, ptr(net._pack16), ptr(x), ptr(xs), ptr(z), ptr(skip), B, T, t_off,
                                        1 if net.compat_zero_prefix else 0, stream_ptr()), "wn16_stack_fwd")
        ctx.net, ctx.t_off, ctx.shape = net, t_off, (B, T, Cr)
        ctx.saved = (x, xs, z) if train else None
        net._last16 = (x, xs, z, skip)                               # (tests look at the intermediates)
        return xs[L - 1], skip

    @staticmethod
    def backward(ctx, dout, dskip):
        net, t_off = ctx.net, ctx.t_off
        B, T, Cr = ctx.shape
        if ctx.saved is None:
            raise _lib.WaveNetHipError("backward through a forward that ran without grad enabled")
        x, xs, z = ctx.saved
        lib = _lib.lib()
        desc = net._stack_desc()
        gt = net._grad_tables()
        nbytes = lib.wn16_stack_bwd_workspace_bytes(desc, B, T, t_off)
        ws = torch.empty((nbytes,), device=x.device, dtype=torch.uint8)
        dout = None if dout is None else dout.contiguous()
        dskip = None if dskip is None else dskip.contiguous()
        dx = torch.empty((B, T, Cr), device=x.device, dtype=torch.bfloat16) if ctx.needs_input_grad[0] else None
        check(lib.wn16_stack_bwd(desc, ptr(net._pack16), ptr(x), ptr(xs), ptr(z), ptr(dout), ptr(dskip), ptr(dx),
                                 gt["wf"], gt["wg"], gt["wp"], gt["ws"], ptr(ws), nbytes, B, T, t_off,
                                 1 if net.compat_zero_prefix else 0,
                                 _lib.default_exec_flags() if net.exec_flags is None else int(net.exec_flags),
                                 stream_ptr()), "wn16_stack_bwd")
        net._last16_ws = ws
        ctx.saved = None
        return dx, None, None, None, None


class _Pointwise16Fn(_Fn):
    """out = W relu(x) + b on bf16 activations; the last head layer writes fp32 logits."""

    @staticmethod
    def forward(ctx, x, W, b, Wb, WbT, act, out_f32, hook):
        lead = x.shape[:-1]
        Cin, Cout = x.shape[-1], W.shape[0]
        x2 = x.reshape(-1, Cin).contiguous()
        out = torch.empty((x2.shape[0], Cout), device=x.device, dtype=torch.float32 if out_f32 else torch.bfloat16)
        check(_lib.lib().wn16_pointwise_fwd(ptr(x2), ptr(Wb), ptr(b), ptr(out), 1 if out_f32 else 0, x2.shape[0], Cin, Cout,
                                            act, stream_ptr()), "wn16_pointwise_fwd")
        ctx.save_for_backward(x2)
        ctx.W, ctx.b, ctx.WbT, ctx.act, ctx.lead = W, b, WbT, act, lead
        return out.view(*lead, Cout)

    @staticmethod
    def backward(ctx, dout):
        (x2,) = ctx.saved_tensors
        W, b = ctx.W, ctx.b
        Cout, Cin = W.shape[0], x2.shape[1]
        N = x2.shape[0]
        lib = _lib.lib()
        d2 = dout.reshape(-1, Cout).contiguous()
        d16 = d32 = scratch = None
        if d2.dtype == torch.float32:
            d32 = d2
            scratch = torch.empty((N, Cout), device=d2.device, dtype=torch.bfloat16)
        else:
            d16 = d2
            if b is not None:                                        # the bias gradient is summed in fp32
                d32 = torch.empty((N, Cout), device=d2.device, dtype=torch.float32)
                check(lib.wn16_cvt_to_f32(ptr(d16), ptr(d32), d16.numel(), stream_ptr()), "wn16_cvt_to_f32")
        dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        nws = lib.wn16_pointwise_bwd_workspace_bytes(N, Cout)      # per-workgroup partials: gradients without float atomics
        ws = torch.empty((nws,), device=d2.device, dtype=torch.uint8)
        check(lib.wn16_pointwise_bwd(ptr(x2), ptr(ctx.WbT), ptr(d16), ptr(d32), ptr(scratch), ptr(dx), ptr(W.grad),
                                     ptr(None if b is None else b.grad), N, Cin, Cout, ctx.act, ptr(ws), nws, stream_ptr()),
              "wn16_pointwise_bwd")
        return (None if dx is None else dx.view(*ctx.lead, Cin)), None, None, None, None, None, None, None


# ----------------------------------------------------------------------------------------------
# links: objects with the attribute names the reference's scripts touch
# ----------------------------------------------------------------------------------------------
class _Link(object):
    """A parameter holder: ``W`` (reference shape) and ``b`` (or None), views of the flat arena."""

    def __init__(self, name: str, wshape: Tuple[int, ...], bshape: Optional[Tuple[int]]):
        self.name, self.wshape, self.bshape = name, wshape, bshape
        self.W: Optional[torch.Tensor] = None
        self.b: Optional[torch.Tensor] = None


class DilatedConvolution1D(_Link):
    """Drop-in for wavenet.py:263-342 (``__call__`` = whole window, ``_forward`` = newest column)."""

    def __init__(self, net, name, in_channels, out_channels, ksize, filter_width=2, dilation=1, nobias=False):
        shape = (out_channels, in_channels) + tuple(ksize)
        super().__init__(name, shape, None if nobias else (out_channels,))
        self.net = net
        self.in_channels, self.out_channels = in_channels, out_channels
        self.filter_width, self.dilation = filter_width, dilation

    def __call__(self, x):
        x = self.net.to_variable(x)
        _need_gpu(x)
        xb = _to_btc(x)
        Z = self.net._Z(xb.shape[1], self.dilation)
        return _as_view(_ConvFn.apply(xb, self.W, self.b, self.filter_width, self.dilation, Z, self.net._hook))

    def _forward(self, x_batch_data):
        """Newest column only, batch 0 (wavenet.py:281-292); returns (1, Cout, 1, 1)."""
        x = self.net.to_variable(x_batch_data)
        _need_gpu(x)
        need = (self.filter_width - 1) * self.dilation + 1
        if x.shape[3] < need:
            raise IndexError("window of %d columns is shorter than the layer's reach %d" % (x.shape[3], need))
        cols = [x.shape[3] - 1 - (self.filter_width - 1 - k) * self.dilation for k in range(self.filter_width)]
        taps = x[0:1, :, :, cols]                                   # (1, Cin, 1, fw), oldest first
        with torch.no_grad():
            out = _ConvFn.apply(_to_btc(taps), self.W, self.b, self.filter_width, 1, 0, None)
        return _as_view(out[:, -1:, :])


class Convolution1x1(_Link):
    def __init__(self, net, name, in_channels, out_channels, nobias=False):
        super().__init__(name, (out_channels, in_channels, 1, 1), None if nobias else (out_channels,))
        self.net = net
        self.in_channels, self.out_channels = in_channels, out_channels

    def __call__(self, x, act="none"):
        x = self.net.to_variable(x)
        _need_gpu(x)
        return _as_view(_PointwiseFn.apply(_to_btc(x), self.W, self.b, ACT[act], self.net))


class ResidualConvLayer(object):
    """wavenet.py:344-368.  ``__call__`` returns (output, projection_softmax) like the reference."""

    def __init__(self, net):
        self.net = net
        self.wf = self.wg = self.projection_block = self.projection_softmax = None

    @property
    def cd(self):
        return self.wf.out_channels

    @property
    def fw(self):
        return self.wf.filter_width

    @property
    def dilation(self):
        return self.wf.dilation

    def _run(self, xb, save):
        B, T, Cr = xb.shape
        out = torch.empty_like(xb)
        z = torch.empty((B, T, self.cd), device=xb.device, dtype=torch.float32)
        check(_lib.lib().wn_layer_fwd(ptr(xb), ptr(self.wf.W), ptr(self.wf.b), ptr(self.wg.W), ptr(self.wg.b),
                                      ptr(self.projection_block.W), ptr(self.projection_block.b), ptr(out), ptr(z),
                                      None, None, B, T, Cr, self.cd, self.fw, self.dilation,
                                      self.net._Z(T, self.dilation), self.net._exec(B), stream_ptr()), "wn_layer_fwd")
        return out, z

    def __call__(self, x):
        """Inference-only single-layer call (training goes through forward_residual_block)."""
        x = self.net.to_variable(x)
        _need_gpu(x)
        with torch.no_grad():
            out, z = self._run(_to_btc(x).contiguous(), False)
            skip = _PointwiseFn.apply(z, self.projection_softmax.W, self.projection_softmax.b, ACT["none"], self.net)
        return _as_view(out), _as_view(skip)

    def _forward(self, x):
        """Newest column only (wavenet.py:350-356): (1,Cr,1,1), (1,Cs,1,1)."""
        x = self.net.to_variable(x)
        _need_gpu(x)
        fw, d = self.fw, self.dilation
        cols = [x.shape[3] - 1 - (fw - 1 - k) * d for k in range(fw)]
        taps = _to_btc(x[0:1, :, :, cols]).contiguous()             # (1, fw, Cr)
        with torch.no_grad():
            B, T, Cr = taps.shape
            out = torch.empty_like(taps)
            z = torch.empty((B, T, self.cd), device=taps.device, dtype=torch.float32)
            check(_lib.lib().wn_layer_fwd(ptr(taps), ptr(self.wf.W), ptr(self.wf.b), ptr(self.wg.W), ptr(self.wg.b),
                                          ptr(self.projection_block.W), ptr(self.projection_block.b), ptr(out),
                                          ptr(z), None, None, B, T, Cr, self.cd, fw, 1, 0, self.net._exec(B),
                                          stream_ptr()), "wn_layer_fwd")
            skip = _PointwiseFn.apply(z[:, -1:, :], self.projection_softmax.W, self.projection_softmax.b,
                                      ACT["none"], self.net)
        return _as_view(out[:, -1:, :]), _as_view(skip)


# ----------------------------------------------------------------------------------------------
# the model
# ----------------------------------------------------------------------------------------------
class _StepPlan(object):
    """A step plan handle (include/wavenet_hip.h, "step plan") and the device memory it lays its images out in."""

    def __init__(self, device, nbytes: int):
        self.mem = torch.empty((nbytes,), device=device, dtype=torch.uint8)
        h = C.c_void_p()
        check(_lib.lib().wn_plan_create(C.byref(h), self.mem.data_ptr(), nbytes), "wn_plan_create")
        self.handle = h

    def __del__(self):
        try:
            _lib.lib().wn_plan_destroy(self.handle)
        except Exception:
            pass


class WaveNet(object):
    """Drop-in for wavenet.py:370-639."""

    head_activation = "relu"          # wavenet.py:588

    def __init__(self, params, compat_zero_prefix: bool = True, seed: Optional[int] = None, storage: str = "fp32"):
        """``storage="bf16"`` (BASELINE config 5; not in the reference): activations in bfloat16, fp32 accumulation, fp32
        master weights; for 128 residual / dilation channels, filter width 2, a multiple of 256 skip channels."""
        params.check()
        if storage not in ("fp32", "bf16"):
            raise Exception("storage must be 'fp32' or 'bf16'")
        self.params = params
        self.storage = storage
        self.gemm_precision = None          # None: the module default (wavenet_amd.set_gemm_precision) at call time
        self.exec_flags = None              # None: _lib.default_exec_flags(); else WN_EXEC_* bits for every call of this model
        self.fuse_head_loss = os.environ.get("WAVENET_HIP_NO_FUSED_HEAD_LOSS") != "1"      # head_cross_entropy: one launch when covered
        self.fwd_t1_min_blocks = None       # None: _lib.default_fwd_t1_min_blocks() (WnExec.fwd_t1_min_blocks)
        self._scratch, self._scratch_keep = {}, []
        self._plan, self._plan_obj, self._plan_on = None, None, False     # step plan (wn_plan_*): see plan_begin
        self.use_step_plan = os.environ.get("WAVENET_HIP_NO_STEP_PLAN") != "1"   # TrainStepGraph: weight preparation in two launches
        self._unit_upstream = False         # True while a caller guarantees d loss = 1 (TrainStepGraph)
        self._pack16 = None
        self._w16_stale = True
        self.compat_zero_prefix = compat_zero_prefix
        self._gpu = False
        self._hook = None
        self._anchor = torch.zeros((1,), requires_grad=True)
        self._dp_group = None
        self._last_layer_inputs = None
        self.create_network()
        self._allocate(seed)
        self.setup_optimizer()

    # -- topology (wavenet.py:379-455) --------------------------------------------------------
    def create_network(self):
        p = self.params
        self.causal_conv_layers: List[DilatedConvolution1D] = []
        fw = p.causal_conv_filter_width
        chans = [p.quantization_steps] + list(p.causal_conv_channels)
        for i in range(len(chans) - 1):
            self.causal_conv_layers.append(DilatedConvolution1D(
                self, "causal_%d" % i, chans[i], chans[i + 1], (1, fw), filter_width=fw, dilation=1,
                nobias=p.causal_conv_no_bias))
        self.residual_blocks: List[List[ResidualConvLayer]] = []
        fw = p.residual_conv_filter_width
        n_in = p.causal_conv_channels[-1]
        n_skip = p.softmax_conv_channels[0]
        for blk in range(p.residual_num_blocks):
            layers = []
            for li, n_out in enumerate(p.residual_conv_channels):
                ksize = (1, fw) if li == 0 else (fw, 1)              # wavenet.py:418-424
                lay = ResidualConvLayer(self)
                pre = "residual_%d_block_%d_" % (blk, li)
                lay.wf = DilatedConvolution1D(self, pre + "wf", n_in, n_out, ksize, filter_width=fw,
                                              dilation=fw ** li, nobias=p.residual_conv_dilation_no_bias)
                lay.wg = DilatedConvolution1D(self, pre + "wg", n_in, n_out, ksize, filter_width=fw,
                                              dilation=fw ** li, nobias=p.residual_conv_dilation_no_bias)
                lay.projection_block = Convolution1x1(self, pre + "projection_block", n_out, n_in,
                                                      nobias=p.residual_conv_projection_no_bias)
                lay.projection_softmax = Convolution1x1(self, pre + "projection_softmax", n_out, n_skip,
                                                        nobias=p.residual_conv_projection_no_bias)
                layers.append(lay)
            self.residual_blocks.append(layers)
        self.softmax_conv_layers: List[Convolution1x1] = []
        sc = p.softmax_conv_channels
        for i in range(len(sc) - 1):
            self.softmax_conv_layers.append(Convolution1x1(self, "softmax_%d" % i, sc[i], sc[i + 1],
                                                           nobias=p.softmax_conv_no_bias))
        self._flat_layers = [lay for blk in self.residual_blocks for lay in blk]
        self._Cr, self._Cs = n_in, n_skip

    def links(self) -> List[_Link]:
        """All parameter holders in the reference's registration order (wavenet.py:461-472)."""
        out: List[_Link] = list(self.causal_conv_layers)
        for lay in self._flat_layers:
            out += [lay.wf, lay.wg, lay.projection_block, lay.projection_softmax]
        return out + list(self.softmax_conv_layers)

    # -- parameters: one flat arena (what the DP all-reduce and the optimiser kernel see) -----
    def _allocate(self, seed):
        rs = np.random.RandomState(seed) if seed is not None else np.random
        spans = []
        off = 0
        for ln in self.links():
            n = int(np.prod(ln.wshape))
            spans.append((ln, "W", off, n, ln.wshape))
            off += (n + 63) // 64 * 64                               # keep every tensor 256-byte aligned
            if ln.bshape is not None:
                spans.append((ln, "b", off, ln.bshape[0], ln.bshape))
                off += (ln.bshape[0] + 63) // 64 * 64
        self._spans = spans
        host = np.zeros((off,), dtype=np.float32)
        for ln, kind, o, n, shape in spans:
            if kind == "W":                                         # LeCunNormal like Chainer's default
                fan_in = shape[1] * shape[2] * shape[3]
                host[o:o + n] = (rs.standard_normal(n) / math.sqrt(fan_in)).astype(np.float32)
        self._arena = torch.from_numpy(host)
        self._grad_arena = torch.zeros_like(self._arena)
        self._bind()

    def _bind(self):
        self._any_requires_grad = True
        for ln, kind, o, n, shape in self._spans:
            t = self._arena[o:o + n].view(shape)
            t.requires_grad_(True)
            t.grad = self._grad_arena[o:o + n].view(shape)
            setattr(ln, kind, t)

    @property
    def receptive_field(self) -> int:
        """train_audio/train.py:36-38."""
        p = self.params
        return (p.residual_conv_filter_width ** len(p.residual_conv_channels) - 1) * p.residual_num_blocks + 1

    @property
    def input_width(self) -> int:
        """Receptive field plus one column per causal layer (train_audio/train.py:42-44)."""
        return self.receptive_field + len(self.params.causal_conv_channels)

    @property
    def num_parameters(self) -> int:
        return sum(n for _, _, _, n, _ in self._spans)

    def state_dict(self) -> Dict[str, np.ndarray]:
        """Reference checkpoint keys (``<link>/W``, ``<link>/b``) and shapes."""
        return {"%s/%s" % (ln.name, kind): self._arena[o:o + n].view(shape).detach().cpu().numpy().copy()
                for ln, kind, o, n, shape in self._spans}

    def load_state_dict(self, sd: Dict[str, np.ndarray]):
        with torch.no_grad():
            for ln, kind, o, n, shape in self._spans:
                key = "%s/%s" % (ln.name, kind)
                if key not in sd:
                    raise KeyError(key)
                a = np.asarray(sd[key], dtype=np.float32)
                if a.shape != tuple(shape):
                    raise Exception("shape of %s is %s, expected %s" % (key, a.shape, tuple(shape)))
                self._arena[o:o + n].copy_(torch.from_numpy(a.reshape(-1)))
        self._weights_changed()

    def _weights_changed(self):
        self._w16_stale = True

    def _exec(self, B: int = 8, T: int = 0):
        """WnExec for a library call on the current stream: this model's GEMM precision (``self.gemm_precision``, or the
        module default), its flags, and a scratch buffer owned by (model, stream), sized by ``wn_exec_workspace_bytes``
        for the largest (B, T) seen so far.  Buffers are never freed or moved once handed out -- a captured graph keeps
        the pointer -- a larger batch or a longer window gets a new, larger one."""
        lib = _lib.lib()
        key = (stream_ptr() or 0, self._arena.device.index)
        ent = self._scratch.get(key)
        if ent is None or ent[1] < B or ent[2] < T:
            p = self.params
            hc = int_array(p.softmax_conv_channels)
            B, T = max(B, 8, ent[1] if ent else 0), max(T, ent[2] if ent else 0)
            nbytes = lib.wn_exec_workspace_bytes(self._stack_desc(), p.quantization_steps, p.causal_conv_channels[0],
                                                 p.causal_conv_filter_width, hc, len(p.softmax_conv_channels), B, T)
            buf = torch.empty((nbytes,), device=self._arena.device, dtype=torch.uint8)
            self._scratch_keep.append(buf)
            ent = (buf, B, T)
            self._scratch[key] = ent
        ex = _lib.WnExec()
        ex.flags = _lib.default_exec_flags() if self.exec_flags is None else int(self.exec_flags)
        prec = self.gemm_precision or _lib.get_gemm_precision()
        ex.precision = _lib.GEMM_PRECISIONS.index("fp32" if ex.flags & _lib.WN_EXEC_FORCE_GENERIC else prec)
        ex.ws, ex.ws_bytes = ent[0].data_ptr(), ent[0].numel()
        ex.fwd_t1_min_blocks = (_lib.default_fwd_t1_min_blocks() if self.fwd_t1_min_blocks is None
                                else int(self.fwd_t1_min_blocks))
        ex.plan = self._plan if self._plan_on else None
        return C.byref(ex)

    # -- step plan (include/wavenet_hip.h, "step plan"): the weight-only preparation of a training step in two launches --------
    def plan_begin(self, nbytes: int = 16 << 20):
        """A NEW plan (its own device memory: a captured graph keeps pointers into it) in the RECORDING state: the next training
        step runs as ever and registers its weight-only preparation work.  Returns the plan object; the model's entry points carry
        it until :meth:`plan_off`."""
        self._plan_obj = _StepPlan(self._arena.device, nbytes)
        self._plan = self._plan_obj.handle
        check(_lib.lib().wn_plan_record(self._plan), "wn_plan_record")
        self._plan_on = True
        return self._plan_obj

    def plan_use(self, plan_obj):
        """Carry an existing plan again (the caller vouches that wn_plan_prepare opens every step that does)."""
        self._plan_obj, self._plan, self._plan_on = plan_obj, plan_obj.handle, True

    def plan_finish(self):
        check(_lib.lib().wn_plan_finish(self._plan, stream_ptr()), "wn_plan_finish")

    def plan_prepare(self, zero_grads: bool = True):
        """First call of a step that carries the READY plan: every weight image of the step, and (``zero_grads``) cleargrads."""
        n = self._grad_arena.numel() if zero_grads else 0
        n4 = n // 4 * 4
        check(_lib.lib().wn_plan_prepare(self._plan, ptr(self._grad_arena) if n4 else None, n4, stream_ptr()), "wn_plan_prepare")
        if n4 < n:
            self._grad_arena[n4:].zero_()

    def plan_off(self):
        """Entry points stop carrying the plan (its images go stale with the next weight update a prepare does not follow)."""
        self._plan_on = False

    def plan_stats(self):
        out = (C.c_int64 * 8)()
        check(_lib.lib().wn_plan_stats(self._plan, out), "wn_plan_stats")
        return dict(zip(("state", "weight_images", "layer_images", "plan_words", "device_bytes", "prepare_calls", "served", "not_served"),
                        [int(v) for v in out]))

    def _pack16_if_stale(self):
        """bf16 operand images of the current weights (one pack per optimiser step; captured with the step in a graph)."""
        if not self._w16_stale:
            return
        lib = _lib.lib()
        desc = self._stack_desc()
        if self._pack16 is None or self._pack16.device != self._arena.device:
            if not lib.wn16_supported(desc):
                lib.wn16_pack_stack(desc, None, None)                  # sets the error text
                raise _lib.WaveNetHipError("storage='bf16': %s" % lib.wn_last_error().decode())
            self._pack16 = torch.empty((lib.wn16_pack_elems(desc),), device=self._arena.device, dtype=torch.bfloat16)
            self._head16 = [(torch.empty(l.W.shape[:2], device=self._arena.device, dtype=torch.bfloat16),
                             torch.empty((l.W.shape[1], l.W.shape[0]), device=self._arena.device, dtype=torch.bfloat16))
                            for l in self.softmax_conv_layers]
        check(lib.wn16_pack_stack(desc, ptr(self._pack16), stream_ptr()), "wn16_pack_stack")
        for l, (wb, wbt) in zip(self.softmax_conv_layers, self._head16):
            check(lib.wn16_pack_pointwise(ptr(l.W), ptr(wb), ptr(wbt), l.W.shape[0], l.W.shape[1], stream_ptr()),
                  "wn16_pack_pointwise")
        self._w16_stale = False

    def _stack_desc(self):
        """WnStackDesc over the current arena (rebuilt when the arena moves, e.g. to_gpu)."""
        key = self._arena.data_ptr()
        if getattr(self, "_desc_key", None) != key:
            L = self._flat_layers
            keep = dict(
                cd=int_array([l.cd for l in L]), dil=int_array([l.dilation for l in L]),
                Wf=ptr_array([l.wf.W for l in L]), bf=ptr_array([l.wf.b for l in L]),
                Wg=ptr_array([l.wg.W for l in L]), bg=ptr_array([l.wg.b for l in L]),
                Wp=ptr_array([l.projection_block.W for l in L]), bp=ptr_array([l.projection_block.b for l in L]),
                Ws=ptr_array([l.projection_softmax.W for l in L]), bs=ptr_array([l.projection_softmax.b for l in L]))
            d = _lib.WnStackDesc()
            d.n_layers, d.Cr, d.Cs, d.fw = len(L), self._Cr, self._Cs, self.params.residual_conv_filter_width
            d.cd, d.dilation = keep["cd"], keep["dil"]
            for k in ("Wf", "bf", "Wg", "bg", "Wp", "bp", "Ws", "bs"):
                setattr(d, k, C.cast(keep[k], C.POINTER(C.c_void_p)))
            g = lambda t: None if t is None else t.grad
            gt = dict(
                wf=ptr_array([l.wf.W.grad for l in L]), bf=ptr_array([g(l.wf.b) for l in L]),
                wg=ptr_array([l.wg.W.grad for l in L]), bg=ptr_array([g(l.wg.b) for l in L]),
                wp=ptr_array([l.projection_block.W.grad for l in L]), bp=ptr_array([g(l.projection_block.b) for l in L]),
                ws=ptr_array([l.projection_softmax.W.grad for l in L]),
                bs=ptr_array([g(l.projection_softmax.b) for l in L]))
            self._desc_key, self._sdesc, self._desc_keep, self._gt = key, d, keep, gt
        return C.byref(self._sdesc)

    def _grad_tables(self):
        self._stack_desc()
        return self._gt

    # -- optimiser (wavenet.py:457-519) ---------------------------------------------------------
    def setup_optimizer(self):
        p = self.params
        name = str(p.optimizer).lower()
        if name == "adam":
            self.optimizer = AdamState(self, alpha=0.0001, beta1=p.momentum)     # wavenet.py:475, get_optimizer 81-84
        elif name == "eve":
            self.optimizer = EveState(self, alpha=0.0001, beta1=p.momentum)      # wavenet.py:85-86
        elif name in RuleState.RULES:
            self.optimizer = RuleState(self, name, lr=0.0001, momentum=p.momentum)   # wavenet.py:87-96
        else:
            raise Exception("unknown optimizer %r" % p.optimizer)                # wavenet.py:97

    def update_laerning_rate(self, lr):
        """wavenet.py:482-493: Adam/Eve take it as alpha, AdaDelta has no learning rate, everything else as lr."""
        opt = self.optimizer
        if isinstance(opt, RuleState):
            if opt.name != "adadelta":
                opt.lr = lr
            return
        opt.alpha = lr

    def update_momentum(self, momentum):
        """wavenet.py:495-513: beta1 / rho / momentum / alpha by optimizer; SGD and AdaGrad have none.  (For MomentumSGD the
        reference assigns a misspelt attribute, which changes nothing; here the momentum is really updated.)"""
        opt = self.optimizer
        if isinstance(opt, RuleState):
            if opt.name in ("adadelta", "nesterov", "nesterovag", "rmsprop", "momentumsgd"):
                opt.hyper = momentum
            return
        opt.beta1 = momentum

    def zero_grads(self):
        self._grad_arena.zero_()

    def backprop(self, loss):
        """cleargrads -> backward -> [DP all-reduce] -> WeightDecay, GradientClipping -> Adam."""
        self.zero_grads()
        if callable(loss):
            loss = loss()
        loss.backward()
        gmult = 1.0
        if self._dp_group is not None:
            gmult = self._dp_group.all_reduce_grads(self._grad_arena)
        if isinstance(self.optimizer, EveState):
            lv = float(loss.detach())                                    # Eve feeds the loss back (wavenet.py:73-79)
            if self._dp_group is not None:
                lv = self._dp_group.mean_loss(lv)                        # ... the global batch's, identical on every rank
            self.optimizer.update(gmult, loss=lv)
        else:
            self.optimizer.update(gmult)
        self._weights_changed()

    def last_update_applied(self) -> bool:
        """False when the most recent optimiser step was SKIPPED on the device because the global gradient norm was not
        finite (wn_adam_step, ABI 4: the weights and the optimiser state are untouched by such a step).  Reads one word back
        (a host synchronisation): for the training loop's occasional health check, not for every step.  A caller that sees
        False after a step of the multi-layer backward can set ``exec_flags |= WN_EXEC_NO_MULTI_LAYER_BWD`` and go on --
        the only thing in the library that produces a NaN on purpose is a dataflow wait of that launch that gave up
        (workgroups not co-resident: another process on the GPU, a CU mask).  Two things to know: the guard lives in the
        clipping hook, so with ``gradient_clipping <= 0`` no norm exists and a NaN gradient reaches m and v as in Chainer; and
        the host-side step counter ``t`` (Adam's bias correction) advances for a skipped step too -- un-advancing it would cost
        a host synchronisation per step; the effect is a step size off by the factor sqrt(1-b2^(t+1))/sqrt(1-b2^t) ~ 1.
        ``train_audio.train`` calls this every 100 updates and stops a run whose checks keep failing."""
        nrm = getattr(self.optimizer, "_norm", None)
        if nrm is None or not self.params.gradient_clipping or self.params.gradient_clipping <= 0:
            return True                                                  # no clipping hook, no norm: nothing is ever skipped
        return bool(torch.isfinite(nrm[0]).item())

    # -- device --------------------------------------------------------------------------------
    def to_gpu(self, device=None):
        if not torch.cuda.is_available():
            raise _lib.WaveNetHipError("to_gpu(): no HIP device is visible")
        _lib.lib()
        dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self._arena = self._arena.detach().to(dev)
        self._grad_arena = self._grad_arena.to(dev)
        self._bind()
        self._anchor = torch.zeros((1,), device=dev, requires_grad=True)
        self.optimizer.to(dev)
        self._gpu = True
        self._weights_changed()

    @property
    def gpu_enabled(self):
        return self._gpu and torch.cuda.is_available()

    @property
    def device(self):
        return self._arena.device

    # -- small tensor helpers (wavenet.py:531-554) ---------------------------------------------
    def slice_1d(self, x, cut=0):
        if cut < 1:
            raise Exception("CausalSlice1d: cut cannot be less than one.")      # wavenet.py:235-236
        return self.to_variable(x)[:, :, :, cut:]

    def padding_1d(self, x, pad=0):
        return torch.nn.functional.pad(self.to_variable(x), (pad, 0))

    def to_variable(self, x):
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(np.ascontiguousarray(x))
        if not isinstance(x, torch.Tensor):
            raise Exception("expected a numpy array or a torch tensor")
        if self._gpu and not x.is_cuda:
            x = x.to(self.device, non_blocking=True)
        return x

    def to_numpy(self, x):
        if isinstance(x, torch.Tensor):
            return x.detach().cpu().numpy()
        return x

    def get_batchsize(self, x):
        return x.shape[0]

    def _Z(self, T, d):
        fw = self.params.residual_conv_filter_width
        return zero_prefix(T, d, fw) if self.compat_zero_prefix else 0

    # -- forward (wavenet.py:556-593) -----------------------------------------------------------
    def forward_one_step(self, x_batch, apply_softmax=True, as_numpy=False):
        causal_output = self.forward_causal_block(x_batch)
        _, sum_skip_connections = self.forward_residual_block(causal_output)
        softmax_output = self.forward_softmax_block(sum_skip_connections, apply_softmax=apply_softmax)
        if as_numpy:
            return self.to_numpy(softmax_output)
        return softmax_output

    def forward_causal_block(self, x_batch):
        """Accepts the reference's one-hot image (B,Q,1,T) float32 *or* integer tokens (B,T): for
        tokens the first layer is a row gather (out = W[:,idx[t-1],0] + W[:,idx[t],1]) instead of a
        dense conv over 256 mostly-zero channels."""
        x = self.to_variable(x_batch)
        _need_gpu(x)
        layers = self.causal_conv_layers
        if self.storage == "bf16":
            if x.is_floating_point() or x.dim() != 2 or len(layers) != 1:
                raise Exception("storage='bf16' takes integer (B, T) tokens and one causal layer")
            l0 = layers[0]
            out = _Embed16Fn.apply(x.to(torch.int32).contiguous(), l0.W, l0.b, l0.filter_width, self)
            self._last_causal_outputs = [out]
            return _as_view(out)
        if not x.is_floating_point():
            if x.dim() != 2:
                raise Exception("integer input must be (B, T) tokens")
            l0 = layers[0]
            out = _EmbedFn.apply(x.to(torch.int32).contiguous(), l0.W, l0.b, l0.filter_width, self)
            start = 1
        else:
            out = _to_btc(x.to(torch.float32))
            start = 0
        outs = [out] if start == 1 else []
        for lay in layers[start:]:
            out = _ConvFn.apply(out, lay.W, lay.b, lay.filter_width, 1, 0, self._hook)
            outs.append(out)
        self._last_causal_outputs = outs                          # FasterWaveNet seeds its rings from these
        return _as_view(out)

    def forward_residual_block(self, x_batch, t_off: int = 0, window_only: bool = False):
        """(output, sum_skip_connections).  ``t_off`` > 0 (an extension) computes the skip sum for
        columns t_off.. only -- what train.py:73 keeps -- instead of slicing it afterwards.
        ``window_only`` (training, where train.py:72 discards the residual output): columns that cannot influence
        ``skip[t_off:]`` are not computed at all, so the returned residual output is UNDEFINED below the window's
        receptive field and must not be used; loss and gradients are unchanged."""
        x = self.to_variable(x_batch)
        _need_gpu(x)
        if self.storage == "bf16":
            self._pack16_if_stale()
            out, skip = _Stack16Fn.apply(_to_btc(x), self._anchor, self, int(t_off), torch.is_grad_enabled())
            return _as_view(out), _as_view(skip)
        out, skip = _StackFn.apply(_to_btc(x), self._anchor, self, int(t_off), torch.is_grad_enabled(),
                                   bool(window_only))
        return _as_view(out), _as_view(skip)

    def forward_softmax_block(self, x_batch, apply_softmax=True, activation: Optional[str] = None):
        x = self.to_variable(x_batch)
        _need_gpu(x)
        act = ACT[activation or self.head_activation]
        out = _to_btc(x)
        if self.storage == "bf16":
            self._pack16_if_stale()
            n = len(self.softmax_conv_layers)
            for i, (lay, (wb, wbt)) in enumerate(zip(self.softmax_conv_layers, self._head16)):
                out = _Pointwise16Fn.apply(out, lay.W, lay.b, wb, wbt, act, i == n - 1, self._hook)
            if apply_softmax:
                out = _SoftmaxFn.apply(out)
            return _as_view(out)
        for lay in self.softmax_conv_layers:
            out = _PointwiseFn.apply(out, lay.W, lay.b, act, self)
        if apply_softmax:
            out = _SoftmaxFn.apply(out)
        return _as_view(out)

    # -- loss (wavenet.py:597-617) ----------------------------------------------------------------
    def cross_entropy(self, raw_network_output, target_signal_data):
        if isinstance(target_signal_data, torch.Tensor) and target_signal_data.requires_grad:
            raise Exception("target_signal_data cannot be Variable")
        raw = self.to_variable(raw_network_output)
        _need_gpu(raw)
        n_norm = -1            # device-resident targets: the rows that count (label != -1) are counted on the device
        if not isinstance(target_signal_data, torch.Tensor):
            # a host array (what the reference requires): labels are checked here -- Chainer type-checks them too -- and
            # -1 is chainer's ignore_label: such rows carry no loss and do not count in the mean
            lab = np.asarray(target_signal_data)
            Q = raw.shape[1]
            if lab.size and (lab.min() < -1 or lab.max() >= Q):
                raise Exception("cross_entropy: labels must lie in [0, %d) or be -1 (ignored)" % Q)
            n_norm = max(int((lab != -1).sum()), 1)
        tgt = self.to_variable(np.asarray(target_signal_data) if not isinstance(target_signal_data, torch.Tensor)
                               else target_signal_data)
        if raw.shape[3] != tgt.shape[1]:
            raise Exception("raw_network_output.width != target.width")
        rows = _to_btc(raw)                                        # row b*T'+t, as after the reference's transpose
        B, Tw, Q = rows.shape
        return _XentFn.apply(rows.reshape(B * Tw, Q).contiguous(), tgt.to(torch.int32).reshape(-1).contiguous(), n_norm)

    def head_cross_entropy(self, sum_skip, target_signal_data):
        """``cross_entropy(forward_softmax_block(sum_skip, apply_softmax=False), target)`` (wavenet.py:584-617), with the LAST head
        convolution and the loss as one launch when the library covers it (fp32 storage, fp16x2 arithmetic, 256 quantisation
        steps: ``wn_head_xent``) -- the (B, Q, 1, T') logits then never reach memory (200 MB less traffic per step at config 2).
        Same loss, same gradients (the products are the skip path's fp16x2 split instead of the head's six-term split: 1e-6);
        anything the fused launch does not cover runs the two calls.  New capability: the reference has no such method; the
        captured training step (``TrainStepGraph`` / ``graph.default_loss``) uses it."""
        x = self.to_variable(sum_skip)
        _need_gpu(x)
        tensor_target = isinstance(target_signal_data, torch.Tensor)
        if tensor_target and target_signal_data.requires_grad:
            raise Exception("target_signal_data cannot be Variable")
        lay = self.softmax_conv_layers[-1]
        n_rows = int(x.shape[0]) * int(x.shape[-1])                 # (B, C, 1, T') -> B * T' rows of the loss
        fused = (self.storage != "bf16" and self.fuse_head_loss and
                 _lib.lib().wn_head_xent_supported(n_rows, lay.W.shape[1], lay.W.shape[0], self._exec()) == 1)
        if not fused:
            return self.cross_entropy(self.forward_softmax_block(x, apply_softmax=False), target_signal_data)
        act = ACT[self.head_activation]
        out = _to_btc(x)
        for l2 in self.softmax_conv_layers[:-1]:
            out = _PointwiseFn.apply(out, l2.W, l2.b, act, self)
        n_norm = -1
        if not tensor_target:
            lab = np.asarray(target_signal_data)
            Q = lay.W.shape[0]
            if lab.size and (lab.min() < -1 or lab.max() >= Q):
                raise Exception("cross_entropy: labels must lie in [0, %d) or be -1 (ignored)" % Q)
            n_norm = max(int((lab != -1).sum()), 1)
        tgt = self.to_variable(np.asarray(target_signal_data) if not tensor_target else target_signal_data)
        if out.shape[1] != tgt.shape[1]:
            raise Exception("raw_network_output.width != target.width")
        return _HeadXentFn.apply(out, lay.W, lay.b, tgt.to(torch.int32).reshape(-1).contiguous(), act, n_norm, self)

    # -- the deferred skip projection -----------------------------------------------------------
    def _skip_sum(self, zs: Sequence[torch.Tensor], skip: torch.Tensor, B, T, t_off, Tw):
        L = self._flat_layers
        check(_lib.lib().wn_skip_sum_fwd(
            len(L), ptr_array(zs), ptr_array([lay.projection_softmax.W for lay in L]),
            ptr_array([lay.projection_softmax.b for lay in L]), int_array([lay.cd for lay in L]), ptr(skip), B, T,
            t_off, Tw, self._Cs, 0, self._exec(B), stream_ptr()), "wn_skip_sum_fwd")

    # -- checkpoints (wavenet.py:619-639): reference key names; .npz for weights + optimizer, and the reference's own
    #    HDF5 container for the weights (read and written through h5py when importable, else through the HDF5 C library:
    #    hdf5_io.py) --
    def save(self, model_dir="./"):
        """When an HDF5 library is at hand, the weights as ``wavenet.model`` in the reference's own container -- the file its
        ``load`` opens (wavenet.py:627-633) -- and THEN ``wavenet.model.npz`` + ``wavenet.opt.npz``: written last, the .npz is
        the newer of the two weight files, so :meth:`load` takes it after every ``save`` and needs no HDF5 library for a
        checkpoint this package wrote."""
        os.makedirs(model_dir, exist_ok=True)
        from . import hdf5_io
        if hdf5_io.available():
            self.save_hdf5(os.path.join(model_dir, "wavenet.model"))
        np.savez(os.path.join(model_dir, "wavenet.model.npz"), **self.state_dict())
        np.savez(os.path.join(model_dir, "wavenet.opt.npz"), **self.optimizer.state_dict())

    def save_hdf5(self, filename):
        """The weights in the file format and layout of the reference's ``serializers.save_hdf5(model_dir +
        "/wavenet.model", self.chain)`` (wavenet.py:619-625): one group per link, datasets ``W`` and ``b``, gzip 4 --
        a file the reference's ``load`` (wavenet.py:627-639) reads."""
        from . import hdf5_io
        hdf5_io.write_datasets(filename, self.state_dict(), compression=4)

    def load_hdf5(self, filename):
        """Weights saved by the reference itself (``serializers.save_hdf5(model_dir + "/wavenet.model", self.chain)``,
        wavenet.py:619-625): Chainer writes one dataset per parameter at ``<link name>/W`` and ``<link name>/b`` -- the
        key names of :meth:`state_dict`.  Read with h5py when it is importable, otherwise with the HDF5 C library through
        ``hdf5_io`` (ImportError, naming what was looked for, when neither exists).  A file that lacks one of the model's
        parameters, or holds it in another shape, is refused (KeyError / Exception from :meth:`load_state_dict`) -- the
        reference's deserializer fails on such a file too."""
        try:
            import h5py
        except ImportError:
            h5py = None
        if h5py is not None:
            sd = {}
            with h5py.File(filename, "r") as f:
                def visit(name, obj):
                    if isinstance(obj, h5py.Dataset):
                        sd[name] = np.asarray(obj)
                f.visititems(visit)
        else:
            from . import hdf5_io
            sd = hdf5_io.read_datasets(filename)
        self.load_state_dict({k: v for k, v in sd.items() if k in self.state_dict()})

    def load(self, model_dir="./"):
        """wavenet.py:627-639.  Two weight files may sit in the directory: the reference's own HDF5 ``wavenet.model`` and this
        package's ``wavenet.model.npz``.  The HDF5 file is loaded when it is the only one or the NEWER one (modification time:
        a reference-written checkpoint dropped next to an older .npz must not be ignored silently -- that case is announced);
        :meth:`save` writes the .npz last, so a directory this package saved loads from the .npz without a word and without an
        HDF5 library.  If the HDF5 file is newer but no HDF5 library can be found, the .npz next to it is loaded with a
        warning instead of failing."""
        h5 = os.path.join(model_dir, "wavenet.model")
        nz = os.path.join(model_dir, "wavenet.model.npz")
        have_h5, have_nz = os.path.isfile(h5), os.path.isfile(nz)
        use_h5 = have_h5 and (not have_nz or os.path.getmtime(h5) > os.path.getmtime(nz))
        if use_h5 and have_nz:
            print("both %s and %s exist: loading the newer one (%s)" % (h5, nz, h5))
        if use_h5:
            print("loading", h5, "...")
            try:
                self.load_hdf5(h5)
            except ImportError as e:
                if not have_nz:
                    raise
                print("cannot read %s (%s): loading the OLDER %s instead" % (h5, e, nz))
                use_h5 = False
        if not use_h5 and have_nz:                                              # silently skipped when absent, like the reference
            print("loading", nz, "...")
            with np.load(nz) as z:
                self.load_state_dict({k: z[k] for k in z.files})
        fn = os.path.join(model_dir, "wavenet.opt.npz")
        if os.path.isfile(fn):
            print("loading", fn, "...")
            with np.load(fn) as z:
                self.optimizer.load_state_dict({k: z[k] for k in z.files})

    # -- data parallel (new capability) -----------------------------------
    def enable_data_parallel(self, group=None, always_reduce: bool = False):
        from .dp import DataParallel
        self._dp_group = DataParallel(self, group, always_reduce=always_reduce)
        return self._dp_group


class AdamState(object):
    """Chainer's Adam on the flat arena, with the reference's hook order (WeightDecay, then
    GradientClipping; wavenet.py:477-480), as two kernel launches per step."""

    def __init__(self, net: WaveNet, alpha=0.001, beta1=0.9, beta2=0.999, eps=1e-8):
        self.net = net
        self.alpha, self.beta1, self.beta2, self.eps = alpha, beta1, beta2, eps
        self.t = 0
        self.m = torch.zeros_like(net._arena)
        self.v = torch.zeros_like(net._arena)
        self._norm = torch.zeros((_lib.SQNORM_WORDS,), dtype=torch.float32)

    def to(self, dev):
        self.m, self.v, self._norm = self.m.to(dev), self.v.to(dev), self._norm.to(dev)

    @property
    def lr(self):
        fix1 = 1.0 - self.beta1 ** self.t
        fix2 = 1.0 - self.beta2 ** self.t
        return self.alpha * math.sqrt(fix2) / fix1

    def update(self, grad_mult: float = 1.0, lr_dev=None):
        """One optimiser step.  ``lr_dev`` (1-element device tensor) makes the kernel read the bias-corrected
        step size from device memory instead of taking ``self.lr`` by value (graph.TrainStepGraph)."""
        net = self.net
        _need_gpu(net._arena)
        self.t += 1
        p = net.params
        lib, st = _lib.lib(), stream_ptr()
        n = net._arena.numel()
        clip = float(p.gradient_clipping) if p.gradient_clipping and p.gradient_clipping > 0 else 0.0
        wd = float(p.weight_decay) if p.weight_decay and p.weight_decay > 0 else 0.0
        norm_ptr = None
        if clip > 0:
            check(lib.wn_sqnorm(ptr(net._grad_arena), ptr(net._arena), n, grad_mult, wd, ptr(self._norm), st),
                  "wn_sqnorm")
            norm_ptr = ptr(self._norm)
        if lr_dev is not None:
            check(lib.wn_adam_step_dev(ptr(net._arena), ptr(net._grad_arena), ptr(self.m), ptr(self.v), n, ptr(lr_dev),
                                       self.beta1, self.beta2, self.eps, wd, norm_ptr, clip, grad_mult, st),
                  "wn_adam_step_dev")
            return
        check(lib.wn_adam_step(ptr(net._arena), ptr(net._grad_arena), ptr(self.m), ptr(self.v), n, self.lr,
                               self.beta1, self.beta2, self.eps, wd, norm_ptr, clip, grad_mult, st), "wn_adam_step")

    def state_dict(self):
        return {"t": np.array(self.t), "alpha": np.array(self.alpha), "m": self.m.cpu().numpy(),
                "v": self.v.cpu().numpy()}

    def _hooks(self, grad_mult):
        """WeightDecay then GradientClipping (wavenet.py:477-480): returns (norm pointer or None, clip, wd)."""
        net = self.net
        p = net.params
        clip = float(p.gradient_clipping) if p.gradient_clipping and p.gradient_clipping > 0 else 0.0
        wd = float(p.weight_decay) if p.weight_decay and p.weight_decay > 0 else 0.0
        norm_ptr = None
        if clip > 0:
            check(_lib.lib().wn_sqnorm(ptr(net._grad_arena), ptr(net._arena), net._arena.numel(), grad_mult, wd,
                                       ptr(self._norm), stream_ptr()), "wn_sqnorm")
            norm_ptr = ptr(self._norm)
        return norm_ptr, clip, wd

    def load_state_dict(self, sd):
        self.t = int(sd["t"])
        self.alpha = float(sd["alpha"])
        self.m.copy_(torch.from_numpy(np.asarray(sd["m"])))
        self.v.copy_(torch.from_numpy(np.asarray(sd["v"])))


class EveState(AdamState):
    """The reference's Eve optimizer (wavenet.py:10-79; Koushik & Hayashi 2016) on the flat arena: Adam's moments, the
    update divided by ``d * sqrt(v) + eps`` where the scalar ``d`` tracks the relative change of the (clamped) loss.
    Every parameter of the reference carries its own copy of (d, f) and updates it with the same loss at the same t
    (wavenet.py:27-44 is called per parameter), so they are one pair of host scalars here."""

    def __init__(self, net, alpha=0.001, beta1=0.9, beta2=0.999, beta3=0.999, eps=1e-8, lower_threshold=0.1,
                 upper_threshold=10):
        super().__init__(net, alpha, beta1, beta2, eps)
        self.beta3, self.lower_threshold, self.upper_threshold = beta3, lower_threshold, upper_threshold
        self.d, self.f = 1.0, 0.0                                    # wavenet.py:25-26

    def _update_d_and_f(self, loss: float):
        d32, f32 = np.float32(self.d), np.float32(self.f)            # the reference keeps both in float32 arrays
        if self.t > 1:
            old_f = float(f32)
            if loss > old_f:
                delta, Delta = self.lower_threshold + 1.0, self.upper_threshold + 1.0
            else:
                delta, Delta = 1.0 / (self.upper_threshold + 1.0), 1.0 / (self.lower_threshold + 1.0)
            c = min(max(delta, loss / (old_f + 1e-12)), Delta)
            new_f = c * old_f
            r = abs(new_f - old_f) / (min(new_f, old_f) + 1e-12)
            d32 = np.float32(d32 + np.float32((1 - self.beta3) * (r - float(d32))))
            f32 = np.float32(new_f)
        else:
            f32 = np.float32(loss)
        self.d, self.f = float(d32), float(f32)

    def update(self, grad_mult: float = 1.0, loss=None, lr_dev=None):
        if loss is None:
            raise RuntimeError("Eve.update requires the loss value")            # wavenet.py:75-76
        if lr_dev is not None:
            raise RuntimeError("Eve needs the loss on the host every step: it cannot run inside a replayed graph")
        net = self.net
        _need_gpu(net._arena)
        self.t += 1
        self._update_d_and_f(float(loss))
        norm_ptr, clip, wd = self._hooks(grad_mult)
        check(_lib.lib().wn_eve_step(ptr(net._arena), ptr(net._grad_arena), ptr(self.m), ptr(self.v), net._arena.numel(),
                                     self.lr, self.beta1, self.beta2, self.eps, self.d, wd, norm_ptr, clip, grad_mult,
                                     stream_ptr()), "wn_eve_step")

    def state_dict(self):
        sd = super().state_dict()
        sd.update(d=np.array(self.d), f=np.array(self.f))
        return sd

    def load_state_dict(self, sd):
        super().load_state_dict(sd)
        self.d, self.f = float(sd["d"]), float(sd["f"])


class RuleState(object):
    """The other optimizers ``get_optimizer`` names (wavenet.py:87-96): Chainer's SGD, MomentumSGD, AdaGrad, AdaDelta,
    NesterovAG and RMSprop on the flat arena, behind the same hooks as Adam (``wn_rule_step``).  Constructor arguments
    follow get_optimizer: ``lr`` where the rule has one, the model's ``momentum`` as momentum / rho (AdaDelta) / alpha
    (RMSprop).  The reference's "momentumsgd" line misspells two names and raises NameError before it can train; the
    rule itself is implemented here."""

    #            rule id, uses momentum hyper, eps (Chainer defaults)
    RULES = {"sgd": (0, False, 0.0), "momentumsgd": (1, True, 0.0), "adagrad": (2, False, 1e-8),
             "adadelta": (3, True, 1e-6), "nesterov": (4, True, 0.0), "nesterovag": (4, True, 0.0),
             "rmsprop": (5, True, 1e-8)}

    def __init__(self, net, name: str, lr=0.0001, momentum=0.9):
        self.net, self.name = net, name.lower()
        self.rule, uses_hyper, self.eps = self.RULES[self.name]
        self.lr = lr
        self.hyper = momentum if uses_hyper else 0.0
        self.t = 0
        # m = first state array (v / h / ms / msg), v = second (AdaDelta's msdx): the names TrainStepGraph snapshots
        self.m = torch.zeros_like(net._arena) if self.rule != 0 else torch.zeros((1,), dtype=torch.float32)
        self.v = torch.zeros_like(net._arena) if self.rule == 3 else torch.zeros((1,), dtype=torch.float32)
        self._norm = torch.zeros((_lib.SQNORM_WORDS,), dtype=torch.float32)

    def to(self, dev):
        self.m, self.v, self._norm = self.m.to(dev), self.v.to(dev), self._norm.to(dev)

    _hooks = AdamState._hooks

    def update(self, grad_mult: float = 1.0, lr_dev=None):
        net = self.net
        _need_gpu(net._arena)
        self.t += 1
        norm_ptr, clip, wd = self._hooks(grad_mult)
        check(_lib.lib().wn_rule_step(self.rule, ptr(net._arena), ptr(net._grad_arena),
                                      ptr(self.m) if self.rule != 0 else None, ptr(self.v) if self.rule == 3 else None,
                                      net._arena.numel(), self.lr, ptr(lr_dev) if lr_dev is not None else None,
                                      self.hyper, self.eps, wd, norm_ptr, clip, grad_mult, stream_ptr()), "wn_rule_step")

    def state_dict(self):
        return {"t": np.array(self.t), "lr": np.array(self.lr), "hyper": np.array(self.hyper),
                "m": self.m.cpu().numpy(), "v": self.v.cpu().numpy()}

    def load_state_dict(self, sd):
        self.t, self.lr, self.hyper = int(sd["t"]), float(sd["lr"]), float(sd["hyper"])
        self.m.copy_(torch.from_numpy(np.asarray(sd["m"])))
        self.v.copy_(torch.from_numpy(np.asarray(sd["v"])))
