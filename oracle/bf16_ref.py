"""CPU restatement of the bf16-storage arithmetic (BASELINE config 5) -- TEST INFRASTRUCTURE ONLY (oracle/__init__.py).

Same operation sequence as oracle/wavenet_ref.py's closed form (wavenet.py:358-368, 556-617 and Chainer's backward over
it), with a bfloat16 rounding (``wavenet_ref.bf16_round``: nearest even, bit-identical to torch's conversion) at every
point where the MI355X path stores an activation or a gradient in HBM or hands an operand to a matrix core, and wide
(float64) accumulation everywhere else.  Two formulations that must agree (tests/test_oracle.py):

* :func:`train_step` -- numpy, forward and a hand-written backward, every intermediate returned (what the GPU tests
  compare the kernels' buffers with);
* :func:`train_step_autograd` -- torch autograd over the same forward with straight-through rounding
  (``x + (bf16(x) - x).detach()``) and gradient rounding nodes at the same places.

Rounding points (forward): embedding output; every layer's z and output; the skip sum; weights as contraction operands.
Backward: d loss / d logits, dskip, every layer's dz_skip, [da | dg] and dx.  tanh / sigmoid, the softmax and all sums are
not rounded.  Scope: what the bf16 kernels cover (filter width 2, no conv / projection biases, ReLU head, one causal layer).
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from .wavenet_ref import bf16_round, conv_pad_and_prefix, input_width  # noqa: F401


def rb(a: np.ndarray) -> np.ndarray:
    return bf16_round(np.asarray(a, dtype=np.float32)).astype(np.float64)


def _sig(x):
    return 0.5 * np.tanh(0.5 * x) + 0.5


def _layers(p):
    fw = p["residual_conv_filter_width"]
    assert fw == 2 and p["causal_conv_filter_width"] == 2 and len(p["causal_conv_channels"]) == 1
    for blk in range(p["residual_num_blocks"]):
        for li in range(len(p["residual_conv_channels"])):
            yield "residual_%d_block_%d_" % (blk, li), fw ** li


def _Z(T, d, compat):
    return conv_pad_and_prefix(T, d, 2)[1] if compat else 0


def _shift(x, d):
    """x[b, t - d] with zeros before the clip start; x is (B, T, C)."""
    if d == 0:
        return x
    out = np.zeros_like(x)
    if d < x.shape[1]:
        out[:, d:] = x[:, :-d]
    return out


def _unshift(x, d):
    """x[b, t + d] with zeros beyond the clip end."""
    out = np.zeros_like(x)
    if d < x.shape[1]:
        out[:, :-d] = x[:, d:]
    return out


def layer_fwd(x, Wf, Wg, Wp, d, Z):
    """One residual layer (wavenet.py:358-368) on a stored (bf16-valued) input x (B, T, C): (z, out), both rounded as stored.
    Weights in the reference's shapes, fp32."""
    T, C = x.shape[1], x.shape[2]
    Wf, Wg, Wp = rb(Wf).reshape(-1, C, 2), rb(Wg).reshape(-1, C, 2), rb(Wp)[:, :, 0, 0]
    xo = _shift(x, d)
    live = (np.arange(T) >= Z)[None, :, None]
    a = (xo @ Wf[:, :, 0].T + x @ Wf[:, :, 1].T) * live
    g = (xo @ Wg[:, :, 0].T + x @ Wg[:, :, 1].T) * live
    z = rb(np.tanh(a) * _sig(g))
    return z, rb(x + z @ Wp.T)


def train_step(p: dict, weights: Dict[str, np.ndarray], idx: np.ndarray, target: np.ndarray, compat_zero_prefix=True,
               keep: dict = None, skip_override: np.ndarray = None):
    """(loss, logits (B,Tw,Q), grads {name: array}).  ``keep`` receives the intermediates, all (B, T, C) float64 holding
    bf16-representable values where the device stores bf16: x0, per layer out / z / dzs / dadg / dx, skip, dskip.
    ``skip_override`` (B, Tw, Cs): the head and everything behind it (loss, every gradient) are evaluated on this skip
    sum instead of the one computed here -- the device's own, in the GPU tests: the ReLU in front of the head turns a
    last-bit difference of a skip value near 0 into a gradient element that is wholly present or absent, which no
    tolerance on a small batch can absorb; with the same skip sum on both sides the comparison is about the kernels."""
    B, T = idx.shape
    Tw = target.shape[1]
    t_off = T - Tw
    Q = p["quantization_steps"]
    w = {k: np.asarray(v, np.float64) for k, v in weights.items()}
    wr = {k: rb(v) for k, v in weights.items()}
    # ---- forward ----
    We = w["causal_0/W"][:, :, 0, :]                       # (C, Q, 2): the embedding is a lookup of fp32 weights
    x0 = We[:, idx, 1].transpose(1, 2, 0).copy()           # (B, T, C)
    x0[:, 1:] += We[:, idx[:, :-1], 0].transpose(1, 2, 0)
    if "causal_0/b" in w:
        x0 += w["causal_0/b"]
    x0 = rb(x0)
    xs, zs, fs, gs, lay = [x0], [], [], [], list(_layers(p))
    tt = np.arange(T)
    for pre, d in lay:
        x = xs[-1]
        Wf, Wg = wr[pre + "wf/W"].reshape(-1, x.shape[2], 2), wr[pre + "wg/W"].reshape(-1, x.shape[2], 2)
        xo = _shift(x, d)
        live = (tt >= _Z(T, d, compat_zero_prefix))[None, :, None]
        a = (xo @ Wf[:, :, 0].T + x @ Wf[:, :, 1].T) * live
        g = (xo @ Wg[:, :, 0].T + x @ Wg[:, :, 1].T) * live
        f, s = np.tanh(a), _sig(g)
        z = rb(f * s)
        out = rb(x + z @ wr[pre + "projection_block/W"][:, :, 0, 0].T)
        fs.append(f); gs.append(s); zs.append(z); xs.append(out)
    skip = np.zeros((B, Tw, p["softmax_conv_channels"][0]))
    for (pre, d), z in zip(lay, zs):
        skip += z[:, t_off:] @ wr[pre + "projection_softmax/W"][:, :, 0, 0].T
    skip = rb(skip)
    skip_own = skip
    if skip_override is not None:
        skip = rb(skip_override)
    nh = len(p["softmax_conv_channels"]) - 1
    hs = [skip]
    for i in range(nh):
        h = np.maximum(hs[-1], 0) @ wr["softmax_%d/W" % i][:, :, 0, 0].T
        if "softmax_%d/b" % i in w:
            h = h + w["softmax_%d/b" % i]
        hs.append(h if i == nh - 1 else rb(h))
    logits = hs[-1]
    m = logits.max(axis=2, keepdims=True)
    e = np.exp(logits - m)
    sm = e / e.sum(axis=2, keepdims=True)
    N = B * Tw
    bi, ti = np.meshgrid(np.arange(B), np.arange(Tw), indexing="ij")
    loss = float(-np.log(sm[bi, ti, target]).mean())
    # ---- backward ----
    grads = {k: np.zeros(v.shape, np.float64) for k, v in weights.items()}
    dl = sm.copy()
    dl[bi, ti, target] -= 1.0
    dl /= N
    g_f32 = dl.astype(np.float32).astype(np.float64)       # the loss kernel writes fp32
    dh = g_f32
    for i in reversed(range(nh)):
        dh_r = rb(dh)
        hin = hs[i]
        grads["softmax_%d/W" % i][:, :, 0, 0] = np.einsum("btq,btc->qc", dh_r, np.maximum(hin, 0))
        if "softmax_%d/b" % i in w:
            grads["softmax_%d/b" % i] = dh.sum(axis=(0, 1))
        dh = rb((dh_r @ wr["softmax_%d/W" % i][:, :, 0, 0]) * (hin > 0))
    dskip = dh
    Cr = x0.shape[2]
    dout = np.zeros((B, T, Cr))
    have_dout = False
    kk = dict(x0=x0, xs=xs[1:], zs=zs, skip=skip_own, dskip=dskip, dzs=[None] * len(lay), dadg=[None] * len(lay),
              dx=[None] * len(lay))
    for l in reversed(range(len(lay))):
        pre, d = lay[l]
        x, z, f, s = xs[l], zs[l], fs[l], gs[l]
        Ws = wr[pre + "projection_softmax/W"][:, :, 0, 0]
        Wp = wr[pre + "projection_block/W"][:, :, 0, 0]
        Wf, Wg = wr[pre + "wf/W"].reshape(-1, Cr, 2), wr[pre + "wg/W"].reshape(-1, Cr, 2)
        grads[pre + "projection_softmax/W"][:, :, 0, 0] = np.einsum("bts,btc->sc", dskip, z[:, t_off:])
        dzs = rb(dskip @ Ws)
        dz = np.zeros_like(z)
        dz[:, t_off:] = dzs
        if have_dout:
            dz += dout @ Wp
            grads[pre + "projection_block/W"][:, :, 0, 0] = np.einsum("bto,btc->oc", dout, z)
        live = (tt >= _Z(T, d, compat_zero_prefix))[None, :, None]
        da = rb(dz * s * (1 - f * f) * live)
        dg = rb(dz * f * s * (1 - s) * live)
        xo = _shift(x, d)
        gWf = np.stack([np.einsum("bto,btc->oc", da, xo), np.einsum("bto,btc->oc", da, x)], axis=2)
        gWg = np.stack([np.einsum("bto,btc->oc", dg, xo), np.einsum("bto,btc->oc", dg, x)], axis=2)
        grads[pre + "wf/W"] = gWf.reshape(weights[pre + "wf/W"].shape)
        grads[pre + "wg/W"] = gWg.reshape(weights[pre + "wg/W"].shape)
        dx = dout + da @ Wf[:, :, 1] + dg @ Wg[:, :, 1] + _unshift(da, d) @ Wf[:, :, 0] + _unshift(dg, d) @ Wg[:, :, 0]
        dx = rb(dx)
        kk["dzs"][l], kk["dadg"][l], kk["dx"][l] = dzs, np.concatenate([da, dg], axis=2), dx
        dout, have_dout = dx, True
    gWe = np.zeros((Cr, Q, 2))
    np.add.at(gWe[:, :, 1].T, idx.reshape(-1), dout.reshape(-1, Cr))
    np.add.at(gWe[:, :, 0].T, idx[:, :-1].reshape(-1), dout[:, 1:].reshape(-1, Cr))
    grads["causal_0/W"] = gWe.reshape(weights["causal_0/W"].shape)
    if "causal_0/b" in w:
        grads["causal_0/b"] = dout.sum(axis=(0, 1))
    if keep is not None:
        keep.update(kk)
    return loss, logits, grads


# ----------------------------------------------------------------------------------------------------------------------
# the same arithmetic through torch autograd (cross-check of the hand-written backward)
# ----------------------------------------------------------------------------------------------------------------------
def _rq(t: torch.Tensor) -> torch.Tensor:
    """forward: round to bf16; backward: identity (straight through)"""
    return t + (t.detach().to(torch.float32).to(torch.bfloat16).to(t.dtype) - t.detach())


class _GradRound(torch.autograd.Function):
    """forward: identity; backward: round the gradient to bf16 (a gradient tensor the device stores in bf16)"""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.float32).to(torch.bfloat16).to(g.dtype)


def train_step_autograd(p, weights, idx, target, compat_zero_prefix=True):
    B, T = idx.shape
    Tw = target.shape[1]
    t_off = T - Tw
    W = {k: torch.tensor(np.asarray(v, np.float64), requires_grad=True) for k, v in weights.items()}
    Wr = {k: _rq(v) for k, v in W.items()}
    ti = torch.as_tensor(idx.astype(np.int64))
    We = W["causal_0/W"][:, :, 0, :]
    x = We[:, ti, 1].permute(1, 2, 0)
    x = torch.cat([x[:, :1], x[:, 1:] + We[:, ti[:, :-1], 0].permute(1, 2, 0)], dim=1)
    if "causal_0/b" in W:
        x = x + W["causal_0/b"]
    x = _GradRound.apply(_rq(x))
    tt = torch.arange(T)
    skip = 0
    for pre, d in _layers(p):
        Cr = x.shape[2]
        Wf, Wg = Wr[pre + "wf/W"].reshape(-1, Cr, 2), Wr[pre + "wg/W"].reshape(-1, Cr, 2)
        xo = torch.cat([torch.zeros_like(x[:, :d]), x[:, :-d]], dim=1) if d < T else torch.zeros_like(x)
        live = (tt >= _Z(T, d, compat_zero_prefix))[None, :, None].to(x.dtype)
        a = (xo @ Wf[:, :, 0].T + x @ Wf[:, :, 1].T) * live
        g = (xo @ Wg[:, :, 0].T + x @ Wg[:, :, 1].T) * live
        # [da | dg] are stored in bf16: round the gradient that reaches the pre-activations
        a, g = _GradRound.apply(a), _GradRound.apply(g)
        z = _rq(torch.tanh(a) * torch.sigmoid(g))
        zs = _GradRound.apply(z[:, t_off:])                # dz_skip is stored in bf16
        skip = skip + zs @ Wr[pre + "projection_softmax/W"][:, :, 0, 0].T
        x = _GradRound.apply(_rq(x + z @ Wr[pre + "projection_block/W"][:, :, 0, 0].T))
    h = _GradRound.apply(_rq(skip))
    nh = len(p["softmax_conv_channels"]) - 1
    for i in range(nh):
        h = torch.relu(h) @ Wr["softmax_%d/W" % i][:, :, 0, 0].T
        if "softmax_%d/b" % i in W:
            # the bias gradient is summed from the fp32 gradient, the matrix products see its bf16 rounding
            h = _GradRound.apply(h) + W["softmax_%d/b" % i]
        else:
            h = _GradRound.apply(h)
        if i < nh - 1:
            h = _rq(h)
    logits = h
    loss = torch.nn.functional.cross_entropy(logits.reshape(B * Tw, -1), torch.as_tensor(target.astype(np.int64)).reshape(-1))
    loss.backward()
    grads = {k: (v.grad.numpy() if v.grad is not None else np.zeros(v.shape)) for k, v in W.items()}
    return float(loss.detach()), logits.detach().numpy(), grads
