"""CPU restatement of the reference forward / loss / fast-generation path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED against
Chainer itself; every function cites the reference lines it follows.

Two formulations are kept on purpose so they can cross-check each other:

* ``*_literal``  -- the op sequence the reference executes (torch-CPU tensors,
  ``F.pad`` / ``reshape`` / ``F.conv2d``), including every intermediate copy.
  This is also what ``bench.py`` times as the "Chainer-equivalent restatement".
* ``*_closed``   -- the closed form the trick reduces to (numpy einsum), which
  is what the HIP kernels implement.

Chainer behaviours relied on (documented behaviour of the third-party
dependency, README.md:19 "Chainer 2", not in the tree): Convolution2D is a
cross-correlation ``y[b,o,i,j] = sum W[o,c,u,v] x[b,c,i+u,j+v] + b[o]``;
``F.sigmoid(x) = tanh(x/2)/2 + 1/2``; ``F.elu`` has alpha 1; ``F.softmax`` is
over axis 1 with max subtraction; ``F.softmax_cross_entropy`` is the mean over
rows of ``-log softmax(x)[target]``; initial W ~ N(0, 1/fan_in), bias 0.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# hyper-parameters (field names of wavenet.py:100-146)
# --------------------------------------------------------------------------

DEFAULTS = dict(
    quantization_steps=256,
    sampling_rate=8000,
    causal_conv_no_bias=True,
    causal_conv_filter_width=2,
    causal_conv_channels=[128],
    residual_conv_dilation_no_bias=True,
    residual_conv_projection_no_bias=True,
    residual_conv_filter_width=2,
    residual_conv_channels=[32] * 9,
    residual_num_blocks=2,
    softmax_conv_no_bias=False,
    softmax_conv_channels=[128, 256],
)


def make_params(**over) -> dict:
    p = {k: (list(v) if isinstance(v, list) else v) for k, v in DEFAULTS.items()}
    for k, v in over.items():
        if k not in p:
            raise KeyError(k)
        p[k] = v
    return p


def receptive_field(p: dict) -> int:
    """train_audio/train.py:36-38."""
    per_unit = p["residual_conv_filter_width"] ** len(p["residual_conv_channels"])
    return (per_unit - 1) * p["residual_num_blocks"] + 1


def input_width(p: dict) -> int:
    """train_audio/train.py:42-44."""
    return receptive_field(p) + len(p["causal_conv_channels"])


# --------------------------------------------------------------------------
# parameter creation (wavenet.py:379-455, names wavenet.py:461-472)
# --------------------------------------------------------------------------

def weight_specs(p: dict) -> List[Tuple[str, Tuple[int, ...], Optional[Tuple[int]]]]:
    """[(link name, W shape, b shape or None)] in the reference's creation order."""
    specs = []
    fw = p["causal_conv_filter_width"]
    chans = [p["quantization_steps"]] + list(p["causal_conv_channels"])
    for i in range(len(chans) - 1):                       # wavenet.py:388-396
        cin, cout = chans[i], chans[i + 1]
        specs.append(("causal_%d" % i, (cout, cin, 1, fw),
                      None if p["causal_conv_no_bias"] else (cout,)))
    fw = p["residual_conv_filter_width"]
    cr = p["causal_conv_channels"][-1]                    # wavenet.py:403
    cs = p["softmax_conv_channels"][0]                    # wavenet.py:404
    for blk in range(p["residual_num_blocks"]):           # wavenet.py:412-444
        for li, cd in enumerate(p["residual_conv_channels"]):
            shape = (cd, cr, 1, fw) if li == 0 else (cd, cr, fw, 1)   # wavenet.py:418-424
            bd = None if p["residual_conv_dilation_no_bias"] else (cd,)
            pre = "residual_%d_block_%d_" % (blk, li)
            specs.append((pre + "wf", shape, bd))
            specs.append((pre + "wg", shape, bd))
            nb = p["residual_conv_projection_no_bias"]
            specs.append((pre + "projection_block", (cr, cd, 1, 1), None if nb else (cr,)))
            specs.append((pre + "projection_softmax", (cs, cd, 1, 1), None if nb else (cs,)))
    sc = p["softmax_conv_channels"]
    for i in range(len(sc) - 1):                          # wavenet.py:451-455
        specs.append(("softmax_%d" % i, (sc[i + 1], sc[i], 1, 1),
                      None if p["softmax_conv_no_bias"] else (sc[i + 1],)))
    return specs


def init_weights(p: dict, seed: int = 1234, bias_scale: float = 0.0) -> Dict[str, np.ndarray]:
    """N(0, 1/fan_in) weights in creation order from ``RandomState(seed)``.

    ``bias_scale`` > 0 draws non-zero biases (Chainer initialises them to 0; a
    non-zero value is used by tests so that bias handling is actually observable).
    """
    rs = np.random.RandomState(seed)
    sd = {}
    for name, ws, bs in weight_specs(p):
        fan_in = ws[1] * ws[2] * ws[3]
        sd[name + "/W"] = (rs.standard_normal(ws) / math.sqrt(fan_in)).astype(np.float32)
        if bs is not None:
            if bias_scale > 0:
                sd[name + "/b"] = (rs.standard_normal(bs) * bias_scale).astype(np.float32)
            else:
                sd[name + "/b"] = np.zeros(bs, dtype=np.float32)
    return sd


# --------------------------------------------------------------------------
# dilated causal convolution
# --------------------------------------------------------------------------

def conv_pad_and_prefix(T: int, d: int, fw: int) -> Tuple[int, int]:
    """(pad, Z) of wavenet.py:303-340 for a d>1 layer; (fw-1, 0) for d==1."""
    if d == 1:
        return fw - 1, 0
    pad = (-T) % d                                       # wavenet.py:308-311
    height = (T + pad) // d                              # wavenet.py:314 (exact)
    if height < fw:                                      # wavenet.py:315-317
        pad += (fw - height) * d
    return pad, max(0, (fw - 1) * d - pad)               # cut<0 -> left pad of -cut zeros


def dilated_conv_literal(x: torch.Tensor, W: torch.Tensor, b: Optional[torch.Tensor],
                         d: int, fw: int) -> torch.Tensor:
    """DilatedConvolution1D.__call__ (wavenet.py:294-342), op for op."""
    B, cin, _, T = x.shape
    cout = W.shape[0]
    if d == 1:                                           # wavenet.py:298-301
        return F.conv2d(F.pad(x, (fw - 1, 0)), W.reshape(cout, cin, 1, fw), b)
    pad, _ = conv_pad_and_prefix(T, d, fw)
    xp = F.pad(x, (pad, 0)) if pad > 0 else x            # CausalPadding1d, wavenet.py:216-227
    xp = xp.reshape(B, cin, -1, d)                       # wavenet.py:325
    out = F.conv2d(xp, W.reshape(cout, cin, fw, 1), b)   # wavenet.py:330
    out = out.reshape(B, cout, 1, -1)                    # wavenet.py:333
    cut = out.shape[3] - T                               # wavenet.py:336
    if cut > 0:
        out = out[:, :, :, cut:]                         # CausalSlice1d, wavenet.py:249-254
    elif cut < 0:
        out = F.pad(out, (-cut, 0))
    return out


def dilated_conv_closed(x: np.ndarray, W: np.ndarray, b: Optional[np.ndarray],
                        d: int, fw: int, compat_zero_prefix: bool = True) -> np.ndarray:
    """out[t] = sum_k W[:,:,k] x[t-(fw-1-k)d] (+b) for t >= Z, exactly 0 for t < Z."""
    B, cin, _, T = x.shape
    cout = W.shape[0]
    Wk = W.reshape(cout, cin, fw)
    out = np.zeros((B, cout, 1, T), dtype=x.dtype)
    for k in range(fw):
        s = (fw - 1 - k) * d
        if s >= T:
            continue
        out[:, :, 0, s:] += np.einsum("oc,bct->bot", Wk[:, :, k], x[:, :, 0, :T - s])
    if b is not None:
        out += b.reshape(1, cout, 1, 1)
    if compat_zero_prefix:
        _, Z = conv_pad_and_prefix(T, d, fw)
        out[:, :, :, :Z] = 0
    return out


def dilated_conv_column(window: np.ndarray, W: np.ndarray, b: Optional[np.ndarray],
                        d: int, fw: int) -> np.ndarray:
    """DilatedConvolution1D._forward (wavenet.py:281-292): newest column only, batch 0."""
    cout, cin = W.shape[0], W.shape[1]
    Wk = W.reshape(cout, cin, fw)
    acc = np.zeros((cout,), dtype=window.dtype)
    for n in range(fw):                                  # tap fw-1-n reads column -d*n-1
        acc += Wk[:, :, fw - 1 - n] @ window[0, :, 0, -d * n - 1]
    if b is not None:
        acc = acc + b
    return acc.reshape(1, cout, 1, 1)


# --------------------------------------------------------------------------
# activations
# --------------------------------------------------------------------------

def _sigmoid_t(x: torch.Tensor) -> torch.Tensor:
    return torch.tanh(x * 0.5) * 0.5 + 0.5


def _sigmoid_n(x: np.ndarray) -> np.ndarray:
    return np.tanh(x * x.dtype.type(0.5)) * x.dtype.type(0.5) + x.dtype.type(0.5)


def _elu_n(x: np.ndarray) -> np.ndarray:
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0))).astype(x.dtype)


# --------------------------------------------------------------------------
# the model, literal
# --------------------------------------------------------------------------

class RefWaveNet:
    """Literal restatement of WaveNet (wavenet.py:370-617) on torch-CPU tensors."""

    def __init__(self, p: dict, weights: Dict[str, np.ndarray], dtype=torch.float32,
                 requires_grad: bool = False):
        if p["quantization_steps"] != p["softmax_conv_channels"][-1]:   # wavenet.py:172-173
            raise Exception("quantization_steps != softmax_conv_channels[-1]")
        self.p = p
        self.dtype = dtype
        self.w = {k: torch.tensor(v, dtype=dtype, requires_grad=requires_grad)
                  for k, v in weights.items()}

    # -- helpers -----------------------------------------------------------
    def _W(self, name):
        return self.w[name + "/W"], self.w.get(name + "/b")

    def layers(self):
        fw = self.p["residual_conv_filter_width"]
        for blk in range(self.p["residual_num_blocks"]):
            for li in range(len(self.p["residual_conv_channels"])):
                yield blk, li, fw ** li, "residual_%d_block_%d_" % (blk, li)

    # -- forward -----------------------------------------------------------
    def forward_causal_block(self, x):                   # wavenet.py:565-570
        fw = self.p["causal_conv_filter_width"]
        out = x
        for i in range(len(self.p["causal_conv_channels"])):
            W, b = self._W("causal_%d" % i)
            out = dilated_conv_literal(out, W, b, 1, fw)
        return out

    def residual_layer(self, x, pre, d):                 # ResidualConvLayer.__call__, wavenet.py:358-368
        fw = self.p["residual_conv_filter_width"]
        Wf, bf = self._W(pre + "wf")
        Wg, bg = self._W(pre + "wg")
        z = torch.tanh(dilated_conv_literal(x, Wf, bf, d, fw)) * \
            _sigmoid_t(dilated_conv_literal(x, Wg, bg, d, fw))
        Wp, bp = self._W(pre + "projection_block")
        Ws, bs = self._W(pre + "projection_softmax")
        return F.conv2d(z, Wp, bp) + x, F.conv2d(z, Ws, bs), z

    def forward_residual_block(self, x, record=None):    # wavenet.py:572-582
        total = 0
        out = x
        for blk, li, d, pre in self.layers():
            out, skip, z = self.residual_layer(out, pre, d)
            if record is not None:
                record.append((out, skip, z))
            total = total + skip                         # int 0 seed, fixed order
        return out, total

    def forward_softmax_block(self, x, apply_softmax=True, act="relu", first_relu_mask=None):   # wavenet.py:584-593
        """``first_relu_mask`` (a 0/1 tensor shaped like x; a checker's device, not a reference feature): the ReLU in front of
        the first head convolution is applied as a product with this mask instead of being re-decided from x.  Used by
        the large-window parity test: a skip value within rounding distance of 0 makes a whole gradient term present in
        one implementation and absent in the other, so gradients are compared at the SAME mask (the device's)."""
        out = x
        for i in range(len(self.p["softmax_conv_channels"]) - 1):
            if i == 0 and first_relu_mask is not None and act == "relu":
                out = out * first_relu_mask
            else:
                out = F.relu(out) if act == "relu" else F.elu(out)   # faster_wavenet.py:108
            W, b = self._W("softmax_%d" % i)
            out = F.conv2d(out, W, b)
        if apply_softmax:
            out = F.softmax(out, dim=1)
        return out

    def forward_one_step(self, x, apply_softmax=True):   # wavenet.py:556-563
        c = self.forward_causal_block(x)
        _, s = self.forward_residual_block(c)
        return self.forward_softmax_block(s, apply_softmax=apply_softmax)

    def cross_entropy(self, raw, target):                # wavenet.py:597-617
        B, Q, _, Tw = raw.shape
        if Tw != target.shape[1]:
            raise Exception("raw_network_output.width != target.width")
        rows = raw.permute(0, 3, 2, 1).reshape(B * Tw, Q)
        tgt = torch.as_tensor(np.asarray(target).reshape(-1).astype(np.int64))
        return F.cross_entropy(rows, tgt)

    def train_loss(self, onehot, target, first_relu_mask=None):
        """Loop body of train_audio/train.py:66-78 (forward + loss)."""
        tw = target.shape[1]
        out = self.forward_causal_block(onehot)
        out, skip = self.forward_residual_block(out)
        skip = skip[:, :, :, skip.shape[3] - tw:]        # slice_1d, train.py:73
        self.last_skip = skip.detach()
        logits = self.forward_softmax_block(skip, apply_softmax=False, first_relu_mask=first_relu_mask)
        return self.cross_entropy(logits, target), logits


def onehot_t(idx: np.ndarray, Q: int, dtype=torch.float32) -> torch.Tensor:
    """data.py:61-68 as a torch tensor (B,Q,1,T)."""
    from .data_ref import onehot_pixel_image
    return torch.tensor(onehot_pixel_image(idx, Q), dtype=dtype)


def train_step_grads(p, weights, idx_in, target, dtype=torch.float32, first_relu_mask=None, keep=None):
    """loss, logits and d loss / d every parameter (wavenet.py:515-519 backward).  ``first_relu_mask`` (B, Cs, 1, Tw):
    see RefWaveNet.forward_softmax_block; ``keep`` (a dict) receives the skip sum of the loss window."""
    net = RefWaveNet(p, weights, dtype=dtype, requires_grad=True)
    mask = None if first_relu_mask is None else torch.as_tensor(np.asarray(first_relu_mask), dtype=dtype)
    loss, logits = net.train_loss(onehot_t(idx_in, p["quantization_steps"], dtype), target, first_relu_mask=mask)
    if keep is not None:
        keep["skip"] = net.last_skip.numpy()
    loss.backward()
    grads = {k: (v.grad.numpy().copy() if v.grad is not None else np.zeros(v.shape, v.detach().numpy().dtype))
             for k, v in net.w.items()}
    return float(loss.detach()), logits.detach().numpy(), grads


# --------------------------------------------------------------------------
# the model, closed form (numpy) -- independent second formulation
# --------------------------------------------------------------------------

def bf16_round(a: np.ndarray) -> np.ndarray:
    """float32 -> nearest bfloat16 (ties to even), returned as float32: what a matrix core sees of an operand in
    BASELINE config 5's arithmetic (bf16 operands, fp32 accumulation)."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = ((u >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)
    return ((u + r) & np.uint32(0xFFFF0000)).view(np.float32)


def forward_closed(p, weights, x, dtype=np.float32, compat_zero_prefix=True, head_act="relu",
                   apply_softmax=False, keep=None, round_operands=None):
    """Same maths as RefWaveNet.forward_one_step via the closed form; returns
    (causal_out, residual_out, sum_skip, head_out).  ``keep`` (a list) receives
    per-layer (out, z) when given.  ``round_operands`` (e.g. :func:`bf16_round`) is applied to both operands of every
    channel contraction after the causal embedding (which is a table lookup, not a product); sums stay in ``dtype``."""
    if round_operands is not None:
        rq = round_operands
        ro = dict(compat_zero_prefix=compat_zero_prefix, head_act=head_act, apply_softmax=apply_softmax, keep=keep)
        return _forward_closed_rounded(p, weights, x, dtype, rq, **ro)
    w = {k: v.astype(dtype) for k, v in weights.items()}
    out = x.astype(dtype)
    fw = p["causal_conv_filter_width"]
    for i in range(len(p["causal_conv_channels"])):
        out = dilated_conv_closed(out, w["causal_%d/W" % i], w.get("causal_%d/b" % i), 1, fw)
    causal = out
    fw = p["residual_conv_filter_width"]
    total = None
    for blk in range(p["residual_num_blocks"]):
        for li in range(len(p["residual_conv_channels"])):
            pre = "residual_%d_block_%d_" % (blk, li)
            d = fw ** li
            a = dilated_conv_closed(out, w[pre + "wf/W"], w.get(pre + "wf/b"), d, fw, compat_zero_prefix)
            g = dilated_conv_closed(out, w[pre + "wg/W"], w.get(pre + "wg/b"), d, fw, compat_zero_prefix)
            z = np.tanh(a) * _sigmoid_n(g)
            Wp = w[pre + "projection_block/W"][:, :, 0, 0]
            Ws = w[pre + "projection_softmax/W"][:, :, 0, 0]
            o = np.einsum("oc,bcht->boht", Wp, z) + out
            s = np.einsum("oc,bcht->boht", Ws, z)
            if pre + "projection_block/b" in w:
                o = o + w[pre + "projection_block/b"].reshape(1, -1, 1, 1)
                s = s + w[pre + "projection_softmax/b"].reshape(1, -1, 1, 1)
            out = o.astype(dtype)
            total = s if total is None else total + s
            if keep is not None:
                keep.append((out, z))
    h = total
    for i in range(len(p["softmax_conv_channels"]) - 1):
        h = np.maximum(h, 0) if head_act == "relu" else _elu_n(h)
        h = np.einsum("oc,bcht->boht", w["softmax_%d/W" % i][:, :, 0, 0], h)
        if "softmax_%d/b" % i in w:
            h = h + w["softmax_%d/b" % i].reshape(1, -1, 1, 1)
    if apply_softmax:
        h = softmax_axis1(h)
    return causal, out, total, h


def _forward_closed_rounded(p, weights, x, dtype, rq, compat_zero_prefix, head_act, apply_softmax, keep):
    w = {k: v.astype(dtype) for k, v in weights.items()}
    out = x.astype(dtype)
    fw = p["causal_conv_filter_width"]
    for i in range(len(p["causal_conv_channels"])):
        out = dilated_conv_closed(out, w["causal_%d/W" % i], w.get("causal_%d/b" % i), 1, fw)
    causal = out
    fw = p["residual_conv_filter_width"]
    total = None
    for blk in range(p["residual_num_blocks"]):
        for li in range(len(p["residual_conv_channels"])):
            pre = "residual_%d_block_%d_" % (blk, li)
            d = fw ** li
            xr = rq(out)
            a = dilated_conv_closed(xr, rq(w[pre + "wf/W"]), w.get(pre + "wf/b"), d, fw, compat_zero_prefix)
            g = dilated_conv_closed(xr, rq(w[pre + "wg/W"]), w.get(pre + "wg/b"), d, fw, compat_zero_prefix)
            z = (np.tanh(a) * _sigmoid_n(g)).astype(dtype)
            zr = rq(z)
            o = np.einsum("oc,bcht->boht", rq(w[pre + "projection_block/W"][:, :, 0, 0]), zr) + out
            s = np.einsum("oc,bcht->boht", rq(w[pre + "projection_softmax/W"][:, :, 0, 0]), zr)
            if pre + "projection_block/b" in w:
                o = o + w[pre + "projection_block/b"].reshape(1, -1, 1, 1)
                s = s + w[pre + "projection_softmax/b"].reshape(1, -1, 1, 1)
            out = o.astype(dtype)
            total = s if total is None else total + s
            if keep is not None:
                keep.append((out, z))
    h = total
    for i in range(len(p["softmax_conv_channels"]) - 1):
        h = np.maximum(h, 0) if head_act == "relu" else _elu_n(h)
        h = np.einsum("oc,bcht->boht", rq(w["softmax_%d/W" % i][:, :, 0, 0]), rq(h.astype(dtype)))
        if "softmax_%d/b" % i in w:
            h = h + w["softmax_%d/b" % i].reshape(1, -1, 1, 1)
    if apply_softmax:
        h = softmax_axis1(h)
    return causal, out, total, h


def softmax_axis1(x: np.ndarray) -> np.ndarray:
    e = np.exp(x - x.max(axis=1, keepdims=True))
    return (e / e.sum(axis=1, keepdims=True)).astype(x.dtype)


# --------------------------------------------------------------------------
# fast generation, literal (faster_wavenet.py:13-113)
# --------------------------------------------------------------------------

class RefFasterWaveNet:
    """FasterWaveNet restated on numpy: full-window caches, physically rolled."""

    def __init__(self, p: dict, weights: Dict[str, np.ndarray], fast_head_act: str = "elu"):
        self.p = p
        self.w = {k: np.asarray(v, dtype=np.float32) for k, v in weights.items()}
        self.net = RefWaveNet(p, weights)
        self.fast_head_act = fast_head_act               # F.elu at faster_wavenet.py:108
        self.prev_causal_outputs = None
        self.prev_residual_outputs = None

    def forward_one_step(self, x: np.ndarray, apply_softmax=True) -> np.ndarray:
        """faster_wavenet.py:13-47: full forward that records every layer's window."""
        xt = torch.tensor(x, dtype=torch.float32)
        self.prev_causal_outputs = []
        out = xt
        fw = self.p["causal_conv_filter_width"]
        for i in range(len(self.p["causal_conv_channels"])):
            W, b = self.net._W("causal_%d" % i)
            out = dilated_conv_literal(out, W, b, 1, fw)
            self.prev_causal_outputs.append(out.numpy().copy())
        rec = []
        _, skip = self.net.forward_residual_block(out, record=rec)
        nl = len(self.p["residual_conv_channels"])
        self.prev_residual_outputs = []
        for blk in range(self.p["residual_num_blocks"]):
            self.prev_residual_outputs.append(
                [[rec[blk * nl + li][0].numpy().copy(), rec[blk * nl + li][1].numpy().copy()]
                 for li in range(nl)])
        return self.net.forward_softmax_block(skip, apply_softmax=apply_softmax).numpy()

    def _layer_column(self, window, pre, d):             # ResidualConvLayer._forward, wavenet.py:350-356
        fw = self.p["residual_conv_filter_width"]
        w = self.w
        a = dilated_conv_column(window, w[pre + "wf/W"], w.get(pre + "wf/b"), d, fw)
        g = dilated_conv_column(window, w[pre + "wg/W"], w.get(pre + "wg/b"), d, fw)
        z = np.tanh(a) * _sigmoid_n(g)
        o = w[pre + "projection_block/W"][:, :, 0, 0] @ z[0, :, 0, 0]
        s = w[pre + "projection_softmax/W"][:, :, 0, 0] @ z[0, :, 0, 0]
        if pre + "projection_block/b" in w:
            o = o + w[pre + "projection_block/b"]
            s = s + w[pre + "projection_softmax/b"]
        o = o + window[0, :, 0, -1]
        return o.astype(np.float32).reshape(1, -1, 1, 1), s.astype(np.float32).reshape(1, -1, 1, 1)

    def _forward_one_step(self, x: np.ndarray, apply_softmax=True) -> np.ndarray:
        """faster_wavenet.py:50-113; returns the full-window output like the reference."""
        if self.prev_causal_outputs is None:
            return self.forward_one_step(x, apply_softmax=apply_softmax)
        inp = x
        fw = self.p["causal_conv_filter_width"]
        for i in range(len(self.p["causal_conv_channels"])):          # faster_wavenet.py:65-78
            col = dilated_conv_column(inp, self.w["causal_%d/W" % i], self.w.get("causal_%d/b" % i), 1, fw)
            prev = np.roll(self.prev_causal_outputs[i], -1, axis=3)
            prev[0, :, 0, -1] = col[0, :, 0, 0]
            self.prev_causal_outputs[i] = prev
            inp = prev
        total = 0
        rfw = self.p["residual_conv_filter_width"]
        for blk in range(self.p["residual_num_blocks"]):              # faster_wavenet.py:80-103
            for li in range(len(self.p["residual_conv_channels"])):
                pre = "residual_%d_block_%d_" % (blk, li)
                o, s = self._layer_column(inp, pre, rfw ** li)
                po, ps = self.prev_residual_outputs[blk][li]
                po = np.roll(po, -1, axis=3)
                po[0, :, 0, -1] = o[0, :, 0, 0]
                ps = np.roll(ps, -1, axis=3)
                ps[0, :, 0, -1] = s[0, :, 0, 0]
                self.prev_residual_outputs[blk][li] = [po, ps]
                total = total + ps
                inp = po
        h = total                                                     # faster_wavenet.py:105-113
        for i in range(len(self.p["softmax_conv_channels"]) - 1):
            h = _elu_n(h) if self.fast_head_act == "elu" else np.maximum(h, 0)
            h = np.einsum("oc,bcht->boht", self.w["softmax_%d/W" % i][:, :, 0, 0], h)
            if "softmax_%d/b" % i in self.w:
                h = h + self.w["softmax_%d/b" % i].reshape(1, -1, 1, 1)
        h = h.astype(np.float32)
        return softmax_axis1(h) if apply_softmax else h


# --------------------------------------------------------------------------
# sampling and the generate loop (train_audio/generate.py:21-43)
# --------------------------------------------------------------------------

def choice_from_uniform(prob: np.ndarray, u: float) -> int:
    """``numpy.random.RandomState.choice(arange(Q), p=prob)`` given its one
    ``random_sample()`` draw ``u``: float64 cumsum, normalise by the last entry,
    ``searchsorted(side='right')`` (numpy legacy generator; generate.py:39)."""
    cdf = np.cumsum(np.asarray(prob, dtype=np.float64))
    cdf /= cdf[-1]
    return int(np.searchsorted(cdf, u, side="right"))


def generate(p, weights, n_emit: int, uniforms: np.ndarray, fast: bool, fast_head_act="elu",
             trace=None) -> np.ndarray:
    """generate_audio's loop: window of ``input_width`` tokens, silence = 127
    (generate.py:21), one categorical draw per step; returns the emitted tokens."""
    from .data_ref import onehot_pixel_image
    iw = input_width(p)
    Q = p["quantization_steps"]
    sil = 127 if Q > 127 else Q // 2
    buf = np.full((iw,), sil, dtype=np.int32)
    model = RefFasterWaveNet(p, weights, fast_head_act) if fast else RefWaveNet(p, weights)
    for step in range(n_emit):
        x = onehot_pixel_image(buf[-iw:].reshape(1, -1), Q)
        if fast:
            sm = model._forward_one_step(x, apply_softmax=True)
        else:
            sm = model.forward_one_step(torch.tensor(x), apply_softmax=True).numpy()
        prob = sm[0, :, 0, -1]
        if trace is not None:
            trace.append(prob.copy())
        buf = np.append(buf, [choice_from_uniform(prob, uniforms[step])]).astype(np.int32)
    return buf[iw:]


# --------------------------------------------------------------------------
# Eve (wavenet.py:10-79), restated literally: one state per parameter array, float32 d / f arrays
# --------------------------------------------------------------------------

class EveRef(object):
    """TEST INFRASTRUCTURE ONLY.  ``update(params, grads, loss)`` performs what Eve.update does to every parameter
    (wavenet.py:46-53: _update_d_and_f then the moment / parameter update), hooks excluded."""

    def __init__(self, alpha=0.001, beta1=0.9, beta2=0.999, beta3=0.999, eps=1e-8, lower_threshold=0.1,
                 upper_threshold=10):
        self.alpha, self.beta1, self.beta2, self.beta3, self.eps = alpha, beta1, beta2, beta3, eps
        self.lower_threshold, self.upper_threshold = lower_threshold, upper_threshold
        self.t = 0
        self.states = {}

    @property
    def lr(self):                                                   # wavenet.py:67-71
        fix1 = 1. - self.beta1 ** self.t
        fix2 = 1. - self.beta2 ** self.t
        return self.alpha * math.sqrt(fix2) / fix1

    def _update_d_and_f(self, state, loss):                         # wavenet.py:27-44
        d, f = state["d"], state["f"]
        if self.t > 1:
            old_f = float(f[0])
            if loss > old_f:
                delta = self.lower_threshold + 1.
                Delta = self.upper_threshold + 1.
            else:
                delta = 1. / (self.upper_threshold + 1.)
                Delta = 1. / (self.lower_threshold + 1.)
            c = min(max(delta, loss / (old_f + 1e-12)), Delta)
            new_f = c * old_f
            r = abs(new_f - old_f) / (min(new_f, old_f) + 1e-12)
            d += (1 - self.beta3) * (r - d)
            f[:] = new_f
        else:
            f[:] = loss

    def update(self, params, grads, loss):
        self.t += 1
        for k in params:
            if k not in self.states:                                # init_state, wavenet.py:20-26
                self.states[k] = dict(m=np.zeros_like(params[k]), v=np.zeros_like(params[k]),
                                      d=np.ones(1, dtype=params[k].dtype), f=np.zeros(1, dtype=params[k].dtype))
            st = self.states[k]
            self._update_d_and_f(st, loss)
            st["m"] += (1. - self.beta1) * (grads[k] - st["m"])
            st["v"] += (1. - self.beta2) * (grads[k] * grads[k] - st["v"])
            params[k] -= self.lr * st["m"] / (st["d"] * np.sqrt(st["v"]) + self.eps)


# --------------------------------------------------------------------------
# the remaining get_optimizer names (wavenet.py:87-96): Chainer's published update rules, float32, hooks excluded
# (Chainer is a dependency that is absent here: parity unpinned, like the rest of this file)
# --------------------------------------------------------------------------

def rule_step_ref(name, p, g, s1, s2, lr, hyper):
    """TEST INFRASTRUCTURE ONLY.  One in-place update of float32 arrays ``p`` with state ``s1`` (v / h / ms / msg) and
    ``s2`` (AdaDelta's msdx).  ``hyper`` = momentum (MomentumSGD, NesterovAG) / alpha (RMSprop) / rho (AdaDelta)."""
    f = np.float32
    lr, hyper = f(lr), f(hyper)
    name = name.lower()
    if name == "sgd":                                       # chainer.optimizers.SGD
        p -= lr * g
    elif name == "momentumsgd":                             # v = momentum v - lr g; p += v
        s1[:] = hyper * s1 - lr * g
        p += s1
    elif name == "adagrad":                                 # eps = 1e-8
        s1 += g * g
        p -= lr * g / (np.sqrt(s1) + f(1e-8))
    elif name == "adadelta":                                # eps = 1e-6
        s1 += (f(1) - hyper) * (g * g - s1)
        dx = np.sqrt((s2 + f(1e-6)) / (s1 + f(1e-6))) * g
        s2 += (f(1) - hyper) * (dx * dx - s2)
        p -= dx
    elif name in ("nesterov", "nesterovag"):                # v = momentum v - lr g; p += momentum^2 v - (1+momentum) lr g
        s1[:] = hyper * s1 - lr * g
        p += hyper * hyper * s1 - (f(1) + hyper) * lr * g
    elif name == "rmsprop":                                 # eps = 1e-8
        s1 += (f(1) - hyper) * (g * g - s1)
        p -= lr * g / (np.sqrt(s1) + f(1e-8))
    else:
        raise Exception()                                   # wavenet.py:97
