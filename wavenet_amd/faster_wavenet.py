"""`FasterWaveNet` with the reference's face (faster_wavenet.py:11-113), backed by the HIP decoder.

The reference caches every layer's full-window output and rolls all caches by one column per
generated sample.  Here the state is a handle owned by libwavenet_hip.so: per-layer rings of the
last (fw-1)*d input columns in HBM, advanced by one persistent kernel (csrc/decoder.hip).

What a caller can observe:
* ``_forward_one_step`` returns the reference's ``(1, Q, 1, W)`` -- every column of the rolled window under the ELU head,
  the newest last (faster_wavenet.py:105-113) -- from a device-side ring of the window's logits (``keep_window``, on by
  default since round 6).  The reference's caller reads ``[0, :, 0, -1]`` only (train_audio/generate.py:38);
  ``full_window=False`` (or ``keep_window = False`` before the prefill) returns just that newest column as
  ``(1, Q, 1, 1)`` and skips the per-call window concatenation + W-row softmax.  ``generate()`` never builds the window.
* only the newest token of ``x_batch_data`` is read (the reference also reads nothing else of it:
  wavenet.py:286), and an integer token may be passed instead of the one-hot window.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, ptr_array, int_array, stream_ptr, ACT
from .wavenet import WaveNet, _as_view, _need_gpu


class _RingState(object):
    """Marker stored in prev_causal_outputs / prev_residual_outputs (the state itself is on the GPU)."""

    def __init__(self, what):
        self.what = what

    def __repr__(self):
        return "<device ring state: %s>" % self.what


class FasterWaveNet(WaveNet):
    fast_head_activation = "elu"       # faster_wavenet.py:108 (the normal head is ReLU, wavenet.py:588)

    def __init__(self, params, compat_zero_prefix: bool = True, seed: Optional[int] = None, storage: str = "fp32"):
        self._dec = None
        self._dec_keep = None
        self._dec_stale = True
        self._batch_decs = []              # decoder handles of generate_batch (one per utterance), created on demand
        self._batch_stale = []
        self.prev_causal_outputs = None
        self.prev_residual_outputs = None
        self.keep_window = True            # keep the logits of the whole window on the device: _forward_one_step's reference shape
        self._hist = None                  # (W, Q) ring of ELU-head logits, oldest column at _hist_pos
        self._hist_pos = 0
        super().__init__(params, compat_zero_prefix=compat_zero_prefix, seed=seed, storage=storage)

    def __del__(self):
        try:
            if self._dec is not None:
                _lib.lib().wn_decoder_destroy(self._dec)
            for h in getattr(self, "_batch_decs", []):
                _lib.lib().wn_decoder_destroy(h)
        except Exception:
            pass

    def _weights_changed(self):
        super()._weights_changed()
        self._dec_stale = True
        self._batch_stale = [True] * len(getattr(self, "_batch_decs", []))

    # -- decoder handle -------------------------------------------------------------------------
    def _desc(self):
        p = self.params
        L = self._flat_layers
        keep = dict(
            causal_ch=int_array(p.causal_conv_channels), cd=int_array(p.residual_conv_channels),
            head_ch=int_array(p.softmax_conv_channels),
            causal_W=ptr_array([l.W for l in self.causal_conv_layers]),
            causal_b=ptr_array([l.b for l in self.causal_conv_layers]),
            Wf=ptr_array([l.wf.W for l in L]), bf=ptr_array([l.wf.b for l in L]),
            Wg=ptr_array([l.wg.W for l in L]), bg=ptr_array([l.wg.b for l in L]),
            Wp=ptr_array([l.projection_block.W for l in L]), bp=ptr_array([l.projection_block.b for l in L]),
            Ws=ptr_array([l.projection_softmax.W for l in L]), bs=ptr_array([l.projection_softmax.b for l in L]),
            head_W=ptr_array([l.W for l in self.softmax_conv_layers]),
            head_b=ptr_array([l.b for l in self.softmax_conv_layers]))
        d = _lib.WnDecoderDesc()
        d.Q, d.fw_causal, d.n_causal = p.quantization_steps, p.causal_conv_filter_width, len(p.causal_conv_channels)
        d.fw, d.n_blocks, d.n_layers = p.residual_conv_filter_width, p.residual_num_blocks, len(p.residual_conv_channels)
        d.Cr, d.Cs, d.n_head = self._Cr, self._Cs, len(self.softmax_conv_layers)
        d.causal_channels, d.cd, d.head_channels = keep["causal_ch"], keep["cd"], keep["head_ch"]
        for k in ("causal_W", "causal_b", "Wf", "bf", "Wg", "bg", "Wp", "bp", "Ws", "bs", "head_W", "head_b"):
            setattr(d, k, C.cast(keep[k], C.POINTER(C.c_void_p)))
        d.head_act = ACT[self.fast_head_activation]
        d.flags = _lib.default_exec_flags() if self.exec_flags is None else int(self.exec_flags)
        return d, keep

    def _decoder(self):
        lib = _lib.lib()
        if self._dec is None:
            d, keep = self._desc()
            h = C.c_void_p()
            check(lib.wn_decoder_create(C.byref(h), C.byref(d), stream_ptr()), "wn_decoder_create")
            self._dec, self._dec_stale = h, False
        elif self._dec_stale:
            d, keep = self._desc()
            check(lib.wn_decoder_update_weights(self._dec, C.byref(d), stream_ptr()), "wn_decoder_update_weights")
            self._dec_stale = False
        return self._dec

    # -- the reference's face -------------------------------------------------------------------
    def forward_one_step(self, x_batch, apply_softmax=True, as_numpy=False):
        """Full forward over the window that also seeds the decoder state (faster_wavenet.py:13-47)."""
        x = self.to_variable(x_batch)
        _need_gpu(x)
        if x.shape[0] != 1:
            raise Exception("FasterWaveNet generates one utterance at a time (batch 1), like the reference")
        storage, self.storage = self.storage, "fp32"               # the decoder state is seeded from fp32 activations
        try:
            return self._prefill(x, apply_softmax, as_numpy)
        finally:
            self.storage = storage

    def _prefill(self, x, apply_softmax, as_numpy):
        with torch.no_grad():
            causal_output = self.forward_causal_block(x)
            _, sum_skip = self.forward_residual_block(causal_output)
            out = self.forward_softmax_block(sum_skip, apply_softmax=apply_softmax)
            tokens = (x if not x.is_floating_point() else x[:, :, 0, :].argmax(dim=1)).to(torch.int32).contiguous()
            W = tokens.shape[1]
            dec = self._decoder()
            check(_lib.lib().wn_decoder_load_state(
                dec, ptr(tokens), W, ptr_array([t.contiguous() for t in self._last_causal_outputs]),
                ptr_array(self._last_layer_inputs), stream_ptr()), "wn_decoder_load_state")
            self._hist = None
            if self.keep_window:
                # the reference's next step re-evaluates EVERY cached column with the fast path's ELU head
                # (faster_wavenet.py:100-112): the window's logits under that head, once, from the skip sum just computed
                lg = self.forward_softmax_block(sum_skip, apply_softmax=False, activation=self.fast_head_activation)
                self._hist = lg[0, :, 0, :].t().contiguous()                     # (W, Q), column 0 = oldest
                self._hist_pos = 0
        self.prev_causal_outputs = _RingState("causal")
        self.prev_residual_outputs = _RingState("residual")
        return self.to_numpy(out) if as_numpy else out

    def _forward_one_step(self, x_batch_data, apply_softmax=True, as_numpy=False, full_window=None):
        """One incremental step (faster_wavenet.py:50-63); falls back to the full forward when the
        state was reset by ``prev_causal_outputs = None``.  Returns the reference's ``(1, Q, 1, W)`` -- every column of the
        rolled window under the ELU head, the newest last (faster_wavenet.py:105-113; one concatenation + one softmax over W
        rows per call) -- unless ``full_window=False`` (or ``keep_window`` was False at the prefill): then the newest
        column only, ``(1, Q, 1, 1)``; ``[0, :, 0, -1]`` reads the same values from either."""
        if full_window is None:
            full_window = bool(self.keep_window)
        if full_window and not self.keep_window:
            raise Exception("full_window=True needs keep_window = True before the first (prefill) call")
        if getattr(self, "prev_causal_outputs", None) is None:
            return self.forward_one_step(x_batch_data, apply_softmax=apply_softmax, as_numpy=as_numpy)
        if isinstance(x_batch_data, (int, np.integer)):
            token = int(x_batch_data)
        else:
            x = x_batch_data
            if isinstance(x, np.ndarray):
                token = int(x[0, -1]) if x.ndim == 2 else int(np.argmax(x[0, :, 0, -1]))
            else:
                token = int(x[0, -1]) if x.dim() == 2 else int(x[0, :, 0, -1].argmax())
        Q = self.params.quantization_steps
        lib = _lib.lib()
        prob = torch.empty((1, 1, Q), device=self.device, dtype=torch.float32)
        if self._hist is None:
            if full_window:
                raise Exception("the window history was dropped (generate() advanced the decoder on the device, or keep_window "
                                "was False at the prefill): pass full_window=False for the newest column, or set "
                                "prev_causal_outputs = None and prefill again")
            check(lib.wn_decoder_step(self._decoder(), token, ptr(prob), 1 if apply_softmax else 0, stream_ptr()),
                  "wn_decoder_step")
            out = _as_view(prob)
            return self.to_numpy(out) if as_numpy else out
        # window history: the step leaves its logits in the ring slot of the column that drops out of the window
        W = self._hist.shape[0]
        slot = self._hist[self._hist_pos]
        check(lib.wn_decoder_step(self._decoder(), token, ptr(slot), 0, stream_ptr()), "wn_decoder_step")
        self._hist_pos = (self._hist_pos + 1) % W
        if not full_window:
            if apply_softmax:
                check(lib.wn_softmax_fwd(ptr(slot), ptr(prob), 1, Q, stream_ptr()), "wn_softmax_fwd")
            else:
                prob.view(-1).copy_(slot)
            out = _as_view(prob)
            return self.to_numpy(out) if as_numpy else out
        win = torch.cat((self._hist[self._hist_pos:], self._hist[:self._hist_pos]), dim=0)      # oldest ... newest
        if apply_softmax:
            res = torch.empty_like(win)
            check(lib.wn_softmax_fwd(ptr(win), ptr(res), W, Q, stream_ptr()), "wn_softmax_fwd")
            win = res
        out = _as_view(win.view(1, W, Q))
        return self.to_numpy(out) if as_numpy else out

    # -- the whole generate loop on the device (train_audio/generate.py:9-60 with --fast) --------
    def generate(self, n_samples: int, uniforms, initial_tokens=None, return_probs: bool = False):
        """Emit ``n_samples`` tokens.  Step 1 is the full forward over the initial window (ReLU head,
        like the reference's first ``_forward_one_step`` call); steps 2.. run inside one persistent
        kernel with the ELU head.  ``uniforms[i]`` is the float64 draw numpy's ``choice`` would make
        at step i (``RandomState.random_sample``)."""
        p = self.params
        Q = p.quantization_steps
        iw = self.input_width
        if initial_tokens is None:
            initial_tokens = np.full((iw,), 127 if Q > 127 else Q // 2, dtype=np.int32)   # generate.py:21
        tok = torch.as_tensor(np.asarray(initial_tokens, dtype=np.int32).reshape(1, -1)).to(self.device)
        u = torch.as_tensor(np.asarray(uniforms, dtype=np.float64)).to(self.device)
        if u.numel() < n_samples:
            raise Exception("need one uniform per emitted sample")
        lib = _lib.lib()
        self.prev_causal_outputs = None
        keep, self.keep_window = self.keep_window, False             # the run below never builds the window (and drops it)
        try:
            p0 = self.forward_one_step(tok, apply_softmax=True)      # (1,Q,1,W)
        finally:
            self.keep_window = keep
        first_prob = p0[0, :, 0, -1].contiguous().view(1, Q)
        out = torch.empty((n_samples,), device=self.device, dtype=torch.int32)
        check(lib.wn_sample_categorical(ptr(first_prob), ptr(u), ptr(out), 1, Q, stream_ptr()),
              "wn_sample_categorical")
        probs = torch.empty((n_samples, Q), device=self.device, dtype=torch.float32) if return_probs else None
        if return_probs:
            probs[0] = first_prob[0]
        if n_samples > 1:
            first = int(out[0].item())
            check(lib.wn_decoder_run(self._decoder(), first, ptr(u[1:]), n_samples - 1, ptr(out[1:]),
                                     ptr(probs[1:]) if return_probs else None, stream_ptr()), "wn_decoder_run")
            # the nine-workgroup run reports a wait that gave up (workgroups not all resident) instead of trapping: its
            # tokens would be void.  One 8-byte read-back; generate() hands tokens to the host anyway.
            check(lib.wn_decoder_status(self._decoder(), stream_ptr()), "wn_decoder_status")
            # the decoder advanced n_samples - 1 steps on the device (its rings are current: step-by-step decoding may go
            # on from here), but the host-side window history -- what _forward_one_step(..., full_window=True) returns
            # the older columns from -- still holds the prefill state: drop it rather than answer with a stale window
            self._hist = None
        return (out, probs) if return_probs else out

    # -- N utterances at once (new capability: the reference generates one utterance per process) ------------------------
    def generate_batch(self, n_samples: int, uniforms, initial_tokens=None):
        """``uniforms``: (N, n_samples) float64 -- N independent utterances from the same initial window, utterance u drawing
        with ``uniforms[u]``; returns (N, n_samples) int32 tokens on the device.  A single utterance is a strict
        sample-to-sample chain (generate.py:9-60, batch 1: wavenet.py:286,290,354) and occupies nine of the GPU's CUs; the
        batched launch (``wn_decoder_run_batch``) runs up to ``wn_decoder_batch_max()`` = 28 such chains side by side, each with
        a decoder state of its own.  Row u equals ``generate(n_samples, uniforms[u])`` bit for bit.  Step 1 -- the full forward
        over the initial window -- is the same for every utterance and runs once."""
        p = self.params
        Q = p.quantization_steps
        iw = self.input_width
        u_np = np.ascontiguousarray(np.asarray(uniforms, dtype=np.float64))
        if u_np.ndim != 2 or u_np.shape[1] < n_samples:
            raise Exception("uniforms must be (N, >= n_samples)")
        N = u_np.shape[0]
        lib = _lib.lib()
        if n_samples < 1:
            raise Exception("generate_batch: n_samples must be positive")
        # what the batched launch does not cover runs as a loop over generate() -- same tokens, one utterance at a time --
        # decided BEFORE any work is done: n_samples == 2 (the launch runs two steps or more), more utterances than
        # wn_decoder_batch_max(), a model the specialised decoder does not take, or WN_DECODER_ONE_WORKGROUP
        flags = _lib.default_exec_flags() if self.exec_flags is None else int(self.exec_flags)
        batched = (1 <= N <= lib.wn_decoder_batch_max() and n_samples != 2 and self.storage != "bf16" and
                   not (flags & (_lib.WN_DECODER_ONE_WORKGROUP | _lib.WN_EXEC_FORCE_GENERIC)))
        if N < 1:
            raise Exception("generate_batch: no utterance")
        if not batched:
            return self._generate_batch_loop(n_samples, u_np, initial_tokens)
        if initial_tokens is None:
            initial_tokens = np.full((iw,), 127 if Q > 127 else Q // 2, dtype=np.int32)   # generate.py:21
        tok = torch.as_tensor(np.asarray(initial_tokens, dtype=np.int32).reshape(1, -1)).to(self.device)
        u = torch.as_tensor(u_np).to(self.device)
        # one prefill, N decoder states seeded from it
        self.prev_causal_outputs = None
        storage, self.storage = self.storage, "fp32"
        try:
            with torch.no_grad():
                causal_output = self.forward_causal_block(tok)
                _, sum_skip = self.forward_residual_block(causal_output)
                p0 = self.forward_softmax_block(sum_skip, apply_softmax=True)
        finally:
            self.storage = storage
        tokens = tok.to(torch.int32).contiguous()
        while len(self._batch_decs) < N:
            d, keep = self._desc()
            h = C.c_void_p()
            check(lib.wn_decoder_create(C.byref(h), C.byref(d), stream_ptr()), "wn_decoder_create")
            self._batch_decs.append(h)
            self._batch_stale.append(False)
        for i in range(N):
            if self._batch_stale[i]:
                d, keep = self._desc()
                check(lib.wn_decoder_update_weights(self._batch_decs[i], C.byref(d), stream_ptr()), "wn_decoder_update_weights")
                self._batch_stale[i] = False
            check(lib.wn_decoder_load_state(
                self._batch_decs[i], ptr(tokens), tokens.shape[1], ptr_array([t.contiguous() for t in self._last_causal_outputs]),
                ptr_array(self._last_layer_inputs), stream_ptr()), "wn_decoder_load_state")
        first_prob = p0[0, :, 0, -1].contiguous().view(1, Q).expand(N, Q).contiguous()
        out = torch.empty((N, n_samples), device=self.device, dtype=torch.int32)
        first = torch.empty((N,), device=self.device, dtype=torch.int32)
        check(lib.wn_sample_categorical(ptr(first_prob), ptr(u[:, 0].contiguous()), ptr(first), N, Q, stream_ptr()),
              "wn_sample_categorical")
        out[:, 0] = first
        if n_samples > 1:
            firsts = (C.c_int32 * N)(*[int(v) for v in first.cpu().tolist()])
            handles = (C.c_void_p * N)(*[h.value for h in self._batch_decs[:N]])
            rest = [u[i, 1:].contiguous() for i in range(N)]
            outs = [torch.empty((n_samples - 1,), device=self.device, dtype=torch.int32) for _ in range(N)]
            same = 0 if os.environ.get("WAVENET_HIP_BATCH_OWN_WEIGHTS") == "1" else 1      # the handles ARE copies of this model's weights
            rc = lib.wn_decoder_run_batch(handles, N, firsts, ptr_array(rest), n_samples - 1, ptr_array(outs), None, same,
                                          stream_ptr())
            if rc == _lib.WN_ESHAPE:         # a shape the batched launch does not take (e.g. not config 4's): one by one
                return self._generate_batch_loop(n_samples, u_np, initial_tokens)
            check(rc, "wn_decoder_run_batch")
            for i in range(N):
                check(lib.wn_decoder_status(self._batch_decs[i], stream_ptr()), "wn_decoder_status")
                out[i, 1:] = outs[i]
        return out

    def _generate_batch_loop(self, n_samples, u_np, initial_tokens):
        """generate_batch for what ``wn_decoder_run_batch`` does not cover: ``generate()`` per utterance (row u is
        ``generate(n_samples, uniforms[u])`` by definition)."""
        return torch.stack([self.generate(n_samples, u_np[i], initial_tokens=initial_tokens) for i in range(u_np.shape[0])])
