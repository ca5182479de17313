"""Helpers shared by the GPU parity tests."""
import numpy as np
import torch

from oracle import wavenet_ref as R
from wavenet_amd import FasterWaveNet, Params, WaveNet

CFG1 = dict(quantization_steps=256, causal_conv_channels=[16], residual_conv_channels=[16] * 4,
            residual_num_blocks=1, softmax_conv_channels=[32, 256])
CFG2 = dict(quantization_steps=256, causal_conv_channels=[32], residual_conv_channels=[32] * 10,
            residual_num_blocks=4, softmax_conv_channels=[256, 256])


def build(over, seed=1234, bias_scale=0.0, cls=WaveNet, **kw):
    """(oracle params dict, oracle weights, GPU model with the same weights)."""
    p = R.make_params(**over)
    w = R.init_weights(p, seed, bias_scale=bias_scale)
    pp = Params(p)
    pp.gradient_clipping = kw.pop("gradient_clipping", 1.0)
    net = cls(pp, seed=0, **kw)
    net.load_state_dict(w)
    net.to_gpu()
    return p, w, net


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def btc(a):
    """oracle (B,C,1,T) numpy -> (B,T,C) numpy"""
    return np.ascontiguousarray(a[:, :, 0, :].transpose(0, 2, 1))


def to_np(t):
    return t.detach().cpu().numpy()


_SCRATCH = {}


def EX(precision=None, nbytes=160 << 20, flags=None, t1_min_blocks=None):
    """WnExec for tests that call the C ABI directly: the module-default GEMM precision (or the given one) and a scratch
    buffer that lives as long as the test process."""
    import ctypes as C
    from wavenet_amd import _lib
    if nbytes not in _SCRATCH:
        _SCRATCH[nbytes] = torch.empty((nbytes,), device="cuda", dtype=torch.uint8)
    ex = _lib.WnExec()
    ex.precision = _lib.GEMM_PRECISIONS.index(precision or _lib.get_gemm_precision())
    ex.flags = _lib.default_exec_flags() if flags is None else flags
    ex.fwd_t1_min_blocks = _lib.default_fwd_t1_min_blocks() if t1_min_blocks is None else t1_min_blocks
    ex.ws, ex.ws_bytes = _SCRATCH[nbytes].data_ptr(), nbytes
    _SCRATCH["last"] = ex
    return C.byref(ex)
