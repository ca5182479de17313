"""ctypes binding of libwavenet_hip.so (include/wavenet_hip.h).

There is no CPU fallback: if the library is missing or a call fails, the caller gets an exception.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Sequence, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WAVENET_HIP_LIB") or os.path.join(_HERE, "libwavenet_hip.so")   # (override: same-box A/B of two builds)
ABI_VERSION = 5
XENT_LOSS_WORDS = 2056      # WN_XENT_LOSS_WORDS: loss[0] + per-workgroup sums of wn_softmax_xent
SQNORM_WORDS = 1040          # WN_SQNORM_WORDS: out[0] + per-workgroup partial sums of wn_sqnorm

WN_ACT_NONE, WN_ACT_RELU, WN_ACT_ELU = 0, 1, 2
ACT = {"none": WN_ACT_NONE, None: WN_ACT_NONE, "relu": WN_ACT_RELU, "elu": WN_ACT_ELU}

_p = C.c_void_p
_i = C.c_int
_f = C.c_float
_i64 = C.c_int64
_pp = C.POINTER(C.c_void_p)
_ip = C.POINTER(C.c_int)
_ex = None      # C.POINTER(WnExec), set below the struct definition


class WnDecoderDesc(C.Structure):
    _fields_ = [
        ("Q", _i), ("fw_causal", _i), ("n_causal", _i), ("fw", _i), ("n_blocks", _i), ("n_layers", _i),
        ("Cr", _i), ("Cs", _i), ("n_head", _i),
        ("causal_channels", _ip), ("cd", _ip), ("head_channels", _ip),
        ("causal_W", _pp), ("causal_b", _pp),
        ("Wf", _pp), ("bf", _pp), ("Wg", _pp), ("bg", _pp), ("Wp", _pp), ("bp", _pp), ("Ws", _pp), ("bs", _pp),
        ("head_W", _pp), ("head_b", _pp),
        ("head_act", _i), ("flags", C.c_uint),
    ]


class WnExec(C.Structure):
    """Per-call options of the entry points that hold a channel GEMM or need scratch (include/wavenet_hip.h)."""
    _fields_ = [("precision", _i), ("flags", C.c_uint), ("ws", _p), ("ws_bytes", C.c_size_t),
                ("fwd_t1_min_blocks", _i), ("reserved", _i), ("plan", _p)]


_ex = C.POINTER(WnExec)


class WnStackDesc(C.Structure):
    _fields_ = [
        ("n_layers", _i), ("Cr", _i), ("Cs", _i), ("fw", _i), ("cd", _ip), ("dilation", _ip),
        ("Wf", _pp), ("bf", _pp), ("Wg", _pp), ("bg", _pp), ("Wp", _pp), ("bp", _pp), ("Ws", _pp), ("bs", _pp),
    ]


_SIGS = {
    "wn_abi_version": (_i, []),
    "wn_last_error": (C.c_char_p, []),
    "wn_layer_fast_path": (_i, [_i, _i, _i]),
    "wn_embed_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "wn_embed_bwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _ex, _p]),
    "wn_conv_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "wn_conv_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "wn_layer_fwd": (_i, [_p] * 11 + [_i] * 7 + [_ex, _p]),
    "wn_layer_bwd": (_i, [_p] * 16 + [_i] * 7 + [_ex, _p]),
    "wn_layer_bwd_workspace_floats": (C.c_size_t, [_i, _i, _i, _i, _i]),
    "wn_pointwise_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _ex, _p]),
    "wn_pointwise_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _ex, _p]),
    "wn_skip_sum_fwd": (_i, [_i, _pp, _pp, _pp, _ip, _p, _i, _i, _i, _i, _i, _i, _ex, _p]),
    "wn_skip_sum_bwd_dz": (_i, [_i, _pp, _ip, _p, _pp, _i, _i, _i, _i, _i, _ex, _p]),
    "wn_skip_sum_bwd_dw": (_i, [_i, _pp, _ip, _p, _pp, _pp, _i, _i, _i, _i, _i, _ex, _p]),
    "wn_stack_fwd": (_i, [C.POINTER(WnStackDesc), _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _ex, _p]),
    "wn_stack_saves_tanh": (_i, [C.POINTER(WnStackDesc), _ex]),
    "wn_stack_bwd_workspace_bytes": (C.c_size_t, [C.POINTER(WnStackDesc), _i, _i]),
    "wn_stack_bwd": (_i, [C.POINTER(WnStackDesc)] + [_p] * 8 + [_pp] * 8 + [_p, C.c_size_t, _i, _i, _i, _i, _ex, _p]),
    "wn_exec_workspace_bytes": (C.c_size_t, [C.POINTER(WnStackDesc), _i, _i, _i, _ip, _i, _i, _i]),
    "wn_softmax_fwd": (_i, [_p, _p, _i, _i, _p]),
    "wn_softmax_xent": (_i, [_p, _p, _p, _p, _i, _i, _i64, _p]),
    "wn_head_xent_supported": (_i, [C.c_int64, _i, _i, C.POINTER(WnExec)]),
    "wn_head_xent": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i64, C.POINTER(WnExec), _p]),
    "wn_nchw_to_btc": (_i, [_p, _p, _i, _i, _i, _p]),
    "wn_btc_to_nchw": (_i, [_p, _p, _i, _i, _i, _p]),
    "wn_decoder_create": (_i, [_pp, C.POINTER(WnDecoderDesc), _p]),
    "wn_decoder_destroy": (_i, [_p]),
    "wn_decoder_update_weights": (_i, [_p, C.POINTER(WnDecoderDesc), _p]),
    "wn_decoder_load_state": (_i, [_p, _p, _i, _pp, _pp, _p]),
    "wn_decoder_step": (_i, [_p, C.c_int32, _p, _i, _p]),
    "wn_decoder_run": (_i, [_p, C.c_int32, _p, _i, _p, _p, _p]),
    "wn_decoder_status": (_i, [_p, _p]),
    "wn_decoder_batch_max": (_i, []),
    "wn_decoder_run_batch": (_i, [_p, _i, _p, _p, _i, _p, _p, _i, _p]),
    "wn_sample_categorical": (_i, [_p, _p, _p, _i, _i, _p]),
    "wn_sqnorm": (_i, [_p, _p, _i64, _f, _f, _p, _p]),
    "wn_adam_step": (_i, [_p, _p, _p, _p, _i64, _f, _f, _f, _f, _f, _p, _f, _f, _p]),
    "wn_adam_step_dev": (_i, [_p, _p, _p, _p, _i64, _p, _f, _f, _f, _f, _p, _f, _f, _p]),
    "wn_mulaw_encode_pcm16": (_i, [_p, _p, _p, _i64, _p]),
    "wn_mulaw_decode": (_i, [_p, _p, _p, _i64, _i, _p]),
    "wn_eve_step": (_i, [_p, _p, _p, _p, _i64, _f, _f, _f, _f, _f, _f, _p, _f, _f, _p]),
    "wn_rule_step": (_i, [_i, _p, _p, _p, _p, _i64, _f, _p, _f, _f, _f, _p, _f, _f, _p]),
    "wn_scale_by_dev": (_i, [_p, _p, _i64, _p]),
    "wn16_supported": (_i, [C.POINTER(WnStackDesc)]),
    "wn16_pack_elems": (C.c_size_t, [C.POINTER(WnStackDesc)]),
    "wn16_pack_stack": (_i, [C.POINTER(WnStackDesc), _p, _p]),
    "wn16_embed_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "wn16_embed_bwd_workspace_bytes": (C.c_size_t, [_i, _i]),
    "wn16_embed_bwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, C.c_size_t, _p]),
    "wn16_cvt_to_bf16": (_i, [_p, _p, _i64, _p]),
    "wn16_cvt_to_f32": (_i, [_p, _p, _i64, _p]),
    "wn16_stack_fwd": (_i, [C.POINTER(WnStackDesc), _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "wn16_stack_bwd_workspace_bytes": (C.c_size_t, [C.POINTER(WnStackDesc), _i, _i, _i]),
    "wn16_stack_bwd": (_i, [C.POINTER(WnStackDesc)] + [_p] * 7 + [_pp] * 4 + [_p, C.c_size_t, _i, _i, _i, _i, C.c_uint, _p]),
    "wn16_pack_pointwise": (_i, [_p, _p, _p, _i, _i, _p]),
    "wn16_pointwise_fwd": (_i, [_p, _p, _p, _p, _i, _i64, _i, _i, _i, _p]),
    "wn16_pointwise_bwd_workspace_bytes": (C.c_size_t, [_i64, _i]),
    "wn16_pointwise_bwd": (_i, [_p] * 8 + [_i64, _i, _i, _i, _p, C.c_size_t, _p]),
    "wn_plan_create": (_i, [C.POINTER(_p), _p, C.c_size_t]),
    "wn_plan_destroy": (_i, [_p]),
    "wn_plan_record": (_i, [_p]),
    "wn_plan_finish": (_i, [_p, _p]),
    "wn_plan_prepare": (_i, [_p, _p, _i64, _p]),
    "wn_plan_stats": (_i, [_p, C.POINTER(C.c_int64)]),
    "wn_prof_enable": (_i, [_i]),
    "wn_prof_report": (_i, [C.c_char_p, _i]),
}

EXPORTS = tuple(_SIGS)
_lib: Optional[C.CDLL] = None


class WaveNetHipError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load the library once; raise if it is absent (there is deliberately no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise WaveNetHipError(
                "%s not found: build it with `python -m wavenet_amd.build` (hipcc, gfx950)" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)          # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        if l.wn_abi_version() != ABI_VERSION:
            raise WaveNetHipError("ABI mismatch: library %d, binding %d" % (l.wn_abi_version(), ABI_VERSION))
        _lib = l
    return _lib


class profile(object):
    """``with profile() as prof: ...`` then ``prof.result()`` -> {entry point: (calls, total_ms, min_ms,
    max_ms)}: HIP events recorded by the library around each per-op entry point's kernels, on the
    stream they are launched on (include/wavenet_hip.h: wn_prof_enable / wn_prof_report)."""

    def __enter__(self):
        lib().wn_prof_enable(1)
        self._res = None
        return self

    def __exit__(self, *exc):
        self.result()
        lib().wn_prof_enable(0)

    def result(self) -> Dict[str, Tuple[int, float, float, float]]:
        if self._res is None:
            n = lib().wn_prof_report(None, 0)
            buf = C.create_string_buffer(n + 16)
            lib().wn_prof_report(buf, n + 16)
            res = {}
            for line in buf.value.decode().splitlines():
                name, calls, tot, mn, mx = line.split()
                res[name] = (int(calls), float(tot), float(mn), float(mx))
            self._res = res
        return self._res


WN_OK, WN_EARG, WN_ESHAPE, WN_EHIP, WN_ETIMEOUT = 0, -1, -2, -3, -4      # include/wavenet_hip.h


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().wn_last_error().decode("utf-8", "replace")
        raise WaveNetHipError("%s failed (%d): %s" % (what or "libwavenet_hip call", rc, msg))


def ptr(t) -> Optional[int]:
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def ptr_array(tensors: Sequence) -> "C.Array":
    arr = (C.c_void_p * max(1, len(tensors)))()
    for n, t in enumerate(tensors):
        arr[n] = None if t is None else t.data_ptr()
    return arr


def int_array(vals: Sequence[int]) -> "C.Array":
    return (C.c_int * max(1, len(vals)))(*[int(v) for v in vals])


def stream_ptr() -> Optional[int]:
    import torch
    return torch.cuda.current_stream().cuda_stream or None


GEMM_PRECISIONS = ("fp32", "bf16x3", "bf16", "fp16x2")
WN_EXEC_FORCE_GENERIC, WN_EXEC_NO_FUSED_WIDE, WN_EXEC_NO_FWD_GROUPS, WN_EXEC_NO_PIPELINED_GEMM = 1, 2, 4, 8
WN_EXEC_NO_MULTI_LAYER_BWD = 16
WN_EXEC_BF16_MULTI_LAYER_BWD = 32     # bf16 storage: the layer backward of layers L-2 .. 1 in one launch (opt-in; no faster, bit-identical)
WN_DECODER_ONE_WORKGROUP = 64          # WnDecoderDesc.flags: wn_decoder_run on one workgroup instead of nine (other summation order: ~1e-7)


def default_exec_flags() -> int:
    """WnExec.flags for models that do not set ``net.exec_flags`` themselves.  The library reads no environment variable
    (ABI 3); these two diagnostic switches are host policy, read here, and travel with every call:
    WAVENET_HIP_FORCE_GENERIC=1 (any-shape correctness kernels everywhere), WAVENET_HIP_NO_FUSED_WIDE=1."""
    f = 0
    if os.environ.get("WAVENET_HIP_FORCE_GENERIC") == "1":
        f |= WN_EXEC_FORCE_GENERIC
    if os.environ.get("WAVENET_HIP_NO_FUSED_WIDE"):
        f |= WN_EXEC_NO_FUSED_WIDE
    if os.environ.get("WAVENET_HIP_NO_FWD_GROUPS") == "1":
        f |= WN_EXEC_NO_FWD_GROUPS
    if os.environ.get("WAVENET_HIP_NO_PIPELINED_GEMM") == "1":
        f |= WN_EXEC_NO_PIPELINED_GEMM
    if os.environ.get("WAVENET_HIP_NO_MULTI_LAYER_BWD") == "1":
        f |= WN_EXEC_NO_MULTI_LAYER_BWD
    if os.environ.get("WAVENET_HIP_BF16_MULTI_LAYER_BWD") == "1":
        f |= WN_EXEC_BF16_MULTI_LAYER_BWD
    if os.environ.get("WAVENET_HIP_DECODER_ONE_WORKGROUP") == "1":
        f |= WN_DECODER_ONE_WORKGROUP
    return f


def default_fwd_t1_min_blocks() -> int:
    """WnExec.fwd_t1_min_blocks default: 0 (the library's own threshold) or WAVENET_HIP_FWD_T1_MIN_BLOCKS."""
    return int(os.environ.get("WAVENET_HIP_FWD_T1_MIN_BLOCKS", "0") or 0)

_default_precision = {"fp32": "fp32", "bf16": "bf16", "bf16x3": "bf16x3"}.get(os.environ.get("WAVENET_HIP_GEMM", ""), "fp16x2")


def set_gemm_precision(name: str) -> None:
    """Default arithmetic of the channel GEMMs for models that do not set ``net.gemm_precision`` themselves: "fp32" (fp32
    MFMA), "bf16x3" (three-way bf16 split, six products, fp32-accurate), "fp16x2" (the skip-path contractions on a two-way
    fp16 split with power-of-two scaling from the operands' measured range, three products, fp32-accurate to 2^-21;
    everything else as bf16x3 -- the start value, or WAVENET_HIP_GEMM) or "bf16" (operands rounded to bf16, fp32
    accumulation).  Pure host state: every library call carries its precision as an argument
    (WnExec), so models of different precision coexist in one process."""
    global _default_precision
    if name not in GEMM_PRECISIONS:
        raise ValueError("precision must be one of %r" % (GEMM_PRECISIONS,))
    if name != "fp32" and os.environ.get("WAVENET_HIP_FORCE_GENERIC") == "1":
        raise WaveNetHipError("WAVENET_HIP_FORCE_GENERIC=1 pins the fp32 kernels")
    _default_precision = name


def get_gemm_precision() -> str:
    return "fp32" if os.environ.get("WAVENET_HIP_FORCE_GENERIC") == "1" else _default_precision
