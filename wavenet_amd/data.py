"""Boundary data formats of the hot path (data.py of the reference): mu-law tokens, the one-hot
"1 x W image", and the wav files either side of them (load + silence trim, inverse mu-law + save;
SURVEY.md section 8f rank 2).  Host-side numpy / scipy.io.wavfile, as in the reference.

The reference's accidents are reproduced on purpose where a file written or read by it must match, and
can be switched off (``compat=False``):
  * 8-bit PCM scale ``1<<8 - 1`` == 128 (data.py:16,52; operator precedence);
  * the inverse mu-law of save_audio_file lacks the ``- 1`` and divides the token by Q, not mu (data.py:39,43);
  * the silence trim drops the last non-silent sample as well (``[start:-end]``, data.py:27-33);
  * mono files: ``signal /= max`` on the integer array is Python-2 floor division (data.py:7-9,17: only the
    stereo branch converts to float) -- every sample becomes -1 or 0.  Off by default here
    (``compat_mono_int_division``): it destroys the audio."""
from __future__ import annotations

import numpy as np

_LUT16 = {}


def mulaw_encode(signal, quantization_steps: int = 256) -> np.ndarray:
    """float signal in [-1, 1] -> int32 tokens (data.py:18-23): float64 mu-law companding followed
    by ``int32(clip(0.5 s + 0.5, 0, 1) * mu)`` (truncation)."""
    mu = quantization_steps - 1
    s = np.asarray(signal, dtype=np.float64)
    s = np.sign(s) * np.log(1 + mu * np.absolute(s)) / np.log(1 + mu)
    return (np.clip(s * 0.5 + 0.5, 0, 1) * mu).astype(np.int32)


def mulaw_encode_pcm16(pcm, quantization_steps: int = 256) -> np.ndarray:
    """int16 PCM -> tokens through a 65,536-entry table built with :func:`mulaw_encode` on
    ``v / 32768`` (data.py:11-17 normalisation), so it is bit-exact with it by construction."""
    lut = _LUT16.get(quantization_steps)
    if lut is None:
        lut = mulaw_encode(np.arange(-32768, 32768, dtype=np.float64) / 32768.0, quantization_steps)
        _LUT16[quantization_steps] = lut
    return lut[np.asarray(pcm).astype(np.int64) + 32768]


def mulaw_encode_pcm16_device(pcm, quantization_steps: int = 256):
    """int16 PCM device tensor -> int32 token device tensor (``wn_mulaw_encode_pcm16``: the same 65,536-entry table as
    :func:`mulaw_encode_pcm16`, looked up on the GPU)."""
    import torch
    from . import _lib
    if not (isinstance(pcm, torch.Tensor) and pcm.is_cuda and pcm.dtype == torch.int16):
        raise _lib.WaveNetHipError("mulaw_encode_pcm16_device needs an int16 tensor on a HIP device")
    mulaw_encode_pcm16(np.zeros(1, np.int16), quantization_steps)            # builds the table
    lut = torch.as_tensor(_LUT16[quantization_steps]).to(pcm.device)
    pcm = pcm.contiguous()
    out = torch.empty(pcm.shape, dtype=torch.int32, device=pcm.device)
    _lib.check(_lib.lib().wn_mulaw_encode_pcm16(_lib.ptr(pcm), _lib.ptr(lut), _lib.ptr(out), pcm.numel(), _lib.stream_ptr()),
               "wn_mulaw_encode_pcm16")
    return out


def mulaw_decode_device(tokens, quantization_steps: int = 256, compat: bool = True):
    """int32 token device tensor -> float32 signal device tensor (``wn_mulaw_decode``; table from :func:`mulaw_decode`)."""
    import torch
    from . import _lib
    if not (isinstance(tokens, torch.Tensor) and tokens.is_cuda and tokens.dtype == torch.int32):
        raise _lib.WaveNetHipError("mulaw_decode_device needs an int32 tensor on a HIP device")
    table = torch.as_tensor(mulaw_decode(np.arange(quantization_steps), quantization_steps, compat).astype(np.float32))
    table = table.to(tokens.device)
    tokens = tokens.contiguous()
    out = torch.empty(tokens.shape, dtype=torch.float32, device=tokens.device)
    _lib.check(_lib.lib().wn_mulaw_decode(_lib.ptr(tokens), _lib.ptr(table), _lib.ptr(out), tokens.numel(), quantization_steps,
                                         _lib.stream_ptr()), "wn_mulaw_decode")
    return out


_PCM = {"16bit_pcm": (1 << 15, np.int16), "32bit_pcm": (1 << 31, np.int32), "8bit_pcm": (1 << 7, np.uint8)}


def _pcm_format(fmt: str, compat: bool):
    if fmt not in _PCM:
        raise Exception("unknown PCM format: %s" % fmt)
    scale, dtype = _PCM[fmt]
    if fmt == "8bit_pcm" and not compat:
        scale = (1 << 8) - 1
    return scale, dtype


def trim_silence(quantized_signal, silence_threshold: int = 1, compat: bool = True) -> np.ndarray:
    """Strip leading / trailing tokens within ``silence_threshold`` of 127 (data.py:25-33).  With ``compat`` the
    slice is the reference's ``[start:-end]``, which also drops the last non-silent sample."""
    q = np.asarray(quantized_signal)
    loud = np.nonzero(np.abs(q.astype(np.int64) - 127) > silence_threshold)[0]
    if q.size == 0:
        return q
    # the reference's loops leave start = size-1 (no break) / end = size-1 when nothing is loud
    start = int(loud[0]) if loud.size else q.size - 1
    if q.size == 1:
        end = 1                                        # xrange(1, 1) is empty: `end` would be unbound in the reference
    else:
        tail = np.nonzero(np.abs(q[:0:-1].astype(np.int64) - 127) > silence_threshold)[0]   # q[-1], q[-2], ... q[1]
        end = int(tail[0]) + 1 if tail.size else q.size - 1
    if compat:
        return q[start:-end]
    return q[start:q.size - end + 1]


def load_audio_file(filename, quantization_steps: int = 256, format: str = "16bit_pcm", compat: bool = True,
                    compat_mono_int_division: bool = False):
    """wav file -> (mu-law tokens int32 with silence trimmed, sampling rate)   (data.py:5-35)."""
    from scipy.io import wavfile
    sampling_rate, signal = wavfile.read(filename)
    scale, _ = _pcm_format(format, compat)
    if signal.ndim > 1:
        signal = signal[:, 0].astype(float)            # left channel only
        signal = signal / scale
    elif compat_mono_int_division and np.issubdtype(signal.dtype, np.integer):
        signal = np.floor_divide(signal.astype(np.int64), scale).astype(float)
    else:
        signal = signal.astype(float) / scale
    return trim_silence(mulaw_encode(signal, quantization_steps), 1, compat), sampling_rate


def mulaw_decode(quantized_signal, quantization_steps: int = 256, compat: bool = True) -> np.ndarray:
    """tokens -> float signal.  ``compat``: the reference's formula (data.py:37-43)
    ``x = (q / Q - 0.5) * 2;  s = sign(x) (1 + mu)^|x| / mu``; otherwise the textbook inverse of
    :func:`mulaw_encode`, ``x = 2 (q + 0.5) / mu - 1;  s = sign(x) ((1 + mu)^|x| - 1) / mu``."""
    mu = quantization_steps - 1
    q = np.asarray(quantized_signal).astype(float)
    if compat:
        x = (q / quantization_steps - 0.5) * 2.0
        return np.sign(x) * ((1 + mu) ** np.absolute(x)) / mu
    x = 2.0 * (q + 0.5) / mu - 1.0
    return np.sign(x) * ((1 + mu) ** np.absolute(x) - 1.0) / mu


def save_audio_file(filename, quantized_signal, quantization_steps: int = 256, format: str = "16bit_pcm",
                    sampling_rate: int = 48000, compat: bool = True):
    """tokens -> wav file (data.py:37-58): inverse mu-law, PCM scale, the mono signal duplicated into two channels."""
    from scipy.io import wavfile
    scale, dtype = _pcm_format(format, compat)
    s = mulaw_decode(quantized_signal, quantization_steps, compat) * scale
    if not compat:
        info = np.iinfo(dtype)
        s = np.clip(s, info.min, info.max)
    audio = np.repeat(s.reshape((-1, 1)).astype(dtype), 2, axis=1)
    wavfile.write(filename, sampling_rate, audio)


def onehot_pixel_image(quantized_signal_batch, quantization_steps: int = 256) -> np.ndarray:
    """(B, T) tokens -> (B, Q, 1, T) float32 one-hot image (data.py:61-68).  The engine also takes
    the tokens directly (WaveNet.forward_causal_block), which skips this 134 MB tensor at B=8,T=16k."""
    idx = np.asarray(quantized_signal_batch)
    B, T = idx.shape
    image = np.zeros((B, quantization_steps, 1, T), dtype=np.float32)
    b, t = np.meshgrid(np.arange(B), np.arange(T), indexing="ij")
    image[b.reshape(-1), idx.reshape(-1), 0, t.reshape(-1)] = 1
    return image


def create_batch(signal, batch_size, input_width, target_width, rng=np.random):
    """Random crops with next-sample targets (train_audio/train.py:14-22)."""
    starts = rng.randint(0, signal.size - target_width - input_width - 1, size=batch_size)
    x = np.empty((batch_size, input_width + target_width), dtype=np.int32)
    t = np.empty((batch_size, target_width), dtype=np.int32)
    for n, s in enumerate(starts):
        x[n] = signal[s:s + input_width + target_width]
        t[n] = signal[s + input_width + 1:s + input_width + target_width + 1]
    return x, t


def synthetic_waveform(B: int, n: int, sr: int, b0: int = 0, Btot=None) -> np.ndarray:
    """Synthetic clips for benchmarks: two sines + noise in [-1, 1], float64.
    Clip ``b`` uses phase ``2 pi (b0+b)/Btot`` and its own noise row, so data-parallel shards differ."""
    Btot = B if Btot is None else Btot
    t = np.arange(n, dtype=np.float64) / sr
    noise = np.random.RandomState(0).standard_normal((Btot, n))
    out = np.empty((B, n), dtype=np.float64)
    for b in range(B):
        ph = 2 * np.pi * (b0 + b) / Btot
        out[b] = 0.6 * np.sin(2 * np.pi * 220.0 * t + ph) + 0.3 * np.sin(2 * np.pi * 554.37 * t) + 0.05 * noise[b0 + b]
    return np.clip(out, -1.0, 1.0)
