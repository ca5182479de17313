"""CPU restatement of the boundary data formats (data.py, train_audio/train.py).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
from __future__ import annotations

import numpy as np


def mulaw_quantize(signal: np.ndarray, quantization_steps: int = 256) -> np.ndarray:
    """data.py:19-23: mu-law compand a float64 signal in [-1,1], then truncate to
    ``int32((clip(0.5 s + 0.5, 0, 1)) * mu)``."""
    mu = quantization_steps - 1
    s = np.asarray(signal, dtype=np.float64)
    s = np.sign(s) * np.log(1 + mu * np.absolute(s)) / np.log(1 + mu)
    return (np.clip(s * 0.5 + 0.5, 0, 1) * mu).astype(np.int32)


def mulaw_quantize_pcm16(pcm: np.ndarray, quantization_steps: int = 256) -> np.ndarray:
    """16-bit PCM path of data.py:11-23 (``signal /= 1<<15`` then the above)."""
    return mulaw_quantize(np.asarray(pcm, dtype=np.float64) / float(1 << 15), quantization_steps)


def onehot_pixel_image(idx: np.ndarray, quantization_steps: int = 256) -> np.ndarray:
    """data.py:61-68: (B,T) int -> (B,Q,1,T) float32 with a single 1 per column."""
    idx = np.asarray(idx)
    B, T = idx.shape
    img = np.zeros((B * T, quantization_steps), dtype=np.float32)
    img[np.arange(B * T), idx.reshape(-1)] = 1
    return np.ascontiguousarray(img.reshape(B, T, quantization_steps, 1).transpose(0, 2, 3, 1))


def create_batch(signal: np.ndarray, starts: np.ndarray, input_width: int, target_width: int):
    """train_audio/train.py:14-22 with the random crop starts supplied by the
    caller: input = sig[s : s+iw+tw], target = sig[s+iw+1 : s+iw+tw+1]."""
    B = len(starts)
    x = np.empty((B, input_width + target_width), dtype=np.int32)
    t = np.empty((B, target_width), dtype=np.int32)
    for n, s in enumerate(starts):
        x[n] = signal[s:s + input_width + target_width]
        t[n] = signal[s + input_width + 1:s + input_width + target_width + 1]
    return x, t


def synthetic_waveform(B: int, n: int, sr: int, b0: int = 0, Btot: int | None = None) -> np.ndarray:
    """SURVEY.md section 8(d) synthetic clips: two sines + noise, float64 in [-1,1].
    Clip ``b`` uses phase ``2 pi (b0+b)/Btot`` so data-parallel shards differ."""
    Btot = B if Btot is None else Btot
    t = np.arange(n, dtype=np.float64) / sr
    rs = np.random.RandomState(0)
    noise = rs.standard_normal((Btot, n))
    out = np.empty((B, n), dtype=np.float64)
    for b in range(B):
        ph = 2 * np.pi * (b0 + b) / Btot
        out[b] = 0.6 * np.sin(2 * np.pi * 220.0 * t + ph) + 0.3 * np.sin(2 * np.pi * 554.37 * t) \
            + 0.05 * noise[b0 + b]
    return np.clip(out, -1.0, 1.0)


# --------------------------------------------------------------------------
# wav side of the path (data.py:5-58), restated on arrays: literal loops, Python-2 semantics spelled out
# --------------------------------------------------------------------------

def load_audio_ref(signal: np.ndarray, quantization_steps: int = 256, fmt: str = "16bit_pcm",
                   py2_mono_int_division: bool = False) -> np.ndarray:
    """What data.py:5-35 does to the array wavfile.read returned (TEST INFRASTRUCTURE ONLY)."""
    if len(signal.shape) > 1:
        signal = signal[:, 0].astype(float)                       # data.py:7-9
    mx = {"16bit_pcm": 1 << 15, "32bit_pcm": 1 << 31, "8bit_pcm": 1 << 8 - 1}[fmt]   # data.py:11-16 (1<<7 for 8 bit)
    if np.issubdtype(signal.dtype, np.integer):
        # data.py:17 on an integer array: Python 2 `/=` is floor division; the sane reading converts first
        signal = (signal.astype(np.int64) // mx).astype(float) if py2_mono_int_division else signal.astype(float) / mx
    else:
        signal = signal / mx
    q = mulaw_quantize(signal, quantization_steps)                # data.py:18-23
    start = 0
    for start in range(q.size):                                   # data.py:27-29
        if abs(int(q[start]) - 127) > 1:
            break
    end = None
    for end in range(1, q.size):                                  # data.py:30-32
        if abs(int(q[-end]) - 127) > 1:
            break
    if end is None:
        end = 1
    return q[start:-end]                                          # data.py:33


def save_audio_ref(quantized_signal: np.ndarray, quantization_steps: int = 256, fmt: str = "16bit_pcm") -> np.ndarray:
    """The (N, 2) integer array data.py:37-58 hands to wavfile.write (TEST INFRASTRUCTURE ONLY)."""
    q = quantized_signal.astype(float)
    normalized = (q / quantization_steps - 0.5) * 2.0             # data.py:39
    mu = quantization_steps - 1
    s = np.sign(normalized) * ((1 + mu) ** np.absolute(normalized)) / mu      # data.py:43
    mx, ty = {"16bit_pcm": (1 << 15, np.int16), "32bit_pcm": (1 << 31, np.int32), "8bit_pcm": (1 << 8 - 1, np.uint8)}[fmt]
    s = s * mx
    audio = s.reshape((-1, 1)).astype(ty)
    return np.repeat(audio, 2, axis=1)                            # data.py:56-57
