"""Boundary data formats of the hot path (data.py of the reference): mu-law tokens and the
one-hot "1 x W image".  Host-side numpy; file I/O (wav load/save, silence trimming) is out of scope
for this round."""
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
