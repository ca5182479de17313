"""CPU: wav load / silence trim / inverse mu-law / wav save against the oracle's literal
restatement of data.py, on files written and read with scipy.io.wavfile as the reference does."""
import os
import warnings

import numpy as np
import pytest
from scipy.io import wavfile

from oracle import data_ref as D
from wavenet_amd import data


def _signal(n=6000, seed=0, silent_head=300, silent_tail=500):
    rs = np.random.RandomState(seed)
    t = np.arange(n) / 16000.0
    s = 0.5 * np.sin(2 * np.pi * 330 * t) + 0.05 * rs.standard_normal(n)
    s[:silent_head] = 0.0
    s[n - silent_tail:] = 0.0
    return np.clip(s, -1, 1)


@pytest.mark.parametrize("stereo", [True, False])
def test_load_matches_the_literal_restatement(tmp_path, stereo):
    pcm = (_signal() * 32767).astype(np.int16)
    arr = np.stack([pcm, pcm[::-1]], axis=1) if stereo else pcm          # the right channel must be ignored
    f = str(tmp_path / "a.wav")
    wavfile.write(f, 16000, arr)
    q, sr = data.load_audio_file(f)
    assert sr == 16000 and q.dtype == np.int32
    np.testing.assert_array_equal(q, D.load_audio_ref(arr))
    # silence (tokens 126..128) is gone from both ends; the reference's slice also drops the last loud sample
    assert abs(int(q[0]) - 127) > 1
    full = data.mulaw_encode(pcm.astype(float) / 32768)
    loud = np.nonzero(np.abs(full - 127) > 1)[0]
    np.testing.assert_array_equal(q, full[loud[0]:loud[-1]])
    np.testing.assert_array_equal(data.load_audio_file(f, compat=False)[0], full[loud[0]:loud[-1] + 1])


def test_mono_python2_integer_division_quirk(tmp_path):
    pcm = (_signal(2000, 1, 0, 0) * 32767).astype(np.int16)
    f = str(tmp_path / "m.wav")
    wavfile.write(f, 8000, pcm)
    q, _ = data.load_audio_file(f, compat_mono_int_division=True)
    np.testing.assert_array_equal(q, D.load_audio_ref(pcm, py2_mono_int_division=True))
    assert set(np.unique(q)) <= {0, 127}                                   # -1 -> token 0, 0 -> token 127


@pytest.mark.parametrize("q", [np.full(50, 127), np.array([127, 127, 3, 127]), np.array([5]), np.array([127, 9]),
                               np.array([9, 127]), np.array([], dtype=np.int64)])
def test_trim_silence_edge_cases_follow_the_reference_loops(q):
    def ref(q):
        if q.size == 0:
            return q
        start = 0
        for start in range(q.size):
            if abs(int(q[start]) - 127) > 1:
                break
        end = 1
        for end in range(1, q.size):
            if abs(int(q[-end]) - 127) > 1:
                break
        return q[start:-end]
    np.testing.assert_array_equal(data.trim_silence(q), ref(q))


@pytest.mark.parametrize("fmt", ["16bit_pcm", "32bit_pcm", "8bit_pcm"])
def test_save_writes_what_the_reference_would(tmp_path, fmt):
    tok = np.random.RandomState(3).randint(0, 256, 4000)
    f = str(tmp_path / "o.wav")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = D.save_audio_ref(tok, 256, fmt)
        data.save_audio_file(f, tok, 256, fmt, sampling_rate=16000)
    sr, got = wavfile.read(f)
    assert sr == 16000 and got.shape == (4000, 2)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(got[:, 0], got[:, 1])                  # the mono signal, twice


def test_textbook_decode_inverts_encode_and_compat_decode_does_not():
    tok = np.arange(256)
    s = data.mulaw_decode(tok, 256, compat=False)
    np.testing.assert_array_equal(data.mulaw_encode(s), tok)              # bin centres map back to their bin
    c = data.mulaw_decode(tok, 256, compat=True)
    assert c[128] == 0.0 and np.all(np.abs(np.delete(c, 128)) >= 1.0 / 255 - 1e-12)   # the missing "- 1": a floor of 1/mu
    assert np.mean(data.mulaw_encode(c) == tok) < 0.9                     # and the round trip is lossy


def test_round_trip_through_files(tmp_path):
    pcm = (_signal(8000, 2) * 32767).astype(np.int16)
    f1, f2 = str(tmp_path / "in.wav"), str(tmp_path / "out.wav")
    wavfile.write(f1, 16000, np.stack([pcm, pcm], axis=1))
    q, sr = data.load_audio_file(f1, compat=False)
    data.save_audio_file(f2, q, sampling_rate=sr, compat=False)
    q2, _ = data.load_audio_file(f2, compat=False)
    n = min(q.size, q2.size)
    assert np.abs(q[:n].astype(int) - q2[:n].astype(int)).max() <= 1      # one 16-bit rounding in between
