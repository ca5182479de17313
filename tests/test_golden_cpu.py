"""CPU: the oracle reproduces the committed golden fixtures (guards the checker against drift)."""
import os

import numpy as np
import torch

from oracle import data_ref as D
from oracle import wavenet_ref as R

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CFG1 = dict(quantization_steps=256, causal_conv_channels=[16], residual_conv_channels=[16] * 4,
            residual_num_blocks=1, softmax_conv_channels=[32, 256])


def test_kat1_fixture():
    z = np.load(os.path.join(G, "kat1_dilated_conv.npz"))
    np.testing.assert_array_equal(z["out"][0, 0, 0], [0, 0, 0, 0, 0, 0, 12, 20, 28, 20])
    np.testing.assert_array_equal(R.dilated_conv_closed(z["x"], z["W"], None, 4, 4), z["out"])


def test_cfg1_forward_fixture_closed_form():
    z = np.load(os.path.join(G, "cfg1_forward.npz"))
    p = R.make_params(**CFG1)
    w = R.init_weights(p, 1234)
    idx = z["idx"].astype(np.int32)
    np.testing.assert_array_equal(idx, np.random.RandomState(0).randint(0, 256, (1, 8000)))
    keep = []
    _, o, s, h = R.forward_closed(p, w, D.onehot_pixel_image(idx, 256), keep=keep)
    cols = z["cols"]
    np.testing.assert_allclose(h[0, :, 0, :][:, cols], z["logits_cols"], atol=2e-5)
    np.testing.assert_allclose(s[0, :, 0, :][:, cols], z["skip_cols"], atol=2e-5)
    np.testing.assert_allclose(o[0, :, 0, :][:, cols], z["out_cols"], atol=2e-5)
    np.testing.assert_allclose([float(k[0].astype(np.float64).sum()) for k in keep], z["layer_out_sum"], rtol=1e-4, atol=1e-2)
    # F3: the zero prefix is visible in the fixture (layer with d=8 leaves skip columns < 8 bias-free zero taps)
    assert 0 in cols and 7 in cols


def test_cfg1_train_fixture():
    z = np.load(os.path.join(G, "cfg1_train_step.npz"))
    p = R.make_params(**CFG1)
    w = R.init_weights(p, 1234)
    loss, _, g = R.train_step_grads(p, w, z["idx"].astype(np.int32), z["target"].astype(np.int32))
    assert abs(loss - float(z["loss"])) < 1e-5
    for k, v in g.items():
        np.testing.assert_allclose(v, z["grad:" + k], atol=1e-6, rtol=1e-4)


def test_fastgen_fixture():
    p = R.make_params(quantization_steps=256, causal_conv_channels=[16], residual_conv_channels=[16] * 4,
                      residual_num_blocks=2, softmax_conv_channels=[32, 256])
    w = R.init_weights(p, 1234)
    for act in ("elu", "relu"):
        z = np.load(os.path.join(G, "fastgen_%s.npz" % act))
        tr = []
        toks = R.generate(p, w, 16, z["uniforms"], fast=True, fast_head_act=act, trace=tr)
        np.testing.assert_array_equal(toks, z["tokens"][:16])
        np.testing.assert_allclose(np.array(tr), z["probs"][:16], atol=1e-6)
    a, b = np.load(os.path.join(G, "fastgen_elu.npz")), np.load(os.path.join(G, "fastgen_relu.npz"))
    np.testing.assert_allclose(a["probs"][0], b["probs"][0], atol=1e-7)      # first step: ReLU path in both
    assert np.abs(a["probs"][1:] - b["probs"][1:]).max() > 1e-5              # then ELU vs ReLU (F4)


def test_mulaw_fixture():
    z = np.load(os.path.join(G, "mulaw_pcm16.npz"))
    np.testing.assert_array_equal(D.mulaw_quantize_pcm16(np.arange(-32768, 32768)), z["table"].astype(np.int32))


def test_cfg4_decode_fixture_first_steps_and_weights():
    """The config-4 trace (4 x 10 layers, window 4094, all 16,000 samples BASELINE configs[3] names): its weights are the
    product's seeded CPU initialisation, equal to the oracle's; the oracle reproduces its first tokens; every stored
    uniform keeps the stated margin; the number of margin-replaced draws is what the margin predicts (a draw lands within
    2e-5 of one of 255 inner CDF boundaries with probability ~ 255 x 2 x 2e-5 = 1.0 %) and must not grow."""
    from wavenet_amd import Params, WaveNet
    z = np.load(os.path.join(G, "cfg4_decode_trace.npz"))
    p = R.make_params(quantization_steps=256, causal_conv_channels=[32], residual_conv_channels=[32] * 10,
                      residual_num_blocks=4, softmax_conv_channels=[256, 256])
    w = R.init_weights(p, 1234)
    sd = WaveNet(Params(p), seed=1234).state_dict()
    assert set(sd) == set(w) and all(np.array_equal(sd[k], w[k]) for k in w)
    tr = []
    toks = R.generate(p, w, 3, z["uniforms"], fast=True, fast_head_act="elu", trace=tr)
    np.testing.assert_array_equal(toks, z["tokens"][:3].astype(np.int32))
    np.testing.assert_allclose(tr[0], z["probs_every8"][0], atol=1e-6)
    assert z["tokens"].shape == (16000,) and z["uniforms"].shape == (16000,)
    assert z["probs_every8"].shape == (256, 256) and z["probs_every128"].shape == (125, 256)
    np.testing.assert_array_equal(z["probs_every8"][::16], z["probs_every128"][:16])       # the two thinnings agree where they meet
    assert int(z["replaced"]) <= 0.0102 * 16000 + 3 * np.sqrt(0.0102 * 16000)              # expectation + 3 sigma of the margin rule
    assert int(z["tokens"].astype(np.int64).sum()) == 1991402
    cdf = np.cumsum(tr[2].astype(np.float64)); cdf /= cdf[-1]
    assert np.abs(cdf - z["uniforms"][2]).min() >= float(z["margin"])


def test_cfg2_bench_step_fixture():
    """tests/golden/cfg2_bench_step.npz (the step bench.py times, train_audio/train.py:58-80 at BASELINE config 2's size): the
    tokens the PRODUCT's host code makes for the bench batch are the ones the fixture was computed on, the oracle reproduces
    the fixture's loss / logits probes / gradient norms (one full oracle step, ~20 s and 7.4 GB)."""
    from wavenet_amd import Params, WaveNet, data
    z = np.load(os.path.join(G, "cfg2_bench_step.npz"))
    p = R.make_params(quantization_steps=256, causal_conv_channels=[32], residual_conv_channels=[32] * 10,
                      residual_num_blocks=4, softmax_conv_channels=[256, 256])
    w = R.init_weights(p, 1234)
    sd = WaveNet(Params(p), seed=1234).state_dict()
    assert set(sd) == set(w) and all(np.array_equal(sd[k], w[k]) for k in w)
    iw = R.input_width(p)
    tok = data.mulaw_encode(data.synthetic_waveform(8, 16385, 16000, b0=0, Btot=8))
    idx, tgt = tok[:, :16384].astype(np.int32), tok[:, iw + 1:16385].astype(np.int32)
    assert [int(idx.astype(np.int64).sum()), int(tgt.astype(np.int64).sum())] == z["tokens_checksum"].tolist()
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    keep = {}
    loss, logits, g = R.train_step_grads(p, w, idx, tgt, keep=keep)
    assert abs(loss - float(z["loss"])) < 2e-6
    pb, pt = z["probe_b"], z["probe_t"]
    np.testing.assert_allclose(logits[pb, :, 0, pt], z["logits_probes"], atol=2e-6)
    np.testing.assert_allclose(keep["skip"][pb, :, 0, pt], z["skip_probes"], atol=2e-6)
    assert int((keep["skip"] > 0).sum()) == int(z["relu_live"])
    names = [str(k) for k in z["grad_names"]]
    assert sorted(g) == names
    for k, l2 in zip(names, z["grad_l2"]):
        assert abs(np.sqrt((g[k].astype(np.float64) ** 2).sum()) - l2) <= 1e-5 * l2 + 1e-12, k
