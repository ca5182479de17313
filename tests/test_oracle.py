"""The oracle checked against itself and against what can be pinned."""
import numpy as np
import pytest
import torch

from oracle import data_ref as D
from oracle import wavenet_ref as R

TINY = dict(quantization_steps=16, causal_conv_channels=[8], residual_conv_channels=[8, 8, 8],
            residual_num_blocks=2, softmax_conv_channels=[12, 16])


def test_kat1_dilated_conv_hand_derived():
    # _tests_/dilated_conv/test_conv.py:8-16 under the current semantics (d = fw**1 = 4)
    cin, cout, fw, T = 4, 3, 4, 10
    x = np.mod(np.arange(cin * T), 5).reshape(1, cin, 1, T).astype(np.float32)
    W = np.ones((cout, cin, fw, 1), np.float32)
    want = np.array([0, 0, 0, 0, 0, 0, 12, 20, 28, 20], np.float32)
    lit = R.dilated_conv_literal(torch.tensor(x), torch.tensor(W), None, 4, fw).numpy()
    clo = R.dilated_conv_closed(x, W, None, 4, fw)
    assert R.conv_pad_and_prefix(T, 4, fw) == (6, 6)
    for o in range(cout):
        np.testing.assert_array_equal(lit[0, o, 0], want)
        np.testing.assert_array_equal(clo[0, o, 0], want)


@pytest.mark.parametrize("B,C,O,T,d,fw", [
    (1, 3, 2, 16, 2, 2), (2, 4, 5, 17, 4, 2), (1, 2, 3, 7, 8, 2), (2, 3, 3, 100, 9, 3),
    (1, 5, 4, 26, 27, 3), (1, 2, 2, 33, 16, 2), (3, 2, 2, 64, 1, 2), (1, 2, 2, 5, 1, 3),
    (1, 4, 4, 600, 8, 2), (1, 4, 4, 513, 512, 2), (1, 2, 2, 40, 16, 4), (1, 2, 2, 3, 4, 2)])
def test_kat2_closed_form_equals_literal(B, C, O, T, d, fw):
    rs = np.random.RandomState(B * 1000 + T)
    x = rs.standard_normal((B, C, 1, T))
    W = rs.standard_normal((O, C, 1, fw) if d == 1 else (O, C, fw, 1))
    b = rs.standard_normal((O,))
    for bias in (None, b):
        lit = R.dilated_conv_literal(torch.tensor(x), torch.tensor(W),
                                     None if bias is None else torch.tensor(bias), d, fw).numpy()
        clo = R.dilated_conv_closed(x, W, bias, d, fw)
        assert lit.shape == (B, O, 1, T)
        np.testing.assert_allclose(clo, lit, rtol=0, atol=1e-12)
    pad, Z = R.conv_pad_and_prefix(T, d, fw)
    assert np.all(lit[..., :Z] == 0)


def test_zero_prefix_values():
    assert R.conv_pad_and_prefix(16384, 512, 2) == (0, 512)
    assert R.conv_pad_and_prefix(16000, 512, 2) == (384, 128)
    assert R.conv_pad_and_prefix(8000, 8, 2) == (0, 8)
    assert R.conv_pad_and_prefix(7, 8, 2) == (9, 0)
    assert R.conv_pad_and_prefix(4094, 512, 2) == (2, 510)


def test_kat3_onehot():
    idx = np.random.RandomState(0).randint(0, 256, (3, 50)).astype(np.int32)
    img = D.onehot_pixel_image(idx, 256)
    assert img.shape == (3, 256, 1, 50) and img.dtype == np.float32
    q = np.arange(256).reshape(1, 256, 1, 1)
    np.testing.assert_array_equal(img, (q == idx.reshape(3, 1, 1, 50)).astype(np.float32))


def test_kat4_mulaw_exhaustive():
    v = np.arange(-32768, 32768)
    q = D.mulaw_quantize_pcm16(v)
    assert q.dtype == np.int32 and q.min() == 0 and q.max() <= 255
    s = v / 32768.0
    want = np.floor(np.clip(0.5 * np.sign(s) * np.log1p(255 * np.abs(s)) / np.log(256.0) + 0.5, 0, 1) * 255)
    # log1p vs log(1+x) may differ in the last bit only at a truncation boundary
    assert np.count_nonzero(q != want) <= 2
    assert D.mulaw_quantize(np.array([0.0]))[0] == 127
    assert D.mulaw_quantize(np.array([1.0]))[0] == 255
    assert D.mulaw_quantize(np.array([-1.0]))[0] == 0
    assert np.all(np.diff(q) >= 0)


def test_kat5_receptive_field():
    p = R.make_params(residual_conv_channels=[32] * 10, residual_num_blocks=4, causal_conv_channels=[32])
    assert R.receptive_field(p) == 4093 and R.input_width(p) == 4094
    p = R.make_params(residual_conv_channels=[16] * 4, residual_num_blocks=1, causal_conv_channels=[16])
    assert R.receptive_field(p) == 16 and R.input_width(p) == 17


def test_weight_specs_shapes_and_count():
    p = R.make_params(residual_conv_channels=[32] * 10, residual_num_blocks=4, causal_conv_channels=[32],
                      softmax_conv_channels=[256, 256])
    sd = R.init_weights(p)
    assert sum(v.size for v in sd.values()) == 614656 + 0  # SURVEY section 8 A9 (bias 256 included)
    assert sd["residual_0_block_0_wf/W"].shape == (32, 32, 1, 2)
    assert sd["residual_0_block_1_wf/W"].shape == (32, 32, 2, 1)
    assert sd["softmax_0/b"].shape == (256,)


@pytest.mark.parametrize("bias", [False, True])
def test_literal_model_equals_closed_model(bias):
    p = R.make_params(**TINY)
    if bias:
        p.update(causal_conv_no_bias=False, residual_conv_dilation_no_bias=False,
                 residual_conv_projection_no_bias=False)
    w = R.init_weights(p, 5, bias_scale=0.3)
    idx = np.random.RandomState(1).randint(0, 16, (2, 45)).astype(np.int32)
    x = D.onehot_pixel_image(idx, 16)
    net = R.RefWaveNet(p, w, dtype=torch.float64)
    xt = torch.tensor(x, dtype=torch.float64)
    c = net.forward_causal_block(xt)
    o, s = net.forward_residual_block(c)
    h = net.forward_softmax_block(s, apply_softmax=True)
    cc, oc, sc, hc = R.forward_closed(p, w, x, dtype=np.float64, apply_softmax=True)
    for a, b_ in ((c, cc), (o, oc), (s, sc), (h, hc)):
        np.testing.assert_allclose(a.numpy(), b_, rtol=0, atol=1e-12)
    # without the zero-prefix quirk the early columns differ, late columns do not
    _, _, s2, _ = R.forward_closed(p, w, x, dtype=np.float64, compat_zero_prefix=False)
    rf = R.receptive_field(p)
    assert np.abs(s2 - sc)[..., :4].max() > 1e-6
    np.testing.assert_allclose(s2[..., rf:], sc[..., rf:], atol=1e-12)


def test_sampler_equals_numpy_choice():
    # the reference's call is np.random.choice(np.arange(Q), p=float32 softmax) (generate.py:39)
    rs = np.random.RandomState(3)
    for trial in range(500):
        logits = (rs.standard_normal(256) * 3).astype(np.float32)
        p = R.softmax_axis1(logits.reshape(1, 256, 1, 1))[0, :, 0, 0]
        assert p.dtype == np.float32
        want = np.random.RandomState(trial).choice(np.arange(256), p=p)
        u = np.random.RandomState(trial).random_sample()
        assert R.choice_from_uniform(p, u) == want


@pytest.mark.parametrize("fw,act", [(2, "relu"), (2, "elu"), (3, "relu")])
def test_kat6_fast_equals_slow(fw, act):
    p = R.make_params(quantization_steps=12, causal_conv_channels=[6], residual_conv_channels=[5, 5, 5],
                      residual_num_blocks=2, softmax_conv_channels=[7, 12],
                      residual_conv_filter_width=fw, causal_conv_filter_width=fw)
    w = R.init_weights(p, 11, bias_scale=0.2)
    u = np.random.RandomState(7).random_sample(12)
    tr_f, tr_s = [], []
    out_f = R.generate(p, w, 12, u, fast=True, fast_head_act=act, trace=tr_f)
    # slow path with the same head activation after the first step
    iw = R.input_width(p)
    buf = np.full((iw,), 6, np.int32)
    net = R.RefWaveNet(p, w)
    for step in range(12):
        x = torch.tensor(D.onehot_pixel_image(buf[-iw:].reshape(1, -1), 12))
        c = net.forward_causal_block(x)
        _, s = net.forward_residual_block(c)
        pr = net.forward_softmax_block(s, True, act="relu" if step == 0 else act).numpy()[0, :, 0, -1]
        tr_s.append(pr)
        buf = np.append(buf, [R.choice_from_uniform(pr, u[step])]).astype(np.int32)
    np.testing.assert_array_equal(out_f, buf[iw:])
    np.testing.assert_allclose(np.array(tr_f), np.array(tr_s), atol=2e-6)
    if act == "elu":   # F4: ELU head and ReLU head really differ
        tr_r = []
        R.generate(p, w, 3, u, fast=True, fast_head_act="relu", trace=tr_r)
        assert np.abs(np.array(tr_r)[1:] - np.array(tr_f)[1:3]).max() > 1e-4


def test_train_step_alignment_and_grads_fd():
    p = R.make_params(**TINY)
    w = R.init_weights(p, 2)
    iw = R.input_width(p)
    sig = np.random.RandomState(4).randint(0, 16, 300).astype(np.int32)
    sig = np.insert(sig, 0, np.full((iw,), 7, np.int32))          # train.py:53
    x, t = D.create_batch(sig, np.array([3, 50]), iw, 20)
    assert x.shape == (2, iw + 20) and t.shape == (2, 20)
    np.testing.assert_array_equal(t[:, :-1], x[:, iw + 1:])        # column iw+j predicts sample iw+j+1
    loss, logits, g = R.train_step_grads(p, w, x, t, dtype=torch.float64)
    assert logits.shape == (2, 16, 1, 20)
    # last layer's projection_block gets no gradient
    assert np.all(g["residual_1_block_2_projection_block/W"] == 0)
    # finite differences on a few entries
    rs = np.random.RandomState(0)
    for name in ("causal_0/W", "residual_0_block_1_wf/W", "residual_1_block_0_projection_softmax/W", "softmax_0/b"):
        for _ in range(3):
            i = tuple(rs.randint(0, s) for s in w[name].shape)
            wp = {k: v.astype(np.float64).copy() for k, v in w.items()}
            wm = {k: v.astype(np.float64).copy() for k, v in w.items()}
            wp[name][i] += 1e-5
            wm[name][i] -= 1e-5
            net = R.RefWaveNet(p, wp, dtype=torch.float64)
            lp = float(net.train_loss(R.onehot_t(x, 16, torch.float64), t)[0])
            net = R.RefWaveNet(p, wm, dtype=torch.float64)
            lm = float(net.train_loss(R.onehot_t(x, 16, torch.float64), t)[0])
            assert abs((lp - lm) / 2e-5 - g[name][i]) < 1e-7


def test_bf16_storage_restatement_hand_backward_equals_autograd():
    """oracle/bf16_ref.py (the checker of the bf16-storage kernels, BASELINE config 5): the numpy forward + hand-written
    backward and the torch-autograd formulation with straight-through rounding agree; and the bf16 arithmetic stays
    within bf16 distance of the fp32 restatement."""
    from oracle import bf16_ref as Q
    p = R.make_params(quantization_steps=32, causal_conv_channels=[16], residual_conv_channels=[16] * 3,
                      residual_num_blocks=2, softmax_conv_channels=[24, 32])
    w = R.init_weights(p, 3)
    w["softmax_0/b"] = (np.random.RandomState(2).standard_normal(32) * 0.3).astype(np.float32)
    rs = np.random.RandomState(4)
    iw = R.input_width(p)
    idx = rs.randint(0, 32, (2, iw + 21)).astype(np.int32)
    tgt = rs.randint(0, 32, (2, 21)).astype(np.int32)
    keep = {}
    l1, lg1, g1 = Q.train_step(p, w, idx, tgt, keep=keep)
    l2, lg2, g2 = Q.train_step_autograd(p, w, idx, tgt)
    l0, lg0, g0 = R.train_step_grads(p, w, idx, tgt)
    assert abs(l1 - l2) < 1e-6 and abs(l1 - l0) < 2e-2
    np.testing.assert_allclose(lg1, lg2, atol=1e-5)
    for k in g1:
        n = np.linalg.norm(g2[k]) + 1e-12
        assert np.linalg.norm(g1[k] - g2[k]) <= 3e-3 * n, (k, np.linalg.norm(g1[k] - g2[k]) / n)
        assert np.linalg.norm(g1[k] - g0[k]) <= 0.1 * (np.linalg.norm(g0[k]) + 1e-12), k
    # stored tensors hold bf16 values
    for t in [keep["x0"], keep["skip"], keep["dskip"]] + keep["zs"] + keep["xs"] + keep["dadg"] + keep["dx"] + keep["dzs"]:
        np.testing.assert_array_equal(Q.rb(t), t)
