"""Generate the golden fixtures under tests/golden/ from the CPU oracle (oracle/).

The reference itself cannot run here (Python 2 + Chainer, neither present), so these vectors are
outputs of the build's own restatement -- PARITY UNPINNED against Chainer, see oracle/__init__.py.
They pin the oracle against drift and give the GPU tests fixed inputs/outputs that travel to the
GPU box.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import data_ref as D          # noqa: E402
from oracle import wavenet_ref as R       # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)

CFG1 = dict(quantization_steps=256, causal_conv_channels=[16], residual_conv_channels=[16] * 4,
            residual_num_blocks=1, softmax_conv_channels=[32, 256])


def kat1():
    x = np.mod(np.arange(40), 5).reshape(1, 4, 1, 10).astype(np.float32)
    W = np.ones((3, 4, 4, 1), np.float32)
    out = R.dilated_conv_literal(torch.tensor(x), torch.tensor(W), None, 4, 4).numpy()
    np.savez(os.path.join(OUT, "kat1_dilated_conv.npz"), x=x, W=W, out=out)


def cfg1_forward():
    p = R.make_params(**CFG1)
    w = R.init_weights(p, 1234)
    idx = np.random.RandomState(0).randint(0, 256, (1, 8000)).astype(np.int32)
    net = R.RefWaveNet(p, w)
    with torch.no_grad():
        x = R.onehot_t(idx, 256)
        c = net.forward_causal_block(x)
        rec = []
        o, s = net.forward_residual_block(c, record=rec)
        logits = net.forward_softmax_block(s, apply_softmax=False)
    cols = np.unique(np.concatenate([np.arange(0, 40), np.arange(7990, 8000),
                                     np.random.RandomState(1).randint(0, 8000, 250)]))
    np.savez_compressed(
        os.path.join(OUT, "cfg1_forward.npz"), idx=idx.astype(np.uint8), cols=cols.astype(np.int32),
        logits_cols=logits.numpy()[0, :, 0, :][:, cols], skip_cols=s.numpy()[0, :, 0, :][:, cols],
        out_cols=o.numpy()[0, :, 0, :][:, cols],
        layer_out_sum=np.array([float(r[0].double().sum()) for r in rec]),
        layer_skip_sum=np.array([float(r[1].double().sum()) for r in rec]),
        logits_sum=np.array(float(logits.double().sum())), logits_abs_sum=np.array(float(logits.double().abs().sum())))


def cfg1_train_step():
    p = R.make_params(**CFG1)
    w = R.init_weights(p, 1234)
    iw = R.input_width(p)
    tw = 600 - iw
    idx = np.random.RandomState(5).randint(0, 256, (2, 600)).astype(np.int32)
    tgt = np.random.RandomState(6).randint(0, 256, (2, tw)).astype(np.int32)
    loss, logits, g = R.train_step_grads(p, w, idx, tgt)
    np.savez_compressed(os.path.join(OUT, "cfg1_train_step.npz"), idx=idx.astype(np.uint8), target=tgt.astype(np.uint8),
                        loss=np.array(loss, np.float64), **{"grad:" + k: v for k, v in g.items()})


def fastgen_trace():
    p = R.make_params(quantization_steps=256, causal_conv_channels=[16], residual_conv_channels=[16] * 4,
                      residual_num_blocks=2, softmax_conv_channels=[32, 256])
    w = R.init_weights(p, 1234)
    u = np.random.RandomState(7).random_sample(64)
    for act in ("elu", "relu"):
        tr = []
        toks = R.generate(p, w, 64, u, fast=True, fast_head_act=act, trace=tr)
        np.savez_compressed(os.path.join(OUT, "fastgen_%s.npz" % act), tokens=toks.astype(np.int32),
                            probs=np.array(tr, np.float32), uniforms=u)


CFG2 = dict(quantization_steps=256, causal_conv_channels=[32], residual_conv_channels=[32] * 10,
            residual_num_blocks=4, softmax_conv_channels=[256, 256])


def cfg4_decode_trace(n=16000, margin=2e-5):
    """BASELINE config 4 (train_audio/generate.py:24-43 with --fast at config 2's 4 x 10 topology, window 4094): the
    oracle's literal queue-cached generation for all 16,000 samples the config names (round 2: 256 steps, round 3: 2,048;
    each is a prefix of the next -- the uniform stream and the replacement stream are consumed in order).  Weights: the product's own seeded initialisation
    (``WaveNet(Params, seed=1234)``, CPU only -- identical to the oracle's ``init_weights(p, 1234)``, asserted) so that
    bench.py can reproduce them without importing the oracle.  Uniforms start as ``RandomState(7).random_sample``; a draw
    that lands within ``margin`` of a boundary of the step's cumulative distribution is replaced by the next draw of a
    second stream, so that the committed token sequence does not hinge on the last bit of a probability."""
    from wavenet_amd import Params, WaveNet
    p = R.make_params(**CFG2)
    w = WaveNet(Params(p), seed=1234).state_dict()
    w0 = R.init_weights(p, 1234)
    assert all(np.array_equal(w[k], w0[k]) for k in w0) and len(w) == len(w0)
    iw = R.input_width(p)
    u = np.random.RandomState(7).random_sample(n)
    spare = np.random.RandomState(8)
    model = R.RefFasterWaveNet(p, w, "elu")
    buf = np.full((iw,), 127, dtype=np.int32)
    probs, replaced = [], 0
    for step in range(n):
        x = D.onehot_pixel_image(buf[-iw:].reshape(1, -1), 256)
        prob = model._forward_one_step(x, apply_softmax=True)[0, :, 0, -1]
        cdf = np.cumsum(prob.astype(np.float64)); cdf /= cdf[-1]
        while np.abs(cdf - u[step]).min() < margin:
            u[step] = spare.random_sample(); replaced += 1
        probs.append(prob.copy())
        buf = np.append(buf, [R.choice_from_uniform(prob, u[step])]).astype(np.int32)
    toks = buf[iw:]
    np.savez_compressed(os.path.join(OUT, "cfg4_decode_trace.npz"), tokens=toks.astype(np.uint8), uniforms=u,
                        probs_every8=np.array(probs[:2048:8], np.float32),        # the first 2,048 steps, as before
                        probs_every128=np.array(probs[::128], np.float32),       # the whole trace, thinner
                        margin=np.array(margin), replaced=np.array(replaced))
    print("cfg4 trace: %d draws replaced, token sum %d" % (replaced, int(toks.sum())))


def cfg2_bench_step(probes=384):
    """The step bench.py times (BASELINE config 2, train_audio/train.py:58-80): 4 x 10 layers, 32 / 256 channels, the
    bench's own batch -- ``bench.make_batch(0, 1, iw)``: 8 synthetic clips x 16,384 tokens, targets = the next sample, loss
    over the last 12,290 columns -- and the product's seeded initial weights (``WaveNet(Params, seed=1234)`` ==
    ``init_weights(p, 1234)``, asserted).  One oracle step (13.5 s, 7.4 GB with 8 threads): the loss, the logits and the
    skip sum at ``probes`` (clip, column) positions, and of every gradient tensor its L2 norm, its largest magnitude and
    its sum -- at the oracle's OWN ReLU mask (the GPU test that differentiates at the device's mask runs the oracle live).
    bench.py holds its first captured step against ``loss`` before the timed region (``golden_loss_match``)."""
    from wavenet_amd import Params, WaveNet, data
    p = R.make_params(**CFG2)
    w = WaveNet(Params(p), seed=1234).state_dict()
    w0 = R.init_weights(p, 1234)
    assert all(np.array_equal(w[k], w0[k]) for k in w0) and len(w) == len(w0)
    iw = R.input_width(p)
    B, T = 8, 16384
    tok = data.mulaw_encode(data.synthetic_waveform(B, T + 1, 16000, b0=0, Btot=B))
    assert np.array_equal(tok, D.mulaw_quantize(D.synthetic_waveform(B, T + 1, 16000)))      # product host code == oracle
    idx, tgt = tok[:, :T].astype(np.int32), tok[:, iw + 1:T + 1].astype(np.int32)
    keep = {}
    loss, logits, g = R.train_step_grads(p, w, idx, tgt, keep=keep)
    Tw = T - iw
    rs = np.random.RandomState(11)
    pb = rs.randint(0, B, probes).astype(np.int32)
    pt = np.concatenate([[0, 1, Tw - 2, Tw - 1], rs.randint(0, Tw, probes - 4)]).astype(np.int32)
    names = sorted(g)
    np.savez_compressed(
        os.path.join(OUT, "cfg2_bench_step.npz"), loss=np.array(loss, np.float64),
        tokens_checksum=np.array([int(idx.astype(np.int64).sum()), int(tgt.astype(np.int64).sum())]),
        probe_b=pb, probe_t=pt, logits_probes=logits[pb, :, 0, pt].astype(np.float32),
        skip_probes=keep["skip"][pb, :, 0, pt].astype(np.float32),
        logits_sum=np.array(float(logits.astype(np.float64).sum())), logits_abs_sum=np.array(float(np.abs(logits.astype(np.float64)).sum())),
        relu_live=np.array(int((keep["skip"] > 0).sum())),
        grad_names=np.array(names), grad_l2=np.array([float(np.sqrt((g[k].astype(np.float64) ** 2).sum())) for k in names]),
        grad_absmax=np.array([float(np.abs(g[k]).max()) for k in names]),
        grad_sum=np.array([float(g[k].astype(np.float64).sum()) for k in names]))
    print("cfg2 bench step: loss %.9f" % loss)


def mulaw_table():
    q = D.mulaw_quantize_pcm16(np.arange(-32768, 32768))
    np.savez_compressed(os.path.join(OUT, "mulaw_pcm16.npz"), table=q.astype(np.uint8))


if __name__ == "__main__":
    only = sys.argv[1:]
    for fn in (kat1, cfg1_forward, cfg1_train_step, fastgen_trace, mulaw_table, cfg4_decode_trace, cfg2_bench_step):
        if not only or fn.__name__ in only:
            fn()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
