"""GPU parity on BASELINE.json's own configurations (the shapes bench.py times), against the CPU oracle.

config 2: 4 blocks x 10 dilations, 32 residual / 256 skip channels -- one train step, loss and EVERY gradient, launched op
          by op and through the replayed TrainStepGraph (train_audio/train.py:58-80);
config 4: the queue-cached decoder at that topology, window 4094: 256 generated tokens bit-exact against the committed
          oracle trace tests/golden/cfg4_decode_trace.npz (train_audio/generate.py:24-43);
config 3: see tests/test_gpu_dp.py (two ranks sharing the GPU).
"""
import os

import numpy as np
import pytest
import torch

from oracle import wavenet_ref as R
from wavenet_amd import FasterWaveNet, Params, TrainStepGraph, WaveNet

from gpu_util import CFG2, build, dev, to_np

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _check_grads(net, g, rel, tag):
    worst = 0.0
    for ln, kind, off, n, shape in net._spans:
        want = g["%s/%s" % (ln.name, kind)]
        got = to_np(net._grad_arena[off:off + n].view(shape))
        scale = max(float(np.abs(want).max()), 1e-6)
        err = float(np.abs(got - want).max())
        worst = max(worst, err / scale)
        assert err <= rel * scale + 1e-7, (tag, ln.name, kind, err, scale)
    return worst


@pytest.mark.parametrize("B,extra", [(2, 200), (1, 333)])
def test_cfg2_topology_train_step_loss_and_every_gradient(B, extra):
    """The stack the bench times (4 x 10 layers, d = 1..512, 1280-deep skip contraction, the 40-layer partial-tile
    reduction, the chained backward's live ranges over four blocks) at T = input_width + extra: loss within 1e-4, every
    gradient tensor within 1e-4 of its largest entry, against the oracle's autograd -- eager and graph replay."""
    p, w, net = build(CFG2)
    iw = R.input_width(p)
    assert iw == 4094 and len(net._flat_layers) == 40
    T = iw + extra
    rs = np.random.RandomState(17 + extra)
    idx = rs.randint(0, 256, (B, T)).astype(np.int32)
    tgt = rs.randint(0, 256, (B, extra)).astype(np.int32)
    loss_ref, logits_ref, g = R.train_step_grads(p, w, idx, tgt)
    x, t = dev(idx), dev(tgt)
    # op by op
    c = net.forward_causal_block(x)
    _, s = net.forward_residual_block(c, t_off=T - extra)
    logits = net.forward_softmax_block(s, apply_softmax=False)
    loss = net.cross_entropy(logits, t)
    net.zero_grads()
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - loss_ref) < 1e-4
    np.testing.assert_allclose(to_np(logits), logits_ref, atol=1e-4)
    worst = _check_grads(net, g, 1e-4, "eager")
    assert float(net.residual_blocks[-1][-1].projection_block.W.grad.abs().sum()) == 0
    # the same step captured and replayed (the gradient arena is what the optimiser graph consumed)
    net.update_laerning_rate(1e-4)
    w0 = to_np(net._arena).copy()
    gr = TrainStepGraph(net, x, t)
    np.testing.assert_array_equal(to_np(net._arena), w0)
    lg = gr.step()
    torch.cuda.synchronize()
    assert abs(float(lg) - loss_ref) < 1e-4
    _check_grads(net, g, 1e-4, "graph")
    assert np.abs(to_np(net._arena) - w0).max() > 0          # and the optimiser half of the graph ran
    assert worst < 1e-4


def test_cfg4_decoder_256_steps_match_the_committed_oracle_trace():
    """cfg4: 4 x 10 layers, window 4094, ELU head after the first (ReLU) step: 256 tokens bit-exact, probabilities of
    every eighth step within 2e-5 of the oracle's.  The fixture's uniforms keep 2e-5 of distance to every boundary of
    the oracle's cumulative distributions (tests/golden/make_golden.py::cfg4_decode_trace)."""
    z = np.load(os.path.join(G, "cfg4_decode_trace.npz"))
    net = FasterWaveNet(Params(R.make_params(**CFG2)), seed=1234)       # the weights the fixture was generated with
    net.to_gpu()
    n = int(z["tokens"].shape[0])
    toks, probs = net.generate(n, z["uniforms"], return_probs=True)
    np.testing.assert_allclose(to_np(probs)[::8], z["probs_every8"], atol=2e-5)
    np.testing.assert_array_equal(to_np(toks), z["tokens"].astype(np.int32))
