"""GPU parity of the bf16-storage path (BASELINE config 5: 128 residual / dilation channels, bf16 activations in HBM, fp32
accumulation) against oracle/bf16_ref.py, the CPU restatement that rounds to bfloat16 exactly where the kernels store or
feed a matrix core (forward AND backward).  Tolerances: stored bf16 tensors within 2 bf16 ulp of the oracle's (an fp32
vs float64 sum that lands on the other side of a rounding tie), logits 2e-2, loss 1e-3, every gradient tensor within
1e-2 of the oracle's in the 2-norm (VERDICT r1 next #1b); reference ops: wavenet.py:358-368, 556-617."""
import numpy as np
import pytest
import torch

from oracle import bf16_ref as Q
from oracle import wavenet_ref as R
from wavenet_amd import Params, TrainStepGraph, WaveNet

from gpu_util import dev, to_np

pytestmark = pytest.mark.gpu


def build16(over, seed=1234, head_bias=0.3):
    p = R.make_params(**over)
    w = R.init_weights(p, seed)
    if head_bias:
        for i in range(len(p["softmax_conv_channels"]) - 1):
            w["softmax_%d/b" % i] = (np.random.RandomState(seed + i).standard_normal(w["softmax_%d/b" % i].shape) *
                                     head_bias).astype(np.float32)
    net = WaveNet(Params(p), seed=0, storage="bf16")
    net.load_state_dict(w)
    net.to_gpu()
    return p, w, net


def close_bf16(got, want, what, ulps=2.0, floor=1e-3):
    got = to_np(got.float()) if isinstance(got, torch.Tensor) else got
    tol = ulps * 2.0 ** -8 * np.abs(want) + floor * max(1.0, float(np.abs(want).max())) * 2.0 ** -8
    bad = np.abs(got - want) > tol
    assert not bad.any(), (what, int(bad.sum()), got.size, float(np.abs(got - want).max()), np.argwhere(bad)[:4].tolist())


def grads_close(net, g, rel, tag="", noise=None):
    """every gradient tensor within ``rel`` (2-norm, relative) of the oracle's -- or, when ``noise`` (a second oracle
    run on weights perturbed by 1e-6) is given, within 1.25 x the oracle's own distance to that run + ``rel``."""
    worst = 0.0
    for ln, kind, off, n, shape in net._spans:
        k = "%s/%s" % (ln.name, kind)
        want = np.asarray(g[k], np.float64).reshape(-1)
        got = to_np(net._grad_arena[off:off + n]).astype(np.float64)
        nw = np.linalg.norm(want)
        err = np.linalg.norm(got - want) / (nw + 1e-30) if nw > 0 else np.abs(got).max()
        worst = max(worst, err)
        bar = rel
        if noise is not None and nw > 0:
            bar = rel + 1.25 * np.linalg.norm(np.asarray(noise[k], np.float64).reshape(-1) - want) / nw
        assert err <= bar, (tag, ln.name, kind, err, bar, nw)
    return worst


def _ws_views(net, L, B, T, Tw):
    ws = net._last16_ws
    al = lambda n: (n + 255) // 256 * 256
    o = 0
    dzs = ws[o:o + L * B * Tw * 128 * 2].view(torch.bfloat16).view(L, B, Tw, 128); o += al(L * B * Tw * 128 * 2)
    dadg = ws[o:o + L * B * T * 256 * 2].view(torch.bfloat16).view(L, B, T, 256); o += al(L * B * T * 256 * 2)
    dx0 = ws[o:o + B * T * 128 * 2].view(torch.bfloat16).view(B, T, 128); o += al(B * T * 128 * 2)
    dx1 = ws[o:o + B * T * 128 * 2].view(torch.bfloat16).view(B, T, 128)
    return dzs, dadg, (dx0, dx1)


def _live_bounds(dilations, t_off):
    """Per stack layer (host logic of wn16_stack_bwd, w16_api.hip): the loss reaches the stack through skip[t_off:] only, so
    layer l carries gradient at columns >= t_off - (reach of the layers above).  Returns (zero_g, live_g, live_x): [da | dg] of
    layer l is WRITTEN from row zero_g[l] on (zeros up to live_g[l]) and unspecified below; its dx from live_x[l] on (layer 0:
    everywhere, the embedding backward reads it all)."""
    L = len(dilations)
    zero_g, live_g, live_x = [0] * L, [0] * L, [0] * L
    t_live = t_off
    for l in range(L - 1, -1, -1):
        live_g[l] = t_live // 32 * 32
        zero_g[l] = t_live // 64 * 64
        t_live = max(t_live - dilations[l], 0)
        live_x[l] = t_live // 32 * 32
    return zero_g, live_g, live_x


SMALL = dict(quantization_steps=256, causal_conv_channels=[128], residual_conv_channels=[128] * 3, residual_num_blocks=2,
             softmax_conv_channels=[256, 256])


@pytest.mark.parametrize("B,T,tw", [(2, 150, 90), (1, 333, 64), (3, 64, 33), (2, 1000, 611), (1, 96, 96)])
def test_bf16_stack_every_intermediate_against_the_rounding_oracle(B, T, tw):
    """6 layers (d = 1, 2, 4, 1, 2, 4), ragged T: layer outputs, z, skip, logits, loss, dz_skip, [da | dg], dx of the two
    lowest layers and every gradient."""
    p, w, net = build16(SMALL, seed=9)
    rs = np.random.RandomState(T)
    idx = rs.randint(0, 256, (B, T)).astype(np.int32)
    tgt = rs.randint(0, 256, (B, tw)).astype(np.int32)
    c = net.forward_causal_block(dev(idx))
    _, s = net.forward_residual_block(c, t_off=T - tw)
    x, xs, z, skip = net._last16
    keep = {}
    # the head and the backward are compared on the device's own skip sum (see oracle/bf16_ref.py::train_step)
    loss_ref, logits_ref, g = Q.train_step(p, w, idx, tgt, keep=keep, skip_override=to_np(skip.float()))
    close_bf16(c[:, :, 0, :].permute(0, 2, 1), keep["x0"], "embedding")
    L = len(keep["zs"])
    # every layer on the input the device itself produced (a 1-ulp difference upstream re-rounds everything downstream, so
    # only a teacher-forced comparison can be held to rounding-tie tolerance), then the free-running chain loosely
    lay = list(Q._layers(p))
    for l in range(L):
        pre, d = lay[l]
        xin = to_np((x if l == 0 else xs[l - 1]).float()).astype(np.float64)
        zw, ow = Q.layer_fwd(xin, w[pre + "wf/W"], w[pre + "wg/W"], w[pre + "projection_block/W"], d, Q._Z(T, d, True))
        close_bf16(z[l], zw, "z of layer %d" % l, floor=0.05)
        close_bf16(xs[l], ow, "output of layer %d" % l, floor=0.1)
        for got, want in ((z[l], keep["zs"][l]), (xs[l], keep["xs"][l])):
            assert np.abs(to_np(got.float()) - want).max() <= 2.0 ** -6 * np.abs(want).max(), l
    zall = np.stack([to_np(z[l].float())[:, T - tw:].astype(np.float64) for l in range(L)])
    sw = sum(zall[l] @ Q.rb(w[lay[l][0] + "projection_softmax/W"][:, :, 0, 0]).T for l in range(L))
    close_bf16(skip, Q.rb(sw), "skip sum", floor=0.1)
    logits = net.forward_softmax_block(s, apply_softmax=False)
    np.testing.assert_allclose(to_np(logits)[:, :, 0, :].transpose(0, 2, 1), logits_ref, atol=2e-2)
    loss = net.cross_entropy(logits, tgt)
    assert abs(float(loss.detach()) - loss_ref) < 1e-3
    net.zero_grads()
    loss.backward()
    torch.cuda.synchronize()
    dzs, dadg, dxb = _ws_views(net, L, B, T, tw)
    # backward buffers: 2-norm comparisons (a ReLU mask that flips where the skip sum is ~0, or an upstream rounding tie,
    # changes single elements completely; the tensors as a whole agree to bf16 accuracy)
    def nclose(got, want, what, rel=2e-2):
        err = np.linalg.norm(to_np(got.float()).astype(np.float64) - want) / (np.linalg.norm(want) + 1e-30)
        assert err <= rel, (what, err)
    # rows no gradient reaches are not written at all (round 5): compared from the first written row on, where the oracle has
    # its zeros / its values
    zero_g, live_g, live_x = _live_bounds([d for _, d in lay], T - tw)
    for l in range(L):
        nclose(dzs[l], keep["dzs"][l], "dz_skip of layer %d" % l)
        assert np.abs(keep["dadg"][l][:, :live_g[l]]).max(initial=0.0) == 0        # the oracle agrees that nothing lives below
        nclose(dadg[l][:, zero_g[l]:], keep["dadg"][l][:, zero_g[l]:], "[da | dg] of layer %d" % l)
    for l in (1, 2):                      # the ping-pong buffers still hold the dx of layers 2 and 1
        assert np.abs(keep["dx"][l][:, :live_x[l]]).max(initial=0.0) == 0
        nclose(dxb[l & 1][:, live_x[l]:], keep["dx"][l][:, live_x[l]:], "dx of layer %d" % l)
    worst = grads_close(net, g, 1e-2)
    assert float(net.residual_blocks[-1][-1].projection_block.W.grad.abs().sum()) == 0
    assert worst > 1e-6                   # and it is not the fp32 path


def test_bf16_storage_rejects_what_it_does_not_cover():
    from wavenet_amd import WaveNetHipError
    p = R.make_params(quantization_steps=256, causal_conv_channels=[64], residual_conv_channels=[64] * 2,
                      residual_num_blocks=1, softmax_conv_channels=[256, 256])
    net = WaveNet(Params(p), seed=0, storage="bf16")
    net.to_gpu()
    c = net.forward_causal_block(dev(np.zeros((1, 40), np.int32)))
    with pytest.raises(WaveNetHipError):
        net.forward_residual_block(c)


CFG5 = dict(quantization_steps=256, causal_conv_channels=[128], residual_conv_channels=[128] * 10, residual_num_blocks=4,
            softmax_conv_channels=[512, 256])


def test_cfg5_full_topology_train_step_vs_the_rounding_oracle_eager_and_graph():
    """BASELINE configs[4] as specified: 4 x 10 layers (d = 1..512), 128 / 512 channels, bf16 storage, T = input_width +
    120, against the oracle that rounds forward and backward, launched op by op and through the replayed TrainStepGraph.
    Forty layers of bf16 re-rounding are chaotic in the last bit: the oracle's own gradients move by ~3 % (2-norm) when its
    weights are perturbed by 1e-6 relative.  That measured sensitivity is the bar: every gradient tensor within 1.25 x the
    oracle's distance to its perturbed self + 2e-3 (the 6-layer test above holds a flat 1e-2; the layer kernels
    themselves are held to rounding-tie tolerance there, input for input)."""
    p, w, net = build16(CFG5, seed=5)
    iw = R.input_width(p)
    B, extra = 1, 120
    T = iw + extra
    rs = np.random.RandomState(3)
    idx = rs.randint(0, 256, (B, T)).astype(np.int32)
    tgt = rs.randint(0, 256, (B, extra)).astype(np.int32)
    x, t = dev(idx), dev(tgt)
    c = net.forward_causal_block(x)
    _, s = net.forward_residual_block(c, t_off=T - extra)
    skip = net._last16[3]
    own = {}
    loss_free, _, g_free = Q.train_step(p, w, idx, tgt, keep=own)
    assert np.abs(to_np(skip.float()) - own["skip"]).max() <= 2.0 ** -5 * np.abs(own["skip"]).max()   # free-running forward
    loss_ref, logits_ref, g = Q.train_step(p, w, idx, tgt, skip_override=to_np(skip.float()))
    logits = net.forward_softmax_block(s, apply_softmax=False)
    loss = net.cross_entropy(logits, t)
    net.zero_grads()
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss.detach()) - loss_ref) < 2e-3
    np.testing.assert_allclose(to_np(logits)[:, :, 0, :].transpose(0, 2, 1), logits_ref, atol=5e-2)
    w2 = {k: (v * (1 + 1e-6 * np.sign(np.random.RandomState(1).standard_normal(v.shape)))).astype(np.float32)
          for k, v in w.items()}
    _, _, g_noise = Q.train_step(p, w2, idx, tgt, skip_override=to_np(skip.float()))
    worst = grads_close(net, g, 2e-3, "eager", noise=g_noise)
    assert worst < 5e-2
    grads_close(net, g_free, 0.25, "eager, free-running oracle")      # ReLU-mask flips on 120 columns included: sanity only
    assert abs(loss_free - loss_ref) < 2e-2
    net.update_laerning_rate(1e-4)
    w0 = to_np(net._arena).copy()
    gr = TrainStepGraph(net, x, t)
    np.testing.assert_array_equal(to_np(net._arena), w0)
    lg = gr.step()
    torch.cuda.synchronize()
    assert abs(float(lg) - loss_ref) < 2e-3
    grads_close(net, g, 2e-3, "graph", noise=g_noise)
    assert np.abs(to_np(net._arena) - w0).max() > 0


@pytest.mark.parametrize("B,T", [(1, 70), (2, 1000), (3, 4133), (8, 16384)])
def test_embedding_gradient_on_the_matrix_cores_matches_a_float64_scatter(B, T):
    """wn16_embed_bwd (the backward of data.py:61-68 + wavenet.py:298-301 on tokens, as a one-hot x dout contraction in
    bf16 MFMA with fp32 accumulation): every table entry within 1e-5 (relative to the column scale) of a float64
    scatter-add of the same bf16 gradient, the bias gradient likewise; accumulates into dW (+=); ragged T, T < one stage."""
    from wavenet_amd import _lib
    from wavenet_amd._lib import check, ptr, stream_ptr
    lib = _lib.lib()
    rs = np.random.RandomState(B * 1000 + T)
    idx = rs.randint(0, 256, (B, T)).astype(np.int32)
    idx[:, : T // 3] = (128 + 20 * np.sin(np.arange(T // 3) / 7.0)).astype(np.int32)     # runs of equal tokens, as in audio
    dout = torch.as_tensor(rs.standard_normal((B, T, 128)).astype(np.float32)).to(torch.bfloat16).cuda()
    d64 = dout.float().cpu().numpy().astype(np.float64)
    want = np.zeros((128, 256, 2))
    for b in range(B):
        np.add.at(want[:, :, 1].T, idx[b], d64[b])                                       # tap 1: the current token
        np.add.at(want[:, :, 0].T, idx[b, :-1], d64[b, 1:])                              # tap 0: the token one step back
    dW = torch.full((128, 256, 1, 2), 0.25, device="cuda", dtype=torch.float32)
    db = torch.full((128,), -1.0, device="cuda", dtype=torch.float32)
    nws = lib.wn16_embed_bwd_workspace_bytes(B, T)
    ws = torch.empty((nws,), device="cuda", dtype=torch.uint8)
    check(lib.wn16_embed_bwd(ptr(dev(idx)), ptr(dout), ptr(dW), ptr(db), B, T, 256, 128, 2, ptr(ws), nws, stream_ptr()),
          "wn16_embed_bwd")
    torch.cuda.synchronize()
    got = to_np(dW).reshape(128, 256, 2).astype(np.float64) - 0.25
    scale = np.abs(d64).sum() / (128 * 256) + 1.0
    assert np.abs(got - want).max() <= 1e-5 * scale, (np.abs(got - want).max(), scale)
    gb = to_np(db).astype(np.float64) + 1.0
    assert np.abs(gb - d64.sum((0, 1))).max() <= 1e-5 * (np.abs(d64).sum() / 128 + 1.0)
    rc = lib.wn16_embed_bwd(ptr(dev(idx)), ptr(dout), ptr(dW), None, B, T, 256, 64, 2, ptr(ws), nws, stream_ptr())
    assert rc != 0 and b"128 channels" in lib.wn_last_error()


def test_bf16_graph_replays_repack_the_weight_images_every_step():
    """the captured step must contain the per-step pack of the bf16 operand images -- a replay that
    computes on the images of the capture-time weights trains on stale weights from the second step on.  Five replays
    against five op-by-op `backprop` steps on a twin model: the losses of steps 2-5 (which see the updated weights only
    through the repacked images) agree, and they are far from what the stale images would give (the step-1 loss)."""
    from wavenet_amd.graph import default_loss
    rs = np.random.RandomState(21)
    B, T, tw = 2, 200, 120
    idx = rs.randint(0, 256, (B, T)).astype(np.int32)
    tgt = ((idx[:, T - tw:] + 1) % 256).astype(np.int32)           # learnable: the next token is the current one + 1
    x, t = dev(idx), dev(tgt)
    nets = []
    for _ in range(2):
        _, _, net = build16(SMALL, seed=9)
        net.update_laerning_rate(2e-2)
        net.optimizer.eps = 1e-3
        nets.append(net)
    eager, graphed = nets
    w0 = to_np(eager._arena).copy()
    le = []
    for _ in range(5):
        loss = default_loss(eager, x, t)
        le.append(float(loss.detach()))
        eager.backprop(loss)
    gr = TrainStepGraph(graphed, x, t)
    assert graphed._w16_stale           # whatever ran before the first replay, the next eager forward repacks
    lg = [float(gr.step()) for _ in range(5)]
    torch.cuda.synchronize()
    assert le[0] - le[4] > 0.05, le                                  # the weights moved enough to tell stale from fresh
    for a, b in zip(le, lg):
        assert abs(a - b) < 0.1 * (le[0] - le[4]), (le, lg)
    wa, wb = to_np(eager._arena), to_np(graphed._arena)
    assert np.abs(wa - wb).max() < 0.1 * np.abs(wa - w0).max()
    # and an eager forward after the replays sees the trained weights too
    l5 = float(default_loss(graphed, x, t).detach())
    assert abs(l5 - float(default_loss(eager, x, t).detach())) < 0.1 * (le[0] - le[4])


def test_cfg5_training_steps_are_bit_reproducible():
    """VERDICT r2 next #5c: no float atomics left on the bf16-storage training path either -- the conv-tap and skip-projection
    weight gradients (k16_wgrad: per-workgroup 256 x 256 blocks, k16_wgrad_reduce adds the time slabs in order), the head's
    weight and bias gradients (the same, through wn16_pointwise_bwd's workspace), dWp (partial tiles), the embedding table
    (one-hot contraction with a fixed-order reduce).  Two models run the same three updates at config 5's full 4 x 10
    topology: loss, every gradient and every weight agree bit for bit."""
    rs = np.random.RandomState(5)
    runs = []
    for rep in range(2):
        p, w, net = build16(CFG5, seed=3)
        net.update_laerning_rate(1e-3)
        iw = net.input_width
        if rep == 0:
            tok = rs.randint(0, 256, size=(3, 2, iw + 300)).astype(np.int32)
        rec = []
        for step in range(3):
            x, tgt = dev(tok[step][:, :-1]), dev(tok[step][:, iw:])
            c = net.forward_causal_block(x)
            _, s = net.forward_residual_block(c, t_off=x.shape[1] - tgt.shape[1])
            loss = net.cross_entropy(net.forward_softmax_block(s, apply_softmax=False), tgt)
            net.backprop(loss)
            torch.cuda.synchronize()
            rec.append((to_np(loss.detach()).copy(), to_np(net._grad_arena).copy(), to_np(net._arena).copy()))
        runs.append(rec)
    for (l0, g0, w0), (l1, g1, w1) in zip(*runs):
        assert l0.tobytes() == l1.tobytes()
        assert np.abs(g0).max() > 0
        np.testing.assert_array_equal(g0, g1)
        np.testing.assert_array_equal(w0, w1)


def test_cfg5_full_bench_size_gradients_equal_the_two_clip_slice_and_are_bit_reproducible():
    """VERDICT r3 missing #4: the launch geometry bench.py's `wide_channel` leg times -- B = 8 x T = 16,384, loss over the
    last 12,290 columns: the XCD-aware block orders of k16_cgemm256 / k16_wgrad, slabs sized to one round of 256
    workgroups, 4,096 tiles per layer kernel -- was reached by no check.  Clips are independent and the loss is a mean
    over the rows whose label is not -1 (chainer.functions.softmax_cross_entropy's ignore label), so the full batch with
    the labels of clips 2..7 set to -1 must reproduce the loss and EVERY gradient of its two-clip slice (B = 2, the
    geometry the oracle-checked tests run at: far fewer tiles, other block orders): same per-column arithmetic, only the
    fp32 summation order of the weight gradients may differ.  Bars: loss 1e-5, every gradient tensor within 2e-4 of its
    largest entry (measured ~1e-5; a geometry bug -- a tile skipped, counted twice or read from the wrong clip -- is
    O(1)); and the full batch must be bit-reproducible (two runs, every gradient equal).  The two-clip slice itself is
    held to the rounding oracle by the tests above at T = 4,214."""
    from bench import make_batch
    cfg = dict(CFG5)
    net = WaveNet(Params(R.make_params(**cfg)), seed=1, storage="bf16")
    net.to_gpu()
    iw = net.input_width
    x, tgt = make_batch(0, 1, iw)                                   # (8, 16384), (8, 12290): the bench's own batch
    assert tuple(x.shape) == (8, 16384) and tuple(tgt.shape) == (8, 12290)
    tgt_masked = tgt.clone()
    tgt_masked[2:] = -1

    def step(xx, tt):
        c = net.forward_causal_block(xx)
        _, s = net.forward_residual_block(c, t_off=iw)
        loss = net.cross_entropy(net.forward_softmax_block(s, apply_softmax=False), tt)
        net.zero_grads()
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach()), to_np(net._grad_arena).copy()

    l8a, g8a = step(x, tgt_masked)
    l8b, g8b = step(x, tgt_masked)
    assert l8a == l8b
    np.testing.assert_array_equal(g8a, g8b)                         # bit-reproducible at the benchmarked size
    l2, g2 = step(x[:2].contiguous(), tgt[:2].contiguous())
    assert abs(l8a - l2) < 1e-5, (l8a, l2)
    assert np.isfinite(g8a).all() and np.abs(g2).max() > 0
    for ln, kind, off, n, shape in net._spans:
        a, b = g2[off:off + n], g8a[off:off + n]
        scale = max(float(np.abs(a).max()), 1e-12)
        assert float(np.abs(a - b).max()) <= 2e-4 * scale, (ln.name, kind, float(np.abs(a - b).max()), scale)
    # and the unmasked full batch runs (the step bench.py times): finite loss, every layer receives gradient
    l8, g8 = step(x, tgt)
    assert np.isfinite(l8) and np.isfinite(g8).all()
    for ln, kind, off, n, shape in net._spans:
        if "projection_block" in ln.name and ln is net.residual_blocks[-1][-1].projection_block:
            continue
        assert np.abs(g8[off:off + n]).max() > 0, ln.name


@pytest.mark.parametrize("B,T,tw,cfg", [(2, 1000, 611, "small"), (3, 4133, 1200, "small"), (1, 333, 64, "small"),
                                        (2, 4294, 200, "cfg5"), (8, 16384, 12290, "cfg5")])
def test_one_launch_layer_backward_equals_the_per_layer_launches_bit_for_bit(B, T, tw, cfg):
    """k16_bwd_multi (WN_EXEC_BF16_MULTI_LAYER_BWD, opt-in: measured no faster) -- layers L-2 .. 1 of
    the bf16 layer backward in ONE launch of co-resident workgroups that follow one dataflow word per 32-column tile instead
    of eighty launch boundaries.  Tile code, tile -> workgroup deal and per-workgroup dWp partial tiles are those of
    k16_gate_bwd / k16_dx, so every gradient, the [da | dg] arrays of every layer and the two dx buffers must equal the
    per-layer launches' BIT FOR BIT --
    a tile read before its producer had stored it, or a stale line served for it, shows as a difference.  Sizes: ragged
    last tiles, clips that do not align with the XCD ranges of tile_range(), the full 4 x 10 stack at the oracle-checked
    window and at the bench's own B = 8 x 16,384 (4,096 tiles, 16 per workgroup), three repetitions of the one-launch
    form (its waits differ from run to run, its results must not).  The per-layer form is what the rounding-oracle tests
    above check op by op."""
    from wavenet_amd import _lib
    over = SMALL if cfg == "small" else CFG5
    net = WaveNet(Params(R.make_params(**over)), seed=2, storage="bf16")
    net.to_gpu()
    L = len(net._flat_layers)
    dil = [2 ** i for _ in range(over["residual_num_blocks"]) for i in range(len(over["residual_conv_channels"]))]
    rs = np.random.RandomState(B * 1000 + T)
    x = dev(rs.randint(0, 256, size=(B, T)).astype(np.int32))
    tgt = dev(rs.randint(0, 256, size=(B, tw)).astype(np.int32))
    base = _lib.default_exec_flags() & ~_lib.WN_EXEC_BF16_MULTI_LAYER_BWD

    def step(flags):
        net.exec_flags = flags
        c = net.forward_causal_block(x)
        _, s = net.forward_residual_block(c, t_off=T - tw)
        loss = net.cross_entropy(net.forward_softmax_block(s, apply_softmax=False), tgt)
        net.zero_grads()
        loss.backward()
        torch.cuda.synchronize()
        dzs, dadg, dxb = _ws_views(net, L, B, T, tw)
        # compared on the device as raw bits (2.7 GB of [da | dg] at the bench's size), from each layer's first WRITTEN row on
        # (rows no gradient reaches are left untouched: whatever the allocation held)
        zero_g, live_g, live_x = _live_bounds(dil, T - tw)
        dadg = dadg.view(torch.int16).clone()
        for l in range(L):
            dadg[l, :, :zero_g[l]] = 0
        dx1, dx2 = dxb[1].view(torch.int16).clone(), dxb[0].view(torch.int16).clone()       # dx of layers 1 and 2
        dx1[:, :live_x[1]] = 0
        dx2[:, :live_x[2]] = 0
        return (net._grad_arena.view(torch.int32).clone(), dadg, dx1, dx2)

    ref = step(base)
    g = ref[0].view(torch.float32)
    assert bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0
    for rep in range(3):
        got = step(base | _lib.WN_EXEC_BF16_MULTI_LAYER_BWD)
        for a, b, what in zip(ref, got, ("gradient arena", "[da | dg] of every layer", "dx of layer 1", "dx of layer 2")):
            assert torch.equal(a, b), "%s differs, repetition %d: %d elements" % (what, rep, int((a != b).sum()))
