"""GPU parity on BASELINE.json's own configurations (the shapes bench.py times), against the CPU oracle.

config 2: 4 blocks x 10 dilations, 32 residual / 256 skip channels -- one train step, loss and EVERY gradient, launched op
          by op and through the replayed TrainStepGraph (train_audio/train.py:58-80);
config 4: the queue-cached decoder at that topology, window 4094: all 16,000 generated tokens bit-exact against the committed
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


_ORACLE = {}


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


@pytest.mark.parametrize("prec", ["fp16x2", "bf16x3", "fp32"])
@pytest.mark.parametrize("B,extra,t1", [(2, 200, 0), (1, 333, 1), (2, 12290, 0), (8, 12290, 0)])
def test_cfg2_topology_train_step_loss_and_every_gradient(B, extra, t1, prec):
    """The stack the bench times (4 x 10 layers, d = 1..512, 1280-deep skip contraction, the 40-layer partial-tile
    reduction, the chained backward's live ranges over four blocks) at T = input_width + extra: loss within 1e-4, every
    gradient tensor within 1e-4 of its largest entry, against the oracle's autograd -- eager and graph replay -- in EVERY
    shipped arithmetic mode (fp16x2: the default; bf16x3 / fp32: the exact-fp32 layer kernels with six-term bf16 or fp32
    skip-path contractions).  The third case is the bench's own window: T = 16,384, t_off = 4,094, loss over 12,290
    columns (B = 2: 1,024 tiles per layer, four per wave in the chained backward, the XCD-aware tile order).  The fourth
    case IS the timed workload: B = 8 x T = 16,384 on bench.py's own batch (``bench.make_batch(0, 1, iw)``; 4,096 tiles per
    layer, one workgroup per CU in the multi-layer backward) -- the oracle step on it costs ~15-25 s and 7.4 GB of host
    memory.  t1 = 1 forces the one-tile-per-wave forward kernels, which bench-sized launches take by themselves
    (WnExec.fwd_t1_min_blocks)."""
    p, w, net = build(CFG2)
    net.gemm_precision = prec
    net.fwd_t1_min_blocks = t1
    iw = R.input_width(p)
    assert iw == 4094 and len(net._flat_layers) == 40
    T = iw + extra
    if B == 8:
        from bench import make_batch
        x, t = make_batch(0, 1, iw)
        idx, tgt = to_np(x), to_np(t)
        assert idx.shape == (8, 16384) and tgt.shape == (8, 12290)
        torch.set_num_threads(min(16, os.cpu_count() or 1))     # the oracle's intra-op pool: more threads than that only hurt
    else:
        rs = np.random.RandomState(17 + extra)
        idx = rs.randint(0, 256, (B, T)).astype(np.int32)
        tgt = rs.randint(0, 256, (B, extra)).astype(np.int32)
        x, t = dev(idx), dev(tgt)
    # op by op
    c = net.forward_causal_block(x)
    _, s = net.forward_residual_block(c, t_off=T - extra)
    # The ReLU in front of the head is discontinuous: a skip value within rounding distance of 0 (one or two of the 6.3 M
    # of the large window) makes a whole gradient term present on one side and absent on the other -- a single such
    # element moved dWs by 6e-3 of its largest entry.  The oracle therefore differentiates at the DEVICE's mask (the sign
    # pattern of the device's skip sum, which itself must match the oracle's to 1e-4 with at most a handful of flips).
    mask = (to_np(s) > 0).astype(np.float32)
    key = (B, extra)
    if key not in _ORACLE or not np.array_equal(_ORACLE[key][0], mask):     # one oracle step per shape (and mask)
        _ORACLE.clear()
        keep = {}
        _ORACLE[key] = (mask, R.train_step_grads(p, w, idx, tgt, first_relu_mask=mask, keep=keep), keep["skip"])
    _, (loss_ref, logits_ref, g), skip_ref = _ORACLE[key]
    np.testing.assert_allclose(to_np(s), skip_ref, atol=1e-4)
    # (a skip value lands within rounding distance, ~1e-6, of 0 with probability ~1e-6: at most 8 flips per 6.3 M elements --
    #  the bound of rounds 4-5 at B = 2, the same DENSITY at B = 8's 25.2 M elements)
    assert int(((skip_ref > 0) != (mask > 0)).sum()) <= 8 * int(np.ceil(mask.size / 6.3e6))
    logits = net.forward_softmax_block(s, apply_softmax=False)
    loss = net.cross_entropy(logits, t)
    net.zero_grads()
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss.detach()) - loss_ref) < 1e-4
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


def test_cfg4_decoder_all_16000_samples_match_the_committed_oracle_trace():
    """cfg4 as BASELINE configs[3] names it: 4 x 10 layers, window 4094, ELU head after the first (ReLU) step, ALL 16,000
    emitted samples bit-exact against the oracle's literal queue-cached generation (train_audio/generate.py:21-43,
    faster_wavenet.py:50-113); probabilities of every eighth step of the first 2,048 and of every 128th step of the whole
    trace within 2e-5 of the oracle's.  The fixture's uniforms keep 2e-5 of distance to every boundary of the oracle's
    cumulative distributions (tests/golden/make_golden.py::cfg4_decode_trace); the number of draws that rule replaced is
    bounded by what the margin predicts (~1.0 % of the steps: 255 boundaries x 2 x 2e-5) and must not grow."""
    z = np.load(os.path.join(G, "cfg4_decode_trace.npz"))
    net = FasterWaveNet(Params(R.make_params(**CFG2)), seed=1234)       # the weights the fixture was generated with
    net.to_gpu()
    n = int(z["tokens"].shape[0])
    assert n == 16000
    toks, probs = net.generate(n, z["uniforms"], return_probs=True)
    replaced = int(z["replaced"])
    assert replaced <= 0.0102 * n + 3 * np.sqrt(0.0102 * n), "the fixture replaced %d of %d uniforms for sitting within %g of a CDF boundary; it must not grow" % (
        replaced, n, float(z["margin"]))
    p = to_np(probs)
    np.testing.assert_allclose(p[:2048:8], z["probs_every8"], atol=2e-5)
    np.testing.assert_allclose(p[::128], z["probs_every128"], atol=2e-5)
    got = to_np(toks)
    want = z["tokens"].astype(np.int32)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "first differing sample %d of %d (%d of the fixture's uniforms were margin-replaced)" % (int(bad[0]), n, replaced)


def test_cfg4_decoder_on_nine_workgroups_equals_the_one_workgroup_kernel():
    """k_decode_fast3 (the chain on one CU, the skip rows and their share of the logits on eight more, flag-in-data exchange
    through device memory) against k_decode_fast (WN_DECODER_ONE_WORKGROUP): same products, other summation order for the
    skip rows (eight partial sums per row) and the logits (eight shares): 3,000 tokens equal, probabilities within 1e-6
    (measured ~1e-7) -- a stale or torn exchange entry is a wrong addend, O(1e-2) -- and two runs of the nine-workgroup
    form agree bit for bit (its order is fixed)."""
    from wavenet_amd import _lib
    u = np.random.RandomState(23).random_sample(3000)
    res = []
    for flags in (_lib.WN_DECODER_ONE_WORKGROUP, 0, 0):
        net = FasterWaveNet(Params(R.make_params(**CFG2)), seed=1234)
        net.exec_flags = flags
        net.to_gpu()
        toks, probs = net.generate(3000, u, return_probs=True)
        torch.cuda.synchronize()
        res.append((to_np(toks).copy(), to_np(probs).copy()))
        del net
    np.testing.assert_array_equal(res[1][0], res[2][0])
    np.testing.assert_array_equal(res[1][1], res[2][1])
    np.testing.assert_array_equal(res[0][0], res[1][0])
    np.testing.assert_allclose(res[0][1], res[1][1], rtol=0, atol=1e-6)
    assert np.abs(res[0][1] - res[1][1]).max() > 0            # and it really is the other kernel
    assert len(set(res[0][0].tolist())) > 20


def test_cfg2_full_batch_gradients_fp16x2_against_exact_fp32_on_the_device():
    """An extra at the bench's full size (B = 8 x T = 16,384: 4,096 tiles per layer, one workgroup per CU in the chained
    backward; the oracle itself is held against this size in test_cfg2_topology_train_step_loss_and_every_gradient's
    fourth case and in test_cfg2_bench_step_against_the_committed_oracle_fixture): the default arithmetic against the
    exact-fp32-MFMA mode on the device, where BOTH differentiate the same recorded forward.  Forward: skip sum and logits of the default fp16x2 mode within 1e-4 of it, loss within 1e-5.  Backward: both
    modes differentiate the SAME recorded forward (the exact-fp32 one: same saved activations, same ReLU mask -- see the
    test above for why the mask must be shared), every gradient tensor of fp16x2 within 1e-4 of fp32 relative to the
    tensor's largest entry."""
    from bench import make_batch
    net = WaveNet(Params(R.make_params(**CFG2)), seed=1234)
    net.to_gpu()
    iw = net.input_width
    x, tgt = make_batch(0, 1, iw)
    fwd, grads = {}, {}
    for prec in ("fp16x2", "fp32"):
        net.gemm_precision = prec
        with torch.no_grad():
            _, s = net.forward_residual_block(net.forward_causal_block(x), t_off=iw)
            lg = net.forward_softmax_block(s, apply_softmax=False)
            fwd[prec] = (to_np(s), to_np(lg), float(net.cross_entropy(lg, tgt)))
        del s, lg
    np.testing.assert_allclose(fwd["fp16x2"][0], fwd["fp32"][0], atol=1e-4)
    np.testing.assert_allclose(fwd["fp16x2"][1], fwd["fp32"][1], atol=1e-4)
    assert abs(fwd["fp16x2"][2] - fwd["fp32"][2]) < 1e-5
    del fwd
    for prec in ("fp32", "fp16x2"):
        net.gemm_precision = "fp32"
        c = net.forward_causal_block(x)
        _, s = net.forward_residual_block(c, t_off=iw)
        loss = net.cross_entropy(net.forward_softmax_block(s, apply_softmax=False), tgt)
        net.zero_grads()
        net.gemm_precision = prec                       # the arithmetic of the backward only
        loss.backward()
        torch.cuda.synchronize()
        grads[prec] = to_np(net._grad_arena).copy()
        del c, s, loss
    for ln, kind, off, n, shape in net._spans:
        a, b = grads["fp32"][off:off + n], grads["fp16x2"][off:off + n]
        scale = max(float(np.abs(a).max()), 1e-9)
        assert np.abs(a - b).max() <= 1e-4 * scale, (ln.name, kind, float(np.abs(a - b).max()), scale)
    assert np.abs(grads["fp32"] - grads["fp16x2"]).max() > 0            # and the two really are different arithmetic


@pytest.mark.parametrize("prec", ["fp16x2", "bf16x3", "fp32"])
def test_cfg2_bench_step_against_the_committed_oracle_fixture(prec):
    """tests/golden/cfg2_bench_step.npz (make_golden.py::cfg2_bench_step): the oracle's loss, logits / skip-sum probes and
    per-tensor gradient norms for the step bench.py times -- its batch, the seed-1234 weights -- at the oracle's own ReLU
    mask.  The captured graph's first step (what bench.py reports as ``golden_loss_match``): loss within 1e-4; op by op:
    logits and skip sum at 384 probed (clip, column) positions within 1e-4; every gradient tensor's L2 norm within 1e-3
    relative + its sum within 1e-3 of (norm x sqrt(n)) -- looser than the live-oracle test because a skip value within
    rounding distance of 0 flips its ReLU (see there), which the norms absorb and an element-wise bar would not."""
    from bench import make_batch
    z = np.load(os.path.join(G, "cfg2_bench_step.npz"))
    net = WaveNet(Params(R.make_params(**CFG2)), seed=1234)
    net.gemm_precision = prec
    net.to_gpu()
    iw = net.input_width
    x, t = make_batch(0, 1, iw)
    assert [int(x.sum().item()), int(t.sum().item())] == z["tokens_checksum"].tolist()
    c = net.forward_causal_block(x)
    _, s = net.forward_residual_block(c, t_off=iw)
    logits = net.forward_softmax_block(s, apply_softmax=False)
    loss = net.cross_entropy(logits, t)
    net.zero_grads()
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss.detach()) - float(z["loss"])) < 1e-4
    pb, pt = torch.as_tensor(z["probe_b"]).long().cuda(), torch.as_tensor(z["probe_t"]).long().cuda()
    np.testing.assert_allclose(to_np(logits[pb, :, 0, pt]), z["logits_probes"], atol=1e-4)
    np.testing.assert_allclose(to_np(s[pb, :, 0, pt]), z["skip_probes"], atol=1e-4)
    assert abs(float(logits.detach().double().sum()) - float(z["logits_sum"])) <= 1e-5 * float(z["logits_abs_sum"])
    assert abs(int((s > 0).sum().item()) - int(z["relu_live"])) <= 64
    names = [str(k) for k in z["grad_names"]]
    got = {"%s/%s" % (ln.name, kind): to_np(net._grad_arena[off:off + n]).astype(np.float64) for ln, kind, off, n, _ in net._spans}
    assert sorted(got) == names
    for k, l2, am, sm in zip(names, z["grad_l2"], z["grad_absmax"], z["grad_sum"]):
        g = got[k]
        assert abs(np.sqrt((g ** 2).sum()) - l2) <= 1e-3 * l2 + 1e-12, (k, np.sqrt((g ** 2).sum()), l2)
        assert abs(np.abs(g).max() - am) <= 1e-2 * am + 1e-12, (k, np.abs(g).max(), am)
        assert abs(g.sum() - sm) <= 1e-3 * l2 * np.sqrt(g.size) + 1e-12, (k, g.sum(), sm)
    del c, s, logits, loss
    gr = TrainStepGraph(net, x, t)
    assert abs(float(gr.step()) - float(z["loss"])) < 1e-4


def test_bench_two_ranks_on_one_gpu_runs_the_n_gt_1_branch():
    """`python bench.py --gpus 2` starts its own ranks (no torch.distributed.run on the command line) and, with
    WAVENET_BENCH_SHARE_GPU=1, both share cuda:0 over gloo: the N > 1 branch of the bench -- two-graph data-parallel step,
    barrier-bracketed timing, MAX over ranks, the `dist` record -- executes end to end and prints one JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WAVENET_BENCH_SHARE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--no-decode", "--no-wide"], env=env, capture_output=True, text=True,
                       timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dist"]["world_size"] == 2 and out["dist"]["backend"] == "gloo"
    assert len(out["dist"]["ranks"]) == 2
    for rk in out["dist"]["ranks"]:
        assert "fwd+bwd graph" in rk["launch"], rk
    assert np.isfinite(out["loss"]) and out["value"] > 0
    assert out["config"]["global_batch"] == 16 and out["scaling"] == "weak"


@pytest.mark.parametrize("prec", ["fp16x2", "bf16x3", "fp32"])
@pytest.mark.parametrize("B,extra,window_only", [(2, 200, False), (1, 333, True), (3, 1333, False), (2, 12290, False), (8, 12290, False)])
def test_multi_layer_backward_launch_equals_the_per_layer_launches(B, extra, window_only, prec):
    """k_layer_bwd_chain_multi: every layer below the top one of the 4 x 10 stack in ONE launch of co-resident workgroups,
    synchronised per tile through dataflow words (no grid barrier), the deal of tiles to waves rotated from layer to layer.
    Same tile code and the same per-tile arithmetic as the per-layer launches (WN_EXEC_NO_MULTI_LAYER_BWD); what differs is
    which wave sums which tiles' weight-gradient contributions, i.e. fp32 summation order: every gradient tensor within 1e-6
    of its largest entry (measured 1-2e-7) -- a missed dependency between layers reads a stale or half-written (V, U) tile
    and is O(1e-2) -- and the multi-layer launch is BIT-reproducible: three repetitions agree exactly (its schedule is
    static; only the waiting is dynamic).  Small and ragged windows (idle waves, live ranges that differ per layer, a
    batch that is no multiple of 8: the non-XCD deal), the bench's window at B = 2, and the bench's full batch (256
    workgroups, one per CU).  In every arithmetic mode (round 6: the exact-fp32-MFMA form of the launch,
    k_layer_bwd_chain_multi<true, false>, serves bf16x3 and fp32)."""
    from wavenet_amd import _lib
    net = WaveNet(Params(R.make_params(**CFG2)), seed=1234)
    net.gemm_precision = prec
    net.to_gpu()
    iw = net.input_width
    T = iw + extra
    rs = np.random.RandomState(B * 1000 + extra)
    x = dev(rs.randint(0, 256, (B, T)).astype(np.int32))
    tgt = dev(rs.randint(0, 256, (B, extra)).astype(np.int32))
    base = _lib.default_exec_flags() & ~_lib.WN_EXEC_NO_MULTI_LAYER_BWD
    got = {}
    for name, flags, reps in (("per-layer", base | _lib.WN_EXEC_NO_MULTI_LAYER_BWD, 1), ("multi", base, 3)):
        net.exec_flags = flags
        for rep in range(reps):
            c = net.forward_causal_block(x)
            _, s = net.forward_residual_block(c, t_off=iw, window_only=window_only)
            loss = net.cross_entropy(net.forward_softmax_block(s, apply_softmax=False), tgt)
            net.zero_grads()
            loss.backward()
            torch.cuda.synchronize()
            got[(name, rep)] = to_np(net._grad_arena).copy()
            del c, s, loss
    ref = got[("per-layer", 0)]
    assert np.isfinite(ref).all() and np.abs(ref).max() > 0
    # (exact-fp32 mode: the skip / head weight gradients come from k_wgrad_mfma, which adds its time slabs with float atomics --
    # those tensors are reproducible to summation order only, in either launch form; everything the layer backward itself
    # produces -- wf, wg, projection_block, and the causal table through dx -- must repeat exactly in every mode)
    for rep in (1, 2):
        if prec != "fp32":
            np.testing.assert_array_equal(got[("multi", 0)], got[("multi", rep)])
        for ln, kind, off, n, shape in net._spans:
            if not ("projection_softmax" in ln.name or ln.name.startswith("softmax")):
                np.testing.assert_array_equal(got[("multi", 0)][off:off + n], got[("multi", rep)][off:off + n], err_msg=ln.name)
    m = got[("multi", 0)]
    for ln, kind, off, n, shape in net._spans:
        a, b = ref[off:off + n], m[off:off + n]
        scale = max(float(np.abs(a).max()), 1e-30)
        assert float(np.abs(a - b).max()) <= 1e-6 * scale, (ln.name, kind, float(np.abs(a - b).max()), scale)


def test_multi_layer_backward_soak_two_runs_of_300_replayed_steps_agree_bit_for_bit():
    """The multi-layer backward waits on words other workgroups write (sc1 stores / loads, no barrier): a dependency that
    is missed once in ten thousand tiles would not show in three repetitions.  Two runs of 300 graph-replayed training
    steps of the bench's own step (B = 8 x 16,384; 300 x 39 layer transitions x 256 workgroups each, replays launched back
    to back without host synchronisation -- the condition under which a memset node once failed to reset the words in
    time) must end on the same weights, bit for bit, finite, and with a loss that went down.  (A 2 x 3,000-step soak ran
    the same way during development.)"""
    from bench import make_batch
    ws, losses = [], []
    for rep in range(2):
        net = FasterWaveNet(Params(R.make_params(**CFG2)), seed=1234)
        net.to_gpu()
        net.update_laerning_rate(3e-4)
        x, tgt = make_batch(0, 1, net.input_width)
        g = TrainStepGraph(net, x, tgt)
        first = float(g.step())
        for _ in range(299):
            loss = g.step()
        torch.cuda.synchronize()
        losses.append((first, float(loss)))
        ws.append(to_np(net._arena).copy())
        del g, net
    assert np.isfinite(ws[0]).all() and losses[0][1] < losses[0][0]
    assert losses[0] == losses[1]
    np.testing.assert_array_equal(ws[0], ws[1])


def test_bench_eight_ranks_rehearsed_on_one_gpu():
    """VERDICT r4 next #5a: BASELINE configs[2] (8 GPUs, global batch 64) has never run for want of an 8-GPU node; what one
    GPU can rehearse is the 8-rank LAUNCH: `python bench.py --gpus 8` starts eight child ranks itself (before any GPU call),
    WAVENET_BENCH_SHARE_GPU=1 puts them all on cuda:0 over gloo, every rank captures the two-graph data-parallel step on its own
    shard of the GLOBAL batch (make_batch(rank, 8): clip b carries phi_b = 2 pi b / 64), the gradient arena is all-reduced
    over eight ranks between the graphs, timing is barrier-bracketed with MAX over ranks.  Checked: one JSON line, world size
    8, eight rank records, global batch 64, and every rank's first clip is the clip make_batch gives that rank (all eight
    different).  It measures no scaling -- it removes "first time eight ranks ever ran" from the day a node appears."""
    import json
    import subprocess
    import sys
    import bench
    from wavenet_amd import data
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WAVENET_BENCH_SHARE_GPU="1", OMP_NUM_THREADS="2")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-decode", "--no-wide"], env=env, capture_output=True, text=True,
                       timeout=1500, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["dist"]["world_size"] == 8 and len(out["dist"]["ranks"]) == 8
    assert out["config"]["global_batch"] == 64 and out["config"]["parallelism"] == "dp8" and out["scaling"] == "weak"
    assert np.isfinite(out["loss"]) and out["value"] > 0
    heads = []
    for rk in sorted(out["dist"]["ranks"], key=lambda d: d["rank"]):
        assert "fwd+bwd graph" in rk["launch"], rk
        b0 = rk["rank"] * bench.B_PER_GPU
        assert rk["global_clips"] == [b0, b0 + bench.B_PER_GPU]
        wav = data.synthetic_waveform(1, bench.T + 1, 16000, b0=b0, Btot=8 * bench.B_PER_GPU)  # that rank's first clip
        want = data.mulaw_encode(wav)[0, :16].tolist()
        assert rk["first_tokens"] == want, (rk["rank"], rk["first_tokens"], want)
        heads.append(tuple(rk["first_tokens"]))
    assert len(set(heads)) == 8                                                            # eight different shards


def test_cfg4_batched_decode_equals_the_single_utterance_runs_bit_for_bit():
    """VERDICT r4 next #9 / SURVEY 8(e) "AR decode: replicas only" on ONE GPU: wn_decoder_run_batch runs N independent
    utterances (nine workgroups each, own decoder state) in one launch.  The groups share nothing, so row u of
    generate_batch(n, uniforms) must equal generate(n, uniforms[u]) token for token -- checked for a full device (28
    utterances = 252 workgroups), for an odd count (5), and against the committed oracle trace for the utterance that uses
    the fixture's own uniforms (first 1,500 samples)."""
    z = np.load(os.path.join(G, "cfg4_decode_trace.npz"))
    net = FasterWaveNet(Params(R.make_params(**CFG2)), seed=1234)
    net.to_gpu()
    n = 1500
    rs = np.random.RandomState(11)
    for N in (5, 28):
        u = rs.random_sample((N, n))
        u[0] = z["uniforms"][:n]
        got = to_np(net.generate_batch(n, u))
        assert got.shape == (N, n)
        np.testing.assert_array_equal(got[0], z["tokens"][:n].astype(np.int32))           # the oracle's trace
        for i in (1, N // 2, N - 1):
            want = to_np(net.generate(n, u[i]))
            np.testing.assert_array_equal(got[i], want, err_msg="utterance %d of %d" % (i, N))
        assert len({tuple(r[:64]) for r in got}) == N                                      # and they ARE different utterances


def test_batched_decode_rejects_bad_arguments_without_touching_the_device():
    """wn_decoder_run_batch through the C ABI: a handle given twice, more utterances than wn_decoder_batch_max(), a single
    step (the nine-workgroup form needs two or more) and a decoder created with WN_DECODER_ONE_WORKGROUP are refused with a
    message (WN_EARG / WN_ESHAPE); wn_decoder_status of a handle that never ran the nine-workgroup form is WN_OK."""
    import ctypes as C
    from wavenet_amd import _lib
    lib = _lib.lib()
    nets = []
    for flags in (None, _lib.WN_DECODER_ONE_WORKGROUP):
        net = FasterWaveNet(Params(R.make_params(**CFG2)), seed=1)
        net.to_gpu()
        net.exec_flags = flags
        nets.append(net)
    h0, h1 = nets[0]._decoder(), nets[1]._decoder()
    assert lib.wn_decoder_status(h0, None) == 0
    nmax = lib.wn_decoder_batch_max()
    assert nmax == 28
    u = torch.zeros((nmax + 1, 8), device="cuda", dtype=torch.float64) + 0.5
    out = torch.zeros((nmax + 1, 8), device="cuda", dtype=torch.int32)
    firsts = (C.c_int32 * (nmax + 1))(*([127] * (nmax + 1)))
    ups = (C.c_void_p * (nmax + 1))(*[u[i].data_ptr() for i in range(nmax + 1)])
    ops = (C.c_void_p * (nmax + 1))(*[out[i].data_ptr() for i in range(nmax + 1)])

    def call(handles, n):
        hs = (C.c_void_p * len(handles))(*[h.value for h in handles])
        rc = lib.wn_decoder_run_batch(hs, len(handles), firsts, ups, n, ops, None, 1, None)
        return rc, lib.wn_last_error().decode()

    rc, msg = call([h0, h0], 8)
    assert rc == -1 and "twice" in msg, (rc, msg)
    rc, msg = call([h0] * (nmax + 1), 8)
    assert rc == -2 and "at most" in msg, (rc, msg)
    rc, msg = call([h0], 1)
    assert rc == -2, (rc, msg)
    rc, msg = call([h1], 8)
    assert rc == -2 and "ONE_WORKGROUP" in msg, (rc, msg)
    torch.cuda.synchronize()
    assert int(out.abs().sum()) == 0                                    # nothing ran


def test_generate_batch_falls_back_to_single_runs_where_the_batched_launch_does_not_reach():
    """generate_batch decides BEFORE any work whether wn_decoder_run_batch covers the request and otherwise
    runs generate() per utterance -- n_samples == 2 (the launch needs two steps or more), a model the specialised decoder
    does not take (16 channels), WN_DECODER_ONE_WORKGROUP -- instead of raising after the prefill; and through the C ABI a
    batch whose handles were packed from DIFFERENT weights is refused when same_weights = 1 (it would decode every utterance
    with handle 0's weights)."""
    import ctypes as C
    from wavenet_amd import _lib
    lib = _lib.lib()
    rs = np.random.RandomState(2)
    net = FasterWaveNet(Params(R.make_params(**CFG2)), seed=1234)
    net.to_gpu()
    u = rs.random_sample((3, 2))
    got = to_np(net.generate_batch(2, u))
    assert got.shape == (3, 2) and len(net._batch_decs) == 0                               # no handle was created for nothing
    for i in range(3):
        np.testing.assert_array_equal(got[i], to_np(net.generate(2, u[i])))
    net.exec_flags = _lib.WN_DECODER_ONE_WORKGROUP
    u = rs.random_sample((2, 40))
    got = to_np(net.generate_batch(40, u))
    np.testing.assert_array_equal(got[1], to_np(net.generate(40, u[1])))
    p, w, small = build(dict(quantization_steps=256, causal_conv_channels=[16], residual_conv_channels=[16] * 4,
                             residual_num_blocks=2, softmax_conv_channels=[32, 256]), cls=FasterWaveNet)
    got = to_np(small.generate_batch(20, u[:, :20]))
    for i in range(2):
        np.testing.assert_array_equal(got[i], R.generate(p, w, 20, u[i, :20], fast=True))
    # same_weights = 1 with handles of two different models
    a = FasterWaveNet(Params(R.make_params(**CFG2)), seed=1)
    b = FasterWaveNet(Params(R.make_params(**CFG2)), seed=2)
    a.to_gpu(); b.to_gpu()
    hs = (C.c_void_p * 2)(a._decoder().value, b._decoder().value)
    ud = torch.zeros((2, 8), device="cuda", dtype=torch.float64) + 0.5
    out = torch.zeros((2, 8), device="cuda", dtype=torch.int32)
    firsts = (C.c_int32 * 2)(127, 127)
    ups = (C.c_void_p * 2)(ud[0].data_ptr(), ud[1].data_ptr())
    ops = (C.c_void_p * 2)(out[0].data_ptr(), out[1].data_ptr())
    rc = lib.wn_decoder_run_batch(hs, 2, firsts, ups, 8, ops, None, 1, None)
    assert rc == -1 and "other weights" in lib.wn_last_error().decode()
    assert lib.wn_decoder_run_batch(hs, 2, firsts, ups, 8, ops, None, 0, None) == 0           # own weights each: fine
    torch.cuda.synchronize()


@pytest.mark.parametrize("prec", ["fp16x2", "bf16x3", "fp32"])
@pytest.mark.parametrize("B,extra", [(2, 700), (8, 12290)])
def test_step_plan_graph_equals_the_graph_without_a_plan_bit_for_bit(B, extra, prec):
    """VERDICT r5 next #5 (the launch floor): a step plan (wn_plan_*) hoists the weight-only preparation of the step -- zero a
    range word, measure max |W|, split, per channel GEMM; the fp16 x 2 layer images; the dataflow words of the multi-layer
    backward; the range pass over dskip; cleargrads -- into the two launches of wn_plan_prepare at the start of the graph.  The
    images are built by the same device code from the same weights, so NOTHING may change: loss, every gradient and the
    weights after three optimiser steps equal the plan-less graph's bit for bit (BASELINE config 2's stack, a small window
    and the bench's own batch shape), the plan served the entry points' look-ups (none missed), and the graph is shorter."""
    from wavenet_amd import _lib
    rs = np.random.RandomState(B + extra)
    res = {}
    for use_plan in (False, True):
        net = WaveNet(Params(R.make_params(**CFG2)), seed=1234)
        net.gemm_precision = prec
        net.use_step_plan = use_plan
        net.to_gpu()
        net.update_laerning_rate(1e-3)
        iw = net.input_width
        rs = np.random.RandomState(B + extra)
        xs = [dev(rs.randint(0, 256, (B, iw + extra)).astype(np.int32)) for _ in range(3)]
        ts = [dev(rs.randint(0, 256, (B, extra)).astype(np.int32)) for _ in range(3)]
        g = TrainStepGraph(net, xs[0], ts[0], keep_graph=True)
        losses = []
        for x, t in list(zip(xs, ts))[:1 if prec == "fp32" else 3]:      # (fp32: see below)
            losses.append(float(g.step(x, t)))
        torch.cuda.synchronize()
        res[use_plan] = (losses, to_np(net._grad_arena).copy(), to_np(net._arena).copy(), g.node_counts()["kernel"],
                         net.plan_stats() if use_plan else None)
        del g, net
    a, b = res[False], res[True]
    if prec != "fp32":
        assert a[0] == b[0], (a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
        np.testing.assert_array_equal(a[2], b[2])
    else:
        # exact-fp32 mode is not bit-reproducible from run to run with or without a plan: its skip / head weight gradients add
        # their time slabs with float atomics (k_wgrad_mfma) -- summation order, ~1e-7 of a tensor's largest entry -- and Adam
        # turns that into O(lr) differences of single weights within a few steps (m / sqrt(v) of a tiny gradient): ONE step
        np.testing.assert_allclose(a[0], b[0], rtol=1e-6)
        np.testing.assert_allclose(a[1], b[1], rtol=0, atol=2e-6 * float(np.abs(a[1]).max()))
    st = b[4]
    assert st["state"] == 2 and st["not_served"] == 0 and st["served"] > 0 and st["prepare_calls"] >= 2, st
    if prec == "fp16x2":
        assert st["weight_images"] == 4 and st["layer_images"] == 40 and st["plan_words"] > 2, st
        assert b[3] <= a[3] - 12, (a[3], b[3])          # ~16 launch-floor kernels became 2
    elif prec == "bf16x3":
        assert st["weight_images"] >= 3 and b[3] < a[3], (st, a[3], b[3])
    assert np.isfinite(a[0]).all()

