#!/usr/bin/env python3
"""Benchmark of the WaveNet hot path on MI355X (BASELINE.json metric: audio samples/sec, train
fwd+bwd and AR-generate, 4x10-layer dilated stack).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one training step of config 2 (4 blocks x 10 dilations, 32 residual / 256 skip channels,
fp32, 8 clips x 16,384 samples per GPU, loss over the last 12,290 columns as train_audio/train.py
does): forward, cross-entropy, backward, [RCCL all-reduce], clip + Adam.  Inputs (tokens, targets)
are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.  Besides `value` (train
samples/s over all GPUs) the line carries the stack-forward rate, the AR-decode rate (config 4,
N=1 only), `roofline` for the dominant kernel and `cpu_baseline` (the oracle's literal
"Chainer-equivalent" restatement timed on this box's host cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from wavenet_amd import FasterWaveNet, Params, TrainStepGraph, _lib, data     # noqa: E402

CFG2 = dict(quantization_steps=256, sampling_rate=16000, causal_conv_channels=[32],
            residual_conv_channels=[32] * 10, residual_num_blocks=4, softmax_conv_channels=[256, 256])
B_PER_GPU, T = 8, 16384
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
F32_MFMA_PEAK_TF = 157.3       # v_mfma_f32_32x32x2_f32 dense peak
BF16X3_PEAK_TF = 2500.0 / 6.0  # fp32-equivalent peak of a 6-term bf16 split product on the ~2.5 PF dense bf16 MFMA
# what the matrix cores multiply in each mode of WnExec.precision (storage and accumulation are fp32 in all of them)
GEMM_MODE_TEXT = {
    "fp32": "every contraction on fp32-input MFMA (v_mfma_f32_32x32x2_f32): exact fp32 products",
    "bf16x3": "skip-path and head contractions on bf16x3 split products (three bf16 parts per operand, six "
              "v_mfma_f32_32x32x16_bf16 per product, error <= 3*2^-27 relative per product); the fused 32-channel layer "
              "kernels (dilated convs, gate, residual projection, their backward) on exact fp32-input MFMA",
    "fp16x2": "EVERY contraction of the step except the head's backward -- the fused layer forward, the chained layer "
              "backward, the skip sum, dz, dWs, the head forward (fused with the loss: wn_head_xent) -- on fp16x2 split "
              "products: operands scaled by a power of two (per tile / per chunk from the wave's own maximum in the layer "
              "kernels and the head, from a measured absmax on the skip path) and split into two fp16 parts, three "
              "v_mfma_f32_32x32x16_f16 per product (error <= 2^-21 relative per product + 2^-24 of the tile maximum), fp32 "
              "accumulation; the head's dx / dW contractions on bf16x3 (six terms).  Not the reference's fp32 "
              "products: held to the same parity bars (logits 1e-4, every gradient 1e-4 relative, tokens bit-exact)",
    "bf16": "operands rounded to bf16 once, fp32 accumulation (config 5's arithmetic; not fp32-accurate)",
}


ARITH_SHORT = {
    "fp32": "f32 x f32 products on v_mfma_f32_32x32x2_f32, f32 accumulate (the reference's arithmetic)",
    "bf16x3": "f32 operands split into 3 bf16 parts, 6 v_mfma_f32_32x32x16_bf16 per product (<= 3*2^-27 per product: "
              "fp32-accurate), f32 accumulate; fused layer kernels on f32 MFMA",
    "fp16x2": "f32 operands scaled by a power of two and split into 2 fp16 parts, 3 v_mfma_f32_32x32x16_f16 per product "
              "(<= 2^-21 per product), f32 accumulate; the head's backward contractions on bf16x3",
    "bf16": "operands rounded to bf16 once, f32 accumulate",
}


def newest_profile(pattern):
    """Newest committed profiles/r<N>_<pattern> file (highest round number), or None."""
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_" + pattern)):
        m = re.match(r"r(\d+)_" + re.escape(pattern) + "$", os.path.basename(f))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    return best[1] if best else None


# `dtype` of the JSON line: what the path computes in, said so that the line needs no footnote ("keep
# fp16x2 as `value` but rename `dtype` to say so").  Storage and accumulation are fp32 in every mode.
DTYPE_TEXT = {"fp32": "f32", "bf16x3": "f32 (products: 3 x bf16 split, 6 MFMA terms; fp32-exact layer kernels)",
              "bf16": "f32 storage / bf16 products", "fp16x2": "f32 storage+accumulate / fp16x2 split products (2^-21)"}


def arith_record(mode):
    """What multiplies in the timed step, and how far that arithmetic is from the float64 truth next to the exact-fp32
    mode (measured by tests/test_gpu_arith_error.py on the GPU; its committed output is quoted, not recomputed: the oracle
    may not run inside the bench's timed legs)."""
    rec = {"mode": mode, "storage": "f32", "accumulate": "f32", "multiply": ARITH_SHORT[mode],
           "like_for_like_with_reference": mode == "fp32",
           "note": "value / ms_per_step are measured in `mode`; ms_per_step_by_mode holds the same captured step in every "
                   "shipped mode (fp32 = exact fp32 products, the reference's arithmetic)",
           "why_not_an_fp32_accurate_headline": "measured round 5 (gpurun_out/bymode.log, DESIGN.md): the skip-path contractions "
                   "alone cost + 0.372 ms with six-term (fp32-accurate) products instead of three (1.133 vs 0.761 ms) -- "
                   "+ 12.3 % of the 3.026 ms step before the fused layer kernels are touched (they would need twice the MFMAs "
                   "and ~twice the split instructions; on exact fp32 MFMA they cost another + 0.82 ms) -- so an accurate mode "
                   "cannot come within 12 % of fp16x2; the distance of BOTH from the float64 truth is the same "
                   "(error_vs_float64_truth)"}
    f = newest_profile("arith_error_vs_fp64.json")
    if f:
        doc = json.load(open(f))
        m = doc.get("modes", {}).get(mode)
        if m:
            rec["error_vs_float64_truth"] = {
                "source": os.path.relpath(f, ROOT), "workload": doc.get("workload"),
                "logits_max_abs_err": m["logits_max_abs_err"], "grad_err_largest_over_tensors": m["grad_err_largest_over_tensors"],
                "ratio_to_exact_fp32_mode": m["vs_fp32_mode"],
                "exact_fp32_mode": {k: doc["modes"]["fp32"][k] for k in ("logits_max_abs_err", "grad_err_largest_over_tensors")}}
    return rec


def make_batch(rank, world, iw):
    """Synthetic 16 kHz clips -> mu-law tokens; targets are the next sample (train.py:14-22)."""
    wav = data.synthetic_waveform(B_PER_GPU, T + 1, 16000, b0=rank * B_PER_GPU, Btot=world * B_PER_GPU)
    tok = data.mulaw_encode(wav)
    x = tok[:, :T]
    tgt = tok[:, iw + 1:T + 1]
    return torch.as_tensor(x).cuda(), torch.as_tensor(tgt).cuda()


def train_step(net, x, tgt, iw):
    c = net.forward_causal_block(x)
    # skip sum for the columns train.py:73 keeps.  (window_only=True would also skip the ~12 % of sample-layers the window
    # cannot see -- same loss and gradients -- but buys nothing at this size, so the bench computes every column like the
    # reference does.)
    _, s = net.forward_residual_block(c, t_off=iw)
    # forward_softmax_block(s, apply_softmax=False) + cross_entropy (train.py:75-76) through WaveNet.head_cross_entropy: the last
    # head convolution and the loss in ONE launch where the library covers it (config 2 does), the same two calls where not
    loss = net.head_cross_entropy(s, tgt)
    net.backprop(loss)
    return loss


def wide_channel_step(rank, world, steps=10, warmup=3):
    """BASELINE config 5 as a side measurement (never the headline value): config 2's topology and batch at 128 residual /
    dilation channels and 512 skip channels in bf16 STORAGE (wavenet_amd.WaveNet(storage="bf16"): bf16 activations in HBM,
    fp32 accumulation on the bf16 matrix cores, fp32 master weights), timed as graph replays like the headline.  Reports the
    matrix-core rate against the dense bf16 peak and the per-entry-point times of an op-by-op pass."""
    cfg = dict(CFG2)
    cfg.update(causal_conv_channels=[128], residual_conv_channels=[128] * 10, softmax_conv_channels=[512, 256])
    net = FasterWaveNet(Params(cfg), seed=1, storage="bf16")
    net.to_gpu()
    net.update_laerning_rate(1e-3)
    iw = net.input_width
    x, tgt = make_batch(rank, world, iw)
    graph = TrainStepGraph(net, x, tgt)
    for _ in range(warmup):
        graph.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = graph.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    with _lib.profile() as prof:
        for _ in range(3):
            train_step(net, x, tgt, iw)
        torch.cuda.synchronize()
    per = {k: round(v[1] / 3, 4) for k, v in sorted(prof.result().items(), key=lambda kv: -kv[1][1])}
    nl, n = 40, B_PER_GPU * T
    # 2 (2 fw Cr Cd + Cd Cr + Cd Cs) flop per sample-layer forward, backward = 2x forward
    flop = 3 * 2 * (2 * 2 * 128 * 128 + 128 * 128 + 128 * 512) * nl * n
    # algorithmic HBM bytes per sample-layer of this design: forward 768 (x in, out + z out),
    # gate backward 1,280, dx 1,280, conv weight gradients 768, skip path 3 x 256 + dskip share
    res = {"workload": "cfg5: 4x10 layers, 128 residual/dilation + 512 skip channels, batch %d x %d, train fwd+bwd+clip+Adam"
                       % (B_PER_GPU, T),
           "dtype": "bf16 storage and MFMA operands, f32 accumulate, f32 master weights", "launch": "hipGraph replay",
           "ms_per_step": dt * 1e3, "samples_per_s": n / dt, "loss": float(loss.detach()),
           "entry_point_ms_per_step": per,
           "mfma": {"flop_per_step": flop, "achieved": flop / dt / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                    "frac": flop / dt / 1e12 / 2500.0,
                    "note": "algorithmic flops / step time / dense bf16 peak; the rocprofv3 MFMA-busy counters of the same "
                            "step are in %s" % (os.path.relpath(newest_profile("cfg5_mfma_busy.json") or "profiles/", ROOT))}}
    del graph, net
    torch.cuda.empty_cache()
    return res


def stack_forward(net, c):
    with torch.no_grad():
        return net.forward_residual_block(c)


def timed(fn, steps, warmup, barrier=None):
    for _ in range(warmup):
        fn()
    if barrier:
        barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    if barrier:
        barrier()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def cpu_baseline(budget_s=25.0):
    """The oracle's literal restatement (oracle/wavenet_ref.py) on the host cores: train fwd+bwd at
    config 2's topology, B=1 x T=16384 (CPU samples/s is batch-independent), and the full-window
    fast decode (faster_wavenet.py:50-113 restated: caches rolled every step)."""
    from oracle import wavenet_ref as R
    from oracle import data_ref as D
    host = os.cpu_count() or 1
    p = R.make_params(**{k: v for k, v in CFG2.items() if k != "sampling_rate"})
    w = R.init_weights(p, 1234)
    iw = R.input_width(p)
    tok = D.mulaw_quantize(D.synthetic_waveform(1, T + 1, 16000))
    # Thread count: climb the ladder 8, 16, 32 ... up to every host core and keep the fastest.  os.cpu_count() is the
    # HOST's count; a container may be allowed far fewer, and torch's intra-op pools with more threads than cores it can
    # run on do not degrade gently (measured on a 256-CPU GPU box: 377 s per step with 256 threads against ~0.1 s with
    # 16) -- so the climb stops at the first rung that is slower, and only a small crop is ever run on an untested rung.
    Tp = iw + 512
    xp, tp = tok[:, :Tp], tok[:, iw + 1:Tp + 1]
    tried, cores, pilot = {}, 1, None
    rungs = sorted({min(host, c) for c in (8, 16, 32, 64, 128, host)})
    for i, c in enumerate(rungs):
        torch.set_num_threads(c)
        if i == 0:
            R.train_step_grads(p, w, xp, tp)                                  # warm-up (allocator, thread pool)
        t0 = time.perf_counter()
        R.train_step_grads(p, w, xp, tp)
        dt = time.perf_counter() - t0
        tried[c] = Tp / dt
        if pilot is None or dt < pilot:
            cores, pilot = c, dt
        elif dt > 1.2 * pilot:
            break
    torch.set_num_threads(cores)
    # bounded sample: the pilot sizes the timed crop so that the leg stays within its budget
    Tc = int(min(T, max(iw + 1024, budget_s * 0.2 / max(pilot / Tp, 1e-9))))
    x, tgt = tok[:, :Tc], tok[:, iw + 1:Tc + 1]
    reps, ts = 3, []
    t_leg = time.perf_counter()
    for _ in range(reps):
        t0 = time.perf_counter()
        R.train_step_grads(p, w, x, tgt)
        ts.append(time.perf_counter() - t0)
        if time.perf_counter() - t_leg > 2.0 * budget_s:                      # never let this leg run away
            break
    reps = len(ts)
    train_sps = Tc / float(np.median(ts))
    fast = R.RefFasterWaveNet(p, w)
    buf = np.full((iw,), 127, np.int32)
    fast._forward_one_step(D.onehot_pixel_image(buf.reshape(1, -1), 256))   # prefill
    t0 = time.perf_counter()
    n = 0
    while n < 200 and time.perf_counter() - t0 < budget_s * 0.35:
        buf = np.append(buf[1:], [n % 256]).astype(np.int32)
        fast._forward_one_step(D.onehot_pixel_image(buf.reshape(1, -1), 256))
        n += 1
    dec_sps = n / (time.perf_counter() - t0)
    return {"value": train_sps, "unit": "samples/s", "cores": cores, "host_cpu_count": host, "kind": "port",
            "threads_tried_samples_per_s": {str(k): round(v, 1) for k, v in tried.items()},
            "sample": "oracle literal restatement (Chainer-equivalent op sequence, not Chainer), torch-CPU fp32, "
                      "%d threads: train fwd+bwd cfg2 topology B=1 x T=%d, median of %d steps; "
                      "fast decode %d steps at W=4094" % (cores, Tc, reps, n),
            "decode_value": dec_sps, "decode_unit": "samples/s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--decode-samples", type=int, default=16000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--no-exact-fp32", action="store_true", help="skip the exact-fp32-MFMA timing of the same step")
    ap.add_argument("--no-wide", action="store_true", help="skip the config-5 (128/512 channels, bf16 operands) side measurement")
    ap.add_argument("--no-graph", action="store_true", help="time op-by-op launches instead of hipGraph replays")
    ap.add_argument("--wide-only", action="store_true", help="only the config-5 side measurement (profiling passes)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by itself: start the N ranks as CHILD processes (one per GPU, torch.distributed.run on
        # 127.0.0.1) and pass rank 0's JSON line through.  Nothing in this process has touched the GPU yet (no HIP call,
        # no torch.cuda query), and nothing is exec'ed: the launcher is a subprocess and we exit with its code.
        import socket
        import subprocess
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on this pool (RCCL needs it)
        env.setdefault("OMP_NUM_THREADS", "8")
        raise SystemExit(subprocess.call(cmd, env=env))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node equal to --gpus" % (args.gpus, world))
    if os.environ.get("WAVENET_BENCH_SHARE_GPU") == "1":      # test hook: several ranks on one device over gloo
        local = 0
    torch.cuda.set_device(local)
    barrier = None
    # WAVENET_BENCH_FORCE_DIST=1 (test hook): run the data-parallel branch -- process group on the real backend, two-graph step
    # with the all-reduce between the graphs, MAX over ranks, the `dist` record -- in a group of ONE rank, which is how a
    # one-GPU box executes RCCL itself (tests/test_gpu_dp.py::test_rccl_world_size_one...)
    force_dist = world == 1 and os.environ.get("WAVENET_BENCH_FORCE_DIST") == "1"
    if force_dist:
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("WAVENET_BENCH_SHARE_GPU") == "1":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))   # "nccl" is RCCL on ROCm
        barrier = dist.barrier

    if args.wide_only:
        print(json.dumps({"wide_channel": wide_channel_step(rank, world, steps=args.steps, warmup=args.warmup)}))
        return
    p = Params(CFG2)
    net = FasterWaveNet(p, seed=1234)
    net.to_gpu()
    net.update_laerning_rate(0.001)                                     # train_audio/args.py --lr default
    if world > 1 or force_dist:
        net.enable_data_parallel(always_reduce=force_dist)
    iw = net.input_width
    x, tgt = make_batch(rank, world, iw)
    assert _lib.lib().wn_layer_fast_path(32, 32, 2) == 1

    # ---- the timed region: K training steps ------------------------------------------------
    # Default: the step is captured once into a HIP graph (wavenet_amd.TrainStepGraph: cleargrads, forward, loss,
    # backward, clip + Adam; with N > 1 the RCCL all-reduce runs between a forward/backward graph and an optimiser
    # graph) and each timed step is one replay.  --no-graph times the same step launched op by op.
    graph = None
    # N > 1: forward/backward graph -> RCCL all-reduce (launched normally, never captured) -> optimiser graph, so that every N
    # is measured in the same launch form (WAVENET_BENCH_GRAPH_DP=0 or --no-graph: op-by-op launches).  The graphs hold only
    # this library's kernels; the communicator is created by the first eager all-reduce, outside any capture.
    use_graph = not args.no_graph and (world == 1 or os.environ.get("WAVENET_BENCH_GRAPH_DP", "1") != "0")
    if use_graph:
        try:
            graph = TrainStepGraph(net, x, tgt)
        except Exception as e:                                           # keep the bench alive: fall back to op-by-op
            sys.stderr.write("graph capture failed (%s: %s); timing op-by-op launches\n" % (type(e).__name__, e))
            graph = None
    step_fn = (lambda: graph.step()) if graph is not None else (lambda: train_step(net, x, tgt, iw))
    # Clock spin-up, untimed, before the W warm-up steps the contract asks for: on a cold GPU the first ~50 steps run 2 %
    # slower than the steady state (measured: 4.26 ms with 3 warm-up steps, 4.16 ms with 100), so a short timed region right
    # after a 3-step warm-up measures the ramp, not the kernel.  Same code path as the timed steps.
    spinup = max(0, 60 - args.warmup)
    # The FIRST step (seed-1234 weights, this batch) against the oracle's loss for exactly that step, committed as
    # tests/golden/cfg2_bench_step.npz (tests/golden/make_golden.py::cfg2_bench_step; N = 1 only: the fixture is rank 0's batch
    # of a world of one).  Like golden_tokens_match for the decode: the bench checks the thing it times, before it times it.
    first_loss = float(step_fn().detach())
    golden = None
    gfile = os.path.join(ROOT, "tests", "golden", "cfg2_bench_step.npz")
    if world == 1 and os.path.exists(gfile):
        gz = np.load(gfile)
        if [int(x.sum().item()), int(tgt.sum().item())] == gz["tokens_checksum"].tolist():
            golden = float(gz["loss"])
    for _ in range(max(0, spinup - 1)):
        step_fn()
    for _ in range(args.warmup):
        step_fn()
    if barrier:
        barrier()
    torch.cuda.synchronize()
    # EXACTLY K steps between two barrier + synchronize brackets (the wall mean, MAX over ranks: `ms_per_step_wall_mean`), and a
    # HIP event on the launch stream after every step ("hipEvents, median"): `ms_per_step` is the MEDIAN of the K
    # per-step event intervals (MAX over ranks), `value` follows from it.  (The C ABI is called on torch's current stream, so
    # torch's events are events on that stream.)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(args.steps):
        loss = step_fn()
        evs[i + 1].record()
    if barrier:
        barrier()
    torch.cuda.synchronize()
    wall_dt = (time.perf_counter() - t0) / args.steps
    step_ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps))
    dt = 1e-3 * (step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2]))
    local_dt = dt
    if world > 1 or force_dist:
        tt = torch.tensor([dt, wall_dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, wall_dt = float(tt[0].item()), float(tt[1].item())
    # ---- the same K steps launched op by op under the in-library HIP-event profiler: per-entry-point kernel time
    # (events on the launch stream around every C-ABI call; a replayed graph has no host code to record them)
    with _lib.profile() as prof:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            train_step(net, x, tgt, iw)
        torch.cuda.synchronize()
        eager_dt = (time.perf_counter() - t0) / args.steps
    per_step = {k: v[1] / args.steps for k, v in prof.result().items()}  # ms of each entry point per step
    # the step plan of the timed graph (wn_plan_*: the weight-only preparation of the step in two launches)
    step_plan = None
    if graph is not None and getattr(graph, "_use_plan", False):
        try:
            net.plan_use(graph._plan)
            step_plan = net.plan_stats()
        except Exception as e:
            step_plan = {"error": "%s: %s" % (type(e).__name__, e)}
        finally:
            net.plan_off()
    # what ONE replayed step launches: the node counts of the same step captured once more with the hipGraph_t kept
    graph_nodes = None
    if graph is not None:
        try:
            graph_nodes = TrainStepGraph(net, x, tgt, keep_graph=True).node_counts()
        except Exception as e:
            graph_nodes = {"error": "%s: %s" % (type(e).__name__, e)}
    samples = world * B_PER_GPU * T
    value = samples / dt

    out = {
        "metric": "audio samples/sec: train fwd+bwd, 4x10-layer dilated stack (cfg2)", "value": value,
        "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "spinup_steps": spinup,
        "ms_per_step": dt * 1e3, "ms_per_step_wall_mean": wall_dt * 1e3,
        "ms_per_step_min_max": [step_ms[0], step_ms[-1]],
        "timing": "ms_per_step = median of the K per-step HIP-event intervals on the launch stream (MAX over ranks); "
                  "ms_per_step_wall_mean = wall clock over the K steps between barrier + synchronize brackets / K",
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": DTYPE_TEXT[_lib.get_gemm_precision()], "data": "synthetic",
        "arith": arith_record(_lib.get_gemm_precision()),
        "gemm_mode": GEMM_MODE_TEXT[_lib.get_gemm_precision()],
        "dtype_note": "dtype names STORAGE and ACCUMULATION (fp32 tensors in HBM, fp32 MFMA accumulators); what the matrix "
                      "cores multiply is in gemm_mode; exact_fp32_ms_per_step is the same step with every contraction on "
                      "fp32-input MFMA (v_mfma_f32_32x32x2_f32)",
        "config": {"workload": "cfg2 train step: 4 blocks x 10 dilations (1..512), 32 residual / 256 skip ch, "
                               "16 kHz, %d clips x 16384 samples per GPU, loss over last 12290 columns, "
                               "fwd + bwd + clip + Adam%s" % (B_PER_GPU, " + RCCL all-reduce" if world > 1 or force_dist else ""),
                   "global_batch": world * B_PER_GPU, "seq_len": T, "parallelism": "dp%d" % world},
        "loss": float(loss.detach()),
        "first_step_loss": first_loss, "golden_loss": golden,
        "golden_loss_match": (abs(first_loss - golden) <= 1e-4) if golden is not None else None,
        "golden_loss_source": "tests/golden/cfg2_bench_step.npz: the oracle's loss for this batch and the seed-1234 weights "
                              "(bar 1e-4); `loss` is the loss after the %d optimiser steps of this run" % (spinup + args.warmup + args.steps),
        "launch": ("hipGraph replay, one graph launch per step" + (" (fwd+bwd graph, RCCL all-reduce, optimiser graph)"
                   if world > 1 or force_dist else "")) if graph is not None else "op-by-op launches from Python",
        "eager_ms_per_step": eager_dt * 1e3, "graph_nodes_per_step": graph_nodes,
        "step_plan": step_plan,
        "entry_point_ms_per_step": {k: round(v, 4) for k, v in sorted(per_step.items(), key=lambda kv: -kv[1])},
    }

    if rank == 0 and world == 1 and not force_dist and not args.no_exact_fp32:
        # ---- the SAME captured step in every shipped arithmetic mode: 10 graph replays each after 3 warm-up replays.
        # fp32 = the reference's own arithmetic (exact fp32 products); bf16x3 = fp32-accurate split products on the skip path
        # and the head, exact fp32 MFMA in the layer kernels; fp16x2 = the default (see `arith`).
        by_mode = {}
        default_mode = _lib.get_gemm_precision()
        for mode in ("fp32", "bf16x3", "fp16x2"):
            if mode == default_mode and graph is not None:
                by_mode[mode] = dt * 1e3                                # the timed region above
                continue
            try:
                net.gemm_precision = mode
                gm = TrainStepGraph(net, x, tgt)
                by_mode[mode] = timed(lambda: gm.step(), 10, 3) * 1e3
                del gm
            except Exception as e:
                by_mode[mode] = None
                sys.stderr.write("%s step failed (%s: %s)\n" % (mode, type(e).__name__, e))
            finally:
                net.gemm_precision = None
                torch.cuda.empty_cache()
        out["ms_per_step_by_mode"] = by_mode
        out["samples_per_s_by_mode"] = {k: (samples / (v * 1e-3) if v else None) for k, v in by_mode.items()}
        out["exact_fp32_ms_per_step"] = by_mode.get("fp32")
        out["exact_fp32_samples_per_s"] = out["samples_per_s_by_mode"].get("fp32")
        # the like-for-like figures inside `config`, so that a reader of the parsed line alone sees them next to `value`:
        # bf16x3 = fp32-accurate products (<= 3 * 2^-27), fp32 = the reference's own arithmetic; `value` is in arith_mode
        out["config"].update({"arith_mode": default_mode,
                              "like_for_like_samples_per_s": out["samples_per_s_by_mode"].get("bf16x3"),
                              "like_for_like_ms_per_step": by_mode.get("bf16x3"),
                              "exact_fp32_samples_per_s": out["samples_per_s_by_mode"].get("fp32"),
                              "exact_fp32_ms_per_step": by_mode.get("fp32"),
                              "golden_loss_match": out.get("golden_loss_match")})
    if rank == 0 and world == 1 and not force_dist:
        # ---- fused residual-stack forward (the north star's roofline target) -------------------
        with torch.no_grad():
            c = net.forward_causal_block(x)
        timed(lambda: stack_forward(net, c), 2, 2)
        with _lib.profile() as prof2:
            sdt = timed(lambda: stack_forward(net, c), 10, 0)
        ms2 = prof2.result()
        nl = len(net._flat_layers)
        Cr, Cd, Cs = 32, 32, 256
        alg_bytes_layer = 4 * (2 * Cr + 2 * Cs) * B_PER_GPU * T            # 2,304 B per sample-layer
        alg_flops_layer = 2 * (2 * 2 * Cr * Cd + Cd * Cr + Cd * Cs) * B_PER_GPU * T
        layer_ms = ms2["wn_layer_fwd"][1] / 10 / nl           # one event bracket per stack call around its nl launches
        skip_ms = ms2["wn_skip_sum_fwd"][1] / 10
        stack_ms = layer_ms * nl + skip_ms
        out["stack_forward"] = {
            "samples_per_s": B_PER_GPU * T / sdt, "ms_wall": sdt * 1e3, "ms_kernels": stack_ms,
            "wn_layer_fwd_ms": layer_ms, "wn_skip_sum_fwd_ms": skip_ms,
            "algorithmic_GBps": nl * alg_bytes_layer / (stack_ms * 1e-3) / 1e9,
            "frac_of_hbm_8TBps": nl * alg_bytes_layer / (stack_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "TFLOPs": nl * alg_flops_layer / (stack_ms * 1e-3) / 1e12,
            "note": "algorithmic bytes = SURVEY 8(d)'s 2,304 B per sample-layer, which include the per-layer skip "
                    "read-modify-write; the deferred skip sum does not move those bytes, so frac_of_hbm_8TBps can exceed 1: "
                    "layer_traffic_frac = MEASURED HBM bytes of all the stack's layer launches (the launch plan x the PMC bytes "
                    "per launch of each kernel, layer_traffic_launches) / the time of the bracket around them / 8 TB/s; "
                    "traffic_frac adds the skip contraction's bytes and time",
        }
        # measured HBM bytes per launch: the newest rocprofv3 PMC summary committed under profiles/ (PMC counters cannot be
        # collected from inside the bench); `traffic_source` says which file, and the file says which commit it measured
        import re
        tfile = newest_profile("hbm_traffic.json")
        tdoc = json.load(open(tfile)) if tfile else {}
        pmc = tdoc.get("kernels", {})
        traffic_source = {"file": os.path.relpath(tfile, ROOT), "measured_at_commit": tdoc.get("commit"),
                          "note": tdoc.get("note")} if tfile else None

        def pmc_find(patterns):
            """(kernel name, HBM bytes per launch) of the first pattern that matches a kernel of the PMC summary; among the
            matches of one pattern the kernel with the most profiled launches.  Patterns are regular expressions on the
            demangled name, written so that a new trailing template parameter does not break them."""
            for pat in patterns:
                hits = [(v.get("fetch_launches_averaged", 0), k) for k, v in pmc.items() if re.match(pat, k)]
                if hits:
                    k = max(hits)[1]
                    return k, float(pmc[k]["hbm_bytes_per_launch"])
            return None, None

        def layer_fwd_traffic(save):
            """Measured HBM bytes of the stack's layer launches (inference form save=0, training form save=2): the launch
            plan of wn_stack_fwd (consecutive layers whose dilations add up to <= 32 share ONE grouped launch, at most 8
            layers: mfma_layer_fwd_group_len; every other layer is one launch) x the PMC bytes per launch of the kernel
            each launch runs.  Returns (bytes, [(kernel, launches, bytes per launch)]) or (None, None)."""
            dil = [lay.dilation for lay in net._flat_layers]
            plan, l = [], 0
            while l < len(dil):
                n, tot = 0, 0
                while l + n < len(dil) and n < 8 and tot + dil[l + n] <= 32:
                    tot += dil[l + n]
                    n += 1
                plan.append(n if n >= 2 else 1)
                l += plan[-1]
            parts, total = [], 0.0
            for kind, pat in (("grp", r"wn::k_layer_fwd_h2_grp<%d\b" % save), ("one", r"wn::k_layer_fwd_h2_t1<%d\b" % save)):
                cnt = sum(1 for n in plan if (n >= 2) == (kind == "grp"))
                if not cnt:
                    continue
                k, b = pmc_find([pat])
                if b is None:
                    return None, None
                parts.append({"kernel": k, "launches": cnt, "bytes_per_launch": b})
                total += cnt * b
            return total, parts
        fb, fparts = layer_fwd_traffic(0)
        if fb:
            out["stack_forward"]["layer_traffic_bytes"] = fb
            out["stack_forward"]["layer_traffic_launches"] = fparts
            out["stack_forward"]["layer_traffic_frac"] = fb / (layer_ms * nl * 1e-3) / 1e9 / HBM_PEAK_GBS
            sk, sb = pmc_find([r"wn::k_colgemm_h2q<0, 0", r"wn::k_colgemm_b3<0, 0,"])
            if sb:
                out["stack_forward"]["traffic_bytes"] = fb + sb
                out["stack_forward"]["traffic_frac"] = (fb + sb) / (stack_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        # ---- roofline of the dominant unit of the TIMED REGION (the training step) -------------------
        # Units = the library's per-op entry points; times are HIP-event means over the timed steps.
        # Algorithmic bytes / flops are SURVEY.md section 8(d)'s per-sample-layer figures x the
        # sample-layers one launch processes (B*T columns of one layer; or all layers for the
        # skip contractions).  `traffic` = real HBM bytes per launch from the rocprofv3 PMC passes
        # kept in profiles/ (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), null if absent.
        n_col = B_PER_GPU * T
        n_colw = B_PER_GPU * (T - iw)
        es = 4
        units = {
            # name: (bound, algorithmic amount per launch, launches per step, kernel-name patterns for the PMC traffic)
            "wn_layer_bwd": ("hbm", es * (3 * Cr + Cs + 2 * Cd) * n_col, nl, [r"wn::k_layer_bwd_chain\w*<true\b", r"wn::k_layer_bwd_chain"]),
            "wn_layer_fwd": ("hbm", es * (2 * Cr + 2 * Cs + 2 * Cd) * n_col, nl, [r"wn::k_layer_fwd_h2\w*<2\b", r"wn::k_layer_fwd_mfma32_t1<2\b"]),
            "wn_skip_sum_fwd": ("mfma", 2.0 * Cs * Cd * nl * n_colw, 1, [r"wn::k_colgemm_h2q<0, 0", r"wn::k_colgemm_b3<0, 0,"]),
            "wn_skip_sum_bwd_dw": ("mfma", 2.0 * Cs * Cd * nl * n_colw, 1, [r"wn::k_wgrad_h2p<0", r"wn::k_wgrad_b3w<false, 0,"]),
            "wn_skip_sum_bwd_dz": ("mfma", 2.0 * Cs * Cd * nl * n_colw, 1, [r"wn::k_colgemm_b3<2, 0,"]),
        }
        dom = max(units, key=lambda k: per_step.get(k, 0.0))
        bound, amount, launches, knames = units[dom]
        launch_ms = per_step[dom] / launches
        traffic_kernel, traffic = pmc_find(knames)
        kernel_launch = None
        if dom == "wn_layer_bwd":
            # The chained backward of the stack is ONE launch for every layer below the top one (k_layer_bwd_chain_multi) + the top
            # layer's own launch + the partial-tile reduction.  The roofline unit stays "one layer": bytes = (multi launch + top
            # layer) / layers, time = the bracket around all of it / layers.
            mk, mb = pmc_find([r"wn::k_layer_bwd_chain_multi"])
            tk, tb = pmc_find([r"wn::k_layer_bwd_chainsp<false, false, true"])
            if mb:
                traffic_kernel, traffic = mk, (mb + (tb or 0.0)) / nl
                kernel_launch = {"kernel": mk, "layers_per_launch": nl - 1, "hbm_bytes_per_launch": mb,
                                 "top_layer_kernel": tk, "top_layer_hbm_bytes": tb,
                                 "bracket_ms": per_step[dom],
                                 "note": "launch_ms / traffic / algorithmic bytes of `roofline` are per LAYER (bracket / %d layers); the "
                                         "rocprofv3 average duration of the multi-layer kernel is ~ bracket_ms minus the top layer's "
                                         "launch (~0.026 ms) and the partial-tile reduction (~0.044 ms)" % nl}
        # fp32-equivalent flops of one launch of the dominant unit
        layer_flops = {"wn_layer_bwd": 2 * 2 * (2 * 2 * Cr * Cd + Cd * Cr) * n_col, "wn_layer_fwd": 2 * (2 * 2 * Cr * Cd + Cd * Cr) * n_col}
        # matrix peak of the fused layer kernels: fp32 MFMA, or three f16 MFMAs per product under fp16x2
        layer_peak = 2500.0 / 3.0 if _lib.get_gemm_precision() == "fp16x2" else F32_MFMA_PEAK_TF
        if bound == "hbm":
            alg = amount / (launch_ms * 1e-3) / 1e9
            # primary figure: MEASURED HBM bytes per launch (the PMC passes committed under profiles/) / launch time; the
            # SURVEY 8(d) algorithmic figure is kept beside it (it charges a per-layer dskip read this design never does)
            ach = (traffic / (launch_ms * 1e-3) / 1e9) if traffic else alg
            out["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": ach / HBM_PEAK_GBS, "basis": "measured HBM bytes (PMC)" if traffic else "SURVEY 8(d) algorithmic bytes",
                               "traffic": traffic, "traffic_kernel": traffic_kernel,
                               "traffic_source": traffic_source, "launch_ms": launch_ms, "kernel_launch": kernel_launch,
                               "traffic_frac": (traffic / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                               "survey_8d": {"algorithmic_bytes_per_launch": amount, "achieved": alg, "frac": alg / HBM_PEAK_GBS},
                               "algorithmic_bytes_per_launch": amount,
                               "mfma_frac": layer_flops.get(dom, 0) / (launch_ms * 1e-3) / 1e12 / layer_peak,
                               "mfma_peak_TFLOPs": layer_peak,
                               "note": "frac = achieved / peak with achieved = HBM bytes MEASURED by the PMC passes in profiles/ "
                                       "(FETCH_SIZE x 2 + WRITE_SIZE, per launch) / launch time; survey_8d: SURVEY 8(d)'s "
                                       "algorithmic bytes (1,664 B per sample-layer for the backward, 1,024 B of them the "
                                       "per-layer dskip read that the deferred skip sum never performs) / launch time; "
                                       "mfma_frac: fp32-equivalent flops of the launch / time / mfma_peak_TFLOPs (157.3 fp32 "
                                       "MFMA; 833 = 2.5 PF / 3 with fp16x2 split products)"}
        else:
            ach = amount / (launch_ms * 1e-3) / 1e12
            peak = {"fp32": F32_MFMA_PEAK_TF, "fp16x2": 2500.0 / 3.0}.get(_lib.get_gemm_precision(), BF16X3_PEAK_TF)
            out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": peak,
                               "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic, "traffic_kernel": traffic_kernel,
                               "traffic_source": traffic_source,
                               "launch_ms": launch_ms, "flops_per_launch": amount,
                               "note": "fp32-equivalent flops; peak = dense 16-bit MFMA / terms of the split product"}
        out["mfma_units"] = {k: {"ms": per_step[k], "TFLOPs": units[k][1] / (per_step[k] * 1e-3) / 1e12}
                             for k in units if units[k][0] == "mfma" and k in per_step}
        # ---- AR decode, config 4: 16k samples on one GPU, persistent per-layer state -----------
        if not args.no_decode:
            # decode with the seeded initial weights (not the ones the timed steps just trained), so that the first tokens
            # (all 16,000 since round 5) can be held against the committed oracle trace tests/golden/cfg4_decode_trace.npz
            n = args.decode_samples
            u = np.random.RandomState(7).random_sample(n)
            gold = None
            gf = os.path.join(ROOT, "tests", "golden", "cfg4_decode_trace.npz")
            if os.path.exists(gf):
                gold = np.load(gf)
                ng = int(gold["tokens"].shape[0])
                if n >= ng:
                    u[:ng] = gold["uniforms"]
                else:
                    gold = None
            net.load_state_dict(FasterWaveNet(p, seed=1234).state_dict())
            net.generate(64, u)                                           # warm-up (creates the handle)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            toks = net.generate(n, u)
            torch.cuda.synchronize()
            ddt = time.perf_counter() - t0
            out["ar_generate"] = {"samples_per_s": n / ddt, "seconds": ddt, "samples": n,
                                  "workload": "cfg4: faster_wavenet queue-cached decode, window 4094, 1 GPU",
                                  "token_checksum": int(toks.sum().item())}
            if gold is not None:
                first = toks[:ng].cpu().numpy()
                out["ar_generate"]["golden_tokens_compared"] = ng
                out["ar_generate"]["golden_tokens_match"] = bool(np.array_equal(first, gold["tokens"].astype(np.int32)))
                out["ar_generate"]["golden_checksum"] = [int(first.sum()), int(gold["tokens"].astype(np.int64).sum())]
            # the same decode for 28 independent utterances in ONE launch (wn_decoder_run_batch: nine workgroups each, 252 of
            # the 256 CUs): what one GPU generates when it is not limited to one strict sample-to-sample chain.  Utterance 0
            # uses the golden uniforms again and must reproduce the single-utterance tokens.
            try:
                nb = int(_lib.lib().wn_decoder_batch_max())
                ub = np.random.RandomState(8).random_sample((nb, n))
                ub[0] = u
                net.generate_batch(64, ub[:, :64])                       # warm-up (creates the handles)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                tb = net.generate_batch(n, ub)
                torch.cuda.synchronize()
                bdt = time.perf_counter() - t0
                out["ar_generate_batch"] = {"utterances": nb, "samples_each": n, "seconds": bdt,
                                            "samples_per_s_aggregate": nb * n / bdt, "samples_per_s_per_utterance": n / bdt,
                                            "utterance0_equals_single_run": bool(torch.equal(tb[0], toks)),
                                            "workload": "cfg4 x %d independent utterances in one launch (9 workgroups each)" % nb}
            except Exception as e:                                       # a side record: never the reason a bench run fails
                out["ar_generate_batch"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if not args.no_wide:
            out["wide_channel"] = wide_channel_step(rank, world)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            if "ar_generate" in out:
                out["ar_generate"]["vs_cpu_baseline"] = out["ar_generate"]["samples_per_s"] / \
                    out["cpu_baseline"]["decode_value"]
    if world > 1 or force_dist:
        # what every rank actually ran, so that a scaling record can be trusted at first sight
        forms = [None] * world
        # first_tokens: the head of this rank's first clip -- clip rank * B_PER_GPU of the GLOBAL batch (clip b uses
        # phi_b = 2 pi b / B with the global b), so that a record shows at first sight that the ranks trained on different clips
        dist.all_gather_object(forms, {"rank": rank, "device": torch.cuda.current_device(), "launch": out["launch"],
                                       "ms_per_step_local": local_dt * 1e3, "global_clips": [rank * B_PER_GPU, (rank + 1) * B_PER_GPU],
                                       "first_tokens": x[0, :16].cpu().tolist()})
        out["dist"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks": forms,
                       "collective": "one all-reduce(SUM) of the flat fp32 gradient arena per step (%d floats)"
                                     % net._grad_arena.numel()}
    if rank == 0:
        print(json.dumps(out))
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
