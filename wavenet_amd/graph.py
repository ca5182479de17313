"""A whole training step as a replayed HIP graph (new capability; the reference launches op by op).

One step of train_audio/train.py:60-78 -- cleargrads, forward (causal -> residual stack -> softmax head), softmax
cross-entropy, backward, the optimiser hooks and Adam -- is ~250 kernel launches of 5-70 us each.  Launched one by
one from Python the GPU idles ~8 % of the step between them; captured once (``torch.cuda.graph`` = hipStreamBeginCapture
on the stream our C-ABI calls are issued on) and replayed, the step is one graph launch.

What changes from step to step lives in device memory, never in kernel arguments:
  * the batch: copied into the graph's static input buffers;
  * Adam's bias-corrected step size alpha_t: a device scalar written before each replay (``wn_adam_step_dev``).
With data parallelism the gradient all-reduce stays OUTSIDE the graphs (forward+backward graph -> RCCL all-reduce ->
optimiser graph), so nothing here depends on capturing a collective.
"""
from __future__ import annotations

import torch

from . import _lib


def default_loss(net, x, tgt, window_only: bool = False):
    """train_audio/train.py:60-75: loss over the last ``tgt.shape[1]`` columns of the window.  ``window_only`` also skips
    the columns that window cannot see (WaveNet.forward_residual_block); same loss, same gradients."""
    c = net.forward_causal_block(x)
    _, s = net.forward_residual_block(c, t_off=x.shape[1] - tgt.shape[1], window_only=window_only)
    # forward_softmax_block(apply_softmax=False) + cross_entropy, the last head convolution and the loss in one launch where covered
    return net.head_cross_entropy(s, tgt)


class TrainStepGraph(object):
    """``g = TrainStepGraph(net, x, tgt); loss = g.step(x, tgt)`` -- same result as
    ``net.backprop(default_loss(net, x, tgt))`` for batches of the captured shape."""

    def __init__(self, net, x, tgt, loss_fn=default_loss, warmup: int = 2, keep_graph: bool = False):
        """``keep_graph``: keep the captured hipGraph_t next to the executable graph so that :meth:`node_counts` can walk it
        (measurement aid: bench.py counts the kernel nodes of the step it times)."""
        if not (net.gpu_enabled and x.is_cuda and tgt.is_cuda):
            raise _lib.WaveNetHipError("TrainStepGraph needs the network and the batch on a HIP device")
        self.net, self.loss_fn = net, loss_fn
        self.x = x.clone()
        self.tgt = tgt.clone()
        opt = net.optimizer
        self._lr = torch.zeros((1,), device=x.device, dtype=torch.float32)
        self._one = None
        dp = net._dp_group is not None
        self._gmult = 1.0 / net._dp_group.world if dp else 1.0
        # warm-up on the capture stream (per-stream scratch, function attributes, allocator pools), then put the
        # training state back: the warm-up steps are not training steps
        keep = (net._arena.clone(), opt.m.clone(), opt.v.clone(), opt.t)
        self._stream = torch.cuda.Stream(device=x.device)
        self._keep_graph = bool(keep_graph)
        self._g1 = torch.cuda.CUDAGraph(keep_graph=True) if keep_graph else torch.cuda.CUDAGraph()
        self._g2 = None
        try:
            self._stream.wait_stream(torch.cuda.current_stream())
            # step plan (fp32 storage): the first warm-up step RECORDS the step's weight-only preparation work, every later
            # step -- the captured one included -- starts with wn_plan_prepare (two launches: all weight images, range words,
            # dataflow words, cleargrads) and its entry points launch no preparation of their own
            self._use_plan = bool(getattr(net, "use_step_plan", False)) and net.storage != "bf16"
            with torch.cuda.stream(self._stream):
                if self._use_plan:
                    self._plan = net.plan_begin()          # this object's own plan: the graph keeps pointers into its memory
                    self._planned = False
                    self._fwd_bwd()
                    self._opt()
                    net.plan_finish()
                    self._planned = True
                for _ in range(max(1, warmup)):
                    self._fwd_bwd()
                    self._opt()
            torch.cuda.current_stream().wait_stream(self._stream)
            torch.cuda.synchronize()
            with torch.no_grad():
                net._arena.copy_(keep[0]); opt.m.copy_(keep[1]); opt.v.copy_(keep[2])
            # whatever images the warm-up derived from the weights (the bf16 operand images of storage='bf16') are stale
            # now, and must be rebuilt INSIDE the capture: a replay has no host code that could repack them
            net._weights_changed()
            # thread_local: API calls of other threads (an RCCL watchdog, a data loader) must not invalidate the capture
            with torch.cuda.graph(self._g1, stream=self._stream, capture_error_mode="thread_local"):
                self.loss = self._fwd_bwd()
                if not dp:
                    self._opt()
            if dp:
                self._g2 = torch.cuda.CUDAGraph(keep_graph=True) if keep_graph else torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._g2, stream=self._stream, pool=self._g1.pool(),
                                      capture_error_mode="thread_local"):
                    self._opt()
            self._snap = self._hyper()
        finally:
            # whatever happened above (a failed capture included), the warm-up and capture steps were not training steps:
            # weights, moments and the optimiser clock go back to what the caller handed in
            torch.cuda.synchronize()
            with torch.no_grad():
                net._arena.copy_(keep[0]); opt.m.copy_(keep[1]); opt.v.copy_(keep[2])
            opt.t = keep[3]
            net._weights_changed()
            # outside this object's graphs nobody runs wn_plan_prepare: eager calls must not take the plan's (stale) images
            if getattr(self, "_use_plan", False):
                net.plan_off()

    def _hyper(self):
        """Everything a captured kernel node took BY VALUE: replaying after one of these changed would silently train with
        the old value (only the batch and the learning rate travel through device memory)."""
        opt, p = self.net.optimizer, self.net.params
        return tuple(getattr(opt, k, None) for k in ("beta1", "beta2", "beta3", "eps", "hyper")) + \
            (p.gradient_clipping, p.weight_decay, self.net.gemm_precision or _lib.get_gemm_precision())

    def _fwd_bwd(self):
        if getattr(self, "_use_plan", False) and self._planned:
            self.net.plan_prepare(zero_grads=True)
        else:
            self.net.zero_grads()
        self.net._unit_upstream = True                 # loss.backward(self._one): d loss = 1, the scale launch is skipped
        try:
            return self._fwd_bwd_body()
        finally:
            self.net._unit_upstream = False

    def _fwd_bwd_body(self):
        loss = self.loss_fn(self.net, self.x, self.tgt)
        # the upstream gradient of the loss is a tensor made ONCE (in the warm-up pass, outside the capture): `loss.backward()`
        # would fill a fresh one in every replay -- a kernel at the launch floor (4.6 us) for one float
        if self._one is None or self._one.shape != loss.shape:
            self._one = torch.ones_like(loss)
        loss.backward(self._one)
        return loss.detach()

    def _opt(self):
        self.net.optimizer.update(self._gmult, lr_dev=self._lr)
        self.net._weights_changed()

    def step(self, x=None, tgt=None):
        """One training step on (x, tgt) (default: the batch already in the static buffers).  Returns the loss
        (a device scalar that the next step overwrites)."""
        net, opt = self.net, self.net.optimizer
        if self._hyper() != self._snap:
            raise _lib.WaveNetHipError(
                "momentum / eps / gradient_clipping / weight_decay / GEMM precision changed after the step was captured "
                "(%r -> %r): capture a new TrainStepGraph" % (self._snap, self._hyper()))
        if x is not None:
            self.x.copy_(x, non_blocking=True)
        if tgt is not None:
            self.tgt.copy_(tgt, non_blocking=True)
        opt.t += 1                                   # update() is not called on replay: keep Adam's clock here
        self._lr.fill_(opt.lr)
        self._g1.replay()
        if self._g2 is not None:
            net._dp_group.all_reduce_grads(net._grad_arena)
            self._g2.replay()
        net._weights_changed()
        return self.loss

    def node_counts(self):
        """{"kernel": n, "memcpy": n, "memset": n, "other": n, "total": n} over the captured graph(s) -- what ONE replayed step
        launches (hipGraphGetNodes / hipGraphNodeGetType on the hipGraph_t torch kept: needs ``keep_graph=True``)."""
        import ctypes as C
        if not self._keep_graph:
            raise _lib.WaveNetHipError("node_counts() needs TrainStepGraph(..., keep_graph=True)")
        hip = C.CDLL("libamdhip64.so")
        out = {"kernel": 0, "memcpy": 0, "memset": 0, "other": 0, "total": 0}
        for g in (self._g1, self._g2):
            if g is None:
                continue
            graph = C.c_void_p(g.raw_cuda_graph())
            n = C.c_size_t(0)
            if hip.hipGraphGetNodes(graph, None, C.byref(n)) != 0:
                raise _lib.WaveNetHipError("hipGraphGetNodes failed")
            nodes = (C.c_void_p * max(1, n.value))()
            if hip.hipGraphGetNodes(graph, nodes, C.byref(n)) != 0:
                raise _lib.WaveNetHipError("hipGraphGetNodes failed")
            for i in range(n.value):
                t = C.c_int(-1)
                hip.hipGraphNodeGetType(C.c_void_p(nodes[i]), C.byref(t))
                # hipGraphNodeType: 0 kernel, 1 memcpy, 2 memset, 3 host, 4 graph, 5 empty, ...
                out[{0: "kernel", 1: "memcpy", 2: "memset"}.get(t.value, "other")] += 1
                out["total"] += 1
        return out
