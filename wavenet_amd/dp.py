"""Data-parallel training over the GPUs of one node (new capability; the reference is single-device).

One process per GPU; every rank holds a full weight replica and its own clips.  The loss is a mean
over B*T' rows, so with equal shards the global gradient is the mean of the rank gradients: ONE
all-reduce(SUM) of the flat gradient arena per step (2.46 MB at config 2; RCCL over xGMI through
``torch.distributed`` backend "nccl"), the 1/world factor folded into the optimiser kernel, then the
reference's hooks and Adam run identically on every rank (wavenet.py:477-480, 515-519)."""
from __future__ import annotations

import torch
import torch.distributed as dist


class DataParallel(object):
    def __init__(self, net, group=None, always_reduce: bool = False):
        """``always_reduce``: issue the collective even in a group of one rank (a no-op arithmetically: x -> x).  It is how
        the one GPU of a test box runs the REAL backend -- communicator creation outside graph capture, stream ordering
        between the two replayed graphs and the collective, tear-down -- when no second GPU exists."""
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.net, self.group, self.always_reduce = net, group, always_reduce
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.broadcast_weights()

    def broadcast_weights(self, src: int = 0):
        """Every replica starts from rank ``src``'s weights and optimiser state: the moment arrays, the step counter
        (Adam's bias correction depends on it: a rank that resumed a checkpoint must not run ahead of the others) and the
        host scalars of the rule in use (alpha / lr / momentum, Eve's loss-feedback pair d, f)."""
        opt = self.net.optimizer
        with torch.no_grad():
            dist.broadcast(self.net._arena, src, group=self.group)
            dist.broadcast(opt.m, src, group=self.group)
            dist.broadcast(opt.v, src, group=self.group)
        names = [k for k in ("t", "alpha", "beta1", "beta2", "beta3", "eps", "lr", "hyper", "d", "f")
                 if k in vars(opt)]                        # instance attributes only (Adam's `lr` is a derived property)
        vals = torch.tensor([float(getattr(opt, k)) for k in names], dtype=torch.float64, device=self.net._arena.device)
        dist.broadcast(vals, src, group=self.group)
        for k, v in zip(names, vals.tolist()):
            setattr(opt, k, int(round(v)) if k == "t" else v)
        self.net._weights_changed()

    def mean_loss(self, loss: float) -> float:
        """The global-batch loss (mean over ranks of the equal-shard means): what Eve's feedback term must see on every rank
        (wavenet.py:73-79 feeds it the loss of the whole minibatch)."""
        if self.world == 1:
            return float(loss)
        t = torch.tensor([float(loss)], dtype=torch.float64, device=self.net._arena.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return float(t.item()) / self.world

    def all_reduce_grads(self, flat_grad: torch.Tensor) -> float:
        """Sum the flat gradient buffer over ranks in place; returns the multiplier (1/world) that
        the optimiser kernel applies."""
        if self.world > 1 or self.always_reduce:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
        return 1.0 / self.world

    def shard(self, global_batch: int):
        """Clip indices [lo, hi) of this rank for a global batch (equal shards)."""
        if global_batch % self.world:
            raise ValueError("global batch %d is not divisible by world size %d" % (global_batch, self.world))
        per = global_batch // self.world
        return self.rank * per, (self.rank + 1) * per
