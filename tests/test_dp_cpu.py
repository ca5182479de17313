"""CPU, world_size 2 over gloo: the data-parallel plumbing (shards, one flat all-reduce, 1/world factor,
weight broadcast).  The gradients fed in are the oracle's, so the identity "mean of the shard
gradients == gradient of the global batch"  is checked end to end without a GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import wavenet_ref as R
from wavenet_amd import Params, WaveNet
from wavenet_amd.dp import DataParallel

TINY = dict(quantization_steps=16, causal_conv_channels=[8], residual_conv_channels=[8, 8, 8],
            residual_num_blocks=2, softmax_conv_channels=[12, 16])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = R.make_params(**TINY)
        w = R.init_weights(p, 5)
        net = WaveNet(Params(p), seed=100 + rank)            # different initial weights per rank ...
        if rank == 0:
            net.load_state_dict(w)
        dp = net.enable_data_parallel()                      # ... until rank 0's are broadcast
        for k, v in net.state_dict().items():
            np.testing.assert_array_equal(v, w[k])
        assert dp.world == world and dp.shard(8) == (4 * rank, 4 * rank + 4)
        with pytest.raises(ValueError):
            dp.shard(7)
        # global batch of 4 clips, 2 per rank; equal shards, loss = mean over rows
        iw = R.input_width(p)
        rs = np.random.RandomState(3)
        idx = rs.randint(0, 16, (4, iw + 20)).astype(np.int32)
        tgt = rs.randint(0, 16, (4, 20)).astype(np.int32)
        lo, hi = dp.shard(4)
        _, _, g_local = R.train_step_grads(p, w, idx[lo:hi], tgt[lo:hi], dtype=torch.float64)
        _, _, g_full = R.train_step_grads(p, w, idx, tgt, dtype=torch.float64)
        for ln, kind, off, n, shape in net._spans:
            net._grad_arena[off:off + n] = torch.from_numpy(g_local["%s/%s" % (ln.name, kind)].reshape(-1).astype(np.float32))
        mult = dp.all_reduce_grads(net._grad_arena)          # ONE collective on the flat buffer
        assert mult == 1.0 / world
        for ln, kind, off, n, shape in net._spans:
            got = net._grad_arena[off:off + n].numpy() * mult
            want = g_full["%s/%s" % (ln.name, kind)].reshape(-1)
            np.testing.assert_allclose(got, want, atol=1e-6)
        open(os.path.join(tmp, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_data_parallel_two_ranks_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")
