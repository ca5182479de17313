"""config 3 stand-in on one GPU: two ranks share cuda:0 over gloo and train data-parallel with the two-graph step
(forward/backward graph -> all-reduce of the flat gradient arena -> optimiser graph, wavenet_amd/graph.py); the weights
must follow the single-process step on the concatenated batch (mean of shard gradients == gradient of
the global batch, then identical hooks + Adam on every rank; train_audio/train.py:58-80 is the step)."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

OVER = dict(quantization_steps=256, causal_conv_channels=[32], residual_conv_channels=[32] * 5,
            residual_num_blocks=2, softmax_conv_channels=[256, 256])
CFG2 = dict(quantization_steps=256, causal_conv_channels=[32], residual_conv_channels=[32] * 10,
            residual_num_blocks=4, softmax_conv_channels=[256, 256])          # BASELINE configs[2]'s stack: 614,656 floats
STEPS = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batches(iw, B_PER, EXTRA):
    rs = np.random.RandomState(3)
    T = iw + EXTRA
    return [(rs.randint(0, 256, (2 * B_PER, T)).astype(np.int32), rs.randint(0, 256, (2 * B_PER, EXTRA)).astype(np.int32))
            for _ in range(STEPS)]


def _worker(rank, world, port, tmp, use_graph, over, B_PER, EXTRA):
    import torch.distributed as dist
    from oracle import wavenet_ref as R
    from wavenet_amd import Params, TrainStepGraph, WaveNet
    from wavenet_amd.graph import default_loss
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = R.make_params(**over)
        w = R.init_weights(p, 11)
        net = WaveNet(Params(p), seed=100 + rank)
        if rank == 0:
            net.load_state_dict(w)
        net.to_gpu()
        net.update_laerning_rate(0.01)
        net.optimizer.eps = 1e-3                 # see test_train_step_graph_replay_equals_eager_steps
        net.optimizer.t = 5 if rank == 0 else 0  # a resumed rank 0: the optimiser clock must travel with the weights
        dp = net.enable_data_parallel()
        assert net.optimizer.t == 5
        iw = net.input_width
        lo, hi = dp.shard(2 * B_PER)
        batches = _batches(iw, B_PER, EXTRA)
        dev = lambda a: torch.as_tensor(a).cuda()
        if use_graph:
            g = TrainStepGraph(net, dev(batches[0][0][lo:hi]), dev(batches[0][1][lo:hi]))
            assert g._g2 is not None and net.optimizer.t == 5
        for x, t in batches:
            if use_graph:
                g.step(dev(x[lo:hi]), dev(t[lo:hi]))
            else:
                net.backprop(default_loss(net, dev(x[lo:hi]), dev(t[lo:hi])))
        torch.cuda.synchronize()
        assert net.optimizer.t == 5 + STEPS
        got = net._arena.detach().cpu().numpy()
        # every rank holds the same weights, bit for bit (same reduced gradient, same kernels)
        mine = torch.from_numpy(got.copy())
        other = mine.clone()
        dist.broadcast(other, 0)
        assert torch.equal(mine, other)
        if rank == 0:
            ref = WaveNet(Params(p), seed=0)
            ref.load_state_dict(w)
            ref.to_gpu()
            ref.update_laerning_rate(0.01)
            ref.optimizer.eps = 1e-3
            ref.optimizer.t = 5
            w_init = ref._arena.detach().cpu().numpy().copy()
            for x, t in batches:
                ref.backprop(default_loss(ref, dev(x), dev(t)))
            torch.cuda.synchronize()
            want = ref._arena.detach().cpu().numpy()
            assert np.abs(want - w_init).max() > 1e-3                 # the weights did move
            np.testing.assert_allclose(got, want, atol=2e-5)
        open(os.path.join(tmp, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("use_graph,over,B_PER,EXTRA", [(True, OVER, 2, 150), (False, OVER, 2, 150), (True, CFG2, 1, 100)])
def test_two_ranks_on_one_gpu_follow_the_global_batch_step(tmp_path, use_graph, over, B_PER, EXTRA):
    """The third case is BASELINE configs[2]'s own 4 x 10 stack (the 614,656-float arena all-reduced between the two
    graphs), one clip per rank."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), use_graph, over, B_PER, EXTRA), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")
