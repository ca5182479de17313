"""Training driver (train_audio/train.py:24-135): for every epoch, for every .wav file, ``repeat`` updates on random crops
of ``input_width + train_width`` samples with next-sample targets; checkpoint after every file and every epoch."""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

from .. import TrainStepGraph, data
from ..graph import default_loss
from . import args as _args
from . import model as _model


HEALTH_EVERY = 100      # updates between two reads of WaveNet.last_update_applied() (a host synchronisation each)


def input_width_of(params) -> int:
    """train.py:36-44: receptive field of the residual stack plus one column per causal layer."""
    per_block = params.residual_conv_filter_width ** len(params.residual_conv_channels)
    return (per_block - 1) * params.residual_num_blocks + 1 + len(params.causal_conv_channels)


class _Crops(object):
    """train.py:14-22 with the signal resident on the device: the start offsets are drawn on the host from numpy's
    global generator (the draw the reference makes, so ``--seed`` selects the same crops), the gather runs on the GPU."""

    def __init__(self, signal: np.ndarray, input_width: int, target_width: int, device):
        self.n = int(signal.size)
        self.iw, self.tw = input_width, target_width
        if self.n - target_width - input_width - 1 <= 0:
            raise Exception("signal too short for input_width + train_width")
        self.signal = torch.as_tensor(signal.astype(np.int32)).to(device)
        self.col = torch.arange(input_width + target_width + 1, device=device)

    def draw(self, batch_size: int):
        starts = np.random.randint(0, self.n - self.tw - self.iw - 1, size=batch_size)
        idx = torch.as_tensor(starts).to(self.col.device)[:, None] + self.col[None, :]
        win = self.signal[idx]                                     # (B, iw + tw + 1)
        return win[:, :self.iw + self.tw].contiguous(), win[:, self.iw + 1:].contiguous()


def train_audio(net, params, path_to_file, batch_size=16, train_width=16, repeat=1000, use_graph=True, state=None):
    """One file: returns the summed loss of its ``repeat`` updates (train.py:24-90)."""
    signals, _ = data.load_audio_file(path_to_file, quantization_steps=params.quantization_steps)
    iw = input_width_of(params)
    silence = 127 if params.quantization_steps > 127 else params.quantization_steps // 2
    signals = np.concatenate([np.full((iw,), silence, dtype=np.int32), signals.astype(np.int32)])   # train.py:53
    crops = _Crops(signals, iw, train_width, net.device)
    sum_loss = torch.zeros((), device=net.device, dtype=torch.float64)
    graph = None
    skipped = 0
    name = os.path.basename(path_to_file)
    for batch_index in range(repeat):
        x, tgt = crops.draw(batch_size)
        if use_graph and str(params.optimizer).lower() != "eve":     # Eve needs the loss on the host every update
            key = (batch_size, iw + train_width)
            graph = None if state is None else state.get(key)
            if graph is None:
                graph = TrainStepGraph(net, x, tgt)
                if state is not None:
                    state[key] = graph
            loss = graph.step(x, tgt)
        else:
            loss = default_loss(net, x, tgt)
            net.backprop(loss)
            loss = loss.detach()
        sum_loss += loss                                            # on the device: no host sync per update
        # health check (one word read back, every HEALTH_EVERY updates and after the last one): a step whose gradient norm was
        # not finite is SKIPPED on the device (wn_adam_step, ABI 4) -- say so instead of training on in silence.  The guard
        # lives in the clipping hook: with gradient_clipping <= 0 there is no norm and nothing is ever skipped.
        if (batch_index % HEALTH_EVERY == HEALTH_EVERY - 1 or batch_index == repeat - 1) and not net.last_update_applied():
            skipped += 1
            sys.stdout.write("\n\twarning: update {} of {} was skipped on the device (gradient norm not finite; loss {}); "
                             "{} such checks failed so far\n".format(batch_index, name, float(loss), skipped))
            if skipped >= 5:
                raise Exception("the gradient norm was not finite at {} consecutive health checks: the run is doing no work "
                                "(lower --lr, or set exec_flags |= WN_EXEC_NO_MULTI_LAYER_BWD if another process shares the "
                                "GPU)".format(skipped))
        elif batch_index % HEALTH_EVERY == HEALTH_EVERY - 1:
            skipped = 0
        if batch_index % 10 == 0:
            sys.stdout.write("\r\t{} - {} width; {}/{}".format(name, signals.size, batch_index, repeat))
            sys.stdout.flush()
    return float(sum_loss.item())


def main(argv=None):
    args = _args.parse(argv)
    params, net = _model.build(args)
    np.random.seed(args.seed)
    net.update_laerning_rate(args.lr)
    files = sorted(fn for fn in os.listdir(args.wav_dir) if fn.endswith(".wav"))
    for fn in files:
        print("loading", fn)
    iw = input_width_of(params)
    rf = iw - len(params.causal_conv_channels)
    print("receptive field width:", int(rf * 1000.0 / params.sampling_rate), "[millisecond]")
    print("receptive field width:", rf, "[step]")
    print("files: {} batch_size: {} train_width: {}".format(len(files), args.batch_size, args.train_width))
    if not files:
        raise Exception("no .wav file in {}".format(args.wav_dir))
    start_time = time.time()
    graphs = {}
    average_loss = None
    for epoch in range(1, args.max_epoch):                          # train.py:118: epochs 1 .. max_epoch - 1
        average_loss = 0.0
        for fn in files:
            average_loss += train_audio(net, params, os.path.join(args.wav_dir, fn), batch_size=args.batch_size,
                                        train_width=args.train_width, repeat=args.repeat,
                                        use_graph=not args.no_graph, state=graphs)
            net.save(args.model_dir)
        average_loss /= len(files)
        sys.stdout.write("\033[2K\repoch: {} - {:.4e} loss - {} min\n".format(
            epoch, average_loss, int((time.time() - start_time) / 60)))
        sys.stdout.flush()
        net.save(args.model_dir)
    return average_loss


if __name__ == "__main__":
    main()
