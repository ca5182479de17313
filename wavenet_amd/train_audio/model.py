"""``model/wavenet.json`` -> network (train_audio/model.py:8-58): read the hyper-parameter file if it is there, otherwise
write the reference's starting configuration, build WaveNet or FasterWaveNet, load the checkpoint, move to the device."""
from __future__ import annotations

import json
import os

import torch

from .. import FasterWaveNet, Params, WaveNet


def default_params() -> Params:
    """The configuration model.py:23-43 writes on the first run."""
    p = Params()
    p.quantization_steps = 256
    p.sampling_rate = 8000
    p.causal_conv_no_bias = True
    p.causal_conv_filter_width = 2
    p.causal_conv_channels = [256]
    p.residual_conv_dilation_no_bias = True
    p.residual_conv_projection_no_bias = True
    p.residual_conv_filter_width = 2
    p.residual_conv_channels = [128] * 8
    p.residual_num_blocks = 1
    p.softmax_conv_no_bias = False
    p.softmax_conv_channels = [256, 256]
    p.optimizer = "adam"
    p.momentum = 0.9
    p.weight_decay = 0
    p.gradient_clipping = 1.0
    return p


def load_params(model_dir: str) -> Params:
    os.makedirs(model_dir, exist_ok=True)
    filename = os.path.join(model_dir, "wavenet.json")
    if os.path.isfile(filename):
        print("loading", filename)
        try:
            with open(filename) as f:
                return Params(json.load(f))
        except Exception:
            raise Exception("could not load {}".format(filename))
    params = default_params()
    with open(filename, "w") as f:
        json.dump(params.to_dict(), f, indent=4)
    return params


def build(args):
    """-> (params, wavenet) on ``cuda:<args.gpu_device>``.  There is no CPU mode (``-g -1`` in the reference): the product
    path is the HIP library and fails loudly without a device."""
    params = load_params(args.model_dir)
    net = (FasterWaveNet if args.fast else WaveNet)(params, seed=args.seed)
    params.dump()
    net.load(args.model_dir)
    if args.gpu_device < 0:
        raise Exception("--gpu_device -1 (CPU) is not supported: this engine runs on a HIP device only")
    torch.cuda.set_device(args.gpu_device)
    net.to_gpu(args.gpu_device)
    return params, net
