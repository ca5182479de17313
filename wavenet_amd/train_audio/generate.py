"""Generation driver (train_audio/generate.py:9-63): start from ``input_width`` samples of silence (token 127), draw
``int(sampling_rate * seconds) - 1`` samples one at a time from the softmax, write ``<output_dir>/generated.wav``."""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

from .. import data
from . import args as _args
from . import model as _model
from .train import input_width_of


def generate_audio(net, params, sampling_rate=48000, generate_sec=1.0, fast=False, output_dir="generated_audio"):
    Q = params.quantization_steps
    iw = input_width_of(params)
    n = int(sampling_rate * generate_sec) - 1                       # generate.py:24: time_step runs 1 .. n
    silence = 127 if Q > 127 else Q // 2
    start_time = time.time()
    if n <= 0:
        tokens = np.zeros((0,), np.int32)
    elif fast:
        # one uniform per sample, the draw numpy's choice() makes (generate.py:40); the whole loop runs on the device
        u = np.random.random_sample(n)
        tokens = net.generate(n, u, initial_tokens=np.full((iw,), silence, np.int32)).cpu().numpy()
    else:
        buf = np.full((iw,), silence, dtype=np.int32)
        for time_step in range(1, n + 1):
            x = torch.as_tensor(buf[-iw:].reshape(1, -1)).to(net.device)
            with torch.no_grad():
                softmax = net.forward_one_step(x, apply_softmax=True, as_numpy=True)[0, :, 0, -1]
            buf = np.append(buf, np.random.choice(np.arange(Q), p=softmax))
            if time_step % 10 == 0:
                sys.stdout.write("\rgenerating {:.2f} msec / {:.2f} msec".format(
                    time_step * 1000.0 / sampling_rate, generate_sec * 1000.0))
                sys.stdout.flush()
        tokens = buf[iw:]
    print("\ndone in {:.3f} sec".format(time.time() - start_time))
    os.makedirs(output_dir, exist_ok=True)
    filename = "{}/generated.wav".format(output_dir)
    data.save_audio_file(filename, tokens, Q, format="16bit_pcm", sampling_rate=sampling_rate)
    return filename, tokens


def main(argv=None):
    args = _args.parse(argv)
    params, net = _model.build(args)
    np.random.seed(args.seed)
    return generate_audio(net, params, sampling_rate=params.sampling_rate, generate_sec=args.seconds, fast=args.fast,
                          output_dir=args.output_dir)


if __name__ == "__main__":
    main()
