"""Command-line flags of the reference's train_audio scripts (train_audio/args.py:5-18: same spellings and defaults),
plus the loop sizes train.py:112-114 / 124 hard-codes, so that a short run does not need an edit."""
from __future__ import annotations

import argparse

# (flags, type, default, help); a bool default makes a store_true switch
_REFERENCE_FLAGS = (
    (("-g", "--gpu_device"), int, 0, "HIP device index"),
    (("-w", "--wav-dir"), str, "wav", "directory of .wav files to train on"),
    (("-m", "--model-dir"), str, "model", "wavenet.json + checkpoints"),
    (("-o", "--output_dir"), str, "generated_audio", "where generate writes generated.wav"),
    (("-s", "--seconds"), float, 1.0, "length of the generated audio"),
    (("--lr",), float, 0.001, "learning_rate"),
    (("--fast",), None, False, "FasterWaveNet: queue-cached generation"),
    (("--seed",), int, None, "numpy seed (crops, sampling)"),
)
_LOOP_FLAGS = (
    (("--batch-size",), int, 16, "train.py:112"),
    (("--train-width",), int, 500, "train.py:113"),
    (("--max-epoch",), int, 2000, "train.py:114 (epochs run 1 .. max_epoch - 1)"),
    (("--repeat",), int, 500, "updates per file per epoch (train.py:124)"),
    (("--no-graph",), None, False, "launch every update op by op instead of replaying a HIP graph"),
)


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(description=__doc__)
    for flags, typ, default, text in _REFERENCE_FLAGS + _LOOP_FLAGS:
        if typ is None:
            ap.add_argument(*flags, action="store_true", default=default, help=text)
        else:
            ap.add_argument(*flags, type=typ, default=default, help=text)
    return ap


def parse(argv=None):
    return build_parser().parse_args(argv)
