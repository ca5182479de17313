"""Command-line flags of train_audio/args.py:5-18 (same names and defaults), plus the loop sizes the reference hard-codes
in train.py:112-114 / 121-124 so that a short run does not need an edit."""
from __future__ import annotations

import argparse


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser()
    ap.add_argument("-g", "--gpu_device", type=int, default=0)
    ap.add_argument("-w", "--wav-dir", type=str, default="wav")
    ap.add_argument("-m", "--model-dir", type=str, default="model")
    # generation
    ap.add_argument("-o", "--output_dir", type=str, default="generated_audio")
    ap.add_argument("-s", "--seconds", type=float, default=1.0)
    ap.add_argument("--lr", type=float, default=0.001, help="learning_rate")
    ap.add_argument("--fast", action="store_true", default=False)
    ap.add_argument("--seed", type=int, default=None)
    # the reference's constants (train.py:112-114, 124)
    ap.add_argument("--batch-size", type=int, default=16)
    ap.add_argument("--train-width", type=int, default=500)
    ap.add_argument("--max-epoch", type=int, default=2000)
    ap.add_argument("--repeat", type=int, default=500, help="updates per file per epoch")
    ap.add_argument("--no-graph", action="store_true", default=False, help="launch every step op by op (no HIP graph)")
    return ap


def parse(argv=None):
    return build_parser().parse_args(argv)
