"""The reference's two command-line callers of the hot path (train_audio/train.py, train_audio/generate.py) on this engine.

    python -m wavenet_amd.train_audio.train    -w wav -m model --lr 0.001
    python -m wavenet_amd.train_audio.generate -m model -o generated_audio -s 1.0 --fast

Same flags, same ``model/wavenet.json`` hyper-parameter file, same random crops, same checkpoint names.  What differs is
where the work runs: the file's tokens live on the device, a batch is an index gather there, a training step is one HIP
graph replay (wavenet_amd.TrainStepGraph), the loss is summed on the device and read once per file, and ``--fast``
generation is one persistent decoder kernel instead of a Python loop.
"""
