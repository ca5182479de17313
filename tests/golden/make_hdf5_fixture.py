"""Makes tests/golden/chainer_layout_tiny.h5: a checkpoint in the container and layout of the reference's
``serializers.save_hdf5(model_dir + "/wavenet.model", self.chain)`` (wavenet.py:619-625), written by **h5py** itself.

Chainer is not on this image, so the file is not Chainer's own output; what is reproduced is what its HDF5Serializer does
with a Chain of links (documented behaviour, chainer/serializers/hdf5.py of Chainer 2): ``group.require_group(link name)``
per child link, ``create_dataset(param name, data=array, compression=4)`` per parameter (no compression for arrays of one
element).  The arrays are the seeded tiny model's own state (tests/golden/chainer_layout_tiny.npz, written by step 1).

Two steps, because h5py exists only for the image's second interpreter:
    python tests/golden/make_hdf5_fixture.py npz          # this repo's interpreter: the arrays -> .npz
    /opt/conda/bin/python3.9 tests/golden/make_hdf5_fixture.py h5     # h5py 3.3.0 / HDF5 1.10.6: .npz -> .h5
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
NPZ = os.path.join(HERE, "chainer_layout_tiny.npz")
H5 = os.path.join(HERE, "chainer_layout_tiny.h5")
TINY = dict(quantization_steps=16, causal_conv_channels=[4], residual_conv_channels=[4, 4], residual_num_blocks=2,
            softmax_conv_channels=[8, 16])

if sys.argv[1:] == ["npz"]:
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from wavenet_amd import Params, WaveNet
    sd = WaveNet(Params(TINY), seed=5).state_dict()
    rng = np.random.RandomState(11)                 # biases are initialised to zero: give them values worth comparing
    sd = {k: (v if k.endswith("/W") else rng.standard_normal(v.shape).astype(np.float32)) for k, v in sd.items()}
    np.savez(NPZ, **sd)
    print("wrote", NPZ, len(sd), "arrays")
elif sys.argv[1:] == ["h5"]:
    import h5py
    with np.load(NPZ) as z, h5py.File(H5, "w") as f:
        for key in z.files:
            link, name = key.rsplit("/", 1)
            arr = z[key]
            f.require_group("/" + link).create_dataset(name, data=arr, compression=None if arr.size <= 1 else 4)
    print("wrote", H5, "with h5py", h5py.__version__, "HDF5", h5py.version.hdf5_version)
else:
    sys.exit(__doc__)
