"""CPU: host logic and the C-ABI surface (no compute calls: there is no GPU here)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch

import wavenet_amd
from wavenet_amd import _lib, data
from wavenet_amd.wavenet import Params, WaveNet, zero_prefix
from oracle import data_ref as D
from oracle import wavenet_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "wavenet_hip.h")).read()
    declared = set(re.findall(r"\b(wn(?:16)?_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "libwavenet_hip.so does not export %s" % name
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    # ... and NOTHING else: the library is built with -fvisibility=hidden + an export map, so its dynamic symbol table is the
    # header's list (no mangled wn:: internals, no compiler markers) -- what a third party binds is what the header says
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if ln.strip()}
    assert exported == declared, sorted(exported ^ declared)
    assert _lib.lib().wn_abi_version() == 5 == int(re.search(r"#define WN_ABI_VERSION (\d+)", hdr).group(1))
    assert not hasattr(lib, "wn_set_gemm_precision")          # ABI v2: no process-wide arithmetic mode
    # ABI v3: no switch read from the process environment inside the library (they are WnExec.flags / fields now)
    csrc = os.path.join(ROOT, "wavenet_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".hpp")):
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f
    # the binding's WnExec / WnDecoderDesc have the header's fields, in order
    def fields(name):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        out = []
        for decl in body.split(";"):
            decl = decl.strip()
            if decl:
                for part in decl.split(","):
                    out.append(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", part)[-1])
        return out
    assert fields("WnExec") == [f[0] for f in _lib.WnExec._fields_]
    assert fields("WnDecoderDesc") == [f[0] for f in _lib.WnDecoderDesc._fields_]
    assert fields("WnStackDesc") == [f[0] for f in _lib.WnStackDesc._fields_]
    # scalar results that carry their own scratch: the binding's sizes are the header's
    assert int(re.search(r"#define WN_SQNORM_WORDS (\d+)", hdr).group(1)) == _lib.SQNORM_WORDS
    assert int(re.search(r"#define WN_XENT_LOSS_WORDS (\d+)", hdr).group(1)) == _lib.XENT_LOSS_WORDS


def test_argument_errors_do_not_need_a_gpu():
    lib = _lib.lib()
    rc = lib.wn_layer_fwd(None, None, None, None, None, None, None, None, None, None, None, 1, 1, 1, 1, 1, 1, 0, None, None)
    assert rc == -1 and b"NULL" in lib.wn_last_error()
    with pytest.raises(_lib.WaveNetHipError):
        _lib.check(rc, "wn_layer_fwd")
    assert lib.wn_layer_fast_path(32, 32, 2) == 1 and lib.wn_layer_fast_path(16, 16, 2) == 0


def test_params_defaults_and_check():
    p = Params()
    assert p.quantization_steps == 256 and p.residual_conv_channels == [32] * 9 and p.optimizer == "adam"
    p.check()
    p2 = Params({"quantization_steps": 16, "softmax_conv_channels": [8, 16], "bogus": 1})
    assert not hasattr(p2, "bogus")
    p2.check()
    p2.nonsense = 3
    with pytest.raises(Exception, match="invalid parameter"):
        p2.check()
    p3 = Params({"quantization_steps": 10})
    with pytest.raises(Exception, match="quantization_steps"):
        p3.check()
    assert set(Params().to_dict()) == set(R.DEFAULTS) | {"optimizer", "weight_decay", "momentum", "gradient_clipping"}


@pytest.mark.parametrize("T,d,fw", [(16384, 512, 2), (16000, 512, 2), (8000, 8, 2), (7, 8, 2), (4094, 512, 2),
                                    (100, 9, 3), (26, 27, 3), (10, 4, 4), (5, 1, 2)])
def test_zero_prefix_matches_oracle(T, d, fw):
    assert zero_prefix(T, d, fw) == R.conv_pad_and_prefix(T, d, fw)[1]


def test_model_layout_matches_reference_names_and_shapes():
    p = Params()
    p.causal_conv_channels = [32]
    p.residual_conv_channels = [32] * 10
    p.residual_num_blocks = 4
    p.softmax_conv_channels = [256, 256]
    net = WaveNet(p, seed=0)
    sd = net.state_dict()
    specs = R.weight_specs(R.make_params(causal_conv_channels=[32], residual_conv_channels=[32] * 10,
                                         residual_num_blocks=4, softmax_conv_channels=[256, 256]))
    want = {}
    for name, ws, bs in specs:
        want[name + "/W"] = ws
        if bs is not None:
            want[name + "/b"] = bs
    assert {k: v.shape for k, v in sd.items()} == want
    assert net.num_parameters == 614656
    assert net.receptive_field == 4093 and net.input_width == 4094
    assert len(net.residual_blocks) == 4 and len(net.residual_blocks[0]) == 10
    assert net.residual_blocks[1][3].wf.dilation == 8 and net.residual_blocks[1][3].wf.W.shape == (32, 32, 2, 1)
    assert net.residual_blocks[1][0].wf.W.shape == (32, 32, 1, 2)
    # round trip through the reference-keyed state dict
    w = R.init_weights(R.make_params(causal_conv_channels=[32], residual_conv_channels=[32] * 10,
                                     residual_num_blocks=4, softmax_conv_channels=[256, 256]))
    net.load_state_dict(w)
    for k, v in net.state_dict().items():
        np.testing.assert_array_equal(v, w[k])
    # grads are views of one flat buffer
    assert net.residual_blocks[0][0].wf.W.grad.data_ptr() >= net._grad_arena.data_ptr()


def test_forward_without_gpu_raises_instead_of_falling_back():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    net = WaveNet(Params({"causal_conv_channels": [8]}), seed=0)
    with pytest.raises(wavenet_amd.WaveNetHipError):
        net.forward_causal_block(np.zeros((1, 4), np.int32))
    with pytest.raises(wavenet_amd.WaveNetHipError):
        net.to_gpu()
    with pytest.raises(Exception, match="cut cannot be less than one"):
        net.slice_1d(torch.zeros(1, 1, 1, 4), 0)


def test_no_product_import_of_the_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "wavenet_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src, "%s mentions the oracle" % f


def test_data_formats_match_oracle():
    idx = np.random.RandomState(0).randint(0, 256, (3, 40)).astype(np.int32)
    np.testing.assert_array_equal(data.onehot_pixel_image(idx), D.onehot_pixel_image(idx))
    s = np.random.RandomState(1).uniform(-1, 1, 1000)
    np.testing.assert_array_equal(data.mulaw_encode(s), D.mulaw_quantize(s))
    v = np.arange(-32768, 32768).astype(np.int16)
    np.testing.assert_array_equal(data.mulaw_encode_pcm16(v), D.mulaw_quantize_pcm16(v.astype(np.int64)))
    sig = np.arange(500, dtype=np.int32)
    x, t = data.create_batch(sig, 4, 17, 20, rng=np.random.RandomState(3))
    starts = np.random.RandomState(3).randint(0, 500 - 20 - 17 - 1, size=4)
    xo, to = D.create_batch(sig, starts, 17, 20)
    np.testing.assert_array_equal(x, xo)
    np.testing.assert_array_equal(t, to)


def test_eve_loss_feedback_scalars_follow_the_reference_recurrence():
    """EveState keeps ONE (d, f) pair on the host; the reference keeps one float32 pair per parameter, all updated with the
    same loss at the same t (wavenet.py:27-44).  Same numbers, step for step, including the clamps in both directions."""
    from oracle import wavenet_ref as R
    from wavenet_amd import Params, WaveNet
    from wavenet_amd.wavenet import EveState
    p = Params(dict(quantization_steps=16, causal_conv_channels=[4], residual_conv_channels=[4, 4], residual_num_blocks=1,
                    softmax_conv_channels=[8, 16], optimizer="eve"))
    net = WaveNet(p)
    assert isinstance(net.optimizer, EveState)
    ref = R.EveRef()
    prm = {"w": np.zeros(3, np.float32)}
    losses = [2.5, 2.4, 2.45, 30.0, 0.01, 0.0101, 5.0, 5.0, 4.0, 1e-9]
    for L in losses:
        net.optimizer.t += 1
        net.optimizer._update_d_and_f(L)
        ref.update(prm, {"w": np.zeros(3, np.float32)}, L)
        assert net.optimizer.t == ref.t
        assert net.optimizer.f == float(ref.states["w"]["f"][0])
        assert net.optimizer.d == float(ref.states["w"]["d"][0])
    with pytest.raises(RuntimeError):
        net.optimizer.update(1.0)                                   # Eve.update requires the loss (wavenet.py:75-76)
    with pytest.raises(Exception):                                  # get_optimizer's final `raise Exception()` (wavenet.py:97)
        WaveNet(Params(dict(quantization_steps=16, causal_conv_channels=[4], residual_conv_channels=[4], residual_num_blocks=1,
                            softmax_conv_channels=[8, 16], optimizer="no-such-rule")))


_TINY = dict(quantization_steps=16, causal_conv_channels=[4], residual_conv_channels=[4, 4], residual_num_blocks=2,
             softmax_conv_channels=[8, 16])
_H5 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chainer_layout_tiny.h5")
_H5_NPZ = _H5[:-3] + ".npz"


def _need_hdf5():
    from wavenet_amd import hdf5_io
    if not hdf5_io.available():
        pytest.skip("no HDF5 C library on this machine")
    return hdf5_io


def test_hdf5_checkpoint_written_by_h5py_is_imported():
    """wavenet.py:627-639 reads `wavenet.model`, a Chainer HDF5 file.  tests/golden/chainer_layout_tiny.h5 was written by
    h5py itself in the layout Chainer's HDF5Serializer produces (groups per link, gzip-4 datasets W / b;
    tests/golden/make_hdf5_fixture.py): load_hdf5 must put exactly those arrays into the model."""
    _need_hdf5()
    from wavenet_amd import Params, WaveNet
    net = WaveNet(Params(_TINY), seed=99)                             # other weights than the file's
    with np.load(_H5_NPZ) as z:
        want = {k: z[k] for k in z.files}
    assert any(not np.array_equal(net.state_dict()[k], want[k]) for k in want)
    net.load_hdf5(_H5)
    got = net.state_dict()
    assert sorted(got) == sorted(want)
    for k in want:
        assert got[k].dtype == np.float32 and np.array_equal(got[k], want[k]), k
    assert "causal_0/W" in got and "residual_0_block_0_wf/W" in got and "softmax_0/W" in got      # Chainer's dataset paths


def test_hdf5_checkpoint_round_trip_and_load_prefers_the_reference_file(tmp_path):
    """save_hdf5 writes the container the reference's own load() reads; load() picks a bare `wavenet.model` up."""
    hdf5_io = _need_hdf5()
    from wavenet_amd import Params, WaveNet
    a = WaveNet(Params(_TINY), seed=3)
    a.save_hdf5(str(tmp_path / "wavenet.model"))
    raw = hdf5_io.read_datasets(str(tmp_path / "wavenet.model"))
    assert sorted(raw) == sorted(a.state_dict())
    b = WaveNet(Params(_TINY), seed=4)
    b.load(str(tmp_path))                                             # no .npz there: the HDF5 file is the checkpoint
    for k, v in a.state_dict().items():
        assert np.array_equal(b.state_dict()[k], v), k
    # save() leaves the reference's file next to the .npz pair
    a.save(str(tmp_path / "both"))
    assert sorted(os.listdir(str(tmp_path / "both"))) == ["wavenet.model", "wavenet.model.npz", "wavenet.opt.npz"]
    raw2 = hdf5_io.read_datasets(str(tmp_path / "both" / "wavenet.model"))
    assert all(np.array_equal(raw2[k], v) for k, v in a.state_dict().items())
    # a file of another model is refused, not half-loaded
    c = WaveNet(Params(dict(_TINY, residual_conv_channels=[4, 4, 4])), seed=4)
    with pytest.raises(KeyError):
        c.load_hdf5(str(tmp_path / "wavenet.model"))
    d = WaveNet(Params(dict(_TINY, causal_conv_channels=[6])), seed=4)
    with pytest.raises(Exception, match="shape of"):
        d.load_hdf5(str(tmp_path / "wavenet.model"))
    with pytest.raises(OSError):
        b.load_hdf5(str(tmp_path / "missing.model"))


def test_load_takes_the_newer_of_the_two_weight_files(tmp_path):
    """a reference-written `wavenet.model` dropped next to an OLDER `wavenet.model.npz` must win (load() used
    to prefer the .npz whenever it existed); the other way round the .npz wins."""
    _need_hdf5()
    from wavenet_amd import Params, WaveNet
    old, new = WaveNet(Params(_TINY), seed=5), WaveNet(Params(_TINY), seed=6)
    for newer in ("h5", "npz"):
        d = tmp_path / newer
        d.mkdir()
        (old if newer == "h5" else new).save_hdf5(str(d / "wavenet.model"))            # placeholder, overwritten below
        np.savez(str(d / "wavenet.model.npz"), **(old if newer == "h5" else new).state_dict())
        (new if newer == "h5" else old).save_hdf5(str(d / "wavenet.model"))
        t0 = os.path.getmtime(str(d / "wavenet.model.npz"))
        os.utime(str(d / "wavenet.model"), (t0 + (10 if newer == "h5" else -10),) * 2)
        net = WaveNet(Params(_TINY), seed=7)
        net.load(str(d))
        for k, v in new.state_dict().items():
            assert np.array_equal(net.state_dict()[k], v), (newer, k)


def test_hdf5_files_written_here_are_read_by_h5py(tmp_path):
    """The other direction, when the image's second interpreter (the one that has h5py) exists: h5py must see float32
    datasets at Chainer's paths with the values written."""
    hdf5_io = _need_hdf5()
    py39 = "/opt/conda/bin/python3.9"
    if not os.path.exists(py39) or subprocess.run([py39, "-c", "import h5py"], capture_output=True).returncode != 0:
        pytest.skip("no interpreter with h5py on this machine")
    rng = np.random.RandomState(2)
    arrays = {"causal_0/W": rng.standard_normal((4, 16, 1, 2)).astype(np.float32),
              "causal_0/b": rng.standard_normal(4).astype(np.float32), "t": np.asarray(7)}
    fn = str(tmp_path / "w.model")
    hdf5_io.write_datasets(fn, arrays)
    code = ("import h5py, numpy as np, sys\n"
            "f = h5py.File(sys.argv[1], 'r')\n"
            "w = f['causal_0/W']\n"
            "assert w.dtype == np.float32 and w.shape == (4, 16, 1, 2) and w.compression == 'gzip' and w.compression_opts == 4\n"
            "assert f['t'].shape == () and int(f['t'][()]) == 7\n"
            "np.save(sys.argv[2], np.concatenate([np.asarray(w).ravel(), np.asarray(f['causal_0/b'])]))\n")
    r = subprocess.run([py39, "-c", code, fn, str(tmp_path / "back.npy")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    back = np.load(str(tmp_path / "back.npy"))
    assert np.array_equal(back, np.concatenate([arrays["causal_0/W"].ravel(), arrays["causal_0/b"]]))


def test_get_optimizer_names_and_the_setters():
    """wavenet.py:81-97 (names), 482-513 (which attribute update_laerning_rate / update_momentum touch)."""
    from wavenet_amd.wavenet import AdamState, EveState, RuleState
    base = dict(causal_conv_channels=[8], residual_conv_channels=[8, 8], residual_num_blocks=1, softmax_conv_channels=[8, 256])
    for name, cls in [("adam", AdamState), ("Adam", AdamState), ("eve", EveState), ("adagrad", RuleState),
                      ("AdaDelta", RuleState), ("nesterov", RuleState), ("nesterovag", RuleState), ("rmsprop", RuleState),
                      ("momentumsgd", RuleState), ("sgd", RuleState)]:
        net = WaveNet(Params(dict(base, optimizer=name, momentum=0.7)))
        opt = net.optimizer
        assert type(opt) is cls, name
        net.update_laerning_rate(0.5)
        net.update_momentum(0.3)
        if cls is RuleState:
            assert opt.lr == (0.0001 if name.lower() == "adadelta" else 0.5)
            assert opt.hyper == (0.0 if name in ("sgd", "adagrad") else 0.3)
        else:
            assert opt.alpha == 0.5 and opt.beta1 == 0.3
    with pytest.raises(Exception):
        WaveNet(Params(dict(base, optimizer="lion")))


def test_load_after_save_reads_the_npz_without_an_hdf5_library(tmp_path, capsys, monkeypatch):
    """save() writes the HDF5 copy FIRST and the .npz last, so after every save() the .npz is the newer file:
    load() takes it, says nothing about "both", and works on a box with no HDF5 library; a newer HDF5 file that cannot be read
    for want of a library falls back to the .npz next to it with a warning."""
    _need_hdf5()
    from wavenet_amd import Params, WaveNet
    a = WaveNet(Params(_TINY), seed=8)
    a.save(str(tmp_path))
    assert os.path.getmtime(str(tmp_path / "wavenet.model.npz")) >= os.path.getmtime(str(tmp_path / "wavenet.model"))
    capsys.readouterr()

    def no_lib(self, filename):
        raise ImportError("no HDF5 library (test)")
    monkeypatch.setattr(WaveNet, "load_hdf5", no_lib)
    b = WaveNet(Params(_TINY), seed=9)
    b.load(str(tmp_path))
    out = capsys.readouterr().out
    assert "both" not in out and "wavenet.model.npz" in out
    assert all(np.array_equal(b.state_dict()[k], v) for k, v in a.state_dict().items())
    t0 = os.path.getmtime(str(tmp_path / "wavenet.model.npz"))
    os.utime(str(tmp_path / "wavenet.model"), (t0 + 10,) * 2)
    c = WaveNet(Params(_TINY), seed=10)
    c.load(str(tmp_path))
    assert "OLDER" in capsys.readouterr().out
    assert all(np.array_equal(c.state_dict()[k], v) for k, v in a.state_dict().items())


def test_step_plan_argument_errors_do_not_need_a_gpu():
    """ABI 5's step plan (wn_plan_*): the host-side state machine refuses misuse with a message before any HIP call -- NULL /
    misaligned / too little device memory, preparing or finishing a plan that is not in the right state, an array to zero that is
    not 16-byte aligned or not a multiple of four floats -- and wn_plan_stats reports the state (0 idle, 1 recording)."""
    import ctypes as C
    lib = _lib.lib()
    h = C.c_void_p()
    assert lib.wn_plan_create(C.byref(h), None, 1 << 20) == -1
    assert lib.wn_plan_create(C.byref(h), 0x1000100, 1 << 20) == 0                      # (never dereferenced on the host)
    out = (C.c_int64 * 8)()
    assert lib.wn_plan_stats(h, out) == 0 and list(out)[:3] == [0, 0, 0]
    assert lib.wn_plan_prepare(h, None, 0, None) == -1 and b"not ready" in lib.wn_last_error()
    assert lib.wn_plan_finish(h, None) == -1 and b"not recording" in lib.wn_last_error()
    assert lib.wn_plan_record(h) == 0
    assert lib.wn_plan_stats(h, out) == 0 and out[0] == 1
    assert lib.wn_plan_prepare(h, None, 0, None) == -1                                    # recording, not ready
    assert lib.wn_plan_destroy(h) == 0
    h2 = C.c_void_p()
    assert lib.wn_plan_create(C.byref(h2), 0x1000101, 1 << 20) == -1 and b"aligned" in lib.wn_last_error()
    assert lib.wn_plan_create(C.byref(h2), 0x1000100, 1024) == -1
    assert lib.wn_plan_create(None, 0x1000100, 1 << 20) == -1
