"""CPU oracle for the WaveNet hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy / torch-CPU) of the algorithm the
reference implements in wavenet.py / faster_wavenet.py / data.py /
train_audio/{train,generate}.py.  It is the *checker*: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  Nothing under ``wavenet_amd/`` imports it, and the product path
raises when the HIP library is missing instead of falling back to this code.

PARITY UNPINNED.  The reference's arithmetic lives in Chainer "2" (README.md:19;
not vendored, no version pin), the reference is Python-2-only and none of its
``_tests_`` scripts stores or asserts a value, so there is no golden vector of
the reference itself to check this restatement against.  What *is* pinned:

* the literal restatement (pad -> reshape -> conv2d -> reshape -> cut/pad,
  wavenet.py:294-342) and an independent closed-form restatement agree
  (tests/test_oracle.py), and KAT-1 derived by hand from
  _tests_/dilated_conv/test_conv.py:8-16;
* the sampler restatement equals ``numpy.random.RandomState.choice`` (the call
  the reference makes at train_audio/generate.py:39) draw for draw;
* the mu-law quantiser equals the formula at data.py:19-23 evaluated in
  float64 for all 65,536 int16 inputs.
"""
