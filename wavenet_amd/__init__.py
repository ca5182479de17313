"""wavenet_amd: MI355X-native WaveNet dilated-causal-conv engine with the musyoku/wavenet API.

    from wavenet_amd import WaveNet, FasterWaveNet, Params      # was: from wavenet import ...
    from wavenet_amd import data                                 # was: import data
"""
from .wavenet import Params, WaveNet, zero_prefix      # noqa: F401
from .faster_wavenet import FasterWaveNet              # noqa: F401
from . import data                                     # noqa: F401
from .graph import TrainStepGraph                      # noqa: F401
from ._lib import WaveNetHipError, set_gemm_precision, get_gemm_precision   # noqa: F401
