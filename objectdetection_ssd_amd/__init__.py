"""MI355X (gfx950) SSD300 forward + MultiBox loss + backward + NMS decode.

Drop-in for the hot path of nitishsaDire/objectDetection_ssd: `Model.SSD_300`,
`Losses.ssd`, `Losses.inference` keep the reference's call signatures; the
arithmetic runs in hand-written HIP kernels behind the C ABI of
`include/ssd_gfx950.h` (libssd_gfx950.so).  There is no CPU / eager fallback.

`install_dropin()` registers this package's Model / Losses / Util under the
reference's top-level module names so that the reference's `train_function.py`
(`from Losses import *`, `from Util import *`) and `train.py`
(`from Model import SSD_300`) import them unchanged.
"""
import sys as _sys

__version__ = "0.1.0"


def install_dropin() -> None:
    from . import Dataset, Losses, Model, Util
    _sys.modules["Model"] = Model
    _sys.modules["Losses"] = Losses
    _sys.modules["Util"] = Util
    _sys.modules["Dataset"] = Dataset
