"""MI355X (gfx950) SSD300 forward + MultiBox loss + backward + NMS decode.

Drop-in for the hot path of nitishsaDire/objectDetection_ssd: `Model.SSD_300`,
`Losses.ssd`, `Losses.inference` keep the reference's call signatures; the
arithmetic runs in hand-written HIP kernels behind the C ABI of
`include/ssd_gfx950.h` (libssd_gfx950.so).  There is no CPU / eager fallback.

`install_dropin()` registers this package's Model / Losses / Util / Dataset under the
reference's top-level module names so that the reference's `train_function.py`
(`from Losses import *`, `from Util import *`) and `train.py` (`from Model import SSD_300`,
`from Util import all_images, ...`, `from Dataset import ...`) import them unchanged.
`install_dropin(dataset=False)` swaps the hot path only (Model, Losses) and leaves the
reference's own `Dataset.py` and the `Util.py` it takes its CPU augmentations from in place.
"""
import sys as _sys

__version__ = "0.2.0"


def install_dropin(dataset: bool = True) -> None:
    """dataset=True: Model, Losses, Util and Dataset are this package's (the input pipeline runs on the GPU: workers
    plan, `inputs.to(device)` renders).  dataset=False: only Model and Losses are replaced; `import Util` / `import Dataset`
    keep finding the reference's files (its PIL / torchvision augmentations, its VOC lists)."""
    from . import Losses, Model
    _sys.modules["Model"] = Model
    _sys.modules["Losses"] = Losses
    if dataset:
        from . import Dataset, Util
        _sys.modules["Util"] = Util
        _sys.modules["Dataset"] = Dataset
