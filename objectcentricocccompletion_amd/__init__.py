"""MI355X-native OcOccNet hot path (Ghostish/ObjectCentricOccCompletion).

Host side: Python mirrors of the reference operator interfaces
(mmdet3d/ops/{voxel,spconv,sst,occ}); device side: hand-written HIP kernels for
gfx950 behind the C ABI of include/ococc_hip.h (libococc_hip.so, loaded by
``_lib``).  No CPU fallback exists: importing the package without the built
library raises.
"""
from . import _lib  # noqa: F401  (fails loudly when libococc_hip.so is missing)

__version__ = '0.1.0'
