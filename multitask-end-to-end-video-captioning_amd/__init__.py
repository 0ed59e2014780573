"""MI355X-native S2VT REINFORCE hot path (gfx950 HIP kernels behind the C ABI of include/s2vt.h).

The directory name is not a Python identifier; import it as ``import s2vt_amd`` (alias module at the
repository root) or ``importlib.import_module("multitask-end-to-end-video-captioning_amd")``.
"""
from . import hostglue  # noqa: F401
from ._lib import S2VTChainTimeout, S2VTLibraryError, lib, lib_path  # noqa: F401
