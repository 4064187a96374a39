"""detqmc_amd -- MI355X-native DQMC sweep engine (hot path of crstnbr/detqmc).

Python only binds the C ABI (include/dqmc_hip.h) and the C++ host layer (include/detsdw_host.h)
for tests and the benchmark harness; the product is the shared library.
"""
from ._lib import DqmcError, load, LIB_PATH  # noqa: F401
from .model import DetHubbard, DetSDW, DetSDWBatch, HubbardParams, KernelContext, SDWParams  # noqa: F401
