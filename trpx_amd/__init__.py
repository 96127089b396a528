"""trpx_amd -- MI355X-native TERSE/PROLIX codec hot path (senikm/trpx drop-in for that path).

Layout: ``csrc/`` hand-written HIP kernels (gfx950) + the C ABI (``include/trpx_hip.h``);
``_lib`` ctypes binding; ``codec`` device-tensor entry points (needs torch); ``terse`` the
host-side mirror of the reference ``jpa::Terse`` class (the C++ mirror is
``include/trpx/Terse.hpp``); ``sharded`` one-process-per-GPU frame sharding with the RCCL size
gather.
"""
from ._lib import TrpxError, U8, I8, U16, I16, U32, I32  # noqa: F401
from .terse import Terse  # noqa: F401

__all__ = ["Terse", "TrpxError", "U8", "I8", "U16", "I16", "U32", "I32"]
