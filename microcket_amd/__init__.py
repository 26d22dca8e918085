"""microcket_amd -- MI355X-native implementation of Microcket's sam2pairs hot path.

The product is native: ``libmkt_hip.so`` (hand-written HIP kernels for gfx950 behind the C ABI of
``include/mkt.h``) and the drop-in ``bin/sam2pairs`` executable.  This Python package is a thin
ctypes binding used by the tests and by ``bench.py``; it never computes anything itself and has no
CPU fallback: without the built library or without a GPU every call raises.
"""
from .capi import (  # noqa: F401
    Context,
    EXT_KEYS,
    EXT_LANES,
    MktError,
    MODE_FLASH,
    MODE_UNC,
    PairsSorter,
    TILES_AUTO,
    TILES_FAST,
    TILES_SMALL,
    Stats,
    device_count,
    exe_path,
    lib_path,
    load_library,
    rmdup,
    run_sam2pairs,
    sam_to_bam,
)

__all__ = [
    "Context", "EXT_KEYS", "EXT_LANES", "MktError", "MODE_FLASH", "MODE_UNC", "PairsSorter", "TILES_AUTO", "TILES_FAST", "TILES_SMALL", "Stats",
    "device_count", "exe_path", "lib_path", "load_library", "rmdup", "run_sam2pairs", "sam_to_bam",
]
