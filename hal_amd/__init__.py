"""hal_amd — MI355X-native traversal hot path of HAL (halLiftover block mapping) behind a C ABI.

This package is a thin ctypes binding over ``libhgx.so`` (built in-tree from ``hal_amd/csrc`` by
``__graft_entry__.build()`` / ``make -C hal_amd/csrc lib``).  There is no Python or CPU fallback for
the compute path: if the library is missing, importing the compute entry points raises.
"""
from ._lib import lib, HgxError, LIB_PATH  # noqa: F401
from .api import (  # noqa: F401
    Alignment,
    LiftoverPlan,
    build_phases,
    format_block_results,
    Comm,
    Interval,
    Record,
    RandOptions,
    liftover_convert,
    liftover_convert_bytes,
    liftover_convert_multi,
    liftover_render_blobs,
    alignment_depth_multi,
    maf_export_multi,
    RECORD_DTYPE,
)
