"""dctdomain_amd -- MI355X-native DCT fingerprints (the ``Fingerprint.quantize`` path of
mgtools/DCTdomain) behind the reference's own class surface.

    from dctdomain_amd import Fingerprint          # drop-in for src/fingerprint.py
    from dctdomain_amd import quantize_batch, LayerBatch, PieceTable   # ragged batches
    from dctdomain_amd import quantize_windows                          # ... of sequences given as overlapping windows

The compute lives in ``libdctfp.so`` (hand-written HIP for gfx950, C ABI in
``include/dctfp.h``); importing this package without that library raises ImportError.
"""

from . import _lib

_lib.load()     # fail loudly if the HIP extension has not been built

from .batch import LayerBatch, PieceTable, quantize_batch, quantize_windows, window_geometry  # noqa: E402
from .fingerprint import Fingerprint  # noqa: E402
from ._lib import Context, DctfpError, get_context  # noqa: E402

__all__ = ['Fingerprint', 'LayerBatch', 'PieceTable', 'quantize_batch', 'quantize_windows', 'window_geometry', 'Context', 'DctfpError', 'get_context']
__version__ = '0.1.0'
