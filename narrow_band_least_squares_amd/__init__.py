"""MI355X-native narrow-band least-squares / least-trimmed-squares array processing.

Same Python call surface as amiezzi/narrow_band_least_squares (+ ``lts_array.ltsva``); the
(band x window) hot path runs in hand-written HIP kernels behind the C ABI of
``include/nbls.h`` (ctypes, no PyTorch on the compute path).  There is no CPU fallback.
"""
import sys as _sys

from .narrow_band_least_squares import (narrow_band_least_squares, narrow_band_loop,
                                        narrow_band_least_squares_parallel)
from .lts_array import ltsva
from .helpers import (get_freqlist, get_winlenlist, filter_data, make_float, get_rij,
                      write_txtfile, read_txtfile)
from .stream import Stream, Trace, Stats
from .engine import resident_trace

__all__ = ['narrow_band_least_squares', 'narrow_band_loop', 'narrow_band_least_squares_parallel',
           'ltsva', 'get_freqlist', 'get_winlenlist', 'filter_data', 'make_float', 'get_rij',
           'write_txtfile', 'read_txtfile', 'Stream', 'Trace', 'Stats', 'install_as_reference_modules', 'resident_trace']


def install_as_reference_modules():
    """Register this package's modules under the reference's module names, so that a script
    written like the reference's example.py (``from narrow_band_least_squares import ...``,
    ``from helpers import ...``, ``from lts_array import ltsva``) runs on the GPU path unchanged."""
    # (the package attribute `narrow_band_least_squares` is the function; fetch the modules by name)
    for name in ('narrow_band_least_squares', 'helpers', 'lts_array'):
        _sys.modules[name] = _sys.modules[__name__ + '.' + name]
