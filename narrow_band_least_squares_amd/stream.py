"""Duck-typed ``Stream`` / ``Trace`` / ``Stats``: the part of obspy's data model that the
reference's hot path and its plotting actually touch: ``st[0].data``,
``st[0].stats.sampling_rate`` / ``.npts`` / ``.starttime`` / ``.latitude`` / ``.longitude``,
``st.copy()``, ``len(st)``, ``st[i].data`` assignment, ``st[0].times('matplotlib')`` and the
trace used as an array (plotting.py:76-77).  A real obspy ``Stream`` works with every function
of this package as well; these classes exist because obspy is not installed here.
"""
import numpy as np


def start_datenum(starttime):
    """Matplotlib date number (days since 1970-01-01T00:00, the matplotlib >= 3.3 epoch).

    Accepts a float (already a date number), an object with ``matplotlib_date`` (obspy
    ``UTCDateTime``), a ``datetime``-like with ``timestamp`` or a ``numpy.datetime64``."""
    if hasattr(starttime, 'matplotlib_date'):
        return float(starttime.matplotlib_date)
    if isinstance(starttime, np.datetime64):
        us = (starttime - np.datetime64('1970-01-01T00:00:00')) / np.timedelta64(1, 'us')
        return float(us) / 86400e6
    if hasattr(starttime, 'timestamp') and not isinstance(starttime, (int, float)):
        ts = starttime.timestamp
        ts = ts() if callable(ts) else ts
        return float(ts) / 86400.0
    return float(starttime)


class Stats:
    def __init__(self, sampling_rate=1.0, npts=0, starttime=0.0, latitude=None, longitude=None,
                 network='', station='', location='', channel=''):
        self.sampling_rate = float(sampling_rate)
        self.npts = int(npts)
        self.starttime = starttime
        self.latitude = latitude
        self.longitude = longitude
        self.network = network
        self.station = station
        self.location = location
        self.channel = channel

    @property
    def delta(self):
        return 1.0 / self.sampling_rate

    def copy(self):
        new = Stats.__new__(Stats)
        new.__dict__.update(self.__dict__)
        return new


class Trace:
    def __init__(self, data, stats=None):
        self._data = np.asarray(data)
        self.stats = stats if stats is not None else Stats(npts=len(self._data))
        self.stats.npts = len(self._data)

    @property
    def data(self):
        return self._data

    @data.setter
    def data(self, value):
        self._data = np.asarray(value)
        self.stats.npts = len(self._data)

    def __len__(self):
        return len(self._data)

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self._data, dtype=dtype)

    def copy(self):
        return Trace(self._data.copy(), self.stats.copy())

    def times(self, type='relative'):
        """``'relative'``: seconds from the start; ``'matplotlib'``: matplotlib date numbers
        (obspy ``Trace.times``)."""
        rel = np.arange(len(self._data)) / self.stats.sampling_rate
        if type == 'relative':
            return rel
        if type == 'matplotlib':
            return start_datenum(self.stats.starttime) + rel / 86400.0
        raise ValueError('unsupported times() type %r' % (type,))


class Stream(list):
    def __init__(self, traces=None):
        super().__init__(traces if traces is not None else [])

    def copy(self):
        return Stream(tr.copy() for tr in self)
