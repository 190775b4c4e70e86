"""Host-side plumbing between the distribution classes and the HIP engine.

* `as_device_data` turns the (A, W) ndarray pair the reference passes to every
  call (networks.py:28,69) into a device-resident `DeviceData`, cached by array
  identity so that a sweep loop written against the reference
  (examples/demo_model_comparison.py:69-72) uploads its data once.
* `EngineBacked` keeps a host copy of a component's state (numpy, so that `rvs`,
  `P`, `log_prior` work anywhere) in step with the engine handle: host edits are
  pushed before the next engine call, engine results are pulled on first read.

There is no CPU implementation of resample / log_likelihood here: they need the
engine and raise `_abi.EngineError` without a HIP device.
"""
import collections

import numpy as np

from . import _abi

_DATA_CACHE = collections.OrderedDict()
_DATA_CACHE_SIZE = 4
_IMMUTABLE = {}        # id(array) -> weakref: arrays the caller has declared read-only (identity is then the whole key)


_HASH_WARN_S = 2e-3
_hash_warned = [False]


def _content_hash(a):
    """64-bit hash of the WHOLE buffer.  The reference reads its arrays on every call (networks.py:28,69); the engine keeps a
    device copy, so a copy is only reused when every byte of the host array is what it was when the copy was made - one flipped
    entry is new data.  Computed inside the library on all host threads (gg_hash64: the ctypes call releases the GIL; a
    single-threaded Python-side hash of a 134 MB float64 W costs 17 ms per call), and still milliseconds for a large network:
    the first time it takes more than a millisecond a warning names the ways around it."""
    import time
    buf = a if a.flags.c_contiguous else np.ascontiguousarray(a)
    _abi.lib()                                   # (loaded outside the timed call)
    t0 = time.perf_counter()
    h = _abi.hash64(buf)
    dt = time.perf_counter() - t0
    if dt > _HASH_WARN_S and buf.nbytes >= (8 << 20) and not _hash_warned[0]:
        _hash_warned[0] = True
        import warnings
        warnings.warn("graphistician_amd: hashing a %.0f MB data array took %.1f ms, and a writeable array is hashed on every "
                      "resample / log_likelihood call (it may have been edited in place since the last one). For a fixed network "
                      "make the arrays read-only (`A.setflags(write=False)`), call graphistician_amd.engine.declare_immutable(A, W) "
                      "or pass a DeviceData handle: the device copy is then found by identity alone."
                      % (buf.nbytes / 1e6, dt * 1e3), RuntimeWarning, stacklevel=4)
    return h


def _is_frozen(a):
    """Nobody can write this buffer through numpy: the array and every ndarray it is a view of are read-only (`setflags(write=
    False)`; a read-only view of a writeable base does not count), and a foreign exporter underneath is read-only too.  Such an
    array is keyed by identity alone - for as long as it stays read-only (switching the flag back on, editing, and switching it
    off again between two calls is not seen)."""
    x = a
    while isinstance(x, np.ndarray):
        if x.flags.writeable:
            return False
        x = x.base
    if x is None:
        return True
    try:
        return bool(memoryview(x).readonly)
    except TypeError:
        return False


def _fingerprint(a):
    """Cache key of one host array: identity (id, buffer address, shape, dtype) plus a hash of its full content, so that
    any in-place edit - `A[:] = model.rvs()`, masking a held-out row and column, one flipped entry - is seen as new data
    and uploaded again.  Arrays that cannot be written (`_is_frozen`) and arrays registered with `declare_immutable` skip the
    hash (identity only): that, or passing a `DeviceData` handle, is what a loop over a large fixed network should do."""
    if a is None:
        return None
    ident = (id(a), a.shape, a.dtype.str, a.__array_interface__["data"][0])
    ref = _IMMUTABLE.get(id(a))
    if (ref is not None and ref() is a) or _is_frozen(a):
        return ident + ("immutable",)
    return ident + (_content_hash(a),)


def declare_immutable(*arrays):
    """Promise that these host arrays will not be edited in place while they are used as data: their device copies are
    then found by identity alone (no per-call hash of the buffer).  `invalidate` / `clear_data_cache` take it back."""
    import weakref
    for a in arrays:
        if a is None:
            continue
        a = np.asarray(a)
        _IMMUTABLE[id(a)] = weakref.ref(a, lambda _r, k=id(a): _IMMUTABLE.pop(k, None))


def as_device_data(A, W=None, device=0, immutable=False, missing_ok=False, wide=False):
    """Device-resident network for (A, W); accepts an existing DeviceData.  immutable=True: see `declare_immutable`;
    missing_ok: NaN entries of A are unobserved (eigenmodels only, eigenmodel.py:121-127); wide: reference-precision weight
    storage (two limbs, `DeviceData(..., wide=True)`) for a weight model built with wide_weights=True."""
    if isinstance(A, _abi.DeviceData):
        return A
    if isinstance(A, tuple) and W is None:
        A, W = A
    A = np.asarray(A)
    Wn = None if W is None else np.asarray(W)
    if immutable:
        declare_immutable(A, Wn)
    key = (_fingerprint(A), _fingerprint(Wn), device, bool(missing_ok), bool(wide))
    hit = _DATA_CACHE.get(key)
    if hit is not None:
        _DATA_CACHE.move_to_end(key)
        return hit[0]
    d = None
    if len(_DATA_CACHE) >= _DATA_CACHE_SIZE:
        # A loop that draws new data every iteration (`A = model.rvs(); model.resample(A)`, the reference's Geweke tests)
        # would allocate and free every device buffer each time round: recycle the least recently used handle of the same
        # shape instead (gg_data_update; refused - and then simply not recycled - while sweeps are still pending on it).
        # Only a handle nobody else holds is recycled: one that was handed to a caller keeps the network it was made from.
        import sys
        old_key, entry = next(iter(_DATA_CACHE.items()))
        old, oldW = entry[0], entry[2]
        B = 1 if Wn is None else (1 if Wn.ndim == 2 else Wn.shape[2])
        sole_owner = sys.getrefcount(old) <= 3      # the cache entry, `old`, getrefcount's argument
        if sole_owner and old.N == A.shape[0] and old.device == device and (oldW is None) == (Wn is None) and old.B == B and \
                bool(getattr(old, "wide", False)) == bool(wide):
            try:
                old.update(A, Wn, missing_ok=missing_ok)
                d = old
                del _DATA_CACHE[old_key]
            except _abi.EngineError:
                d = None
        del old, entry
    if d is None:
        d = _abi.DeviceData(A, Wn, device=device, missing_ok=missing_ok, wide=wide)
    _DATA_CACHE[key] = (d, A, Wn)  # keep the arrays alive so ids are not reused
    while len(_DATA_CACHE) > _DATA_CACHE_SIZE:
        _DATA_CACHE.popitem(last=False)
    return d


def clear_data_cache():
    _DATA_CACHE.clear()
    _IMMUTABLE.clear()


def invalidate(A, W=None):
    """Forget the device copy made from these host arrays (and any `declare_immutable` promise about them)."""
    for key in [k for k, v in _DATA_CACHE.items() if v[1] is A or (W is not None and v[2] is W)]:
        del _DATA_CACHE[key]
    for a in (A, W):
        if a is not None:
            _IMMUTABLE.pop(id(a), None)


def draw_seed():
    """Engine seed from numpy's global stream, so np.random.seed(...) fixes the chain."""
    return int(np.random.randint(0, 2 ** 31 - 1)) * 2 ** 31 + int(np.random.randint(0, 2 ** 31 - 1))


class EngineBacked(object):
    """Mixin: state fields mirrored between host numpy arrays and a gg_model handle."""

    # name -> (state id, dtype, shape function(self))
    _fields = {}

    def _init_engine(self, device=0, seed=None, **cfg):
        self._device = device
        self._seed = draw_seed() if seed is None else int(seed)
        self._cfg = dict(cfg, device=device, seed=self._seed)
        self._model = None
        self._host = {}
        self._dirty = set()
        self._stale = set()

    # -- host <-> device -------------------------------------------------------
    def _hget(self, name):
        if name in self._stale:
            sid, dtype, shape = self._fields[name]
            self._host[name] = self._model.get(sid, shape(self), dtype)
            self._stale.discard(name)
        return self._host[name]

    def _hset(self, name, value):
        sid, dtype, shape = self._fields[name]
        v = np.ascontiguousarray(value, dtype=dtype).reshape(shape(self)).copy()
        self._host[name] = v
        self._stale.discard(name)
        self._dirty.add(name)

    def _engine(self):
        if self._model is None:
            self._model = _abi.Model(**self._cfg)   # raises without a HIP device
            self._dirty = set(self._host.keys())
        for name in sorted(self._dirty):
            sid, dtype, shape = self._fields[name]
            self._model.set(sid, self._host[name], dtype)
        self._dirty.clear()
        return self._model

    def _mark_stale(self, *names):
        for n in names:
            self._stale.add(n)

    def _data(self, data, missing_ok=False):
        if isinstance(data, _abi.DeviceData):
            return data
        if isinstance(data, (tuple, list)):
            return as_device_data(data[0], data[1], device=self._device, missing_ok=missing_ok, wide=getattr(self, "_wide", False))
        return as_device_data(data, None, device=self._device, missing_ok=missing_ok)

    @property
    def engine_stats(self):
        return self._engine().stats()
