"""
qingdai_amd/double_buffer.py -- the read/write/swap container contract of
pygcm/numerics/double_buffer.py:47-184, as the host-side mirror of how the
device path treats its state: kernels read the front slot of a field and write the back slot, and
`qd_swap` exchanges the two pointers in O(1) (qd_internal.h).

Contract (identical to the reference's, checked by tests/test_double_buffer_cpu.py, which restates the
reference's own tests/test_double_buffering.py):
  * `read` is what readers see, `write` is the next state, `swap()` flips them without copying and
    invalidates the write side;
  * indexing reads from `read`; item assignment goes to `write`, and the FIRST write after a swap (or
    after construction) first mirrors `read` into `write`, so partial updates keep the rest of the state;
  * `np.asarray(x)` / ufuncs see `read`; a ufunc with `out=x` lands in `write` (same first-write mirror);
  * `x[...] = x` is refused.
"""
from __future__ import annotations

import numpy as np


class DoubleBufferingArray:
    __slots__ = ("_slots", "_front", "_back_is_current", "__weakref__")
    __array_priority__ = 1000          # let our __array_ufunc__ win over ndarray's

    def __init__(self, shape, dtype=np.float64, initial_value=0.0):
        self._slots = [np.full(shape, initial_value, dtype=dtype), np.full(shape, initial_value, dtype=dtype)]
        self._front = 0
        self._back_is_current = False   # has `write` been brought up to date with `read` since the last swap?

    # -- the two sides
    @property
    def read(self):
        return self._slots[self._front]

    @property
    def write(self):
        return self._slots[self._front ^ 1]

    def swap(self):
        self._front ^= 1
        self._back_is_current = False

    @property
    def shape(self):
        return self.read.shape

    @property
    def dtype(self):
        return self.read.dtype

    def zero_write(self):
        self.write[...] = 0
        self._back_is_current = True

    def _prepare_write(self):
        if not self._back_is_current:
            np.copyto(self.write, self.read)
            self._back_is_current = True
        return self.write

    # -- indexing
    def __getitem__(self, key):
        return self.read[key]

    def __setitem__(self, key, value):
        if value is self:
            raise ValueError("DoubleBufferingArray: self-aliasing write is not allowed (dba[...] = dba).")
        self._prepare_write()[key] = value

    # -- NumPy interop
    def __array__(self, dtype=None, copy=None):
        return self.read if dtype is None else np.asarray(self.read, dtype=dtype)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method != "__call__":
            return NotImplemented
        args = [x.read if isinstance(x, DoubleBufferingArray) else x for x in inputs]
        out = kwargs.get("out")
        if out is None:
            return ufunc(*args, **kwargs)
        outs = out if isinstance(out, tuple) else (out,)
        kwargs["out"] = tuple(y._prepare_write() if isinstance(y, DoubleBufferingArray) else y for y in outs)
        return ufunc(*args, **kwargs)

    def __repr__(self):
        return (f"DoubleBufferingArray(shape={self.shape}, dtype={self.dtype}, read=buf{self._front}, "
                f"write=buf{self._front ^ 1})")
