"""CPU: the DoubleBufferingArray contract (pygcm/numerics/double_buffer.py:47-184).  The cases restate the
behaviours the reference's own tests/test_double_buffering.py pins: write isolation until swap, first-write
mirroring, NumPy coercion / ufunc routing with out=, multi-output ufuncs, the self-alias guard, zero_write."""
import numpy as np
import pytest

from qingdai_amd.double_buffer import DoubleBufferingArray as DBA


def test_writes_are_invisible_until_swap_and_partial_writes_keep_the_rest():
    x = DBA((2, 3), dtype=float, initial_value=0.0)
    assert np.all(x.read == 0.0)
    x[...] = 1.0
    assert np.all(x.read == 0.0)
    x.swap()
    assert np.all(x.read == 1.0)
    x[0, :] = 5.0                       # first write after the swap mirrors read -> write first
    assert np.all(x.read[0, :] == 1.0)
    x.swap()
    assert np.all(x.read[0, :] == 5.0) and np.all(x.read[1, :] == 1.0)


def test_getitem_reads_front_setitem_writes_back():
    x = DBA((2, 2), dtype=float, initial_value=2.0)
    assert x[0, 0] == 2.0
    x[1, 1] = 9.0
    assert x.read[1, 1] == 2.0
    x.swap()
    assert x.read[1, 1] == 9.0 and x.shape == (2, 2) and x.dtype == np.float64


def test_numpy_coercion_and_ufunc_routing():
    x = DBA((2, 2), dtype=float, initial_value=0.5)
    assert np.array_equal(np.asarray(x), x.read)
    y = np.sin(x)
    assert isinstance(y, np.ndarray) and np.all(x.read == 0.5)
    np.add(x, 1.0, out=x)
    assert np.all(x.read == 0.5)
    x.swap()
    assert np.all(x.read == 1.5)


def test_multi_output_ufunc_out_tuple():
    a, q, r = DBA((2, 2), dtype=int, initial_value=9), DBA((2, 2), dtype=int), DBA((2, 2), dtype=int)
    a[...] = 9
    a.swap()
    np.divmod(a, 4, out=(q, r))
    assert np.all(q.read == 0) and np.all(r.read == 0)
    q.swap(); r.swap()
    assert np.all(q.read == 2) and np.all(r.read == 1)


def test_self_alias_is_refused():
    x = DBA((2, 2))
    with pytest.raises(ValueError):
        x[...] = x


@pytest.mark.parametrize("dtype", [np.float32, np.float64, int])
def test_zero_write_and_repr(dtype):
    x = DBA((1, 3), dtype=dtype, initial_value=7)
    x.swap()
    x.zero_write()
    x.swap()
    assert np.all(x.read == 0)
    assert "DoubleBufferingArray" in repr(x)
    # zero_write counts as the first write: a following partial write must not resurrect the old front
    x.swap(); x.zero_write(); x[0, 0] = 3
    x.swap()
    assert x.read[0, 0] == 3 and np.all(x.read[0, 1:] == 0)
