"""The minibatch ids are drawn with numpy's global RNG as the reference draws them (mdnn.py:219-222:
np.random.randint(0, n_train, ...), int64 by default); the mirror asks for int32 directly.  That is only
valid while numpy's legacy generator produces the same values, and leaves the same state behind, for both
dtypes -- which is what this test pins."""
import numpy as np
import pytest


@pytest.mark.parametrize('n_train,shape', [(800, (100, 100)), (80000, (122, 8192)), (1, (3, 5)), (7, (11, 13)),
                                           (65536, (50, 64)), (65537, (50, 64))])
def test_int32_draw_is_the_int64_draw(n_train, shape):
    np.random.seed(1234)
    a = np.random.randint(0, n_train, shape)
    after_a = np.random.randint(0, 1 << 30, 4)
    np.random.seed(1234)
    b = np.random.randint(0, n_train, shape, dtype=np.int32)
    after_b = np.random.randint(0, 1 << 30, 4)
    assert a.dtype == np.int64 and b.dtype == np.int32
    assert np.array_equal(a, b)
    assert np.array_equal(after_a, after_b)      # ... and the stream goes on from the same place
