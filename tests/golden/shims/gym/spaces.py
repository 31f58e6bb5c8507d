import numpy as np


class Space(object):
    def __init__(self, shape=None, dtype=None):
        self.shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        low = np.asarray(low)
        high = np.asarray(high)
        if shape is None:
            shape = low.shape
        else:
            low = np.full(shape, low)
            high = np.full(shape, high)
        super().__init__(shape, dtype)
        # gym 0.18 casts the bounds to the space dtype (float64 -> float32 here)
        self.low = low.astype(self.dtype)
        self.high = high.astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and np.all(x >= self.low) and np.all(x <= self.high)


class Discrete(Space):
    def __init__(self, n):
        self.n = n
        super().__init__((), np.int64)
