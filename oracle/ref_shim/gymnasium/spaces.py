import numpy as np


class Space:
    shape = None
    dtype = None

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def contains(self, x):
        return True


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        if shape is None:
            shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
        self.shape = tuple(int(s) for s in shape)
        self.low = np.broadcast_to(np.asarray(low), self.shape).astype(self.dtype) if self.shape else np.asarray(low, dtype=self.dtype)
        self.high = np.broadcast_to(np.asarray(high), self.shape).astype(self.dtype) if self.shape else np.asarray(high, dtype=self.dtype)

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(self.dtype)


class Discrete(Space):
    def __init__(self, n, start=0):
        self.n = int(n)
        self.start = int(start)
        self.shape = ()
        self.dtype = np.dtype(np.int64)

    def sample(self):
        return int(np.random.randint(self.n)) + self.start


class MultiDiscrete(Space):
    def __init__(self, nvec, dtype=np.int64):
        self.nvec = np.asarray(nvec, dtype=dtype)
        self.shape = self.nvec.shape
        self.dtype = np.dtype(dtype)

    def sample(self):
        return (np.random.random(self.nvec.shape) * self.nvec).astype(self.dtype)


class Dict(Space):
    def __init__(self, spaces=None, **kw):
        self.spaces = dict(spaces or {})
        self.spaces.update(kw)

    def __getitem__(self, k):
        return self.spaces[k]

    def __setitem__(self, k, v):
        self.spaces[k] = v

    def __iter__(self):
        return iter(self.spaces)

    def keys(self):
        return self.spaces.keys()

    def items(self):
        return self.spaces.items()


class Tuple(Space):
    def __init__(self, spaces):
        self.spaces = tuple(spaces)

    def __getitem__(self, i):
        return self.spaces[i]

    def __len__(self):
        return len(self.spaces)
