"""Minimal observation/action space objects.

gymnasium is not installed in the build image; when it is importable its ``spaces`` are used so
``PDEEnv`` is a real ``gym.Env``.  Otherwise these two small classes provide the attributes RL
code reads (``shape``, ``dtype``, ``low``, ``high``, ``n``, ``sample``, ``contains``).
"""

from __future__ import annotations

import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        self.dtype = np.dtype(dtype)
        self.shape = tuple(shape) if shape is not None else np.shape(low)
        self.low = np.full(self.shape, low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype)
        self._rng = np.random.default_rng(seed)

    def sample(self):
        u = self._rng.uniform(self.low.astype(np.float64), self.high.astype(np.float64))
        return u.astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"


class Discrete:
    def __init__(self, n, seed=None):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype(np.int64)
        self._rng = np.random.default_rng(seed)

    def sample(self):
        return int(self._rng.integers(self.n))

    def contains(self, x):
        return isinstance(x, (int, np.integer)) and 0 <= int(x) < self.n

    def __repr__(self):
        return f"Discrete({self.n})"


try:  # pragma: no cover - depends on the environment
    import gymnasium as _gym
    from gymnasium import spaces as _gspaces

    Box = _gspaces.Box  # noqa: F811
    Discrete = _gspaces.Discrete  # noqa: F811
    EnvBase = _gym.Env
    HAVE_GYMNASIUM = True
except Exception:  # ModuleNotFoundError in the build image
    EnvBase = object
    HAVE_GYMNASIUM = False
