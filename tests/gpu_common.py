"""Shared helpers of the `-m gpu` test modules (the `hip` engine factory fixture lives in tests/conftest.py)."""
import os

import numpy as np

H = bytes.fromhex
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rand_scalars_bytes(n, seed):
    raw = np.random.default_rng(seed).integers(0, 256, size=(n, 32), dtype=np.uint8)
    raw[:, 0] &= 0x3F                       # < 2^254 < r: canonical
    return raw.tobytes()


def ints(b):
    return [int.from_bytes(b[i:i + 32], "big") for i in range(0, len(b), 32)]
