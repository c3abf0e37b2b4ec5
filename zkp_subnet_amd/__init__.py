"""MI355X-native KZG segment prover: drop-in for the prover seam of apollozkp/zkp-subnet's miner.

Python host code (this package) -> ctypes -> libkzg_mi355x.so (hand-written HIP for gfx950).
See DESIGN.md for the hot path and INTEGRATION.md for the binding into the reference miner.
"""
from ._native import KzgError, lib_available  # noqa: F401
from .engine import HipEngine  # noqa: F401
from .client import Client, Response  # noqa: F401
from .multi import MultiDeviceClient, SegmentedMsm  # noqa: F401
