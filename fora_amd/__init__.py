"""fora_amd -- MI355X-native FORA single-source Personalized PageRank engine.

The product is the C-ABI library libfora_hip.so (include/fora_hip.h) plus the host
CLI `fora query|topk|build`.  This Python package only binds that C ABI with ctypes
for tests and bench.py; there is no Python or CPU compute path.
"""
from .capi import Engine, ForaError, QueryStats, Timing, lib_path  # noqa: F401
