"""linearham_amd -- MI355X-native phylo-HMM log-likelihood hot path of matsengrp/linearham.

The product is the HIP library (csrc/ -> lib/liblinearham_hip.so, C ABI in include/linearham_amd.h)
and the C++ host (csrc/host -> lib/liblinearham_host.so).  This Python package is only the thin
ctypes binding used by tests/ and bench.py; it contains no numerics and no CPU fallback.
"""
from .capi import (HipLibrary, Family, FamilyDesc, JunctionTables, Segments, load_library,  # noqa: F401
                   library_path)
