"""ark_vrf_amd -- MI355X-native batched VRF engine for the hot path of davxy/ark-vrf.

The product is libavrf.so (HIP kernels + C ABI, include/avrf.h); this package is the thin
Python harness binding used by tests and bench.py.  No CPU fallback exists."""
from . import _native  # noqa: F401
from ._native import AvrfError, Context, device_count  # noqa: F401
