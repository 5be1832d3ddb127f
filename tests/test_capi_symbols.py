"""CPU-side checks of the drop-in boundary: libavrf.so loads without a GPU, exports every symbol
declared in include/*.h, and refuses to create a context when no device is visible (no CPU fallback)."""
import ctypes as C
import glob
import os
import re

from conftest import ROOT


def declared_symbols():
    syms = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = open(h).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        syms |= set(re.findall(r"\b(avrf_[a-z0-9_]+)\s*\(", src))
    return syms


def test_library_exports_every_declared_symbol():
    from ark_vrf_amd import _native as nat
    lib = nat.lib()
    syms = declared_symbols()
    assert len(syms) >= 25, syms
    missing = [s for s in sorted(syms) if not hasattr(lib, s)]
    assert not missing, missing
    assert b"avrf" in lib.avrf_version()


def test_no_cpu_fallback_without_device():
    from ark_vrf_amd import _native as nat
    if nat.device_count() > 0:
        return  # on a GPU box the context tests are in the -m gpu suite
    h = C.c_void_p()
    assert nat.lib().avrf_ctx_create(0, 0, C.byref(h)) == nat.ERR_NO_DEVICE
    try:
        nat.Context(0)
    except nat.AvrfError:
        pass
    else:
        raise AssertionError("Context() must fail loudly without a GPU")


def test_error_codes_mirror_reference_enum():
    """include/avrf.h status codes follow ark_vrf::Error (src/lib.rs:135-147) in declaration order."""
    src = open(os.path.join(ROOT, "include", "avrf.h")).read()
    for name, val in [("AVRF_OK", 0), ("AVRF_VERIFICATION_FAILURE", 1), ("AVRF_INVALID_DATA", 2),
                      ("AVRF_RING_CAPACITY_EXCEEDED", 3), ("AVRF_SRS_LOOKUP_FAILED", 4)]:
        assert re.search(rf"{name}\s*=\s*{val}\b", src), name
