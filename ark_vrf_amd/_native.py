"""ctypes binding of libavrf.so (the C ABI declared in include/avrf.h).

There is no CPU fallback: if the shared library is missing or no MI355X is visible, the
calls raise.  The library is built in-tree by `__graft_entry__.build()` /
`make -C ark_vrf_amd/csrc`.
"""
import ctypes as C
import os

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_DIR, "libavrf.so")

OK, VERIFICATION_FAILURE, INVALID_DATA, RING_CAPACITY_EXCEEDED, SRS_LOOKUP_FAILED = 0, 1, 2, 3, 4
ERR_NO_DEVICE, ERR_BAD_ARG = -1, -2

BANDERSNATCH_SHA512_ELL2 = 0
BABYJUBJUB_SHA512_TAI = 1


class AvrfError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AvrfError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        L.avrf_version.restype = C.c_char_p
        L.avrf_batch_last_terms.restype = C.c_size_t
        _lib = L
    return _lib


def _u8(data):
    data = bytes(data)
    return (C.c_uint8 * max(1, len(data))).from_buffer_copy(data.ljust(1, b"\0"))


def _u32(vals):
    vals = list(vals)
    return (C.c_uint32 * max(1, len(vals)))(*vals)


def device_count():
    return lib().avrf_device_count()


class Context:
    """One engine instance = suite + HIP stream + device workspace (avrf_ctx)."""

    def __init__(self, suite=BANDERSNATCH_SHA512_ELL2, device=0):
        self._h = C.c_void_p()
        st = lib().avrf_ctx_create(int(suite), int(device), C.byref(self._h))
        if st != OK:
            raise AvrfError(f"avrf_ctx_create failed with {st} (no MI355X visible? there is no CPU fallback)")
        self.suite = suite

    def close(self):
        if self._h:
            lib().avrf_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- msm_unchecked
    def msm(self, bases_xy, scalars):
        n = len(scalars) // 32
        assert len(bases_xy) == 64 * n
        out = (C.c_uint8 * 64)()
        st = lib().avrf_msm_te(self._h, C.c_size_t(n), _u8(bases_xy), _u8(scalars), out)
        if st != OK:
            raise AvrfError(f"avrf_msm_te -> {st}")
        return bytes(out)

    # -- thin batch
    @staticmethod
    def _pack(items_ios, ads):
        iob = b"".join(i + o for ios in items_ios for i, o in ios)
        return _u8(iob), _u32(len(ios) for ios in items_ios), _u8(b"".join(ads)), _u32(len(a) for a in ads)

    def thin_batch_verify(self, pks_xy, items_ios_xy, ads, proofs):
        iob, cnt, adb, adl = self._pack(items_ios_xy, ads)
        return lib().avrf_thin_batch_verify(self._h, C.c_size_t(len(pks_xy)), _u8(b"".join(pks_xy)), iob, cnt, adb, adl,
                                            _u8(b"".join(proofs)))

    def thin_batch_stage(self, pks_xy, items_ios_xy, ads, proofs):
        iob, cnt, adb, adl = self._pack(items_ios_xy, ads)
        return lib().avrf_thin_batch_stage(self._h, C.c_size_t(len(pks_xy)), _u8(b"".join(pks_xy)), iob, cnt, adb, adl,
                                           _u8(b"".join(proofs)))

    def thin_batch_stage_raw(self, n, pks_xy, ios_xy, io_counts, ads, ad_lens, proofs):
        """Same as thin_batch_stage with pre-packed buffers (bytes / numpy arrays)."""
        return lib().avrf_thin_batch_stage(self._h, C.c_size_t(n), _u8(pks_xy), _u8(ios_xy), _u32(io_counts), _u8(ads),
                                           _u32(ad_lens), _u8(proofs))

    def thin_batch_run(self):
        return lib().avrf_thin_batch_run(self._h)

    def last_terms(self):
        k = lib().avrf_batch_last_terms(self._h, None, None)
        bases, sc = (C.c_uint8 * max(1, 64 * k))(), (C.c_uint8 * max(1, 32 * k))()
        k2 = lib().avrf_batch_last_terms(self._h, bases, sc)
        assert k2 == k
        return bytes(bases)[: 64 * k], bytes(sc)[: 32 * k]

    def last_timing(self):
        out = (C.c_double * 8)()
        lib().avrf_last_timing(self._h, out)
        return list(out)
