"""ctypes binding of libavrf.so (the C ABI declared in include/avrf.h).

There is no CPU fallback: if the shared library is missing or no MI355X is visible, the
calls raise.  The library is built in-tree by `__graft_entry__.build()` /
`make -C ark_vrf_amd/csrc`.
"""
import ctypes as C
import os

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AVRF_LIB_PATH") or os.path.join(_DIR, "libavrf.so")   # override: A/B builds of the same library

OK, VERIFICATION_FAILURE, INVALID_DATA, RING_CAPACITY_EXCEEDED, SRS_LOOKUP_FAILED = 0, 1, 2, 3, 4
ERR_NO_DEVICE, ERR_BAD_ARG = -1, -2

BANDERSNATCH_SHA512_ELL2 = 0
BABYJUBJUB_SHA512_TAI = 1
JUBJUB_SHA512_TAI = 2
ED25519_SHA512_TAI = 3          # Tiny / Thin / Pedersen only (no ring suite)
TESTING_SHA256_TAI = 6           # the crate's own test suite: edwards25519 with HashTranscript<Sha256> (no ring)
BANDERSNATCH_SHAKE128_ELL2 = 5   # suite 0's curve with the SHAKE128 sponge as transcript
SECP256R1_SHA256_TAI = 7        # NIST P-256, a genuinely short-Weierstrass suite (Tiny / Thin / Pedersen; no ring): 33-byte points, identity xy = zeros
BANDERSNATCH_SW_SHA512_TAI = 4  # Bandersnatch, short-Weierstrass presentation: 33-byte compressed points (Context.point_len)

THIN_PROOF_LEN = 96       # R_xy || s
PEDERSEN_PROOF_LEN = 256  # Yb_xy || R_xy || Ok_xy || s || sb


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
    if isinstance(data, C.Array):
        return data
    data = bytes(data)
    return (C.c_uint8 * max(1, len(data))).from_buffer_copy(data.ljust(1, b"\0"))


def _u32(vals):
    if isinstance(vals, C.Array):
        return vals
    vals = list(vals)
    return (C.c_uint32 * max(1, len(vals)))(*vals)


def device_count():
    return lib().avrf_device_count()


def set_blocking_sync(device=0, on=True):
    """avrf_device_set_blocking_sync: waiting host threads sleep instead of spinning (device-wide)."""
    return lib().avrf_device_set_blocking_sync(int(device), 1 if on else 0)


class Batch:
    """Host-side packed batch in the C-ABI layout (see include/avrf.h)."""

    def __init__(self, n, ios_xy, io_counts, ads, ad_lens, pks_xy=None, proofs=None, sks=None):
        self.n = n
        self.ios_xy, self.io_counts, self.ads, self.ad_lens = _u8(ios_xy), _u32(io_counts), _u8(ads), _u32(ad_lens)
        self.pks_xy = _u8(pks_xy) if pks_xy is not None else None
        self.proofs = _u8(proofs) if proofs is not None else None
        self.sks = _u8(sks) if sks is not None else None

    @classmethod
    def from_items(cls, items_ios_xy, ads, pks_xy=None, proofs=None, sks=None):
        iob = b"".join(i + o for ios in items_ios_xy for i, o in ios)
        j = lambda v: None if v is None else b"".join(v)
        return cls(len(ads), iob, [len(x) for x in items_ios_xy], b"".join(ads), [len(a) for a in ads],
                   j(pks_xy), j(proofs), j(sks))


class Context:
    """One engine instance = suite + HIP stream + device workspace (avrf_ctx)."""

    def __init__(self, suite=BANDERSNATCH_SHA512_ELL2, device=0):
        self._h = C.c_void_p()
        st = lib().avrf_ctx_create(int(suite), int(device), C.byref(self._h))
        if st != OK:
            raise AvrfError(f"avrf_ctx_create failed with {st} (no MI355X visible? there is no CPU fallback)")
        self.suite = suite
        try:
            lib().avrf_point_len.restype = C.c_size_t
            self.point_len = lib().avrf_point_len(int(suite))  # serialize_compressed size of the suite's points: 32, or 33 (SW form)
        except AttributeError:                                 # an older build loaded through AVRF_LIB_PATH (A/B runs)
            self.point_len = 32

    def set_validation(self, level):
        """0: caller guarantees on-curve subgroup points (reference's typed-point contract); 1: on-curve check; 2: + subgroup."""
        st = lib().avrf_ctx_set_validation(self._h, int(level))
        if st != OK:
            raise AvrfError(f"avrf_ctx_set_validation -> {st}")

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

    def g1_msm(self, bases_xy, scalars):
        """KZG-side MSM on the suite's pairing curve; points as LE x||y (48+48 / 32+32 bytes)."""
        n = len(scalars) // 32
        fq = 48 if self.suite == 0 else 32
        assert len(bases_xy) == 2 * fq * n
        out = (C.c_uint8 * (2 * fq))()
        st = lib().avrf_g1_msm(self._h, C.c_size_t(n), _u8(bases_xy), _u8(scalars), out)
        if st != OK:
            raise AvrfError(f"avrf_g1_msm -> {st}")
        return bytes(out)

    # -- batch verifiers
    def thin_batch_verify(self, pks_xy, items_ios_xy, ads, proofs):
        b = Batch.from_items(items_ios_xy, ads, pks_xy=pks_xy, proofs=proofs)
        return lib().avrf_thin_batch_verify(self._h, C.c_size_t(b.n), b.pks_xy, b.ios_xy, b.io_counts, b.ads, b.ad_lens, b.proofs)

    def thin_batch_stage(self, b):
        return lib().avrf_thin_batch_stage(self._h, C.c_size_t(b.n), b.pks_xy, b.ios_xy, b.io_counts, b.ads, b.ad_lens, b.proofs)

    def thin_batch_run(self):
        return lib().avrf_thin_batch_run(self._h)

    def pedersen_batch_verify(self, items_ios_xy, ads, proofs):
        b = Batch.from_items(items_ios_xy, ads, proofs=proofs)
        return lib().avrf_pedersen_batch_verify(self._h, C.c_size_t(b.n), b.ios_xy, b.io_counts, b.ads, b.ad_lens, b.proofs)

    def pedersen_batch_stage(self, b):
        return lib().avrf_pedersen_batch_stage(self._h, C.c_size_t(b.n), b.ios_xy, b.io_counts, b.ads, b.ad_lens, b.proofs)

    def pedersen_batch_run(self):
        return lib().avrf_pedersen_batch_run(self._h)

    # the run of the staged batch in three calls (include/avrf.h): one host thread can keep several contexts in flight
    def batch_run_begin(self):
        return lib().avrf_batch_run_begin(self._h)

    def batch_run_hash(self):
        return lib().avrf_batch_run_hash(self._h)

    def batch_run_end(self):
        return lib().avrf_batch_run_end(self._h)

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

    # -- independent per-item calls
    def thin_prove(self, b):
        out = (C.c_uint8 * max(1, 96 * b.n))()
        st = lib().avrf_thin_prove(self._h, C.c_size_t(b.n), b.sks, b.pks_xy, b.ios_xy, b.io_counts, b.ads, b.ad_lens, out)
        if st != OK:
            raise AvrfError(f"avrf_thin_prove -> {st}")
        return bytes(out)[: 96 * b.n]

    def thin_verify(self, b):
        out = (C.c_int32 * max(1, b.n))()
        st = lib().avrf_thin_verify(self._h, C.c_size_t(b.n), b.pks_xy, b.ios_xy, b.io_counts, b.ads, b.ad_lens, b.proofs, out)
        if st != OK:
            raise AvrfError(f"avrf_thin_verify -> {st}")
        return list(out)[: b.n]

    def tiny_prove(self, b):
        out = (C.c_uint8 * max(1, 48 * b.n))()
        st = lib().avrf_tiny_prove(self._h, C.c_size_t(b.n), b.sks, b.pks_xy, b.ios_xy, b.io_counts, b.ads, b.ad_lens, out)
        if st != OK:
            raise AvrfError(f"avrf_tiny_prove -> {st}")
        return bytes(out)[: 48 * b.n]

    def tiny_verify(self, b):
        out = (C.c_int32 * max(1, b.n))()
        st = lib().avrf_tiny_verify(self._h, C.c_size_t(b.n), b.pks_xy, b.ios_xy, b.io_counts, b.ads, b.ad_lens, b.proofs, out)
        if st != OK:
            raise AvrfError(f"avrf_tiny_verify -> {st}")
        return list(out)[: b.n]

    def pedersen_prove(self, b):
        out, bl = (C.c_uint8 * max(1, 256 * b.n))(), (C.c_uint8 * max(1, 32 * b.n))()
        st = lib().avrf_pedersen_prove(self._h, C.c_size_t(b.n), b.sks, b.pks_xy, b.ios_xy, b.io_counts, b.ads, b.ad_lens, out, bl)
        if st != OK:
            raise AvrfError(f"avrf_pedersen_prove -> {st}")
        return bytes(out)[: 256 * b.n], bytes(bl)[: 32 * b.n]

    def pedersen_verify(self, b):
        out = (C.c_int32 * max(1, b.n))()
        st = lib().avrf_pedersen_verify(self._h, C.c_size_t(b.n), b.ios_xy, b.io_counts, b.ads, b.ad_lens, b.proofs, out)
        if st != OK:
            raise AvrfError(f"avrf_pedersen_verify -> {st}")
        return list(out)[: b.n]

    # -- the same calls into caller-owned ctypes buffers (what a binding that keeps its buffers would do; no Python copies):
    # out: c_uint8 * (256 n) / (96 n); blindings: c_uint8 * (32 n) or None; status: c_int32 * n.  Return the call's status.
    def pedersen_prove_into(self, b, out, blindings=None):
        return lib().avrf_pedersen_prove(self._h, C.c_size_t(b.n), b.sks, b.pks_xy, b.ios_xy, b.io_counts, b.ads, b.ad_lens, out, blindings)

    def pedersen_verify_into(self, b, status):
        return lib().avrf_pedersen_verify(self._h, C.c_size_t(b.n), b.ios_xy, b.io_counts, b.ads, b.ad_lens, b.proofs, status)

    def thin_prove_into(self, b, out):
        return lib().avrf_thin_prove(self._h, C.c_size_t(b.n), b.sks, b.pks_xy, b.ios_xy, b.io_counts, b.ads, b.ad_lens, out)

    def thin_verify_into(self, b, status):
        return lib().avrf_thin_verify(self._h, C.c_size_t(b.n), b.pks_xy, b.ios_xy, b.io_counts, b.ads, b.ad_lens, b.proofs, status)

    def scalar_mul_base(self, scalars):
        n = len(scalars) // 32
        out = (C.c_uint8 * max(1, 64 * n))()
        st = lib().avrf_scalar_mul_base(self._h, C.c_size_t(n), _u8(scalars), out)
        if st != OK:
            raise AvrfError(f"avrf_scalar_mul_base -> {st}")
        return bytes(out)[: 64 * n]

    def scalar_mul(self, scalars, points_xy):
        n = len(scalars) // 32
        out = (C.c_uint8 * max(1, 64 * n))()
        st = lib().avrf_scalar_mul(self._h, C.c_size_t(n), _u8(scalars), _u8(points_xy), out)
        if st != OK:
            raise AvrfError(f"avrf_scalar_mul -> {st}")
        return bytes(out)[: 64 * n]

    def output_hash(self, points_xy, n_bytes=32):
        """Output::hash::<N> of output points given as x || y -> list of N-byte strings"""
        n = len(points_xy) // 64
        out = (C.c_uint8 * max(1, n_bytes * n))()
        st = lib().avrf_output_hash(self._h, C.c_size_t(n), _u8(points_xy), C.c_size_t(n_bytes), out)
        if st != OK:
            raise AvrfError(f"avrf_output_hash -> {st}")
        b = bytes(out)
        return [b[n_bytes * i: n_bytes * (i + 1)] for i in range(n)]

    def secret_from_seed(self, seeds, with_public=True):
        """Secret::from_seed for 32-byte seeds -> (secret scalars n x 32, public keys n x 64 or None)"""
        n = len(seeds) // 32
        sks = (C.c_uint8 * max(1, 32 * n))()
        pks = (C.c_uint8 * max(1, 64 * n))() if with_public else None
        st = lib().avrf_secret_from_seed(self._h, C.c_size_t(n), _u8(seeds), sks, pks)
        if st != OK:
            raise AvrfError(f"avrf_secret_from_seed -> {st}")
        return bytes(sks)[: 32 * n], (bytes(pks)[: 64 * n] if with_public else None)

    def hash_to_curve(self, messages):
        """Input::new for a list of byte strings -> (xy bytes n x 64, statuses)."""
        n = len(messages)
        out = (C.c_uint8 * max(1, n * 64))()
        st = (C.c_int32 * max(1, n))()
        rc = lib().avrf_hash_to_curve(self._h, C.c_size_t(n), _u8(b"".join(messages)), _u32([len(m) for m in messages]), out, st)
        if rc != OK:
            raise AvrfError(f"avrf_hash_to_curve -> {rc}")
        return bytes(out)[: n * 64], list(st)[:n]

    def points_decompress(self, comp, validate=False):
        n = len(comp) // self.point_len
        out, st_out = (C.c_uint8 * max(1, 64 * n))(), (C.c_int32 * max(1, n))()
        st = lib().avrf_points_decompress(self._h, C.c_size_t(n), _u8(comp), out, int(validate), st_out)
        if st != OK:
            raise AvrfError(f"avrf_points_decompress -> {st}")
        return bytes(out)[: 64 * n], list(st_out)[:n]

    def points_compress(self, xy):
        n = len(xy) // 64
        out = (C.c_uint8 * max(1, self.point_len * n))()
        st = lib().avrf_points_compress(self._h, C.c_size_t(n), _u8(xy), out)
        if st != OK:
            raise AvrfError(f"avrf_points_compress -> {st}")
        return bytes(out)[: self.point_len * n]


class PinnedBatch:
    """A Batch whose buffers live in page-locked host memory (avrf_host_alloc): staging copies from it are DMA transfers."""

    def __init__(self, n, ios_xy, io_counts, ads, ad_lens, pks_xy=None, proofs=None):
        self.n = n
        self._ptrs = []
        self.ios_xy = self._pin(bytes(ios_xy))
        self.io_counts = self._pin(bytes(_u32(io_counts)))
        self.ads = self._pin(bytes(ads))
        self.ad_lens = self._pin(bytes(_u32(ad_lens)))
        self.pks_xy = self._pin(bytes(pks_xy)) if pks_xy is not None else None
        self.proofs = self._pin(bytes(proofs)) if proofs is not None else None

    def _pin(self, data):
        p = C.c_void_p()
        st = lib().avrf_host_alloc(C.c_size_t(max(1, len(data))), C.byref(p))
        if st != OK or not p:
            raise AvrfError(f"avrf_host_alloc -> {st}")
        C.memmove(p, data, len(data))
        self._ptrs.append(p)
        return p

    def close(self):
        for p in self._ptrs:
            lib().avrf_host_free(p)
        self._ptrs = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Pool:
    """avrf_pool: many BatchVerifier::verify jobs in flight on one device (include/avrf.h).  kind 1 thin, 2 pedersen."""

    def __init__(self, suite=BANDERSNATCH_SHA512_ELL2, device=0, kind=1, slots=8, lanes=4, threads=2, hash_group=1, depth=0):
        self._h = C.c_void_p()
        st = lib().avrf_pool_create(int(suite), int(device), int(kind), int(slots), int(lanes), int(depth), int(threads), int(hash_group), C.byref(self._h))
        if st != OK:
            raise AvrfError(f"avrf_pool_create -> {st}")
        self.kind = kind
        self._keep = {}                                         # ticket -> batch object (its buffers must outlive the run)

    def close(self):
        if self._h:
            lib().avrf_pool_destroy(self._h)
            self._h = C.c_void_p()
            self._keep = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_validation(self, level):
        return lib().avrf_pool_set_validation(self._h, int(level))

    def submit(self, b):
        """b: Batch or PinnedBatch.  Returns the ticket."""
        t = C.c_uint64(0)
        st = lib().avrf_pool_submit(self._h, C.c_size_t(b.n), b.pks_xy, b.ios_xy, b.io_counts, b.ads, b.ad_lens, b.proofs, C.byref(t))
        if st != OK:
            raise AvrfError(f"avrf_pool_submit -> {st}")
        self._keep[t.value] = b
        return t.value

    def submit_wire(self, b, validate=1):
        """b: a Batch / PinnedBatch whose pks_xy / ios_xy / proofs fields hold the WIRE bytes (compressed points)"""
        t = C.c_uint64(0)
        st = lib().avrf_pool_submit_wire(self._h, C.c_size_t(b.n), b.pks_xy, b.ios_xy, b.io_counts, b.ads, b.ad_lens, b.proofs, int(validate), C.byref(t))
        if st != OK:
            raise AvrfError(f"avrf_pool_submit_wire -> {st}")
        self._keep[t.value] = b
        return t.value

    def wait(self, ticket):
        s = C.c_int(0)
        st = lib().avrf_pool_wait(self._h, C.c_uint64(ticket), C.byref(s))
        if st != OK:
            raise AvrfError(f"avrf_pool_wait -> {st}")
        return s.value

    def resubmit(self, ticket, from_host=False):
        t = C.c_uint64(0)
        st = lib().avrf_pool_resubmit(self._h, C.c_uint64(ticket), 1 if from_host else 0, C.byref(t))
        if st != OK:
            raise AvrfError(f"avrf_pool_resubmit -> {st}")
        self._keep[t.value] = self._keep.pop(ticket, None)
        return t.value

    def cycle(self, steps_block, min_seconds=0.0, from_host=False, max_steps=0, expect=0):
        """-> (runs made, runs with another status than `expect`, seconds from first launch to last verdict)"""
        done, bad, sec = C.c_uint64(0), C.c_uint64(0), C.c_double(0)
        st = lib().avrf_pool_cycle(self._h, 1 if from_host else 0, C.c_uint64(steps_block), C.c_double(min_seconds), C.c_uint64(max_steps),
                                   int(expect), C.byref(done), C.byref(bad), C.byref(sec))
        if st != OK:
            raise AvrfError(f"avrf_pool_cycle -> {st}")
        return done.value, bad.value, sec.value

    def stats(self, reset=False):
        out = (C.c_double * 16)()
        st = lib().avrf_pool_stats(self._h, 1 if reset else 0, out, C.c_size_t(16))
        if st != OK:
            raise AvrfError(f"avrf_pool_stats -> {st}")
        k = ["cpu_us_begin", "cpu_us_collect", "cpu_us_hash", "cpu_us_launch", "cpu_us_end", "cpu_us_total", "hash_groups", "hashed", "sleeps",
             "accumulate_ms_total", "accumulate_launches"]
        return {k[i]: out[i] for i in range(len(k))}
