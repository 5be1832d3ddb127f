"""ctypes harness for the Ring-VRF entry points of libavrf.so (include/avrf.h, "Ring VRF" section).

Mirrors `RingSetup` / `RingProverKey` of the reference (src/ring.rs:340-464):
    setup = RingSetup(ctx, srs_bytes, ring_size)      # RingSetup::from_pcs_params
    key, commitment = setup.index(pks_xy)             # prover_key / verifier_key -> ring_proof::index
    proofs = key.prove(key_indices, blindings)        # RingProver::prove (blinding disabled)
"""
import ctypes as C

from . import _native as nat


class RingKey:
    def __init__(self, setup, handle, commitment):
        self.setup, self._h, self.commitment = setup, handle, commitment

    def prove(self, key_indices, blindings, blinding_mode=0):
        n = len(key_indices)
        plen = self.setup.proof_len
        out = (C.c_uint8 * max(1, n * plen))()
        st = nat.lib().avrf_ring_prove(self._h, C.c_size_t(n), nat._u32(key_indices), nat._u8(b"".join(blindings)), int(blinding_mode), out)
        if st != nat.OK:
            raise nat.AvrfError(f"avrf_ring_prove -> {st}")
        b = bytes(out)
        return [b[i * plen: (i + 1) * plen] for i in range(n)]

    def close(self):
        if self._h:
            nat.lib().avrf_ring_key_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RingSetup:
    def __init__(self, ctx, srs_bytes, ring_size, verifier_only=False, seed=None):
        """srs_bytes: the URS (RingSetup / PcsParams, either ark-serialize mode); with verifier_only=True the serialised
        PcsVerifierParams (g1, g2, tau g2) instead -- a setup that verifies but holds no SRS (src/ring.rs:466-482);
        seed (32 bytes, srs_bytes None): RingSetup::from_seed (src/ring.rs:359-366, avrf_ring_setup_from_seed)."""
        L = nat.lib()
        for f in ("avrf_ring_max_ring_size", "avrf_ring_domain_size", "avrf_ring_proof_len", "avrf_ring_commitment_len"):
            getattr(L, f).restype = C.c_size_t
        self.ctx = ctx
        self._h = C.c_void_p()
        if seed is not None:
            assert srs_bytes is None and len(seed) == 32
            fn = "avrf_ring_setup_from_seed"
            st = L.avrf_ring_setup_from_seed(ctx._h, nat._u8(seed), C.c_size_t(ring_size), C.byref(self._h))
        else:
            fn = "avrf_ring_verifier_setup_load" if verifier_only else "avrf_ring_setup_load"
            st = getattr(L, fn)(ctx._h, nat._u8(srs_bytes), C.c_size_t(len(srs_bytes)), C.c_size_t(ring_size), C.byref(self._h))
        self.status = st
        if st != nat.OK:
            self._h = None
            raise nat.AvrfError(f"{fn} -> {st}")
        self.max_ring_size = L.avrf_ring_max_ring_size(self._h)
        self.domain_size = L.avrf_ring_domain_size(self._h)
        self.proof_len = L.avrf_ring_proof_len(self._h)
        self.commitment_len = L.avrf_ring_commitment_len(self._h)

    @classmethod
    def from_seed(cls, ctx, ring_size, seed):
        return cls(ctx, None, ring_size, seed=seed)

    def _ser(self, fn, compress):
        ln = C.c_size_t(0)
        getattr(nat.lib(), fn)(self._h, int(compress), None, C.c_size_t(0), C.byref(ln))
        out = (C.c_uint8 * max(1, ln.value))()
        st = getattr(nat.lib(), fn)(self._h, int(compress), out, C.c_size_t(ln.value), C.byref(ln))
        if st != nat.OK:
            raise nat.AvrfError(f"{fn} -> {st}")
        return bytes(out)[: ln.value]

    def serialize(self, compress=False):
        """CanonicalSerialize for RingSetup (src/ring.rs:484-521): the URS bytes this setup keeps."""
        return self._ser("avrf_ring_setup_serialize", compress)

    def pcs_verifier_params(self, compress=False):
        """RingSetup::pcs_verifier_params (src/ring.rs:435): RawKzgVerifierKey { g1, g2, tau_in_g2 }, serialised."""
        return self._ser("avrf_ring_pcs_verifier_params_serialize", compress)

    def builder_params(self, compress=False):
        """RingBuilderPcsParams (src/ring.rs:523-529): the SRS in Lagrangian form, serialised."""
        return self._ser("avrf_ring_builder_params_serialize", compress)

    def index(self, pks_xy):
        """pks_xy: list of 64-byte keys.  Returns a RingKey (its .commitment = compressed RingCommitment)."""
        h = C.c_void_p()
        com = (C.c_uint8 * self.commitment_len)()
        st = nat.lib().avrf_ring_index(self._h, nat._u8(b"".join(pks_xy)), C.c_size_t(len(pks_xy)), C.byref(h), com)
        if st != nat.OK:
            raise nat.AvrfError(f"avrf_ring_index -> {st}")
        return RingKey(self, h, bytes(com))

    def close(self):
        if self._h:
            nat.lib().avrf_ring_setup_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class VerifierKeyBuilder:
    """VerifierKeyBuilder (src/ring.rs:539-637): ring commitment built by appending keys in batches."""

    def __init__(self, setup):
        L = nat.lib()
        L.avrf_ring_vk_builder_free_slots.restype = C.c_size_t
        self.setup = setup
        self._h = C.c_void_p()
        st = L.avrf_ring_vk_builder_new(setup._h, C.byref(self._h))
        if st != nat.OK:
            self._h = None
            raise nat.AvrfError(f"avrf_ring_vk_builder_new -> {st}")

    def free_slots(self):
        return nat.lib().avrf_ring_vk_builder_free_slots(self._h)

    def append(self, pks_xy):
        """Returns the status (0 ok, 3 RingCapacityExceeded: nothing appended, 2 InvalidData)."""
        return nat.lib().avrf_ring_vk_builder_append(self._h, nat._u8(b"".join(pks_xy)), C.c_size_t(len(pks_xy)))

    def finalize(self):
        out = (C.c_uint8 * self.setup.commitment_len)()
        st = nat.lib().avrf_ring_vk_builder_finalize(self._h, out)
        if st != nat.OK:
            raise nat.AvrfError(f"avrf_ring_vk_builder_finalize -> {st}")
        return bytes(out)

    def close(self):
        if self._h:
            nat.lib().avrf_ring_vk_builder_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ring_batch_verify(setup, ring_commitments, ring_of_item, instances_xy, ring_proofs):
    """RingVerifier::verify / RingBatchVerifier (src/ring.rs:242,682-735) on bare ring proofs.
    ring_commitments: list of compressed RingCommitment bytes; ring_of_item: list of indices into it (or None);
    instances_xy: list of 64-byte key commitments Yb; ring_proofs: list of proof bytes.  Returns the status."""
    n = len(ring_proofs)
    roi = nat._u32(ring_of_item) if ring_of_item is not None else None
    return nat.lib().avrf_ring_batch_verify(setup._h, C.c_size_t(n), nat._u8(b"".join(ring_commitments)), C.c_size_t(len(ring_commitments)),
                                            roi, nat._u8(b"".join(instances_xy)), nat._u8(b"".join(ring_proofs)))


def srs_generate(ctx, suite, tau, g1, g2, ring_size):
    """Kzg::setup with an explicit trapdoor (RingSetup::from_seed / from_rand, src/ring.rs:359-374): URS bytes for a ring
    of `ring_size` keys.  tau: int < r; g1, g2: generator entries in URS encoding (e.g. sliced from an SRS file)."""
    L = nat.lib()
    L.avrf_ring_pcs_domain_size.restype = C.c_size_t
    n_g1 = L.avrf_ring_pcs_domain_size(int(suite), C.c_size_t(ring_size))
    need = 8 + n_g1 * len(g1) + 8 + 2 * len(g2)
    out = (C.c_uint8 * need)()
    ln = C.c_size_t(0)
    st = L.avrf_ring_srs_generate(ctx._h, nat._u8(int(tau).to_bytes(32, "little")), nat._u8(g1), nat._u8(g2), C.c_size_t(n_g1), out, C.c_size_t(need), C.byref(ln))
    if st != nat.OK:
        raise nat.AvrfError(f"avrf_ring_srs_generate -> {st}")
    return bytes(out)[:ln.value]


def pairing_check(setup, a_xy, b_xy):
    """ok[i] = e(A_i, g2) * e(B_i, tau g2) == 1 on the device (avrf_ring_pairing_check); a_xy, b_xy: lists of canonical LE x||y."""
    n = len(a_xy)
    out = (C.c_int32 * max(1, n))()
    st = nat.lib().avrf_ring_pairing_check(setup._h, C.c_size_t(n), nat._u8(b"".join(a_xy)), nat._u8(b"".join(b_xy)), out)
    if st != nat.OK:
        raise nat.AvrfError(f"avrf_ring_pairing_check -> {st}")
    return list(out)[:n]


def ring_verify_each(setup, ring_commitments, ring_of_item, instances_xy, ring_proofs):
    """n x RingVerifier::verify with per-proof statuses (avrf_ring_verify_each): pairings on the device."""
    n = len(ring_proofs)
    roi = nat._u32(ring_of_item) if ring_of_item is not None else None
    out = (C.c_int32 * max(1, n))()
    st = nat.lib().avrf_ring_verify_each(setup._h, C.c_size_t(n), nat._u8(b"".join(ring_commitments)), C.c_size_t(len(ring_commitments)),
                                         roi, nat._u8(b"".join(instances_xy)), nat._u8(b"".join(ring_proofs)), out)
    if st != nat.OK:
        raise nat.AvrfError(f"avrf_ring_verify_each -> {st}")
    return list(out)[:n]
