"""GPU: the same valid inputs must give the same verdicts and bytes every time.  Guards a failure that was sporadic by nature
(DESIGN.md section 7, item 6): with interprocedural register allocation the input-flag word of `k_thin_prove<SuiteBabyJubJub>`,
live across dozens of calls into out-of-line field functions whose asm multipliers clobber fixed registers, came back
corrupted on SOME launches -- InvalidData for valid input, in one suite, depending on what ran before.  Contexts of all suites
are alive at once and their provers / verifiers are interleaved for many rounds."""
import json
import os

import pytest

import oracle as orc

pytestmark = pytest.mark.gpu
NAMES = {0: "bandersnatch_sha-512_ell2", 1: "baby-jubjub_sha-512_tai", 2: "jubjub_sha-512_tai", 3: "ed25519_sha-512_tai",
         5: "bandersnatch_shake128_ell2", 6: "testing_sha-256_tai"}


def test_interleaved_suites_repeat(golden_dir):
    from ark_vrf_amd import _native as nat
    from ark_vrf_amd._native import Batch
    from helpers import xy
    work = {}
    for s, name in NAMES.items():
        vs = json.load(open(os.path.join(golden_dir, name + "_thin.json")))
        c = nat.Context(s)
        sks = [bytes.fromhex(v["sk"]) for v in vs]
        pks = [xy(s, bytes.fromhex(v["pk"])) for v in vs]
        ios = [[(xy(s, bytes.fromhex(v["h"])), xy(s, bytes.fromhex(v["gamma"])))] for v in vs]
        ads = [bytes.fromhex(v["ad"]) for v in vs]
        first = c.thin_prove(Batch.from_items(ios, ads, sks=sks, pks_xy=pks))
        assert [orc.point_compress(s, first[96 * j: 96 * j + 64]).hex() + first[96 * j + 64: 96 * j + 96].hex() for j in range(7)] == \
            [v["proof_r"] + v["proof_s"] for v in vs]
        # items with TWO I/O pairs take the lane-per-item kernels (one-pair calls of this size take the 32-lanes-per-item ones since
        # round 4): both families stay under the interleaving
        ios2 = []
        for j in range(7):
            inp2 = ios[(j + 1) % 7][0][0]                                             # another item's input point, this item's key
            out2 = xy(s, orc.vrf_output(s, sks[j], orc.point_compress(s, inp2)))
            ios2.append([ios[j][0], (inp2, out2)])
        first2 = c.thin_prove(Batch.from_items(ios2, ads, sks=sks, pks_xy=pks))
        work[s] = (c, sks, pks, ios, ads, first, ios2, first2)
    for rnd in range(25):
        for s, (c, sks, pks, ios, ads, first, ios2, first2) in work.items():
            assert c.thin_prove(Batch.from_items(ios2, ads, sks=sks, pks_xy=pks)) == first2, (rnd, s, "two pairs")
            assert c.thin_verify(Batch.from_items(ios2, ads, pks_xy=pks, proofs=[first2[96 * j: 96 * j + 96] for j in range(7)])) == [0] * 7, (rnd, s, "two pairs")
            assert c.thin_prove(Batch.from_items(ios, ads, sks=sks, pks_xy=pks)) == first, (rnd, s, "given")
            assert c.thin_prove(Batch.from_items(ios, ads, sks=sks)) == first, (rnd, s, "derived")
            tp = [first[96 * j: 96 * j + 96] for j in range(7)]
            assert c.thin_verify(Batch.from_items(ios, ads, pks_xy=pks, proofs=tp)) == [0] * 7, (rnd, s)
            pr, _ = c.pedersen_prove(Batch.from_items(ios, ads, sks=sks, pks_xy=pks))
            assert c.pedersen_verify(Batch.from_items(ios, ads, proofs=[pr[256 * j: 256 * j + 256] for j in range(7)])) == [0] * 7, (rnd, s)
    for c, *_ in work.values():
        c.close()


BLOCKING_CHILD = r"""
import json, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from ark_vrf_amd import _native as nat
from ark_vrf_amd._native import Batch
from helpers import xy
s = 0
vs = json.load(open(sys.argv[1]))
sks = [bytes.fromhex(v["sk"]) for v in vs]
pks = [xy(s, bytes.fromhex(v["pk"])) for v in vs]
ios = [[(xy(s, bytes.fromhex(v["h"])), xy(s, bytes.fromhex(v["gamma"])))] for v in vs]
ads = [bytes.fromhex(v["ad"]) for v in vs]
assert nat.set_blocking_sync(99, True) != 0          # no such device
c = nat.Context(s)
spin = c.thin_prove(Batch.from_items(ios, ads, sks=sks, pks_xy=pks))
assert nat.set_blocking_sync(0, True) == 0
assert c.thin_prove(Batch.from_items(ios, ads, sks=sks, pks_xy=pks)) == spin
tp = [spin[96 * j: 96 * j + 96] for j in range(len(vs))]
assert c.thin_verify(Batch.from_items(ios, ads, pks_xy=pks, proofs=tp)) == [0] * len(vs)
assert c.thin_batch_verify(pks, ios, ads, tp) == 0
bad = [tp[0][:95] + bytes([tp[0][95] ^ 1])] + tp[1:]
assert c.thin_batch_verify(pks, ios, ads, bad) != 0
c.close()
print("blocking-wait-ok", spin[:8].hex())
"""


def test_blocking_wait_changes_nothing_but_the_wait(golden_dir):
    """avrf_device_set_blocking_sync: the host threads sleep while they wait; verdicts and bytes are those of the default mode.
    The switch is device-wide and meant to be thrown once, early, by the process that owns the device (bench.py does): the test
    does the same in a process of its own."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", BLOCKING_CHILD, os.path.join(golden_dir, NAMES[0] + "_thin.json")], cwd=root,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "blocking-wait-ok" in r.stdout, r.stderr[-2000:]
