"""The C++ mirror of the reference interface (include/avrf.hpp) against the reference's golden
vectors: compiled with g++ against libavrf.so and run on the GPU box."""
import json
import os
import subprocess

import pytest

import oracle as orc
from conftest import ROOT
from helpers import xy

pytestmark = pytest.mark.gpu
NAMES = {0: "bandersnatch_sha-512_ell2", 1: "baby-jubjub_sha-512_tai", 2: "jubjub_sha-512_tai"}


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("cpp") / "mirror_check")
    lib = os.path.join(ROOT, "ark_vrf_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "mirror_check.cpp"),
                           "-o", out, "-L", lib, "-lavrf", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib"])
    return out


@pytest.mark.parametrize("suite", [0, 1, 2])
def test_mirror_reproduces_vectors(exe, golden_dir, suite):
    vt = json.load(open(os.path.join(golden_dir, NAMES[suite] + "_thin.json")))
    vp = json.load(open(os.path.join(golden_dir, NAMES[suite] + "_pedersen.json")))
    vy = json.load(open(os.path.join(golden_dir, NAMES[suite] + "_tiny.json")))
    for i, (t, p, y) in enumerate(list(zip(vt, vp, vy))[:3]):
        seed = bytes([[1, 2, 3, 4, 5, 5, 6][i]]) + bytes(31)              # src/testing.rs: the vectors' secrets come from these seeds
        res = subprocess.run([exe, str(suite), t["sk"], xy(suite, bytes.fromhex(t["h"])).hex(), t["ad"] or "", seed.hex()],
                             capture_output=True, text=True, check=True).stdout
        kv = dict(line.split("=", 1) for line in res.strip().splitlines())
        comp = lambda k: orc.point_compress(suite, bytes.fromhex(kv[k])).hex()
        assert comp("pk") == t["pk"] and comp("output") == t["gamma"]
        assert kv["beta"] == t["beta"] and kv["seed_sk"] == t["sk"] and comp("seed_pk") == t["pk"]   # Output::hash, Secret::from_seed
        assert kv["tiny_c"] == y["proof_c"] and kv["tiny_s"] == y["proof_s"]
        assert [kv[k] for k in ("tiny_verify", "tiny_verify_bad_ad", "tiny_verify_bad_c")] == ["0", "1", "1"]
        assert comp("thin_r") == t["proof_r"] and kv["thin_s"] == t["proof_s"]
        assert comp("ped_pk_com") == p["proof_pk_com"] and comp("ped_r") == p["proof_r"] and comp("ped_ok") == p["proof_ok"]
        assert kv["ped_s"] == p["proof_s"] and kv["ped_sb"] == p["proof_sb"] and kv["ped_blinding"] == p["blinding"]
        assert [kv[k] for k in ("thin_verify", "thin_verify_bad_ad", "thin_batch_empty", "thin_batch", "thin_batch_bad")] == ["0", "1", "0", "0", "1"]
        assert [kv[k] for k in ("ped_verify", "ped_batch", "ped_batch_bad")] == ["0", "0", "1"]
        assert kv["thin_pool"] == "010"                                   # good / tampered / good through thin::VerifierPool


@pytest.mark.parametrize("suite", [0, 1, 2])
def test_mirror_ring(exe, golden_dir, suite):
    """ring::{RingSetup, prover_key, Prover::prove, Verifier::verify, BatchVerifier} through the C++ mirror reproduce the
    reference's ring vector (commitment + deterministic ring proof) and its accept / reject behaviour."""
    v = json.load(open(os.path.join(golden_dir, NAMES[suite] + "_ring.json")))[0]
    srs = os.path.join(golden_dir, ["bls12-381-srs-2-11-uncompressed-zcash.bin", "bn254-testing-2-9-uncompressed.bin", "bls12-381-srs-2-11-uncompressed-zcash.bin"][suite])
    raw = bytes.fromhex(v["ring_pks"])
    pks = [xy(suite, raw[32 * i: 32 * i + 32]) for i in range(len(raw) // 32)]
    idx = pks.index(xy(suite, bytes.fromhex(v["pk"])))
    res = subprocess.run([exe, str(suite), v["sk"], xy(suite, bytes.fromhex(v["h"])).hex(), v["ad"] or "", srs, b"".join(pks).hex(), str(idx)],
                         capture_output=True, text=True, check=True).stdout
    kv = dict(line.split("=", 1) for line in res.strip().splitlines())
    assert kv["ring_setup_too_big"] == "3" and kv["ring_setup"] == "0" and kv["ring_index"] == "0"
    assert kv["ring_commitment"] == v["ring_pks_com"]
    assert kv["ring_proof"] == v["ring_proof"]
    assert orc.point_compress(suite, bytes.fromhex(kv["ring_ped_pk_com"])).hex() == v["proof_pk_com"]
    assert [kv[k] for k in ("ring_verify", "ring_verify_bad_ad", "ring_verify_bad_proof", "ring_hiding_differs", "ring_batch", "ring_batch_bad")] == ["0", "1", "1", "1", "0", "1"]
