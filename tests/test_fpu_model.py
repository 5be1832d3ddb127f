"""tools/fpu_model.py: the exact-integer model of the unsaturated-limb arithmetic the bucket-accumulation kernels run on
(csrc/fpu.h, fpu_te.h, fpu_g1.h) -- no column of a multiplication leaves the signed 64-bit accumulator, the inductive value
bounds hold in the worst case, mixed additions agree with the affine group laws, the exceptional cases of the XYZZ law are
detected exactly.  The kernels are generated from the same layout constants (UL<F>, G1U<C>); the device-side agreement with the
saturated form is tests/test_gpu_unsat_accumulate.py and tools/ubench_fpu.hip."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_model_checks_pass():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fpu_model.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "all checks passed" in r.stdout
    for name in ("FqBandersnatch", "FqBls12381", "FqBn254", "SuiteBandersnatch", "SuiteBabyJubJub", "SuiteEd25519", "G1Bls12381", "G1Bn254"):
        assert name in r.stdout


def test_layout_constants_match_the_headers():
    """the constants the model derives (3a - 2b = SH, base shifts) are the ones fpu_g1.h asserts"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fpu_model as M
    C = M.parse()
    f381, f254 = M.Field("FqBls12381", C["FqBls12381"]), M.Field("FqBn254", C["FqBn254"])
    g381, g254 = M.G1("G1Bls12381", C["G1Bls12381"], f381), M.G1("G1Bn254", C["G1Bn254"], f254)
    assert (f381.W, f381.L, f381.SH) == (28, 14, 8) and (f254.W, f254.L, f254.SH) == (29, 9, 5)
    assert (g381.a, g381.bb, g381.g, g381.d, g381.sx, g381.sy) == (4, 2, 4, 6, 8, 4)
    assert (g254.a, g254.bb, g254.g, g254.d, g254.sx, g254.sy) == (3, 2, 4, 6, 4, 1)
    hdr = open(os.path.join(ROOT, "ark_vrf_amd", "csrc", "fpu_g1.h")).read()
    assert "a = SH == 8 ? 4 : 3, b = 2, m = 2" in hdr


def test_asm_multiplier_streams_match_the_model_and_the_header_is_current():
    """tools/gen_fpu_asm.py --check: the instruction streams of fu_mul_asm / fu_sqr_asm (csrc/fpu_asm_gen.h), run in its emulator,
    return the limbs of the model's fu_mul on random and extreme operands, honour the asm statement's register contract, and the
    committed header is what the generator writes from the current consts_gen.h"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_fpu_asm.py"), "--check"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    for name in ("FqBandersnatch", "FqBabyJubJub", "FqEd25519", "FqBls12381"):
        assert name in r.stdout
    assert "limb for limb" in r.stdout and "is current" in r.stdout
