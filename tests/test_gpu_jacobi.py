"""The two-symbol macro-step Jacobi of fp256.h (fp_jacobi2_nf: the prime-order-subgroup test of Validate::Yes, src/lib.rs:410-433)
against Euler's criterion and against the single-bit form it replaced: tools/jacobi_probe.hip is compiled with the library's
flags and run on the GPU box (structured inputs -- 0, small values, p - small, 2^k and neighbours, zero low words, values whose top
limbs equal the modulus's, short values -- and random pairs on two fields)."""
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_jacobi_probe(tmp_path):
    exe = str(tmp_path / "jacobi_probe")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-mllvm", "-enable-ipra=0", "-Wno-unused-value",
                           "-I", os.path.join(ROOT, "ark_vrf_amd", "csrc"), "-o", exe, os.path.join(ROOT, "tools", "jacobi_probe.hip")])
    r = subprocess.run([exe, "65536"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "jacobi probe ok" in r.stdout and "mismatches 0" in r.stdout
