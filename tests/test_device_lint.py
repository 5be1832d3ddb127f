"""The build-time fence of DESIGN.md section 7-5 (tools/lint_device_code.py): the shipped library passes, and the shape that
faulted in round 3 -- generic multiply-adds inlined en masse into an out-of-line device function -- is refused."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINT = os.path.join(ROOT, "tools", "lint_device_code.py")


def test_shipped_library_passes_the_lint():
    r = subprocess.run([sys.executable, LINT], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "code objects" in r.stdout and "VIOLATION" not in r.stdout


def test_lint_refuses_the_fragile_shape(tmp_path):
    exe = str(tmp_path / "lint_selftest")
    c = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-o", exe, os.path.join(ROOT, "tools", "lint_selftest.hip")],
                       capture_output=True, text=True, timeout=600)
    assert c.returncode == 0, c.stderr[-3000:]
    r = subprocess.run([sys.executable, LINT, exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 1, r.stdout
    assert "VIOLATION" in r.stdout and "sgpr_carry_multiply_adds_en_masse" in r.stdout and "a_few" not in r.stdout
