"""Arithmetic identities the HIP kernels rely on, proven exhaustively on the CPU (no GPU needed)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_division_free_hswish_is_exact_for_every_f32(tmp_path):
    """tools/check_div6.c walks all 2^32 bit patterns: inside the range guard the kernels' division-free
    hard-swish equals the contract's u / 6.0f bit for bit (cpp-paddle-ocr_amd/csrc/ocr_common.h)."""
    exe = str(tmp_path / "check_div6")
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-o", exe,
                           os.path.join(ROOT, "tools", "check_div6.c"), "-lm"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mismatches inside the guard: 0 " in out.stdout
