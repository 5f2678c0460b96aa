"""The device-capable decimal -> binary32 conversion (Eisel-Lemire, exon_amd/csrc/host/decimal_f32.h) must be
correctly rounded: checked bit-for-bit against glibc strtof on random decimal strings, the float range extremes and
a sweep of %.9g round trips.  CPU test (the same header is compiled into the GPU text parser)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_eisel_lemire_matches_strtof(tmp_path):
    exe = tmp_path / "check_decimal_f32"
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tools", "check_decimal_f32.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe), "3000000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "mismatches 0" in out.stdout
