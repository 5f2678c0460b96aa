"""tests/abi_harness.c: the C ABI driven from plain C the way the Rust shim drives it.  On CPU it must build, link and report
the missing device loudly; on the GPU box it runs the whole open -> push -> finish_arrow -> release sequence."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(tmp_path):
    exe = tmp_path / "abi_harness"
    lib = os.path.join(ROOT, "exon_amd", "lib")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", os.path.join(ROOT, "tests", "abi_harness.c"), "-I", os.path.join(ROOT, "include"),
                           "-L", lib, "-lexon_hip", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", str(exe)])
    return str(exe)


def test_harness_builds_links_and_fails_loudly_without_a_device(tmp_path):
    import ctypes as C
    import exon_amd
    n = C.c_int(-1)
    exon_amd.load().exon_hip_device_count(C.byref(n))
    if n.value > 0:
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([build(tmp_path), "--allow-no-device"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("NO_DEVICE") and "no HIP device" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([build(tmp_path)], capture_output=True, text=True)  # without the flag the missing device is an error
    assert r.returncode == 1 and "exon_hip_ctx_create" in r.stderr


@pytest.mark.gpu
def test_harness_runs_the_shim_call_sequence_on_the_gpu(tmp_path):
    r = subprocess.run([build(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("OK 5 groups, 3 batches moved"), r.stdout + r.stderr
