"""Corrupted inputs through the GPU decode pipelines (tools/fuzz_gpu_decode.py): BAM / BCF / VCF corrupted BEFORE the BGZF
framing (so the CRC passes and the device record splitters / parsers see the damage) and plain SAM.  One process, one ctx,
all formats in sequence -- the recycled device buffers are part of what is tested: a malformed VCF line once left its
FILTER slot undefined and k_remap_filters indexed a table with whatever an earlier scan had left there.
Every input must give the host decoder's answer or be rejected by both paths."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_corrupted_files_give_the_host_answer_or_an_error():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu_decode.py"), "60"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if "same answer" in l]
    assert len(lines) == 8
    for l in lines:
        w = l.split()
        same, both, gpu_only = int(w[3]), int(w[6]), int(w[-1])
        assert same + both + gpu_only == 60 and same > 0, l
