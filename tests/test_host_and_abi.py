"""CPU tests: the C-ABI library loads and exports every symbol include/exon_hip.h declares, and the host-side
planning helpers agree with the oracle.  No compute calls (no GPU here)."""
import os
import re

import pytest

import exon_amd
from exon_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    h = open(os.path.join(ROOT, "include", "exon_hip.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(exon_hip_[a-z0-9_]+)\s*\(", h)))


def test_library_exports_every_declared_symbol():
    lib = exon_amd.load()
    syms = header_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/exon_hip.h but not exported"
    # and the python binding declares a signature for each of them
    assert sorted(_lib.SIGNATURES) == syms


def test_abi_version():
    assert exon_amd.load().exon_hip_abi_version() == 5


def test_no_device_fails_loudly():
    """On a box without a GPU the product refuses to run (no CPU fallback)."""
    import ctypes as C
    lib = exon_amd.load()
    n = C.c_int(-1)
    lib.exon_hip_device_count(C.byref(n))
    if n.value > 0:
        pytest.skip("GPU present")
    with pytest.raises(exon_amd.ExonHipError, match="no HIP device"):
        exon_amd.Context(0)


@pytest.mark.parametrize("region", ["1", "chr1:1-12209145", "1:9999921", "7:50000000-100000000", "a:b", "HLA-A*01:01",
                                    "chr1:0", "chr1:5-", "X:+7-9"])
def test_parse_region_matches_oracle(oracle, region):
    assert exon_amd.parse_region(region) == oracle.parse_region(region)


def test_parse_region_rejects_empty():
    with pytest.raises(exon_amd.ExonHipError):
        exon_amd.parse_region("")


@pytest.mark.parametrize("sizes,target", [([50, 10, 30, 20, 40], 2), ([5, 5, 5], 8), ([7], 3), ([], 4),
                                          ([3, 1, 2, 9, 9, 4, 8, 1], 3), (list(range(100, 0, -1)), 8)])
def test_regroup_files_by_size_matches_oracle(oracle, sizes, target):
    assert exon_amd.regroup_files_by_size(sizes, target) == oracle.regroup_files_by_size(sizes, target)


def test_shard_rows_cover_and_align():
    from exon_amd.distributed import shard_rows
    for n in (0, 1, 7, 8, 1000, 10**9, 10**9 + 3):
        for w in (1, 2, 4, 8):
            prev = 0
            for r in range(w):
                lo, hi = shard_rows(n, r, w)
                assert lo == prev and lo <= hi and (lo % 8 == 0 or lo == n)
                prev = hi
            assert prev == n
