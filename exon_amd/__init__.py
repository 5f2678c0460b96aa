"""exon_amd -- MI355X-native scan -> filter -> aggregate path for Exon-style genomic Arrow data.

The product is the HIP library `exon_amd/lib/libexon_hip.so` behind the C ABI of `include/exon_hip.h`;
this package is the thin host side used by tests, the CLI and bench.py.  Importing the operators
fails loudly when the library is missing -- there is no CPU fallback.
"""
from ._lib import ExonHipError, LIB_PATH, build, load  # noqa: F401
from .engine import (Context, DeviceBuffer, Plan, Scan, Stream, VCFParser, FASTQParser, BAMParser, bgzf_scan, index_query, parse_region,  # noqa: F401
                     regroup_files_by_size)  # noqa: F401

__all__ = ["Context", "DeviceBuffer", "Plan", "Scan", "Stream", "ExonHipError", "parse_region", "index_query",
           "regroup_files_by_size", "build", "load", "LIB_PATH"]
