"""CPU ORACLE -- test infrastructure only (see oracle/exon_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (exon_amd) never does.
"""
from .oracle_c import Oracle, build_oracle  # noqa: F401
